// K2 (split-K query-stationary form): MFMA scan for query batches of any size (one query included) over 1024-element bf16 rows
// (768 < dim_pad <= 1024), gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// The query-stationary dataflow of crag_scan_q16.hip with 32 instead of 64 queries per wave (1024-element queries
// are twice as many registers): four waves (one per SIMD) x 32 stationary queries, corpus tiles of
// 32 rows streamed HBM -> LDS by LDS-DMA, fused threshold filter with the query on the lane (lane & 15 of each of the
// wave's two 16-query blocks; v_mfma_f32_16x16x32_bf16; the accumulators are the scores: pre-scaled queries, C = |x|^2).  A 32-row tile
// of 2-KiB rows does not fit a three-deep LDS ring, so it travels as two half-tiles of 32 rows x 512
// elements (K-parts) through a ring of FOUR 32-KiB buffers, three half-tiles in flight; one generated
// statement (crag_qsplit_kloop.inc, gen_qsplit_kloop.py holds the register contract and the wait counts)
// multiplies both K-parts of a tile into one set of accumulators, with the mid-tile rendezvous inside.  The 64
// query fragments fill all 256 AGPRs (fixed registers outside the compiler's allocation, checked on the
// emitted ISA by build.py).  Candidate appends use a flat per-query buffer + a returning atomic.
//
//   MODE 0 (sample): ordered-float score keys to sample[row][Bpad]  (row-major, coalesced)
//   MODE 1 (filter): s <= tau -> append (score key << 32 | row) to the query's candidate buffer
#include "crag_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int QS_TM = 32;        // corpus rows per tile
constexpr int QS_WAVES = 4;      // waves per workgroup: one per SIMD
constexpr int QS_QPB = 32 * QS_WAVES;
constexpr int QS_D = 1024;       // elements per (padded) row
constexpr int QS_GROWB = QS_D * 2;   // bytes per row in memory
constexpr int QS_HROWB = 1024;       // bytes per row of a half-tile in LDS (512 elements)
constexpr int QS_HALF = QS_TM * QS_HROWB;  // bytes per LDS buffer
constexpr int QS_NBUF = 4;
constexpr int QS_XN_BYTES = 256;
constexpr int QS_XN_BASE = QS_NBUF * QS_HALF;  // |x|^2 slots: [buffer][wave][64 floats]
constexpr int QS_ZERO_BASE = QS_XN_BASE + QS_NBUF * QS_WAVES * QS_XN_BYTES;  // 256 zero bytes: the "|x|^2" of the ip metric
constexpr int QS_DMA_PER_HALF = 9;   // per wave: 8 row pieces + the |x|^2 column

struct QsplitDma {            // one half-tile to prefetch (all wave-uniform)
    const unsigned char* lo;  // first byte of the half-tile's rows 0..15 (row stride QS_GROWB)
    const unsigned char* hi;  // ... rows 16..31
    unsigned lds_dst;         // LDS byte address of this wave's first row piece in the target buffer
    const float* xn;          // |x|^2 of the tile's first row
    unsigned xn_dst;          // LDS byte address of this wave's |x|^2 slot of the target buffer
};

#include "crag_qsplit_kloop.inc"

// NT: one query group per launch, so every tile is read by exactly one workgroup: nontemporal DMA loads (see
// crag_scan_q16.hip)
template <int MODE, bool NT>
__global__ __launch_bounds__(64 * QS_WAVES, 1) void k_scan_qsplit(const unsigned short* __restrict__ rows,
                                                                  const float* __restrict__ xnorm, long long row_begin,
                                                                  long long row_end, const unsigned short* __restrict__ queries /* pre-scaled: -2 q (l2), -q (ip) */,
                                                                  int n_queries, int nqg, int metric,
                                                                  const float* __restrict__ tau, uint32_t* __restrict__ sample,
                                                                  long long sample_stride, unsigned long long* __restrict__ cand,
                                                                  uint32_t* __restrict__ cnt) {
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];  // the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;  // query column of both 16-query blocks; row quarter

    // workgroup -> (query group, row-tile stream); workgroups with equal blockIdx % 8 share an XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int qg = j % nqg;
    const int stream = j / nqg;
    const int n_streams = (int)(gridDim.x >> 3) / nqg;
    const long long n_tiles = (row_end + QS_TM - 1) / QS_TM;
    const long long t_step = (long long)n_streams * 8;
    long long t = row_begin / QS_TM + (long long)stream * 8 + xcd;  // row_begin: a multiple of the tile height

    const int q_base = qg * QS_QPB + wave * 32;
    const bool wave_active = q_base < n_queries;  // wave-uniform
    const int q0 = q_base + fr, q1 = q_base + 16 + fr;
    const bool ok0 = q0 < n_queries, ok1 = q1 < n_queries;
    // stationary B operands: this lane's two queries, k = 32 s + 8 fg + [0,8), in fixed AGPRs
    qsplit_load(queries + (long long)(ok0 ? q0 : n_queries - 1) * QS_D + 8 * fg,
                queries + (long long)(ok1 ? q1 : n_queries - 1) * QS_D + 8 * fg);
    // a lane without a query holds a copy of the batch's last query: its threshold is a NaN, which no score is at or below
    // (-inf would let the -inf scores of a query with an infinite component through, towards a buffer that does not exist)
    float tq0 = __builtin_nanf(""), tq1 = __builtin_nanf("");
    if (MODE == 1 && ok0) tq0 = tau[q0];
    if (MODE == 1 && ok1) tq1 = tau[q1];

    // LDS-DMA: a wave moves rows wave + 4 i and wave + 4 i + 16 (i < 4) of a half-tile, one 1-KiB piece per
    // row; LDS position p of row r holds chunk p ^ (r & 15), rows r and r + 16 swizzle alike
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave + QS_WAVES * i;
        voff[i] = (unsigned)(r * QS_GROWB + ((lane ^ (r & 15)) << 4));
    }
    const unsigned xoff = lane * 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto half_desc = [&](long long tile, int part, int buf) {
        QsplitDma d;
        d.lo = (const unsigned char*)rows + tile * (long long)(QS_TM * QS_GROWB) + part * QS_HROWB;
        d.hi = d.lo + 16 * QS_GROWB;
        d.lds_dst = lds_base + buf * QS_HALF + wave * QS_HROWB;
        d.xn = xnorm + tile * QS_TM;
        d.xn_dst = lds_base + QS_XN_BASE + (buf * QS_WAVES + wave) * QS_XN_BYTES;
        return d;
    };
    // C++ issue of one half-tile (prologue, inactive waves, ragged last tile): exactly QS_DMA_PER_HALF instructions
    auto issue = [&](long long tile, int part, int buf) {
        const long long row0 = tile * QS_TM;
        const unsigned char* tb = (const unsigned char*)rows + row0 * (long long)QS_GROWB + part * QS_HROWB;
        const bool whole = row0 + QS_TM <= row_end;
        const unsigned r_last = (unsigned)(row_end - 1 - row0);  // ragged tile only: its last valid row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned off = voff[i] + h * 16 * QS_GROWB;
                if (!whole) {  // rows past the end re-read the last row (never past the matrix)
                    asm volatile("" : "+v"(off));
                    const unsigned r = off / QS_GROWB;
                    if (r > r_last) off -= (r - r_last) * QS_GROWB;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + off),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * QS_HALF + (wave + QS_WAVES * i + 16 * h) * QS_HROWB),
                                                 16, 0, NT ? 2 : 0);
            }
        }
        const float* gx = xnorm + row0 + lane;  // xnorm is padded by >= 256 floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                         (__attribute__((address_space(3))) void*)(lds + QS_XN_BASE + (buf * QS_WAVES + wave) * QS_XN_BYTES),
                                         4, 0, NT ? 2 : 0);
    };

    const unsigned a_lane = lds_base + fr * QS_HROWB + ((fg ^ fr) << 4);
    // C operand of a tile's first MFMAs: the lane's |x|^2 runs (l2) or zeros (ip: the score is -q.x alone)
    const bool l2 = metric == CRAG_L2;
    const unsigned x_lane = lds_base + (l2 ? QS_XN_BASE + wave * QS_XN_BYTES : QS_ZERO_BASE) + 16 * fg;
    const unsigned x_step = l2 ? QS_WAVES * QS_XN_BYTES : 0;
    if (tid < 64) ((unsigned*)(lds + QS_ZERO_BASE))[tid] = 0u;  // visible to every wave behind the prologue's barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... because the store has left this wave before it reaches that barrier

    // half-tile h of this workgroup's sequence lives in buffer h % 4; tile i = half-tiles 2 i (part 0) and 2 i + 1.
    // prologue: half-tiles 0, 1, 2 in flight, wait for the first
    int cur = 0;  // buffer of the current tile's part 0: 0 or 2
    if (t < n_tiles) {
        issue(t, 0, 0);
        issue(t, 1, 1);
    }
    if (t + t_step < n_tiles) {
        issue(t + t_step, 0, 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QS_DMA_PER_HALF) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    for (; t < n_tiles; t += t_step) {
        const long long t1 = t + t_step, t2 = t + 2 * t_step;
        const bool ex3 = t1 < n_tiles, ex4 = t2 < n_tiles;  // half-tiles h + 3 (next tile, part 1) / h + 4 exist
        const int b1 = cur + 1, b3 = (cur + 3) & 3, b4 = cur;  // buffers of h + 1, h + 3, h + 4
        const bool dma0 = ex3 && wave_active && (t1 + 1) * QS_TM <= row_end;
        const bool dma1 = ex4 && wave_active && (t2 + 1) * QS_TM <= row_end;
        bool drained = false;  // this wave's end-of-tile wait must be vmcnt(0)
        if (wave_active) {
            if (ex3 && !dma0) issue(t1, 1, b3);  // ragged next tile: up front (buffer h + 3 = h - 1 is free)
            const unsigned a0 = a_lane + cur * QS_HALF, a1 = a_lane + b1 * QS_HALF;
            const unsigned xaddr = x_lane + cur * x_step;  // both half-tiles of a tile carry its |x|^2 column
            f32x16_t acc;  // element 8 rb + 4 jb + e = SCORE of row 16 rb + 4 fg + e against the query of block jb
            float mn0, mn1;
            QsplitDma d0 = half_desc(t1, 1, b3), d1 = half_desc(t2, 0, b4);
            if (dma0 && dma1) QsplitTile<true, true, NT>::run(acc, mn0, mn1, a0, a1, xaddr, voff, xoff, d0, d1);
            else if (dma0) QsplitTile<true, false, NT>::run(acc, mn0, mn1, a0, a1, xaddr, voff, xoff, d0, d1);
            else if (dma1) QsplitTile<false, true, NT>::run(acc, mn0, mn1, a0, a1, xaddr, voff, xoff, d0, d1);
            else QsplitTile<false, false, false>::run(acc, mn0, mn1, a0, a1, xaddr, voff, xoff, d0, d1);
            if (ex4 && !dma1) issue(t2, 0, b4);  // ragged: after the mid-tile rendezvous freed buffer h
            drained = !(dma0 && dma1);
            const unsigned mb = (unsigned)t * QS_TM + 4 * fg, re = (unsigned)row_end;  // rows per collection < 2^32
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // r = 8 rb + 4 jb + e
                    const unsigned m = mb + 16 * (r >> 3) + (r & 3);
                    const int q = (r & 4) ? q1 : q0;
                    if (((r & 4) ? ok1 : ok0) && m < re) sample[m * sample_stride + q] = crag_fkey(acc[r]);
                }
                drained = true;
            } else if (__any(mn0 <= tq0 || mn1 <= tq1)) {
                // one returning atomic per lane and query block for all of its passing rows
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const float tq = jb ? tq1 : tq0;
                    if ((jb ? mn1 : mn0) <= tq) {
                        int npass = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {  // r = 4 rb + e
                            const unsigned m = mb + 16 * (r >> 2) + (r & 3);
                            npass += (acc[8 * (r >> 2) + 4 * jb + (r & 3)] <= tq && m < re) ? 1 : 0;
                        }
                        if (npass) {
                            int qa = jb ? q1 : q0;
                            asm volatile("" : "+v"(qa));  // address arithmetic stays on this rare path
                            unsigned pos = atomicAdd(&cnt[CRAG_CNT(qa)], (unsigned)npass);
                            unsigned long long* cq = cand + (long long)qa * CRAG_CAND_CAP;
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const unsigned m = mb + 16 * (r >> 2) + (r & 3);
                                const float sc = acc[8 * (r >> 2) + 4 * jb + (r & 3)];
                                if (sc <= tq && m < re) {
                                    if (pos < CRAG_CAND_CAP) cq[pos] = ((unsigned long long)crag_fkey(sc) << 32) | (unsigned long long)m;
                                    ++pos;
                                }
                            }
                        }
                    }
                }
                drained = true;  // vmcnt counts these stores and atomics too, out of order with the DMA loads
            }
        } else {
            // a wave without queries still moves its share of every half-tile and keeps both rendezvous
            if (ex3) {
                issue(t1, 1, b3);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QS_DMA_PER_HALF) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (ex4) issue(t2, 0, b4);
            drained = !(ex3 && ex4);
        }
        // half-tile h + 2 (the next tile's part 0) must have landed in every wave's view: each wave retires all
        // but the two half-tiles it issued after that one (h + 3 and h + 4), then the rendezvous
        if (drained) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QS_DMA_PER_HALF) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        cur ^= 2;
    }
}

template <int MODE, bool NT>
int launch_one(const MfmaArgs& a, int grid, int nqg, hipStream_t s) {
    const size_t lds = (size_t)QS_NBUF * QS_HALF + (size_t)QS_NBUF * QS_WAVES * QS_XN_BYTES + 256;
    static std::atomic<uint64_t> lds_ok;
    crag_allow_dynamic_lds((const void*)k_scan_qsplit<MODE, NT>, lds, lds_ok);
    hipLaunchKernelGGL((k_scan_qsplit<MODE, NT>), dim3(grid), dim3(64 * QS_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                       a.row_begin, a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau,
                       a.sample, a.sample_stride, a.cand, a.cnt);
    return (int)hipGetLastError();
}

int launch_mode(const MfmaArgs& a, int mode, int grid, int nqg, hipStream_t s) {
    if (mode == 0) return launch_one<0, false>(a, grid, nqg, s);
#ifdef CRAG_QSPLIT_NO_NT
    return launch_one<1, false>(a, grid, nqg, s);
#else
    return nqg == 1 ? launch_one<1, true>(a, grid, nqg, s) : launch_one<1, false>(a, grid, nqg, s);
#endif
}

}  // namespace

bool crag_qsplit_supported(int dim_pad) { return dim_pad == QS_D; }

// sample layout of this kernel: sample[row * sample_stride + query]  (row-major)
int crag_launch_scan_qsplit(const MfmaArgs& a, int mode, int n_cus, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.n_queries <= 0) return 0;
    if (a.row_begin % QS_TM || a.dim_pad != QS_D) return -1;
    const int nqg = (a.n_queries + QS_QPB - 1) / QS_QPB;
    if (nqg > 32) return -1;
    int streams = (n_cus / 8) / nqg;  // row-tile streams per XCD
    if (streams < 1) streams = 1;
    const long long n_tiles = (a.row_end - a.row_begin + QS_TM - 1) / QS_TM;
    while (streams > 1 && (long long)streams * 8 > n_tiles) --streams;
    return launch_mode(a, mode, 8 * streams * nqg, nqg, s);
}
