// K2 (float32 query-stationary form): MFMA scan for large query batches over float32 rows
// (dim_pad in {128, 192, 256, 384} -- 384 is the reference's MiniLM embedding), gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// The one-block query-stationary dataflow (crag_scan_qreg2.hip is the two-block form): four waves (one per SIMD) x 32 stationary queries, corpus tiles of
// 32 rows streamed HBM -> LDS by LDS-DMA through a ring of three buffers, fused threshold filter with the
// query on the lane -- on v_mfma_f32_32x32x2_f32 (two k per instruction, 16 passes: ~157 TFLOP/s peak), so
// the kernel is MFMA-bound at about 2 TB/s-equivalent for up to 128 queries per pass over the corpus, where
// the GEMV needs one pass per 8 queries.  The per-tile statement is generated (crag_qf32_kloop.inc,
// gen_qf32_kloop.py holds the k-slot assignment and the register contract); query registers are fixed AGPRs
// outside the compiler's allocation, checked on the emitted ISA by build.py.  Appends: flat buffer +
// returning atomic.
//
//   MODE 0 (sample): ordered-float score keys to sample[row][Bpad]  (row-major, coalesced)
//   MODE 1 (filter): s <= tau -> append (score key << 32 | row) to the query's candidate buffer
#include "crag_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int QF_TM = 32;      // corpus rows per tile
constexpr int QF_WAVES = 4;    // waves per workgroup: one per SIMD
constexpr int QF_QPB = 32 * QF_WAVES;
constexpr int QF_NSTAGE = 3;
constexpr int QF_XN_BYTES = 256;

struct Qf32Epi {
    float c, w;       // score = c * (q . x) + w * |x|^2   (l2: -2, 1; ip: -1, 0)
    unsigned xaddr;   // LDS address of the lane's |x|^2 run
};
template <int PPW>
struct Qf32Dma {
    unsigned voff[PPW / 2];        // per-lane byte offset of this wave's piece i inside a 16-row half of the tile
    const unsigned char* tile;     // wave-uniform: first byte of the tile in the row matrix
    const unsigned char* tile_hi;  // ... of the tile's rows 16..31
    unsigned lds_dst;              // LDS byte address of this wave's piece 0 in the target buffer
    unsigned xoff;                 // per-lane byte offset into the |x|^2 column (lane * 4)
    const float* xn;               // |x|^2 of the tile's first row
    unsigned xn_dst;               // LDS byte address of this wave's |x|^2 slot
};

#include "crag_qf32_kloop.inc"

template <int DP, int MODE>
__global__ __launch_bounds__(64 * QF_WAVES, 1) void k_scan_qf32(const float* __restrict__ rows, const float* __restrict__ xnorm,
                                                                long long row_begin, long long row_end,
                                                                const float* __restrict__ queries, int n_queries, int nqg,
                                                                int metric, const float* __restrict__ tau,
                                                                uint32_t* __restrict__ sample, long long sample_stride,
                                                                unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt) {
    constexpr int ROWB = DP * 4;               // bytes per row (a multiple of 256)
    constexpr int C = ROWB / 16;               // 16-byte chunks per row
    constexpr int STAGE = QF_TM * ROWB;        // bytes per LDS tile buffer
    constexpr int PPW = ROWB / 128;            // 1-KiB DMA pieces per wave and tile
    constexpr int HP = PPW / 2;                // ... and 16-row half
    constexpr int HALF = 16 * ROWB;
    constexpr int XN_BASE = QF_NSTAGE * STAGE; // |x|^2 slots: [stage][wave][64 floats]
    static_assert(ROWB % 256 == 0 && PPW % 2 == 0, "row geometry");
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];  // the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // workgroup -> (query group, row-tile stream); workgroups with equal blockIdx % 8 share an XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int qg = j % nqg;
    const int stream = j / nqg;
    const int n_streams = (int)(gridDim.x >> 3) / nqg;
    const long long n_tiles = (row_end + QF_TM - 1) / QF_TM;
    const long long t_step = (long long)n_streams * 8;
    long long t = row_begin / QF_TM + (long long)stream * 8 + xcd;  // row_begin: a multiple of the tile height

    const int q_base = qg * QF_QPB + wave * 32;
    const bool wave_active = q_base < n_queries;  // wave-uniform
    const int my_q = q_base + fr;
    const bool q_ok = my_q < n_queries;
    // stationary B operand: register 2c + m = q[4c + 2 fh + m]  (fixed AGPRs, see gen_qf32_kloop.py)
    Qf32Load<DP>::run(queries + (long long)(q_ok ? my_q : n_queries - 1) * DP + 2 * fh);
    float tq = -__builtin_inff();
    if (MODE == 1 && q_ok) tq = tau[my_q];

    // LDS-DMA of one tile, as crag_scan_qreg2.hip: linear 16-byte position P holds global chunk
    // (row P / C, chunk (P % C) ^ (row & 15)); a wave takes pieces wave + WAVES*i of each 16-row half
    unsigned voff[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int P = (wave + QF_WAVES * i) * 64 + lane;
        const int r = P / C, p = P - r * C;
        voff[i] = (unsigned)(r * C + (p ^ (r & 15))) * 16u;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto issue = [&](long long tile, int buf) {
        const long long row0 = tile * QF_TM;
        const unsigned char* tb = (const unsigned char*)rows + row0 * (long long)ROWB;
        const bool whole = row0 + QF_TM <= row_end;
        const unsigned r_last = (unsigned)(row_end - 1 - row0);  // ragged tile only: its last valid row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < HP; ++i) {
                unsigned off = voff[i] + h * HALF;
                if (!whole) {  // rows past the end re-read the last row (never past the matrix)
                    asm volatile("" : "+v"(off));
                    const unsigned r = off / ROWB;
                    if (r > r_last) off -= (r - r_last) * ROWB;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + off),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * STAGE + h * HALF + (wave + QF_WAVES * i) * 1024),
                                                 16, 0, 0);
            }
        }
        const float* gx = xnorm + row0 + lane;  // xnorm is padded by >= 256 floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                         (__attribute__((address_space(3))) void*)(lds + XN_BASE + (buf * QF_WAVES + wave) * QF_XN_BYTES),
                                         4, 0, 0);
    };

    // fragment address of chunk 0 in buffer 0: row fr, swizzle folded in, 8 bytes per k-half
    const unsigned a_lane = lds_base + fr * ROWB + ((fr & 15) << 4) + 8 * fh;
    const unsigned x_lane = lds_base + XN_BASE + wave * QF_XN_BYTES + 16 * fh;
    Qf32Epi e;
    e.c = metric == CRAG_L2 ? -2.f : -1.f;
    e.w = metric == CRAG_L2 ? 1.f : 0.f;

    // prologue: two tiles in flight, wait for the first
    int cur = 0;
    if (t < n_tiles) issue(t, 0);
    if (t + t_step < n_tiles) {
        issue(t + t_step, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    for (; t < n_tiles; t += t_step) {
        const long long nt2 = t + 2 * t_step;
        const bool more = nt2 < n_tiles;  // workgroup-uniform
        int nxt2 = cur + 2;
        if (nxt2 >= QF_NSTAGE) nxt2 -= QF_NSTAGE;
        const bool dma_in_loop = more && wave_active && (nt2 + 1) * QF_TM <= row_end;
        if (more && !dma_in_loop) issue(nt2, nxt2);
        bool drained = !more;
        if (wave_active) {
            const unsigned a0 = a_lane + cur * STAGE;
            e.xaddr = x_lane + cur * (QF_WAVES * QF_XN_BYTES);
            f32x16_t acc, xs;  // xs = w * |x|^2 of this lane's 16 rows (rows 8 g + 4 fh + e)
            float mn;
            if (dma_in_loop) {
                Qf32Dma<PPW> d;
#pragma unroll
                for (int i = 0; i < HP; ++i) d.voff[i] = voff[i];
                d.tile = (const unsigned char*)rows + nt2 * (long long)STAGE;
                d.tile_hi = d.tile + HALF;
                d.lds_dst = lds_base + nxt2 * STAGE + wave * 1024;
                d.xoff = lane * 4;
                d.xn = xnorm + nt2 * QF_TM;
                d.xn_dst = lds_base + XN_BASE + (nxt2 * QF_WAVES + wave) * QF_XN_BYTES;
                Qf32Tile<DP, true>::run(acc, xs, mn, a0, e, d);
            } else {
                Qf32Tile<DP, false>::run(acc, xs, mn, a0, e);
            }
            const unsigned mb = (unsigned)t * QF_TM + 4 * fh, re = (unsigned)row_end;  // rows per collection < 2^32
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned m = mb + 8 * (r >> 2) + (r & 3);
                    if (q_ok && m < re) sample[m * sample_stride + my_q] = crag_fkey(fmaf(e.c, acc[r], xs[r]));
                }
                drained = true;
            } else if (__any(mn <= tq)) {
                // one returning atomic per lane (= per query) for all of its passing rows
                int npass = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned m = mb + 8 * (r >> 2) + (r & 3);
                    npass += (fmaf(e.c, acc[r], xs[r]) <= tq && m < re) ? 1 : 0;
                }
                if (npass) {
                    int qa = my_q;
                    asm volatile("" : "+v"(qa));  // address arithmetic stays on this rare path
                    unsigned pos = atomicAdd(&cnt[CRAG_CNT(qa)], (unsigned)npass);
                    unsigned long long* cq = cand + (long long)qa * CRAG_CAND_CAP;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned m = mb + 8 * (r >> 2) + (r & 3);
                        const float s = fmaf(e.c, acc[r], xs[r]);
                        if (s <= tq && m < re) {
                            if (pos < CRAG_CAND_CAP) cq[pos] = ((unsigned long long)crag_fkey(s) << 32) | (unsigned long long)m;
                            ++pos;
                        }
                    }
                }
                drained = true;  // vmcnt counts these stores and atomics too, out of order with the DMA loads
            }
        }
        // tile t + step must have landed in every wave's view before anyone reads it: each wave retires all
        // but the DMA instructions it issued in THIS iteration (they are its youngest), then the barrier
        if (drained) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == QF_NSTAGE ? 0 : cur + 1;
    }
}

template <int DP>
int launch_dp(const MfmaArgs& a, int mode, int grid, int nqg, hipStream_t s) {
    const size_t lds = (size_t)QF_NSTAGE * QF_TM * DP * 4 + (size_t)QF_NSTAGE * QF_WAVES * QF_XN_BYTES;
    static std::atomic<uint64_t> lds_ok[2];
    crag_allow_dynamic_lds(mode == 0 ? (const void*)k_scan_qf32<DP, 0> : (const void*)k_scan_qf32<DP, 1>, lds, lds_ok[mode]);
    if (mode == 0)
        hipLaunchKernelGGL((k_scan_qf32<DP, 0>), dim3(grid), dim3(64 * QF_WAVES), lds, s, (const float*)a.rows, a.xnorm, a.row_begin,
                           a.row_end, (const float*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample, a.sample_stride,
                           a.cand, a.cnt);
    else
        hipLaunchKernelGGL((k_scan_qf32<DP, 1>), dim3(grid), dim3(64 * QF_WAVES), lds, s, (const float*)a.rows, a.xnorm, a.row_begin,
                           a.row_end, (const float*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample, a.sample_stride,
                           a.cand, a.cnt);
    return (int)hipGetLastError();
}

}  // namespace

// float32 rows this kernel takes (dim_pad of the collection)
bool crag_qf32_supported(int dim_pad) { return dim_pad == 128 || dim_pad == 192 || dim_pad == 256 || dim_pad == 384; }

// sample layout of this kernel: sample[row * sample_stride + query]  (row-major)
int crag_launch_scan_qf32(const MfmaArgs& a, int mode, int n_cus, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.n_queries <= 0) return 0;
    if (a.row_begin % QF_TM) return -1;
    const int nqg = (a.n_queries + QF_QPB - 1) / QF_QPB;
    if (nqg > 32) return -1;
    int streams = (n_cus / 8) / nqg;  // row-tile streams per XCD
    if (streams < 1) streams = 1;
    const long long n_tiles = (a.row_end - a.row_begin + QF_TM - 1) / QF_TM;
    while (streams > 1 && (long long)streams * 8 > n_tiles) --streams;
    const int grid = 8 * streams * nqg;
    switch (a.dim_pad) {
        case 128: return launch_dp<128>(a, mode, grid, nqg, s);
        case 192: return launch_dp<192>(a, mode, grid, nqg, s);
        case 256: return launch_dp<256>(a, mode, grid, nqg, s);
        case 384: return launch_dp<384>(a, mode, grid, nqg, s);
        default: return -1;
    }
}
