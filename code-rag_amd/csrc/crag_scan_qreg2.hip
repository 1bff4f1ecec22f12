// K2 (two-block query-stationary form): MFMA scan for query batches of any size (B >= 2-3 on the auto path),
// bf16, dim_pad <= 768, gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// Query-stationary dataflow (queries never move, 32-row corpus tiles stream HBM -> LDS by
// LDS-DMA through a ring of three buffers, one wave per SIMD, fused threshold filter); a wave
// holds TWO 32-query blocks, so a workgroup owns 256 queries and every A fragment read from LDS
// feeds two MFMAs: per MFMA half the LDS-DMA instructions, half the L2 -> LDS bytes and half the
// LDS reads, and the two accumulator chains alternate in the matrix pipe.  The 96 query fragments of
// D = 768 (384 registers) do not fit the 256 AGPRs: 64 live there, 32 in VGPRs (fixed registers outside
// the compiler's allocation, see gen_qreg2_kloop.py and build.py's ISA check), and the accumulators
// are VGPRs so that the filter reads them with plain VALU instructions.  The whole tile (K-loop,
// prefetch of a later tile, the filter's fast path = minimum score of the lane's rows per block) is
// one generated statement, crag_qreg2_kloop.inc (gen_qreg2_kloop.py holds the register contract).
//
//   MODE 0 (sample): ordered-float score keys to the TILE-major sample [row / 32][Bpad][row % 32]
//   MODE 1 (filter): s <= tau -> append (score key << 32 | row) to the query's candidate buffer
#include "crag_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8_t;

constexpr int Q2_TM = 32;      // corpus rows per tile
constexpr int Q2_WAVES = 4;    // waves per workgroup: one per SIMD
constexpr int Q2_QPW = 64;     // queries per wave: two MFMA blocks
constexpr int Q2_QPB = Q2_WAVES * Q2_QPW;
constexpr int Q2_NSTAGE = 3;   // LDS tile buffers
constexpr int Q2_XN_BYTES = 256;

// score = c * (q . x) + w * |x|^2   (l2: -2, 1; ip: -1, 0) and the LDS address of the lane's |x|^2 run
struct Qreg2Epi {
    float c, w;
    unsigned xaddr;
};

template <int PPW>
struct Qreg2Dma {
    unsigned voff[PPW / 2];     // per-lane byte offset of this wave's piece i inside a 16-row half of the tile block
    const unsigned char* tile;  // wave-uniform: first byte of the tile in the row matrix
    const unsigned char* tile_hi;  // wave-uniform: first byte of the tile's rows 16..31
    unsigned lds_dst;           // wave-uniform: LDS byte address of this wave's piece 0 in the target buffer
    unsigned xoff;              // per-lane byte offset into the |x|^2 column (lane * 4)
    const float* xn;            // wave-uniform: |x|^2 of the tile's first row
    unsigned xn_dst;            // wave-uniform: LDS byte address of this wave's |x|^2 slot
};

#ifdef Q2_PROFILE  // -DQ2_PROF_REALTIME: the constant 100 MHz counter instead of shader cycles
#ifdef Q2_PROF_REALTIME
#define Q2_PROF_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define Q2_PROF_NOW() __builtin_readcyclecounter()
#endif
#endif
#ifndef Q2_KLOOP_INC  // tools/qreg_bench.hip substitutes ablation builds of the statements
#define Q2_KLOOP_INC "crag_qreg2_kloop.inc"
#endif
#include Q2_KLOOP_INC

// slow path of the filter for one query block: this lane's query has at least one passing row.  The lane
// appends to ITS segment of the query's candidate buffer (two segments per workgroup that scans for the
// query -- a query sits on lanes fr and fr + 32, which hold different rows of the tile -- MfmaArgs in
// crag_common.h) with a private counter: no atomic, no round trip to memory.
__device__ __forceinline__ void q2_append(const f32x16_t& acc, const f32x16_t& xs, float c, float tq, unsigned mb,
                                          unsigned row_end, unsigned long long* __restrict__ cq, unsigned seg_cap,
                                          unsigned& pos) {
    // called wave-uniformly; four rows at a time, skipped with one vote when none of them passes on any lane
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = fmaf(c, acc[4 * g + e], xs[4 * g + e]);
        const float mg = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
        if (__any(mg <= tq)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned m = mb + 8 * g + e;
                if (s[e] <= tq && m < row_end) {
                    if (pos < seg_cap) cq[pos] = ((unsigned long long)crag_fkey(s[e]) << 32) | (unsigned long long)m;
                    ++pos;
                }
            }
        }
    }
}

// NT: this launch has one query group, so every tile is read by exactly one workgroup: its DMA loads carry the
// nontemporal bit (the stream does not displace the L2's contents; the batch-1 GEMV gains 12 % of bandwidth from
// the same bit).  With several query groups per tile stream the L2 is what shares the tile: default policy.
template <int KS, int MODE, bool NT>
__global__ __launch_bounds__(64 * Q2_WAVES, 1) void k_scan_qreg2(const unsigned short* __restrict__ rows,
                                                                 const float* __restrict__ xnorm, long long row_begin, long long row_end,
                                                                 const unsigned short* __restrict__ queries, int n_queries,
                                                                 int nqg, int metric, const float* __restrict__ tau,
                                                                 uint32_t* __restrict__ sample, long long sample_stride,
                                                                 unsigned long long* __restrict__ cand,
                                                                 uint32_t* __restrict__ segcnt, int n_seg, int seg_cap, long long cand_stride,
                                                                 uint32_t* __restrict__ progress, int ahead_max) {
    constexpr int C = KS * 2;                  // 16-byte chunks per row
    constexpr int D = KS * 16;                 // elements per row
    constexpr int ROWB = C * 16;               // bytes per row (a multiple of 256)
    constexpr int STAGE = Q2_TM * ROWB;        // bytes per LDS tile buffer
    constexpr int PIECES = Q2_TM * C / 64;     // 1-KiB DMA pieces per tile
    constexpr int PPW = PIECES / Q2_WAVES;     // pieces per wave
    constexpr int XN_BASE = Q2_NSTAGE * STAGE; // |x|^2 slots: [stage][wave][64 floats]
    static_assert(PIECES % Q2_WAVES == 0, "pieces must split evenly over the waves");
    static_assert(ROWB % 256 == 0, "the XOR address form needs 256-byte rows");
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];  // the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
#ifdef Q2_PROFILE
    const unsigned long long prof_entry = Q2_PROF_NOW();
#endif

    // workgroup -> (query group, row-tile stream); workgroups with equal blockIdx % 8 share an XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int qg = j % nqg;
    const int stream = j / nqg;
    const int n_streams = (int)(gridDim.x >> 3) / nqg;
    const long long n_tiles = (row_end + Q2_TM - 1) / Q2_TM;
    const long long t_step = (long long)n_streams * 8;
    long long t = row_begin / Q2_TM + (long long)stream * 8 + xcd;  // row_begin: a multiple of the tile height

    // PACING (launches with several query groups, MODE 1).  The nqg workgroups of one (xcd, stream) pair read the
    // same tiles, and the XCD's L2 is what lets the second to the last reader find a tile the first one fetched --
    // as long as they stay within a few tiles of each other (8 streams x 48 KiB tiles share 4 MiB).  Left alone
    // they drift apart (the filter's slow path costs each query group differently) and the re-reads go to the
    // Infinity Cache or HBM.  Each workgroup publishes its tile count (one sc1 store per tile), wave 0 reads the
    // others' counts with one scalar load per tile inside its tile statement, and a workgroup more than ahead_max
    // tiles ahead of the slowest waits for it.  Waiting costs nothing: the launch ends with its slowest workgroup.
    // Speed only, never correctness: every wait is bounded (a partner that is not resident yet, or that another
    // process keeps off the chip, ends the pacing for this workgroup), and a finished workgroup reports "infinitely far".
    uint32_t* const prog = (MODE == 1 && !NT && nqg > 1 && progress) ? progress + (long long)(stream * 8 + xcd) * 32 : nullptr;
    bool pacing = prog != nullptr && wave == 0;  // wave-uniform
    // wave 0's statement always loads 32 bytes; without pacing from a line of this workgroup's own (every workgroup
    // polling ONE line with scalar glc loads serialises in its L2 channel: the kernel ran 2.6x longer)
    const unsigned* const pp = prog ? (const unsigned*)prog : (const unsigned*)(queries + (long long)(b % n_queries) * D);
    unsigned tiles_done = 0;
    u32x8_t pv;

    const int q_base = qg * Q2_QPB + wave * Q2_QPW;
    const bool wave_active = q_base < n_queries;  // wave-uniform
    const int q0 = q_base + fr, q1 = q_base + 32 + fr;
    const bool ok0 = q0 < n_queries, ok1 = q1 < n_queries;
    // stationary B operands: this lane's two queries, k = 16 s + 8 fh + [0,8)
    // (they live in fixed registers that only the generated statements name, see gen_qreg2_kloop.py)
    Qreg2Load<KS>::run(queries + (long long)(ok0 ? q0 : n_queries - 1) * D + 8 * fh,
                       queries + (long long)(ok1 ? q1 : n_queries - 1) * D + 8 * fh);
#ifdef Q2_PROFILE
    const unsigned long long prof_loaded = Q2_PROF_NOW();
#endif
    float tq0 = -__builtin_inff(), tq1 = -__builtin_inff();
    if (MODE == 1 && ok0) tq0 = tau[q0];
    if (MODE == 1 && ok1) tq1 = tau[q1];
    // this workgroup's segment of its queries' candidate buffers, and the lane's two private counters
    const int seg = (stream * 8 + xcd) * 2 + fh;  // < n_seg
    unsigned pos0 = 0, pos1 = 0;
    if (MODE == 1 && ok0) pos0 = segcnt[(long long)q0 * n_seg + seg];
    if (MODE == 1 && ok1) pos1 = segcnt[(long long)q1 * n_seg + seg];

    // LDS-DMA of one tile: 1-KiB pieces of the lane-linear image; linear 16-byte position P holds global
    // chunk (row P / C, chunk (P % C) ^ (row & 15)).  A wave takes pieces wave + WAVES*i of each 16-row half
    // of the tile: rows r and r + 16 swizzle alike, so both halves share one per-lane offset and differ in
    // the scalar base.  Plus this wave's copy of |x|^2.  Every wave issues exactly PPW + 1 DMA instructions
    // per tile (the counted waits rely on it).
    constexpr int HP = PPW / 2;                 // pieces per wave and half
    constexpr int HALF = 16 * ROWB;             // bytes per half tile (= KS / 2 pieces)
    static_assert(PPW % 2 == 0 && (KS / 2) % Q2_WAVES == 0, "half tiles must split evenly over the waves");
    unsigned voff[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int P = (wave + Q2_WAVES * i) * 64 + lane;
        const int r = P / C, p = P - r * C;
        voff[i] = (unsigned)(r * C + (p ^ (r & 15))) * 16u;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto issue = [&](long long tile, int buf) {
        const long long row0 = tile * Q2_TM;
        const unsigned char* tb = (const unsigned char*)rows + row0 * (long long)ROWB;
        const bool whole = row0 + Q2_TM <= row_end;
        const unsigned r_last = (unsigned)(row_end - 1 - row0);  // ragged tile only: its last valid row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < HP; ++i) {
                unsigned off = voff[i] + h * HALF;
                if (!whole) {  // ragged last tile: rows past the end re-read the last row (never past the matrix)
                    asm volatile("" : "+v"(off));  // keep the row arithmetic here, not hoisted into live registers
                    const unsigned r = off / ROWB;
                    if (r > r_last) off -= (r - r_last) * ROWB;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + off),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * STAGE + h * HALF + (wave + Q2_WAVES * i) * 1024),
                                                 16, 0, NT ? 2 : 0);
            }
        }
        const float* gx = xnorm + row0 + lane;  // xnorm is padded by >= 256 floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                         (__attribute__((address_space(3))) void*)(lds + XN_BASE + (buf * Q2_WAVES + wave) * Q2_XN_BYTES),
                                         4, 0, NT ? 2 : 0);
    };

    // fragment address of k-step 0 in buffer 0 (see gen_qreg2_kloop.py), |x|^2 run of this lane in buffer 0
    const unsigned a_lane = lds_base + fr * ROWB + ((fh ^ (fr & 15)) << 4);
    const unsigned x_lane = lds_base + XN_BASE + wave * Q2_XN_BYTES + 16 * fh;
    Qreg2Epi e;
    e.c = metric == CRAG_L2 ? -2.f : -1.f;
    e.w = metric == CRAG_L2 ? 1.f : 0.f;

    // prologue: two tiles in flight, wait for the first (issuing them ahead of the query load was tried: the
    // prologue stays ~15 us either way and the whole kernel came out 0.7 % slower)
    int cur = 0;
    if (t < n_tiles) issue(t, 0);
    if (t + t_step < n_tiles) {
        issue(t + t_step, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

#ifdef Q2_PROFILE  // tools/qreg_bench.hip: time in the end-of-tile wait + barrier vs the whole loop (100 MHz ticks)
    unsigned long long prof_wait = 0, prof_bar = 0;
    const unsigned long long prof_t0 = Q2_PROF_NOW();
#endif
    for (; t < n_tiles; t += t_step) {
        const long long nt2 = t + 2 * t_step;
        const bool more = nt2 < n_tiles;  // workgroup-uniform
        int nxt2 = cur + 2;
        if (nxt2 >= Q2_NSTAGE) nxt2 -= Q2_NSTAGE;
        // prefetch tile t + 2 steps into the buffer of tile t - step (every wave left it before the last
        // barrier): between the MFMAs of the K-loop when this wave multiplies and the tile is whole,
        // up front otherwise
        const bool dma_in_loop = more && wave_active && (nt2 + 1) * Q2_TM <= row_end;
        if (more && !dma_in_loop) issue(nt2, nxt2);
        if (wave_active) {
            const unsigned a0 = a_lane + cur * STAGE;
            e.xaddr = x_lane + cur * (Q2_WAVES * Q2_XN_BYTES);
            f32x16_t acc0, acc1, xs;  // xs = w * |x|^2 of this lane's 16 rows (rows 8 g + 4 fh + e)
            float mn0, mn1;
            if (dma_in_loop) {
                Qreg2Dma<PPW> d;
#pragma unroll
                for (int i = 0; i < HP; ++i) d.voff[i] = voff[i];
                d.tile = (const unsigned char*)rows + nt2 * (long long)STAGE;
                d.tile_hi = d.tile + HALF;
                d.lds_dst = lds_base + nxt2 * STAGE + wave * 1024;
                d.xoff = lane * 4;
                d.xn = xnorm + nt2 * Q2_TM;
                d.xn_dst = lds_base + XN_BASE + (nxt2 * Q2_WAVES + wave) * Q2_XN_BYTES;
                switch (wave) {  // the DMA slots of the K-loop are staggered by wave
                    case 0:
                        if constexpr (NT) Qreg2KLoop<KS, true, 0, true>::run(acc0, acc1, xs, mn0, mn1, a0, e, d);
                        else Qreg2KLoop<KS, true, 0, false>::run(acc0, acc1, xs, mn0, mn1, a0, e, d, pp, pv);
                        break;
                    case 1: Qreg2KLoop<KS, true, 1, NT>::run(acc0, acc1, xs, mn0, mn1, a0, e, d); break;
                    case 2: Qreg2KLoop<KS, true, 2, NT>::run(acc0, acc1, xs, mn0, mn1, a0, e, d); break;
                    default: Qreg2KLoop<KS, true, 3, NT>::run(acc0, acc1, xs, mn0, mn1, a0, e, d); break;
                }
            } else {
                Qreg2KLoop<KS, false, 0, false>::run(acc0, acc1, xs, mn0, mn1, a0, e);
            }
            const bool hit0 = mn0 <= tq0, hit1 = mn1 <= tq1;
            if (MODE == 0 || __any(hit0 || hit1)) {
                // row numbers fit 32 bits (rows per collection < 2^32, include/coderag_hip.h)
                const unsigned mb = (unsigned)t * Q2_TM + 4 * fh, re = (unsigned)row_end;
                if (MODE == 0) {
                    // tile-major sample: key of (row, query) at ((row / 32) * Bpad + query) * 32 + row % 32, so the
                    // lane stores its four 4-row runs as 16-byte words and k_kth reads whole 128-byte lines
                    // (rows past row_end land in the tile's padding and are never read)
                    uint4* s0 = (uint4*)(sample + ((long long)t * sample_stride + q0) * 32 + 4 * fh);
                    uint4* s1 = (uint4*)(sample + ((long long)t * sample_stride + q1) * 32 + 4 * fh);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint4 k0, k1;
                        k0.x = crag_fkey(fmaf(e.c, acc0[4 * g + 0], xs[4 * g + 0]));
                        k0.y = crag_fkey(fmaf(e.c, acc0[4 * g + 1], xs[4 * g + 1]));
                        k0.z = crag_fkey(fmaf(e.c, acc0[4 * g + 2], xs[4 * g + 2]));
                        k0.w = crag_fkey(fmaf(e.c, acc0[4 * g + 3], xs[4 * g + 3]));
                        k1.x = crag_fkey(fmaf(e.c, acc1[4 * g + 0], xs[4 * g + 0]));
                        k1.y = crag_fkey(fmaf(e.c, acc1[4 * g + 1], xs[4 * g + 1]));
                        k1.z = crag_fkey(fmaf(e.c, acc1[4 * g + 2], xs[4 * g + 2]));
                        k1.w = crag_fkey(fmaf(e.c, acc1[4 * g + 3], xs[4 * g + 3]));
                        if (ok0) s0[2 * g] = k0;
                        if (ok1) s1[2 * g] = k1;
                    }
                } else {
                    {
                        int qa = q0;
                        asm volatile("" : "+v"(qa));  // address arithmetic stays on this rare path (not hoisted)
                        unsigned long long* cq = cand + (long long)qa * cand_stride + (long long)seg * seg_cap;
                        if (__any(hit0)) q2_append(acc0, xs, e.c, tq0, mb, re, cq, (unsigned)seg_cap, pos0);
                        if (__any(hit1)) q2_append(acc1, xs, e.c, tq1, mb, re, cq + 32ll * cand_stride, (unsigned)seg_cap, pos1);
                    }
                    // vmcnt counts these stores too (and they may retire out of order with the DMA loads), but the
                    // counter is never below the number of outstanding loads and loads retire in order, so the
                    // counted wait below still proves that the next tile has landed; it only waits a little longer
                }
            }
        }
        // tile t + step must have landed in every wave's view before anyone reads it: each wave retires all
        // but the DMA instructions it issued in THIS iteration (they are its youngest), then the barrier
#ifdef Q2_PROFILE
        const unsigned long long prof_a = Q2_PROF_NOW();
#endif
        if (more) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (MODE == 1 && !NT) {
            if (pacing) {  // wave 0 of a workgroup that shares its tiles
                ++tiles_done;
                if (lane == 0) __hip_atomic_store(prog + qg, tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dma_in_loop) {  // this tile's statement polled the partners (their counts as of its start)
                    for (int spin = 0;; ++spin) {
                        unsigned slowest = 0xffffffffu;
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (i < nqg && i != qg) slowest = pv[i] < slowest ? pv[i] : slowest;
                        if (tiles_done <= slowest + ((unsigned)ahead_max & 0xffu)) break;
                        if (spin == 96) {  // ~150 us: the partner is not on the chip; stop pacing for good
                            pacing = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(16);
                        asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(pv) : "s"(pp) : "memory");
                    }
                }
            }
        }
#ifdef Q2_PROFILE
        const unsigned long long prof_b = Q2_PROF_NOW();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef Q2_PROFILE
        const unsigned long long prof_c = Q2_PROF_NOW();
        prof_wait += prof_b - prof_a;
        prof_bar += prof_c - prof_b;
#endif
        cur = cur + 1 == Q2_NSTAGE ? 0 : cur + 1;
    }
    if (MODE == 1 && wave_active) {
        if (ok0) segcnt[(long long)q0 * n_seg + seg] = pos0;
        if (ok1) segcnt[(long long)q1 * n_seg + seg] = pos1;
    }
    if (prog && wave == 0 && lane == 0) __hip_atomic_store(prog + qg, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef Q2_PROFILE
    if (lane == 0) {
        uint32_t* pc = segcnt + Q2_PROFILE;  // tools/qreg_bench.hip: counters behind the segment counters
        atomicAdd(&pc[0], (unsigned)prof_wait);
        atomicAdd(&pc[1], (unsigned)prof_bar);
        atomicAdd(&pc[2], (unsigned)(Q2_PROF_NOW() - prof_t0));
        atomicAdd(&pc[3], 1u);
        atomicAdd(&pc[4], (unsigned)(prof_loaded - prof_entry));
        atomicAdd(&pc[5], (unsigned)(prof_t0 - prof_loaded));
        atomicMax(&pc[6], ~(unsigned)prof_entry);        // earliest entry (inverted)
        atomicMax(&pc[7], (unsigned)prof_entry);         // latest entry
        atomicMax(&pc[8], (unsigned)Q2_PROF_NOW());      // latest exit
    }
#endif
}

template <int KS, int MODE, bool NT>
int launch_one(const MfmaArgs& a, int grid, int nqg, size_t lds, hipStream_t s) {
    static std::atomic<uint64_t> lds_ok;
    crag_allow_dynamic_lds((const void*)k_scan_qreg2<KS, MODE, NT>, lds, lds_ok);
    hipLaunchKernelGGL((k_scan_qreg2<KS, MODE, NT>), dim3(grid), dim3(64 * Q2_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                       a.row_begin, a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample,
                       a.sample_stride, a.cand, a.segcnt, a.n_seg, a.seg_cap, a.cand_stride, a.progress, a.ahead_max);
    return (int)hipGetLastError();
}

template <int KS>
int launch_ks(const MfmaArgs& a, int mode, int grid, int nqg, hipStream_t s) {
    const size_t lds = (size_t)Q2_NSTAGE * Q2_TM * KS * 2 * 16 + (size_t)Q2_NSTAGE * Q2_WAVES * Q2_XN_BYTES;
    if (mode == 0) return launch_one<KS, 0, false>(a, grid, nqg, lds, s);  // the sample: a few thousand rows
    return nqg == 1 ? launch_one<KS, 1, true>(a, grid, nqg, lds, s) : launch_one<KS, 1, false>(a, grid, nqg, lds, s);
}

}  // namespace

bool crag_qreg2_supported(int dim_pad) { return dim_pad == 256 || dim_pad == 384 || dim_pad == 512 || dim_pad == 768; }

int crag_qreg2_segments(int n_queries, int n_cus) {
    const int nqg = (n_queries + Q2_QPB - 1) / Q2_QPB;
    const int streams = (n_cus / 8) / nqg;
    return 2 * 8 * (streams < 1 ? 1 : streams);  // two per workgroup: one per half-wave (row half fh)
}

// sample layout of this kernel: tile-major, key of (row, query) at ((row / 32) * sample_stride + query) * 32 + row % 32
int crag_launch_scan_qreg2(const MfmaArgs& a, int mode, int n_cus, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.n_queries <= 0) return 0;
    if (mode == 1 && (a.n_seg != crag_qreg2_segments(a.n_queries, n_cus) || a.seg_cap < 1 ||
                      (long long)a.n_seg * a.seg_cap != a.cand_stride || !a.segcnt))
        return -1;
    if (a.row_begin % Q2_TM) return -1;
    const int nqg = (a.n_queries + Q2_QPB - 1) / Q2_QPB;
    if (nqg > 32) return -1;
    int streams = (n_cus / 8) / nqg;  // row-tile streams per XCD
    if (streams < 1) streams = 1;
    const long long n_tiles = (a.row_end - a.row_begin + Q2_TM - 1) / Q2_TM;
    while (streams > 1 && (long long)streams * 8 > n_tiles) --streams;
    const int grid = 8 * streams * nqg;
#ifdef Q2_ONLY_KS
    return launch_ks<Q2_ONLY_KS>(a, mode, grid, nqg, s);
#else
    switch (a.dim_pad) {
        case 256: return launch_ks<16>(a, mode, grid, nqg, s);
        case 384: return launch_ks<24>(a, mode, grid, nqg, s);
        case 512: return launch_ks<32>(a, mode, grid, nqg, s);
        case 768: return launch_ks<48>(a, mode, grid, nqg, s);
        default: return -1;
    }
#endif
}
