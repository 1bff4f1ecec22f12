// K2 (query-stationary form): MFMA scan for query batches of any size (from 1-5 queries up on the auto path), bf16,
// dim_pad <= 768, gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// Query-stationary dataflow: queries never move -- 64 per wave, 256 per workgroup, held in fixed registers (all 256
// AGPRs + v[128:255] at D = 768) as the MFMA B operand for every k-step; 32-row corpus tiles stream HBM -> LDS by
// LDS-DMA through a ring of three buffers (source-side XOR swizzle, conflict-free ds_read_b128); one wave per SIMD
// (512 registers); fused threshold filter with the query on the lane; segmented, atomic-free candidate buffers;
// pacing between the workgroups that share a tile stream.  Matrix shape v_mfma_f32_16x16x32_bf16: the wave's 32 x 64
// output tile is 2 row blocks x 4 query blocks of 16x16 accumulators.  Round 3 measured it against the 32x32x16 form
// of rounds 1-2 (same tile, LDS image, DMA, registers, flops; profiles/r03_ab_shapes_*.jsonl; git d9d5712 holds both kernels and tools/ab_shapes.py): the
// scan is power-bound at large batches and the chip holds a higher clock on this shape (MI355X_MICROARCH.md, "DVFS
// give-back" item 7; cdna_hip_programming.md section 5.4 rule 28): +5 % with the epilogue unchanged, +11-15 % with the
// score accumulated in place (below), never slower at any batch size or row length -- the 32x32x16 kernel was deleted.
// A query sits on the four lanes fr, fr + 16, fr + 32, fr + 48 (eight rows of the tile each), so a workgroup appends to
// FOUR segments of a query's candidate buffer.  The accumulators ARE the scores: the B operand is the query scaled by
// -2 (l2) / -1 (ip) and the first MFMA of a tile takes C = |x|^2 (k_q16_scale_queries, gen_q16_kloop.py).  The tile
// statements are generated: crag_q16_kloop.inc (gen_q16_kloop.py holds the register contract, build.py checks it on
// the emitted ISA).
//
//   MODE 0 (sample): ordered-float score keys to the TILE-major sample [row / 32][Bpad][row % 32]
//   MODE 1 (filter): s <= tau -> append (score key << 32 | row) to the query's candidate buffer
#include "crag_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8_t;

constexpr int Q16_TM = 32;      // corpus rows per tile
constexpr int Q16_WAVES = 4;    // waves per workgroup: one per SIMD
constexpr int Q16_QPW = 64;     // queries per wave: four 16-query MFMA blocks
constexpr int Q16_QPB = Q16_WAVES * Q16_QPW;
constexpr int Q16_NSTAGE = 3;   // LDS tile buffers
constexpr int Q16_XN_BYTES = 256;

template <int PPW>
struct Q16Dma {
    unsigned voff[PPW / 2];     // per-lane byte offset of this wave's piece i inside a 16-row half of the tile block
    const unsigned char* tile;  // wave-uniform: first byte of the tile in the row matrix
    const unsigned char* tile_hi;  // wave-uniform: first byte of the tile's rows 16..31
    unsigned lds_dst;           // wave-uniform: LDS byte address of this wave's piece 0 in the target buffer
    unsigned xoff;              // per-lane byte offset into the |x|^2 column (lane * 4)
    const float* xn;            // wave-uniform: |x|^2 of the tile's first row
    unsigned xn_dst;            // wave-uniform: LDS byte address of this wave's |x|^2 slot
};

#ifdef Q16_PROFILE  // diagnostic build only (tools/q16_bench.hip): shader-cycle stamps; the shipped kernel executes none
#define Q16_PROF_NOW() __builtin_readcyclecounter()
#endif
#ifndef Q16_KLOOP_INC  // tools/q16_bench.hip substitutes ablation builds of the statements
#define Q16_KLOOP_INC "crag_q16_kloop.inc"
#endif
#include Q16_KLOOP_INC

// slow path of the filter for query block J: this lane's query has at least one passing row.  The lane appends to
// ITS segment of the query's candidate buffer (four segments per workgroup that scans for the query -- a query sits
// on lanes fr + 16 g, g = 0..3, which hold different rows of the tile -- MfmaArgs in crag_common.h) with a private
// counter: no atomic, no round trip to memory.
template <int J>
__device__ __forceinline__ void q16_append(const f32x16_t& acc0, const f32x16_t& acc1, float tq, unsigned mb, unsigned row_end,
                                           unsigned long long* __restrict__ cq, unsigned seg_cap, unsigned& pos) {
    // called wave-uniformly; four rows at a time (one row block), skipped with one vote when none passes on any lane.
    // The accumulators ARE the scores (gen_q16_kloop.py: pre-scaled queries, C = |x|^2)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        float s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = rb ? acc1[4 * J + e] : acc0[4 * J + e];
        const float mg = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
        if (__any(mg <= tq)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned m = mb + 16 * rb + e;
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
__global__ __launch_bounds__(64 * Q16_WAVES, 1) void k_scan_q16(const unsigned short* __restrict__ rows,
                                                                 const float* __restrict__ xnorm, long long row_begin, long long row_end,
                                                                 const unsigned short* __restrict__ queries /* pre-scaled: -2 q (l2), -q (ip) */, int n_queries,
                                                                 int nqg, int metric, const float* __restrict__ tau,
                                                                 uint32_t* __restrict__ sample, long long sample_stride,
                                                                 unsigned long long* __restrict__ cand,
                                                                 uint32_t* __restrict__ segcnt, int n_seg, int seg_cap, long long cand_stride,
                                                                 uint32_t* __restrict__ progress, int ahead_max) {
    constexpr int C = KS * 2;                  // 16-byte chunks per row
    constexpr int D = KS * 16;                 // elements per row
    constexpr int ROWB = C * 16;               // bytes per row (a multiple of 256)
    constexpr int STAGE = Q16_TM * ROWB;        // bytes per LDS tile buffer
    constexpr int PIECES = Q16_TM * C / 64;     // 1-KiB DMA pieces per tile
    constexpr int PPW = PIECES / Q16_WAVES;     // pieces per wave
    constexpr int XN_BASE = Q16_NSTAGE * STAGE; // |x|^2 slots: [stage][wave][64 floats]
    constexpr int ZERO_BASE = XN_BASE + Q16_NSTAGE * Q16_WAVES * Q16_XN_BYTES;  // 256 zero bytes: the "|x|^2" of the ip metric
    static_assert(PIECES % Q16_WAVES == 0, "pieces must split evenly over the waves");
    static_assert(ROWB % 256 == 0, "the XOR address form needs 256-byte rows");
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];  // the ONLY LDS object
#ifdef Q16_PROFILE
    const unsigned long long prof_entry = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;  // query column of every 16-query block; row quarter

    // workgroup -> (query group, row-tile stream); workgroups with equal blockIdx % 8 share an XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int qg = j % nqg;
    const int stream = j / nqg;
    const int n_streams = (int)(gridDim.x >> 3) / nqg;
    const long long n_tiles = (row_end + Q16_TM - 1) / Q16_TM;
    const long long t_step = (long long)n_streams * 8;
    long long t = row_begin / Q16_TM + (long long)stream * 8 + xcd;  // row_begin: a multiple of the tile height

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
    const unsigned ahead = (unsigned)ahead_max & 0xffu, pace_mask = ((unsigned)ahead_max >> 8) & 0xffu;  // MfmaArgs::ahead_max
    u32x8_t pv;

    const int q_base = qg * Q16_QPB + wave * Q16_QPW;
    const bool wave_active = q_base < n_queries;  // wave-uniform
    int qn[4];
    bool ok[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        qn[jb] = q_base + 16 * jb + fr;
        ok[jb] = qn[jb] < n_queries;
    }
    // stationary B operands: this lane's four queries, k = 32 s + 8 fg + [0,8)
    // (they live in fixed registers that only the generated statements name, see gen_q16_kloop.py)
    Q16Load<KS>::run(queries + (long long)(ok[0] ? qn[0] : n_queries - 1) * D + 8 * fg,
                     queries + (long long)(ok[1] ? qn[1] : n_queries - 1) * D + 8 * fg,
                     queries + (long long)(ok[2] ? qn[2] : n_queries - 1) * D + 8 * fg,
                     queries + (long long)(ok[3] ? qn[3] : n_queries - 1) * D + 8 * fg);
    float tq[4];
    unsigned pos[4];
    // this workgroup's segments of its queries' candidate buffers (one per row quarter), the lane's private counters
    const int seg = (stream * 8 + xcd) * 4 + fg;  // < n_seg
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        // a lane without a query holds a COPY of the batch's last query (the clamped load above): its threshold is a NaN,
        // which no score is at or below -- -inf would let the scores of a query with an infinite component through,
        // and the append would land behind the last query's buffer
        tq[jb] = __builtin_nanf("");
        pos[jb] = 0;
        if (MODE == 1 && ok[jb]) {
            tq[jb] = tau[qn[jb]];
            pos[jb] = segcnt[(long long)qn[jb] * n_seg + seg];
        }
    }

    // LDS-DMA of one tile: 1-KiB pieces of the lane-linear image; linear 16-byte position P holds global
    // chunk (row P / C, chunk (P % C) ^ (row & 15)).  A wave takes pieces wave + WAVES*i of each 16-row half
    // of the tile: rows r and r + 16 swizzle alike, so both halves share one per-lane offset and differ in
    // the scalar base.  Plus this wave's copy of |x|^2.  Every wave issues exactly PPW + 1 DMA instructions
    // per tile (the counted waits rely on it).
    constexpr int HP = PPW / 2;                 // pieces per wave and half
    constexpr int HALF = 16 * ROWB;             // bytes per half tile (= KS / 2 pieces)
    static_assert(PPW % 2 == 0 && (KS / 2) % Q16_WAVES == 0, "half tiles must split evenly over the waves");
    unsigned voff[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int P = (wave + Q16_WAVES * i) * 64 + lane;
        const int r = P / C, p = P - r * C;
        voff[i] = (unsigned)(r * C + (p ^ (r & 15))) * 16u;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto issue = [&](long long tile, int buf) {
        const long long row0 = tile * Q16_TM;
        const unsigned char* tb = (const unsigned char*)rows + row0 * (long long)ROWB;
        const bool whole = row0 + Q16_TM <= row_end;
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
                                                 (__attribute__((address_space(3))) void*)(lds + buf * STAGE + h * HALF + (wave + Q16_WAVES * i) * 1024),
                                                 16, 0, NT ? 2 : 0);
            }
        }
        const float* gx = xnorm + row0 + lane;  // xnorm is padded by >= 256 floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                         (__attribute__((address_space(3))) void*)(lds + XN_BASE + (buf * Q16_WAVES + wave) * Q16_XN_BYTES),
                                         4, 0, NT ? 2 : 0);
    };

    // fragment address of k-step 0 in buffer 0 (see gen_q16_kloop.py), |x|^2 run of this lane in buffer 0
    const unsigned a_lane = lds_base + fr * ROWB + ((fg ^ fr) << 4);
    // C operand of a tile's first MFMAs: the lane's |x|^2 runs (l2) or zeros (ip: the score is -q.x alone)
    const bool l2 = metric == CRAG_L2;
    const unsigned x_lane = lds_base + (l2 ? XN_BASE + wave * Q16_XN_BYTES : ZERO_BASE) + 16 * fg;
    const unsigned x_step = l2 ? Q16_WAVES * Q16_XN_BYTES : 0;
    if (tid < 64) ((unsigned*)(lds + ZERO_BASE))[tid] = 0u;  // visible to every wave behind the prologue's barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... because the store has left this wave before it reaches that barrier

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
    if (prog && wave == 0 && lane == 0) {  // pacing line: the slots no query group owns read "finished"
        for (int i = nqg; i < 8; ++i) __hip_atomic_store(prog + i, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_barrier();

#ifdef Q16_PROFILE  // time in the tile statement, the slow path, the end-of-tile wait and the barrier (shader cycles),
    // and the clock the chip holds: shader cycles per tick of the constant 100 MHz counter over the whole tile loop
    unsigned long long prof_stmt = 0, prof_slow = 0, prof_tiles = 0;
    const unsigned long long prof_t0 = Q16_PROF_NOW(), prof_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // what follows a tile statement: the hit test and, rarely, the slow path (appends) or the sample store
    auto after_tile = [&](const f32x16_t& acc0, const f32x16_t& acc1, float mn0, float mn1, float mn2, float mn3, unsigned tile32) {
        const bool hit0 = mn0 <= tq[0], hit1 = mn1 <= tq[1], hit2 = mn2 <= tq[2], hit3 = mn3 <= tq[3];
        if (MODE == 0 || __any(hit0 || hit1 || hit2 || hit3)) {
            // row numbers fit 32 bits (rows per collection < 2^32, include/coderag_hip.h)
            const unsigned mb = tile32 * Q16_TM + 4 * fg, re = (unsigned)row_end;
            if (MODE == 0) {
                // tile-major sample: key of (row, query) at ((row / 32) * Bpad + query) * 32 + row % 32, so the
                // lane stores its 4-row runs as 16-byte words and k_kth reads whole 128-byte lines
                // (rows past row_end land in the tile's padding and are never read)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    uint4* sp = (uint4*)(sample + ((long long)tile32 * sample_stride + qn[jb]) * 32 + 4 * fg);
                    uint4 k0, k1;
                    k0.x = crag_fkey(acc0[4 * jb + 0]);
                    k0.y = crag_fkey(acc0[4 * jb + 1]);
                    k0.z = crag_fkey(acc0[4 * jb + 2]);
                    k0.w = crag_fkey(acc0[4 * jb + 3]);
                    k1.x = crag_fkey(acc1[4 * jb + 0]);
                    k1.y = crag_fkey(acc1[4 * jb + 1]);
                    k1.z = crag_fkey(acc1[4 * jb + 2]);
                    k1.w = crag_fkey(acc1[4 * jb + 3]);
                    if (ok[jb]) {
                        sp[0] = k0;  // rows 4 fg + [0,4)
                        sp[4] = k1;  // rows 16 + 4 fg + [0,4)
                    }
                }
            } else {
                int qa = qn[0];
                asm volatile("" : "+v"(qa));  // address arithmetic stays on this rare path (not hoisted)
                unsigned long long* cq = cand + (long long)qa * cand_stride + (long long)seg * seg_cap;
                if (__any(hit0)) q16_append<0>(acc0, acc1, tq[0], mb, re, cq, (unsigned)seg_cap, pos[0]);
                if (__any(hit1)) q16_append<1>(acc0, acc1, tq[1], mb, re, cq + 16ll * cand_stride, (unsigned)seg_cap, pos[1]);
                if (__any(hit2)) q16_append<2>(acc0, acc1, tq[2], mb, re, cq + 32ll * cand_stride, (unsigned)seg_cap, pos[2]);
                if (__any(hit3)) q16_append<3>(acc0, acc1, tq[3], mb, re, cq + 48ll * cand_stride, (unsigned)seg_cap, pos[3]);
                // vmcnt counts these stores too (and they may retire out of order with the DMA loads), but the
                // counter is never below the number of outstanding loads and loads retire in order, so the counted
                // wait at the end of the next tile still proves that its successor has landed; it only waits a little longer
            }
        }
    };

    // ---- steady state of a filter launch (the waves that hold queries): Q16Fast statements, one per tile --------
    // Runs while the tile two steps ahead exists and is whole (its DMA rides between the MFMAs).  Per tile the wave
    // executes: [wave 0: pacing], one statement (which ends with the counted vmcnt wait, the barrier and the first reads
    // of the next tile), the hit test, and a dozen scalar instructions of bookkeeping -- every instruction outside the
    // MFMA stream costs this wave (one per SIMD) four cycles, so the general loop's 64-bit index arithmetic is kept out.
    long long n_fast = 0;
    if (MODE == 1 && wave_active) {
        const long long last_whole = row_end / Q16_TM - 1;
        if (t + 2 * t_step <= last_whole) n_fast = (last_whole - t - 2 * t_step) / t_step + 1;
    }
    if (n_fast > 0) {
        auto fast = [&](auto ph) {
            constexpr int PH = decltype(ph)::value;
            // the run starts on ring buffer 0 (it follows the prologue directly)
            Q16First<KS>::run(a_lane, x_lane);
            Q16Dma<PPW> d;
#pragma unroll
            for (int i = 0; i < HP; ++i) d.voff[i] = voff[i];
            d.tile = (const unsigned char*)rows + (t + 2 * t_step) * (long long)STAGE;
            d.xoff = lane * 4;
            d.xn = xnorm + (t + 2 * t_step) * Q16_TM;
            const long long tile_inc = t_step * (long long)STAGE;
            const long long xn_inc = t_step * Q16_TM;
            unsigned tile32 = (unsigned)t;
            const unsigned step32 = (unsigned)t_step;
            // one tile on ring buffer CUR: every address that depends on the buffer is a loop-invariant register (the
            // loop below is unrolled over the three buffers), so a tile costs the pointer increments and nothing else
            auto one = [&](auto cur_tag) {
                constexpr int CUR = decltype(cur_tag)::value;
                constexpr int NXT = (CUR + 1) % Q16_NSTAGE, NXT2 = (CUR + 2) % Q16_NSTAGE;
                if constexpr (PH == 0 && !NT) {
                    // wave 0 of a workgroup that shares its tiles, every (pace_mask + 1)-th tile: publish the tile count, then
                    // wait for the slowest partner.  (Every tile cost the whole workgroup ~80 cycles per tile behind wave 0:
                    // the other waves meet it at the statement's barrier.)
                    if (pacing && (tiles_done & pace_mask) == 0) {
                        if (lane == 0) __hip_atomic_store(prog + qg, tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (tiles_done) {  // the previous statement polled the partners (their counts as of its start)
                            for (int spin = 0;; ++spin) {
                                unsigned slowest = pv[0];
#pragma unroll
                                for (int i = 1; i < 8; ++i) slowest = pv[i] < slowest ? pv[i] : slowest;
                                if (tiles_done <= slowest + ahead) break;
                                if (spin == 96) {  // ~150 us: the partner is not on the chip; stop pacing for good
                                    pacing = false;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(16);
                                asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(pv) : "s"(pp) : "memory");
                            }
                        }
                    }
                    ++tiles_done;
                }
                d.tile_hi = d.tile + HALF;
                d.lds_dst = lds_base + NXT2 * STAGE + wave * 1024;
                d.xn_dst = lds_base + XN_BASE + (NXT2 * Q16_WAVES + wave) * Q16_XN_BYTES;
                f32x16_t acc0, acc1;  // row block rb: element 4 jb + e = SCORE of row 16 rb + 4 fg + e against the query of block jb
                float mn0, mn1, mn2, mn3;
#ifdef Q16_PROFILE
                const unsigned long long prof_s0 = Q16_PROF_NOW();
#endif
                if constexpr (PH == 0 && !NT)
                    Q16Fast<KS, 0, false>::run(acc0, acc1, mn0, mn1, mn2, mn3, a_lane + NXT * STAGE, x_lane + NXT * x_step, d, pp, pv);
                else
                    Q16Fast<KS, PH, NT>::run(acc0, acc1, mn0, mn1, mn2, mn3, a_lane + NXT * STAGE, x_lane + NXT * x_step, d);
#ifdef Q16_PROFILE
                const unsigned long long prof_s1 = Q16_PROF_NOW();
#endif
                after_tile(acc0, acc1, mn0, mn1, mn2, mn3, tile32);
#ifdef Q16_PROFILE
                prof_stmt += prof_s1 - prof_s0;
                prof_slow += Q16_PROF_NOW() - prof_s1;
                ++prof_tiles;
#endif
                d.tile += tile_inc;
                d.xn += xn_inc;
                tile32 += step32;
            };
            const unsigned rem = (unsigned)(n_fast % Q16_NSTAGE);  // n_fast < 2^27 tiles
            for (unsigned left = (unsigned)(n_fast / Q16_NSTAGE); left; --left) {
                one(std::integral_constant<int, 0>{});
                one(std::integral_constant<int, 1>{});
                one(std::integral_constant<int, 2>{});
            }
            if (rem >= 1) one(std::integral_constant<int, 0>{});
            if (rem >= 2) one(std::integral_constant<int, 1>{});
            cur = (int)rem;
        };
        switch (wave) {  // the DMA slots of the statement are staggered by wave
            case 0: fast(std::integral_constant<int, 0>{}); break;
            case 1: fast(std::integral_constant<int, 1>{}); break;
            case 2: fast(std::integral_constant<int, 2>{}); break;
            default: fast(std::integral_constant<int, 3>{}); break;
        }
        t += n_fast * t_step;
    }

    // ---- general path: sample launches, the last tiles of a stream, ragged tiles, waves without queries -----------
    // (a wave without queries runs ALL its tiles here: DMA up front, wait, barrier -- one barrier per tile, as the
    // statements of the other waves execute)
    for (; t < n_tiles; t += t_step) {
        const long long nt2 = t + 2 * t_step;
        const bool more = nt2 < n_tiles;  // workgroup-uniform
        int nxt2 = cur + 2;
        if (nxt2 >= Q16_NSTAGE) nxt2 -= Q16_NSTAGE;
        // prefetch tile t + 2 steps into the buffer of tile t - step (every wave left it before the last barrier)
        if (more) issue(nt2, nxt2);
        if (wave_active) {
            f32x16_t acc0, acc1;
            float mn0, mn1, mn2, mn3;
            Q16KLoop<KS>::run(acc0, acc1, mn0, mn1, mn2, mn3, a_lane + cur * STAGE, x_lane + cur * x_step);
            after_tile(acc0, acc1, mn0, mn1, mn2, mn3, (unsigned)t);
        }
        // tile t + step must have landed in every wave's view before anyone reads it: each wave retires all
        // but the DMA instructions it issued in THIS iteration (they are its youngest), then the barrier
        if (more) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == Q16_NSTAGE ? 0 : cur + 1;
    }
    if (MODE == 1 && wave_active) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
            if (ok[jb]) segcnt[(long long)qn[jb] * n_seg + seg] = pos[jb];
    }
    if (prog && wave == 0 && lane == 0) __hip_atomic_store(prog + qg, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef Q16_PROFILE
    if (lane == 0) {  // eight 64-bit words per wave behind the segment counters; nothing else reads them
        unsigned long long* pc = (unsigned long long*)(segcnt + Q16_PROFILE) + ((long long)blockIdx.x * Q16_WAVES + wave) * 8;
        pc[0] = Q16_PROF_NOW() - prof_t0;
        pc[1] = __builtin_amdgcn_s_memrealtime() - prof_r0;
        pc[2] = prof_stmt;
        pc[3] = prof_slow;
        pc[4] = prof_entry;  // absolute 100 MHz stamps: kernel entry, end of the prologue
        pc[5] = prof_r0;
        pc[6] = prof_tiles;
        pc[7] = wave_active ? 1 : 0;
    }
#endif
}

// the scan's B operand: queries scaled by -2 (l2) or -1 (ip), exact in bf16 (NaN stays NaN), so that the accumulators
// are the scores.  Once per search, before the cascade; k_finalize keeps re-scoring from the caller's queries.
__global__ __launch_bounds__(256) void k_q16_scale_queries(const unsigned short* __restrict__ q, unsigned short* __restrict__ out,
                                                           long long n, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f = __builtin_bit_cast(float, (uint32_t)q[i] << 16) * scale;  // a power of two times a bf16: exact
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    out[i] = f != f ? (unsigned short)0x7fc0u : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int KS, int MODE, bool NT>
int launch_one(const MfmaArgs& a, int grid, int nqg, size_t lds, hipStream_t s) {
    static std::atomic<uint64_t> lds_ok;
    crag_allow_dynamic_lds((const void*)k_scan_q16<KS, MODE, NT>, lds, lds_ok);
    hipLaunchKernelGGL((k_scan_q16<KS, MODE, NT>), dim3(grid), dim3(64 * Q16_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                       a.row_begin, a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample,
                       a.sample_stride, a.cand, a.segcnt, a.n_seg, a.seg_cap, a.cand_stride, a.progress, a.ahead_max);
    return (int)hipGetLastError();
}

template <int KS>
int launch_ks(const MfmaArgs& a, int mode, int grid, int nqg, hipStream_t s) {
    const size_t lds = (size_t)Q16_NSTAGE * Q16_TM * KS * 2 * 16 + (size_t)Q16_NSTAGE * Q16_WAVES * Q16_XN_BYTES + 256;
    if (mode == 0) return launch_one<KS, 0, false>(a, grid, nqg, lds, s);  // the sample: a few thousand rows
    return nqg == 1 ? launch_one<KS, 1, true>(a, grid, nqg, lds, s) : launch_one<KS, 1, false>(a, grid, nqg, lds, s);
}

}  // namespace

bool crag_q16_supported(int dim_pad) { return dim_pad == 256 || dim_pad == 384 || dim_pad == 512 || dim_pad == 768; }

// (the selection kernels gather at most 1024 segments per query -- 4 x 8 x 32 streams on the 256 CUs of an MI355X --
// so a larger device is used as 256 CUs)
int crag_q16_segments(int n_queries, int n_cus) {
    n_cus = n_cus > 256 ? 256 : n_cus;
    const int nqg = (n_queries + Q16_QPB - 1) / Q16_QPB;
    const int streams = (n_cus / 8) / nqg;
    return 4 * 8 * (streams < 1 ? 1 : streams);  // four per workgroup: one per row quarter (lane >> 4)
}

// sample layout of this kernel: tile-major, key of (row, query) at ((row / 32) * sample_stride + query) * 32 + row % 32
int crag_launch_scan_q16(const MfmaArgs& a, int mode, int n_cus, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.n_queries <= 0) return 0;
    if (mode == 1 && (a.n_seg != crag_q16_segments(a.n_queries, n_cus) || a.seg_cap < 1 ||
                      (long long)a.n_seg * a.seg_cap != a.cand_stride || !a.segcnt))
        return -1;
    if (a.row_begin % Q16_TM) return -1;
    const int nqg = (a.n_queries + Q16_QPB - 1) / Q16_QPB;
    if (nqg > 32) return -1;
    n_cus = n_cus > 256 ? 256 : n_cus;
    int streams = (n_cus / 8) / nqg;  // row-tile streams per XCD
    if (streams < 1) streams = 1;
    const long long n_tiles = (a.row_end - a.row_begin + Q16_TM - 1) / Q16_TM;
    while (streams > 1 && (long long)streams * 8 > n_tiles) --streams;
    const int grid = 8 * streams * nqg;
#ifdef Q16_ONLY_KS
    return launch_ks<Q16_ONLY_KS>(a, mode, grid, nqg, s);
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

int crag_q16_scale_queries(const void* queries, int n_queries, int dim_pad, int metric, void* out, hipStream_t s) {
    const long long n = (long long)n_queries * dim_pad;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_q16_scale_queries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned short*)queries,
                       (unsigned short*)out, n, metric == CRAG_L2 ? -2.f : -1.f);
    return (int)hipGetLastError();
}
