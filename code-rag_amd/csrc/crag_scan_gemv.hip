// K1: HBM-streaming query-by-corpus scan for small query batches (GEMV regime), gfx950.
//
// Replaces the k-NN half of `col.query(query_texts=[q], n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 with an exact brute-force scan.
//
// Data layout: corpus rows row-major, each row C 16-byte chunks (bf16: 8 elements, fp32: 4).
// A wave streams a contiguous "super-tile" of RT = 64/NQ rows with fully coalesced 1-KiB
// wave loads (lane L of load n reads flat chunk 64n+L).  Each lane accumulates partial dot
// products for the row(s) its chunks fall into; a transposing butterfly (permlane32/16 swaps,
// then DPP row_ror / row_half_mirror / quad_perm) reduces the 64 (row, query) partial vectors
// so that lane l ends up holding the full inner product of row l/NQ with query l%NQ.
// Epilogue, fused (the score vector never goes to HBM in filter mode):
//   mode 0 (sample): write the ordered-float key of the score for rows [0, n_sample)
//   mode 1 (filter): compare against the per-query threshold tau, ballot, and append the rare
//                    survivors (score key << 32 | row) to the per-query candidate buffer.
// score s = |x|^2 - 2 q.x   (l2; |q|^2 is constant per query)   or   -q.x   (ip).
#include "crag_common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b.   v_permlane16_swap: odd 16-lane rows
// of a <-> even rows of b.  Inline asm, because hipcc (ROCm 7.2) folds the builtin's second
// result onto the first (__builtin_amdgcn_permlane32_swap(...)[1] reads vdst, seen on hardware).
// The leading wait states are this code's job -- the compiler inserts no hazard nops inside an asm statement:
//   * a VALU write followed by a permlane swap that reads the register needs 2 wait states;
//   * the first stage's inputs are accumulators that v_dot2c_f32_bf16 wrote, and on gfx940+ a DOT result needs
//     3 wait states before a different VALU instruction reads it and 4 before one overwrites it (the swap does
//     both).  hipcc schedules each swap right behind the last dot of its two accumulators; with only two wait
//     states the LAST pair (rows 31 and 63 of a super-tile) read a stale accumulator whenever nothing else sat
//     in between -- found by tools/fuzz_more.py on 1536-element bf16 rows, one query (round 2; the bug was in
//     round 1's kernel too, where no test reached that variant).  s_nop 4 = five wait states covers all three.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm("s_nop 4\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    asm("s_nop 4\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int DT>
__device__ __forceinline__ float dot_chunk(uint4 x, uint4 q, float acc) {
    if constexpr (DT == CRAG_BF16) {
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x.x), __builtin_bit_cast(bf16x2_t, q.x), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x.y), __builtin_bit_cast(bf16x2_t, q.y), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x.z), __builtin_bit_cast(bf16x2_t, q.z), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x.w), __builtin_bit_cast(bf16x2_t, q.w), acc, false);
    } else {
        acc = fmaf(__builtin_bit_cast(float, x.x), __builtin_bit_cast(float, q.x), acc);
        acc = fmaf(__builtin_bit_cast(float, x.y), __builtin_bit_cast(float, q.y), acc);
        acc = fmaf(__builtin_bit_cast(float, x.z), __builtin_bit_cast(float, q.z), acc);
        acc = fmaf(__builtin_bit_cast(float, x.w), __builtin_bit_cast(float, q.w), acc);
    }
    return acc;
}

// 64 per-lane partial vectors v[0..63] -> lane l holds sum over all lanes of v[l].
__device__ __forceinline__ float butterfly64(float (&v)[64], int lane) {
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {  // xor 32: lanes <32 keep value i, lanes >=32 keep value i+32
        float a = v[i], b = v[i + 32];
        permlane32_swap(a, b);
        w[i] = a + b;
    }
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // xor 16: lane row rho=(l>>4) keeps value i + 16*rho
        float a = w[i], b = w[i + 16];
        permlane16_swap(a, b);
        x[i] = a + b;
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // xor 8 inside a row of 16 (row_ror:8)
        float keep = b3 ? x[i + 8] : x[i];
        float send = b3 ? x[i] : x[i + 8];
        y[i] = keep + dpp_mov<0x128>(send);
    }
    float z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // halves of 8 lanes, partner 7-i (row_half_mirror)
        float keep = b2 ? y[i + 4] : y[i];
        float send = b2 ? y[i] : y[i + 4];
        z[i] = keep + dpp_mov<0x141>(send);
    }
    float t[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // xor 2 (quad_perm [2,3,0,1])
        float keep = b1 ? z[i + 2] : z[i];
        float send = b1 ? z[i] : z[i + 2];
        t[i] = keep + dpp_mov<0x4E>(send);
    }
    float keep = b0 ? t[1] : t[0];
    float send = b0 ? t[0] : t[1];
    return keep + dpp_mov<0xB1>(send);  // xor 1 (quad_perm [1,0,3,2])
}

// Corpus rows are read once per pass by one wave: nontemporal loads (global_load_dwordx4 ... nt) keep them from
// displacing what the L2 is for (queries, |x|^2, descriptors).  Same box, one query, 6-10M rows: 256 / 384 / 512 / 768
// bf16 elements per row +9 / +12 / +9 / +12 % bandwidth (6.18 -> 6.92 TB/s at 768), 1024 / 1536 / 2048 elements
// +12 % each (6.0-6.2 -> 6.8-6.95 TB/s).  An earlier measurement had the long rows LOSE 16 % with nt; that was
// taken on a kernel whose loads the compiler had serialised (see the fence in the dot loop and the xnorm pointer),
// and so were the experiments made on it (waves per CU, loads in flight, start skew, blocked assignment).
// Also measured and dropped after the fix: a two-waves-per-SIMD occupancy hint for the 192/256-chunk variants
// (256 VGPRs instead of 264-270: -1 %).
#ifndef CRAG_GEMV_NT
#define CRAG_GEMV_NT 1
#endif
template <bool NT>
__device__ __forceinline__ uint4 load_row_chunk(const uint4* p) {
    // explicitly GLOBAL: in the long-row variants hipcc otherwise falls back to flat loads, which also count in
    // lgkmcnt and were waited for one at a time
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const __attribute__((address_space(1))) u32x4_t* g = (const __attribute__((address_space(1))) u32x4_t*)p;
    u32x4_t v;
    if constexpr (NT) v = __builtin_nontemporal_load(g);
    else v = *g;
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int DT, int C, int NQ>
__global__ __launch_bounds__(256) void k_scan_gemv(const GroupDesc* __restrict__ groups, int n_groups,
                                                   const uint4* __restrict__ queries, int pass, int metric,
                                                   const float* __restrict__ tau, uint32_t* __restrict__ sample,
                                                   long long sample_stride, unsigned long long* __restrict__ cand,
                                                   uint32_t* __restrict__ cnt) {
    constexpr int PL = crag_lcm(C, 64) / 64;  // loads per period (period = whole rows AND whole wave loads)
    constexpr int PR = crag_lcm(C, 64) / C;   // rows per period
    constexpr int RT = 64 / NQ;               // rows per super-tile
    constexpr int NL = RT * C / 64;           // wave loads per super-tile
    constexpr int U = NL < 6 ? NL : 6;        // loads in flight per wave
    constexpr bool NT = CRAG_GEMV_NT;
    static_assert(RT % PR == 0, "super-tile must hold whole periods");
    static_assert(C >= 32, "a wave load may span at most two rows");

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int bid = blockIdx.x;

    // block -> group (groups are few; blk_first ascending)
    const int mode = pass != 0;  // 0 dense sample, 1 filter
    int g = 0;
    for (int hi = n_groups; hi - g > 1;) {  // the last group whose first block is <= bid (64 groups: 6 dependent loads, not 64)
        const int mid = (g + hi) >> 1;
        if (bid >= groups[mid].blk_first[pass]) g = mid; else hi = mid;
    }
    const GroupDesc* __restrict__ gp = groups + g;
    const long long R = pass == 0 ? gp->n_sample : (pass == 1 ? gp->n_mid : gp->n_rows);
    const long long n_st = (R + RT - 1) / RT;
    const int n_waves = gp->n_blk[pass] * 4;
    const int w = (bid - gp->blk_first[pass]) * 4 + wave;
    const int nq = gp->nq;  // real queries in this group (<= NQ); the rest are padding
    const int q_first = gp->q_first;
    // pointers read from the descriptor are generic to the compiler; a FLAT load of xnorm would stay "pending" for
    // the wait-count pass until the norm is used after the dot loop, and every wait inside the loop then becomes
    // vmcnt(0) (drain all loads in flight) instead of vmcnt(U-1).  Global address space keeps the count exact.
    const __attribute__((address_space(1))) float* __restrict__ xnorm =
        (const __attribute__((address_space(1))) float*)gp->xnorm;

    // per-lane query chunks: the lane's chunk-in-row repeats with period PL loads
    uint4 qreg[PL][NQ];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        const int chunk = (64 * j + lane) % C;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int qi = q_first + (q < nq ? q : 0);
            qreg[j][q] = queries[(long long)qi * C + chunk];
        }
    }
    const int my_q = lane % NQ;
    const bool q_valid = my_q < nq;
    const int qg = q_first + (q_valid ? my_q : 0);
    const float my_tau = mode ? tau[qg] : 0.f;
    const uint4* __restrict__ rows = (const uint4*)gp->rows;

    // a pass skips the rows whose survivors are already in the buffer (from[] holds multiples of 128)
    const long long t_first = gp->from[pass] / RT;
    for (long long t = t_first + w; t < n_st; t += n_waves) {
        const long long row0 = t * RT;
        const uint4* __restrict__ base = rows + row0 * C;
        long long rel_max_ll = (R - row0) * C - 1;
        const int rel_max = rel_max_ll < (long long)(RT * C - 1) ? (int)rel_max_ll : RT * C - 1;
        const long long my_row = row0 + lane / NQ;
        float xn = 0.f;
        if (metric == CRAG_L2) xn = xnorm[my_row < R ? my_row : R - 1];

        float acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = 0.f;

        uint4 buf[U];
#pragma unroll
        for (int n = 0; n < U; ++n) {
            const int rel = 64 * n + lane;
            buf[n] = load_row_chunk<NT>(base + (rel < rel_max ? rel : rel_max));
        }
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const uint4 x = buf[n % U];
            if (n + U < NL) {
                const int rel = 64 * (n + U) + lane;
                buf[n % U] = load_row_chunk<NT>(base + (rel < rel_max ? rel : rel_max));
            }
            // the load above stays ABOVE this chunk's dots: without the fence hipcc's scheduler sinks each load
            // next to its use in some variants (128/192/256 chunks per row, one query) and the wave then waits
            // for one load at a time
            __builtin_amdgcn_sched_barrier(0);
            const int row_lo = (64 * n) / C;
            const int row_hi = (64 * n + 63) / C;
            if (row_lo == row_hi) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[row_lo * NQ + q] = dot_chunk<DT>(x, qreg[n % PL][q], acc[row_lo * NQ + q]);
            } else {
                const bool hi = lane >= row_hi * C - 64 * n;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float p = dot_chunk<DT>(x, qreg[n % PL][q], 0.f);
                    acc[row_lo * NQ + q] += hi ? 0.f : p;
                    acc[row_hi * NQ + q] += hi ? p : 0.f;
                }
            }
        }
        const float ip = butterfly64(acc, lane);  // lane l: row l/NQ, query l%NQ
        const float s = metric == CRAG_L2 ? fmaf(-2.f, ip, xn) : -ip;
        const bool valid = q_valid && my_row < R;
        if (mode == 0) {
            if (valid) sample[(long long)qg * sample_stride + my_row] = crag_fkey(s);
        } else {
            const bool pass = valid && s <= my_tau;
            if (__ballot(pass)) {
                if (pass) {
                    const unsigned pos = atomicAdd(&cnt[CRAG_CNT(qg)], 1u);
                    if (pos < CRAG_CAND_CAP)
                        cand[(long long)qg * CRAG_CAND_CAP + pos] =
                            ((unsigned long long)crag_fkey(s) << 32) | (unsigned long long)(uint32_t)my_row;
                }
            }
        }
    }
}

template <int DT, int C>
int launch_nq(const ScanArgs& a, int pass, int nqt, int blocks, hipStream_t s) {
#define CRAG_GO(NQ)                                                                                              \
    hipLaunchKernelGGL((k_scan_gemv<DT, C, NQ>), dim3(blocks), dim3(256), 0, s, a.groups, a.n_groups,            \
                       (const uint4*)a.queries, pass, a.metric, a.tau, a.sample, a.sample_stride, a.cand, a.cnt)
    switch (nqt) {
        case 1: CRAG_GO(1); break;
        case 2: CRAG_GO(2); break;
        case 4: CRAG_GO(4); break;
        case 8: CRAG_GO(8); break;
        default: return -1;
    }
#undef CRAG_GO
    return (int)hipGetLastError();
}

template <int DT>
int launch_c(const ScanArgs& a, int pass, int nqt, int blocks, hipStream_t s) {
    switch (a.chunks) {
        case 32: return launch_nq<DT, 32>(a, pass, nqt, blocks, s);
        case 48: return launch_nq<DT, 48>(a, pass, nqt, blocks, s);
        case 64: return launch_nq<DT, 64>(a, pass, nqt, blocks, s);
        case 96: return launch_nq<DT, 96>(a, pass, nqt, blocks, s);
        case 128: return launch_nq<DT, 128>(a, pass, nqt, blocks, s);
        case 192: return launch_nq<DT, 192>(a, pass, nqt, blocks, s);
        case 256: return launch_nq<DT, 256>(a, pass, nqt, blocks, s);
        default: return -1;
    }
}

}  // namespace

bool crag_gemv_supported(int dtype, int chunks) {
    (void)dtype;
    return chunks == 32 || chunks == 48 || chunks == 64 || chunks == 96 || chunks == 128 || chunks == 192 ||
           chunks == 256;
}

int crag_launch_scan_gemv(const ScanArgs& a, int pass, int nq_template, int total_blocks, hipStream_t s) {
    if (total_blocks <= 0) return 0;
    return a.dtype == CRAG_BF16 ? launch_c<CRAG_BF16>(a, pass, nq_template, total_blocks, s)
                                : launch_c<CRAG_F32>(a, pass, nq_template, total_blocks, s);
}
