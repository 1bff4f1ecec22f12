// LDS-tiled batched scan, gfx950: bf16 rows of 1536 / 2048 elements and float32 rows of every length.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection, where no
// query-stationary scan applies:
//   * bf16 rows longer than 1024 elements: the queries do not fit a wave's registers (96 / 128 MFMA fragments per 32
//     queries = 384 / 512 registers before the first accumulator), so crag_scan_q16 / qsplit stop at 1024 elements;
//     GEMV groups of 8 queries read the corpus B / 8 times (tools/sweep_auto.py, round 2: 256 queries on 1.3M x 1536
//     took 16 ms, 15x the 768-element case);
//   * float32 rows (the reference's own storage type, embed_ingest.py:93 via Chroma): the float32 matrix pipe is slow
//     enough (16 passes per v_mfma_f32_32x32x2_f32) that operands fed from LDS keep it busy -- this kernel is 3-10 %
//     ahead of the hand-scheduled query-stationary float32 scan it replaced.
//
// Neither operand is stationary: a workgroup owns a tile of 128 corpus rows x TN queries and walks K in steps of
// 128 bytes per row (64 bf16 / 32 float32 elements); both operands go HBM/L2 -> registers (two steps ahead) -> LDS
// (two stages, one barrier per step) and from there as ds_read_b128 fragments into v_mfma_f32_32x32x16_bf16 -- or, for
// float32, into four v_mfma_f32_32x32x2_f32 per fragment: lane (row, half) holds elements 4*(2s + half) + e of its
// row, MFMA e multiplies element e of the A fragment with element e of the B fragment, so both operands see the same
// k order and no shuffle is needed.  Corpus rows are the A operand and queries the B operand, so the accumulator puts
// the QUERY on the lane (column = lane & 31) and 16 rows in the lane's registers: tau is one VGPR per lane and the
// fused filter is 16 compares per accumulator, as in the query-stationary scans.
// 128-byte LDS rows, XOR swizzle slot = chunk ^ ((row >> 1) & 7): the fragment reads are conflict-free.
// Workgroups that share a row tile (more queries than one tile holds) are eight apart in the grid, so they run on
// the same XCD and the second one finds the tile in that XCD's L2.
//
// Same contract as the one-block query-stationary scan (MfmaArgs in crag_common.h):
//   MODE 0 (sample): ordered-float score keys to sample[row][Bpad]  (row-major, coalesced over queries)
//   MODE 1 (filter): s <= tau -> append (score key << 32 | row) to the query's flat candidate buffer
#include "crag_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int QL_TM = 128;  // corpus rows per tile
// queries per tile: TN = 128 (2 x 2 waves of 64 rows x 64 queries), 96 (4 x 1 waves of 32 x 96), 64 (2 x 2 waves of
// 64 x 32) or 32 (4 x 1 waves of 32 x 32).  A tile multiplies all its columns whatever the batch holds, which is what
// small float32 batches feel (the float32 matrix pipe is 16x slower than the bf16 one: a 128-query tile costs 2.2 ms
// on 4 GB of rows, a 32-query tile a quarter); launch_dt picks the width with the fewest columns for the batch.
template <int TN> struct QlGeom;
template <> struct QlGeom<128> { static constexpr int WC = 2, FI = 2, FJ = 2; };
template <> struct QlGeom<96> { static constexpr int WC = 1, FI = 1, FJ = 3; };
template <> struct QlGeom<64> { static constexpr int WC = 2, FI = 2, FJ = 1; };
template <> struct QlGeom<32> { static constexpr int WC = 1, FI = 1, FJ = 1; };
constexpr int QL_KB = 128;  // bytes of every row per stage

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool NT>
__device__ __forceinline__ uint4 load16(const uint4* p) {
    const __attribute__((address_space(1))) u32x4_t* g = (const __attribute__((address_space(1))) u32x4_t*)p;
    u32x4_t v;
    if constexpr (NT) v = __builtin_nontemporal_load(g);
    else v = *g;
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int DT, int MODE, bool NT, int TN>
__global__ __launch_bounds__(256, 2) void k_scan_qlong(const unsigned char* __restrict__ rows,
                                                       const float* __restrict__ xnorm, long long row_begin,
                                                       long long row_end, const unsigned char* __restrict__ queries,
                                                       int n_queries, int n_qtiles, int row_bytes, int metric,
                                                       const float* __restrict__ tau, uint32_t* __restrict__ sample,
                                                       long long sample_stride, unsigned long long* __restrict__ cand,
                                                       uint32_t* __restrict__ cnt) {
    constexpr int WC = QlGeom<TN>::WC, FI = QlGeom<TN>::FI, FJ = QlGeom<TN>::FJ;  // wave grid (4 / WC) x WC, FI x FJ fragments each
    constexpr int NB = TN / 32;  // 16-byte chunks of the query operand per thread and step
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][(QL_TM + TN) * 128];  // [stage]: A 16 KiB, then B
    constexpr int B_OFF = QL_TM * 128;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    // grid index -> (row tile, query tile): the query tiles of one row tile are 8 blocks apart (same XCD)
    const long long bid = blockIdx.x;
    const long long per = 8LL * n_qtiles;
    const long long grp = bid / per;
    const int rem = (int)(bid - grp * per);
    const int qt = rem >> 3;
    const long long rt = grp * 8 + (rem & 7);
    const long long row0 = row_begin + rt * QL_TM;
    if (row0 >= row_end) return;  // the last group of eight row tiles may be ragged (uniform per workgroup)
    const int q0 = qt * TN;

    // staging geometry: thread handles chunks id = tid + 256*i: row = id>>3, chunk = id&7 (4 per step of the corpus
    // tile, NB of the query tile)
    const uint4* gA[4];
    const uint4* gB[NB];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        const int r = id >> 3, c = id & 7;
        long long ar = row0 + r;
        if (ar >= row_end) ar = row_end - 1;
        gA[i] = (const uint4*)(rows + ar * row_bytes) + c;
        soff[i] = lds_off(r, c);
        if (i < NB) {
            int qr = q0 + r;
            if (qr >= n_queries) qr = n_queries - 1;
            gB[i] = (const uint4*)(queries + (long long)qr * row_bytes) + c;
        }
    }
    const int KT = row_bytes / QL_KB;

    f32x16_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K loop, two steps per trip so that the two register sets are static.  Loads run TWO steps ahead of the
    // multiplication (a step's 16 MFMAs per wave last ~0.25 us, a corpus load from HBM 1-2 us: with one step of
    // lookahead and two workgroups per CU only 32 KiB per CU were in flight).  The barrier is the raw s_barrier behind
    // a wait for this wave's LDS traffic only: __syncthreads() would also drain the loads in flight (vmcnt(0)).
    const int fr = lane & 31, fh = lane >> 5;  // fragment reads: per k-step s the chunk is 2s + (lane >> 5)
    uint4 ra0[4], rb0[NB], ra1[4], rb1[NB];
    auto load_step = [&](int kt, uint4 (&ra)[4], uint4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = load16<NT>(gA[i] + kt * 8);
            if (i < NB) rb[i] = load16<false>(gB[i] + kt * 8);
        }
        __builtin_amdgcn_sched_barrier(0);  // the loads are issued HERE, ahead of the step's MFMAs (hipcc sinks them otherwise)
    };
    auto store_step = [&](int buf, const uint4 (&ra)[4], const uint4 (&rb)[NB]) {
        __builtin_amdgcn_sched_barrier(0);  // ... and their wait + LDS stores stay behind the MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(uint4*)(&lds[buf][soff[i]]) = ra[i];
            if (i < NB) *(uint4*)(&lds[buf][B_OFF + soff[i]]) = rb[i];
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto multiply = [&](int cur) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (DT == CRAG_BF16) {
                bf16x8_t a[FI], b[FJ];
#pragma unroll
                for (int i = 0; i < FI; ++i) a[i] = *(const bf16x8_t*)(&lds[cur][lds_off((wr * FI + i) * 32 + fr, 2 * s + fh)]);
#pragma unroll
                for (int j = 0; j < FJ; ++j) b[j] = *(const bf16x8_t*)(&lds[cur][B_OFF + lds_off((wc * FJ + j) * 32 + fr, 2 * s + fh)]);
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                f32x4_t a[FI], b[FJ];
#pragma unroll
                for (int i = 0; i < FI; ++i) a[i] = *(const f32x4_t*)(&lds[cur][lds_off((wr * FI + i) * 32 + fr, 2 * s + fh)]);
#pragma unroll
                for (int j = 0; j < FJ; ++j) b[j] = *(const f32x4_t*)(&lds[cur][B_OFF + lds_off((wc * FJ + j) * 32 + fr, 2 * s + fh)]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < FI; ++i)
#pragma unroll
                        for (int j = 0; j < FJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
    };
    // KT is even (row_bytes is a multiple of 256, checked by the launcher); everything in the loop is unconditional,
    // so the two register sets stay where they are (conditional loads made hipcc copy one set into the other)
    load_step(0, ra0, rb0);
    load_step(1, ra1, rb1);
    store_step(0, ra0, rb0);
    lds_barrier();
    for (int kt = 0; kt < KT - 2; kt += 2) {
        // LDS[0] holds step kt; set 1 holds step kt + 1 (maybe still in flight); set 0 is free
        load_step(kt + 2, ra0, rb0);
        multiply(0);
        store_step(1, ra1, rb1);
        lds_barrier();
        // LDS[1] holds step kt + 1; set 0 holds step kt + 2; set 1 is free
        load_step(kt + 3, ra1, rb1);
        multiply(1);
        store_step(0, ra0, rb0);
        lds_barrier();
    }
    multiply(0);  // step KT - 2 from LDS[0]; set 1 holds the last step
    store_step(1, ra1, rb1);
    lds_barrier();
    multiply(1);

    // fused epilogue: query on the lane, 16 corpus rows per accumulator
    const float c2 = metric == CRAG_L2 ? -2.f : -1.f;
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        const int q = q0 + (wc * FJ + j) * 32 + fr;
        const bool qok = q < n_queries;
        // a lane past the batch multiplied a copy of the last query: NaN, not -inf, so that not even a -inf score passes
        const float tq = (MODE == 1 && qok) ? tau[q] : __builtin_nanf("");
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const long long mb = row0 + (wr * FI + i) * 32 + 4 * fh;
            float sc[16];
            bool hit = false;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (metric == CRAG_L2) xn = *(const float4*)(xnorm + mb + 8 * g4);  // xnorm is padded by 256 floats
                const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sc[g4 * 4 + e] = fmaf(c2, acc[i][j][g4 * 4 + e], xv[e]);
                    hit = hit || sc[g4 * 4 + e] <= tq;
                }
            }
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = mb + 8 * (r >> 2) + (r & 3);
                    if (qok && m < row_end) sample[m * sample_stride + q] = crag_fkey(sc[r]);
                }
            } else if (__any(hit)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = mb + 8 * (r >> 2) + (r & 3);
                    if (sc[r] <= tq && m < row_end) {
                        const unsigned pos = atomicAdd(&cnt[CRAG_CNT(q)], 1u);
                        if (pos < CRAG_CAND_CAP)
                            cand[(long long)q * CRAG_CAND_CAP + pos] =
                                ((unsigned long long)crag_fkey(sc[r]) << 32) | (unsigned long long)(uint32_t)m;
                    }
                }
            }
        }
    }
}

template <int DT, int MODE, bool NT, int TN>
int launch_one(const MfmaArgs& a, int esz, long long blocks, int qtiles, hipStream_t s) {
    hipLaunchKernelGGL((k_scan_qlong<DT, MODE, NT, TN>), dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char*)a.rows,
                       a.xnorm, a.row_begin, a.row_end, (const unsigned char*)a.queries, a.n_queries, qtiles,
                       a.dim_pad * esz, a.metric, a.tau, a.sample, a.sample_stride, a.cand, a.cnt);
    return (int)hipGetLastError();
}

template <int DT, int TN>
int launch_tn(const MfmaArgs& a, int mode, hipStream_t s) {
    const int esz = DT == CRAG_BF16 ? 2 : 4;
    const long long rtiles = (a.row_end - a.row_begin + QL_TM - 1) / QL_TM;
    const int qtiles = (a.n_queries + TN - 1) / TN;
    const long long blocks = (rtiles + 7) / 8 * 8 * qtiles;
    if (blocks > 0x7fffffffLL) return -1;
    const bool nt = qtiles == 1;  // a row tile read by one workgroup only: keep it out of the L2
    if (mode == 0) return nt ? launch_one<DT, 0, true, TN>(a, esz, blocks, qtiles, s) : launch_one<DT, 0, false, TN>(a, esz, blocks, qtiles, s);
    return nt ? launch_one<DT, 1, true, TN>(a, esz, blocks, qtiles, s) : launch_one<DT, 1, false, TN>(a, esz, blocks, qtiles, s);
}

template <int DT>
int launch_dt(const MfmaArgs& a, int mode, hipStream_t s) {
    // the tile width that multiplies the fewest columns for this batch (a tile multiplies all of its columns whatever
    // the batch holds); every tile also stages the 128 corpus rows once more, priced at 16 columns
    int best = 128;
    long long cost = (((long long)a.n_queries + 127) / 128) * (128 + 16);
    for (int tn : {96, 64, 32}) {
        const long long c = (((long long)a.n_queries + tn - 1) / tn) * (tn + 16);
        if (c < cost) {
            cost = c;
            best = tn;
        }
    }
    switch (best) {
        case 32: return launch_tn<DT, 32>(a, mode, s);
        case 64: return launch_tn<DT, 64>(a, mode, s);
        case 96: return launch_tn<DT, 96>(a, mode, s);
        default: return launch_tn<DT, 128>(a, mode, s);
    }
}

}  // namespace

// bf16: 1536 / 2048 elements (shorter rows have query-stationary scans); float32: every row length
bool crag_qlong_supported(int dtype, int dim_pad) {
    if (dtype == CRAG_BF16) return dim_pad > 1024 && dim_pad <= 2048 && dim_pad % 128 == 0;
    return dim_pad >= 128 && dim_pad <= 1024 && dim_pad % 64 == 0;  // any float32 row length the index stores
}

// sample layout of this kernel: sample[row * sample_stride + query]  (row-major)
int crag_launch_scan_qlong(const MfmaArgs& a, int dtype, int mode, int n_cus, hipStream_t s) {
    (void)n_cus;
    if (a.row_end <= a.row_begin || a.n_queries <= 0) return 0;
    if (a.row_begin % 32 || !crag_qlong_supported(dtype, a.dim_pad)) return -1;
    if ((a.dim_pad * (dtype == CRAG_BF16 ? 2 : 4)) % (2 * QL_KB)) return -1;  // the K loop takes two 128-byte steps per trip
    return dtype == CRAG_BF16 ? launch_dt<CRAG_BF16>(a, mode, s) : launch_dt<CRAG_F32>(a, mode, s);
}
