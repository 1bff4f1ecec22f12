// K2: MFMA-tiled query-by-corpus scan for large query batches (GEMM regime), bf16, gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// scores[m][n] = <row m, query n> with corpus rows on the MFMA M index (A operand) and queries on
// the N index (B operand), so the accumulator layout of v_mfma_f32_32x32x16_bf16 puts the QUERY
// on the lane (col = lane & 31) and 16 corpus rows in the lane's registers: the per-query filter
// threshold tau is one VGPR per lane and the fused top-k filter is 16 compares per accumulator.
// The score matrix is never written (mode 1); only rows with s <= tau are appended.
//
// Tile: 128 corpus rows x 128 queries per 256-thread workgroup (2x2 waves of 64x64), BK = 64,
// LDS double-buffered, register-staged global loads, XOR-swizzled 128-byte LDS rows so that
// ds_read_b128 fragment reads are bank-conflict free (slot = chunk ^ ((row >> 1) & 7)).
#include "crag_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int TM = 128;  // corpus rows per tile
constexpr int TN = 128;  // queries per tile
constexpr int BK = 64;   // k elements per stage (128 bytes per row)

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256) void k_scan_mfma(const unsigned short* __restrict__ rows,
                                                   const float* __restrict__ xnorm, long long row_end,
                                                   const unsigned short* __restrict__ queries, int n_queries,
                                                   int n_qtiles, int dim_pad, int mode, int metric,
                                                   const float* __restrict__ tau, uint32_t* __restrict__ sample,
                                                   long long sample_stride, unsigned long long* __restrict__ cand,
                                                   uint32_t* __restrict__ cnt) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][TM * 128];  // [stage][A|B][16 KiB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const long long bid = blockIdx.x;
    const int qt = (int)(bid % n_qtiles);
    const long long rt = bid / n_qtiles;
    const long long row0 = rt * TM;
    const int q0 = qt * TN;

    // staging geometry: thread handles chunks id = tid + 256*i (i<4): row = id>>3, chunk = id&7
    const uint4* gA[4];
    const uint4* gB[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        const int r = id >> 3, c = id & 7;
        long long ar = row0 + r;
        if (ar >= row_end) ar = row_end - 1;
        int qr = q0 + r;
        if (qr >= n_queries) qr = n_queries - 1;
        gA[i] = (const uint4*)(rows + ar * dim_pad) + c;
        gB[i] = (const uint4*)(queries + (long long)qr * dim_pad) + c;
        soff[i] = lds_off(r, c);
    }
    const int KT = dim_pad / BK;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = gA[i][0];
        rb[i] = gB[i][0];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(uint4*)(&lds[0][0][soff[i]]) = ra[i];
        *(uint4*)(&lds[0][1][soff[i]]) = rb[i];
    }
    __syncthreads();

    // fragment read offsets (per k-step s the chunk is 2s + (lane>>5))
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = gA[i][(kt + 1) * 8];
                rb[i] = gB[i][(kt + 1) * 8];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 64 + i * 32 + fr;
                a[i] = *(const bf16x8_t*)(&lds[cur][0][lds_off(r, 2 * s + fh)]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wc * 64 + j * 32 + fr;
                b[j] = *(const bf16x8_t*)(&lds[cur][1][lds_off(r, 2 * s + fh)]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(uint4*)(&lds[cur ^ 1][0][soff[i]]) = ra[i];
                *(uint4*)(&lds[cur ^ 1][1][soff[i]]) = rb[i];
            }
        }
        __syncthreads();
    }

    // fused epilogue: query on the lane, 16 corpus rows per accumulator
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + wc * 64 + j * 32 + fr;
        const bool qok = q < n_queries;
        const float tq = (mode && qok) ? tau[q] : -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long mb = row0 + wr * 64 + i * 32 + 4 * fh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const long long m0 = mb + 8 * g4;
                float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (metric == CRAG_L2) xn = *(const float4*)(xnorm + m0);  // xnorm is padded by >= 128 floats
                const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ip = acc[i][j][g4 * 4 + e];
                    const float s = metric == CRAG_L2 ? fmaf(-2.f, ip, xv[e]) : -ip;
                    const long long m = m0 + e;
                    if (mode == 0) {
                        if (qok && m < row_end) sample[(long long)q * sample_stride + m] = crag_fkey(s);
                    } else {
                        if (s <= tq && m < row_end) {
                            const unsigned pos = atomicAdd(&cnt[CRAG_CNT(q)], 1u);
                            if (pos < CRAG_CAND_CAP)
                                cand[(long long)q * CRAG_CAND_CAP + pos] =
                                    ((unsigned long long)crag_fkey(s) << 32) | (unsigned long long)(uint32_t)m;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

int crag_launch_scan_mfma(const MfmaArgs& a, int mode, hipStream_t s) {
    const long long n = a.row_end;
    if (n <= 0 || a.n_queries <= 0) return 0;
    if (a.dim_pad % BK != 0) return -1;
    const long long rtiles = (n + TM - 1) / TM;
    const int qtiles = (a.n_queries + TN - 1) / TN;
    const long long blocks = rtiles * qtiles;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL(k_scan_mfma, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)a.rows, a.xnorm,
                       n, (const unsigned short*)a.queries, a.n_queries, qtiles, a.dim_pad, mode, a.metric, a.tau,
                       a.sample, a.sample_stride, a.cand, a.cnt);
    return (int)hipGetLastError();
}
