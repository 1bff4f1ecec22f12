// K2 (query-stationary form): MFMA scan for 9 <= B query batches, bf16, dim_pad <= 768, gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// The query block never moves: a 512-thread workgroup owns 256 queries, wave w holds queries
// [32w, 32w+32) as the B operand of v_mfma_f32_32x32x16_bf16 for ALL k-steps in registers
// (dim_pad/16 fragments x 4 VGPRs = 192 VGPRs at D = 768).  Only the corpus streams: 32-row
// tiles (one contiguous 32*D*2-byte block of the row-major matrix) are copied HBM -> LDS by
// LDS-DMA (global_load_lds_dwordx4, no staging registers), double-buffered, and every wave reads
// the same A fragments from LDS (one ds_read_b128 per MFMA = half the LDS read rate).  No query
// traffic, no corpus re-read for B <= 256; for B > 256 the workgroups of one XCD that share a
// row-tile sequence hit that XCD's L2.
//
// LDS image of a tile: linear rows of D*2 bytes (a multiple of 256 B), 16-byte chunk p of row r
// holds global chunk p ^ (r & 15): the DMA destination is lane-linear, so the swizzle is applied
// to the per-lane SOURCE address and again on the fragment read; the 16 lanes of a ds_read_b128
// group then hit 16 distinct 16-byte slots (conflict free).
//
// Accumulator layout: query on the lane (col = lane & 31), 16 corpus rows in registers, so the
// fused top-k filter is one compare per register against a per-lane threshold tau.
//   MODE 0 (sample): ordered-float score keys to sample[row][Bpad]  (row-major, coalesced)
//   MODE 1 (filter): s <= tau -> append (score key << 32 | row) to the query's candidate buffer
#include "crag_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int QR_TM = 32;        // corpus rows per tile
constexpr int QR_WAVES = 4;      // waves per workgroup: one per SIMD, each with the whole 512-register file
constexpr int QR_QB = 1;         // 32-query blocks per wave (each A fragment read feeds QR_QB MFMAs)
constexpr int QR_QPW = 32 * QR_QB;         // queries per wave
constexpr int QR_QPB = QR_WAVES * QR_QPW;  // queries per workgroup
static_assert(QR_QB == 1, "the generated K-loop handles one 32-query block per wave");

#include "crag_qreg_kloop.inc"

template <int KS, int MODE>
__global__ __launch_bounds__(64 * QR_WAVES, 1) void k_scan_qreg(const unsigned short* __restrict__ rows,
                                                      const float* __restrict__ xnorm, long long row_end,
                                                      const unsigned short* __restrict__ queries, int n_queries,
                                                      int nqg, int metric, const float* __restrict__ tau,
                                                      uint32_t* __restrict__ sample, long long sample_stride,
                                                      unsigned long long* __restrict__ cand,
                                                      uint32_t* __restrict__ cnt) {
    constexpr int C = KS * 2;                  // 16-byte chunks per row
    constexpr int D = KS * 16;                 // elements per row
    constexpr int STAGE = QR_TM * C * 16;      // bytes per LDS buffer
    constexpr int PIECES = QR_TM * C / 64;     // 1-KiB DMA pieces per tile
    constexpr int PPW = PIECES / QR_WAVES;     // pieces per wave
    static_assert(PIECES % QR_WAVES == 0, "pieces must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // the ONLY LDS object: [2][STAGE]

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
    const long long n_tiles = (row_end + QR_TM - 1) / QR_TM;
    const long long t_step = (long long)n_streams * 8;
    long long t = (long long)stream * 8 + xcd;

    const int q_base = qg * QR_QPB + wave * QR_QPW;
    const bool wave_active = q_base < n_queries;  // wave-uniform
    // stationary B operands: this lane's QR_QB queries, k = 16 s + 8 fh + [0,8)
    bf16x8_t qf[QR_QB][KS];
    int my_q[QR_QB];
    bool q_ok[QR_QB];
    float tq[QR_QB];
#pragma unroll
    for (int qb = 0; qb < QR_QB; ++qb) {
        my_q[qb] = q_base + 32 * qb + fr;
        q_ok[qb] = my_q[qb] < n_queries;
        const unsigned short* qp = queries + (long long)(q_ok[qb] ? my_q[qb] : n_queries - 1) * D + 8 * fh;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[qb][s] = *(const bf16x8_t*)(qp + 16 * s);
        tq[qb] = -__builtin_inff();
        if (MODE == 1 && q_ok[qb]) tq[qb] = tau[my_q[qb]];
    }

    // LDS-DMA of one tile: piece (wave + 8 i) of the lane-linear image; linear 16-byte position P holds
    // global chunk (row P / C, chunk (P % C) ^ (row & 15))
    auto issue = [&](long long tile, int buf) {
        const long long row0 = tile * QR_TM;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int P = (wave + QR_WAVES * i) * 64 + lane;
            const int r = P / C, p = P - r * C;
            long long gr = row0 + r;
            if (gr >= row_end) gr = row_end - 1;
            const unsigned short* g = rows + gr * D + ((p ^ (r & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (wave + QR_WAVES * i) * 1024),
                                             16, 0, 0);
        }
    };

    // fragment read offsets: row fr, chunk 2 s + fh, swizzled
    const int a_row_off = fr * C * 16;
    const int a_sw = fr & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    int cur = 0;
    if (t < n_tiles) issue(t, 0);
    __syncthreads();
    for (; t < n_tiles; t += t_step) {
        const long long nt = t + t_step;
        if (nt < n_tiles) issue(nt, cur ^ 1);
        if (wave_active) {
            unsigned ad[8];  // LDS byte address of k-step j (j < 8); k-step s adds (s >> 3) * 256
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ad[jj] = lds_base + cur * STAGE + a_row_off + (((2 * jj + fh) ^ a_sw) << 4);
            f32x16_t acc[QR_QB];
#pragma unroll
            for (int qb = 0; qb < QR_QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
            QregKLoop<KS>::run(acc[0], ad, qf[0]);
            // fused epilogue: 16 rows of each of this lane's queries
            const long long mb = t * QR_TM + 4 * fh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const long long m0 = mb + 8 * g4;
                float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (metric == CRAG_L2) xn = *(const float4*)(xnorm + m0);  // xnorm is padded by >= 256 floats
                const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
                for (int qb = 0; qb < QR_QB; ++qb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ip = acc[qb][g4 * 4 + e];
                        const float s = metric == CRAG_L2 ? fmaf(-2.f, ip, xv[e]) : -ip;
                        const long long m = m0 + e;
                        if (MODE == 0) {
                            if (q_ok[qb] && m < row_end) sample[m * sample_stride + my_q[qb]] = crag_fkey(s);
                        } else {
                            if (s <= tq[qb] && m < row_end) {
                                const unsigned pos = atomicAdd(&cnt[my_q[qb]], 1u);
                                if (pos < CRAG_CAND_CAP)
                                    cand[(long long)my_q[qb] * CRAG_CAND_CAP + pos] =
                                        ((unsigned long long)crag_fkey(s) << 32) | (unsigned long long)(uint32_t)m;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();  // next tile landed (vmcnt(0)) and every wave is done with buf[cur]
        cur ^= 1;
    }
}

template <int KS>
int launch_ks(const MfmaArgs& a, int mode, int grid, int nqg, hipStream_t s) {
    const size_t lds = 2 * (size_t)QR_TM * KS * 2 * 16;
    static bool attr_set[2] = {false, false};
    if (!attr_set[mode]) {
        if (mode == 0)
            (void)hipFuncSetAttribute((const void*)k_scan_qreg<KS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else
            (void)hipFuncSetAttribute((const void*)k_scan_qreg<KS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[mode] = true;
    }
    if (mode == 0)
        hipLaunchKernelGGL((k_scan_qreg<KS, 0>), dim3(grid), dim3(64 * QR_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                           a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample,
                           a.sample_stride, a.cand, a.cnt);
    else
        hipLaunchKernelGGL((k_scan_qreg<KS, 1>), dim3(grid), dim3(64 * QR_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                           a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample,
                           a.sample_stride, a.cand, a.cnt);
    return (int)hipGetLastError();
}

}  // namespace

bool crag_qreg_supported(int dim_pad) { return dim_pad == 256 || dim_pad == 384 || dim_pad == 512 || dim_pad == 768; }

// sample layout of this kernel: sample[row * sample_stride + query]  (row-major)
int crag_launch_scan_qreg(const MfmaArgs& a, int mode, int n_cus, hipStream_t s) {
    if (a.row_end <= 0 || a.n_queries <= 0) return 0;
    const int nqg = (a.n_queries + QR_QPB - 1) / QR_QPB;
    if (nqg > 32) return -1;
    int streams = (n_cus / 8) / nqg;  // row-tile streams per XCD
    if (streams < 1) streams = 1;
    const long long n_tiles = (a.row_end + QR_TM - 1) / QR_TM;
    while (streams > 1 && (long long)streams * 8 > n_tiles) --streams;
    const int grid = 8 * streams * nqg;
    switch (a.dim_pad) {
        case 256: return launch_ks<16>(a, mode, grid, nqg, s);
        case 384: return launch_ks<24>(a, mode, grid, nqg, s);
        case 512: return launch_ks<32>(a, mode, grid, nqg, s);
        case 768: return launch_ks<48>(a, mode, grid, nqg, s);
        default: return -1;
    }
}
