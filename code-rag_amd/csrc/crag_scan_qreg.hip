// K2 (query-stationary form): MFMA scan for 3 <= B <= 128 query batches, bf16, dim_pad <= 768, gfx950.
//
// Replaces the k-NN half of `col.query(..., n_results=2*top_k)` at
// /root/reference/retriever/hybrid_search.py:60-64 for a batch of queries against one collection.
//
// The query block never moves: a 256-thread workgroup (one wave per SIMD, each with the whole
// 512-register file) owns 128 queries; wave w holds queries [32w, 32w+32) as the B operand of
// v_mfma_f32_32x32x16_bf16 for ALL k-steps in AGPRs (dim_pad/16 fragments x 4 = 192 AGPRs at
// D = 768).  Only the corpus streams: 32-row tiles (one contiguous 32*D*2-byte block of the
// row-major matrix) are copied HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging
// registers) into a ring of three buffers (two tiles in flight behind the one being multiplied,
// counted s_waitcnt vmcnt + raw s_barrier), and every wave reads the same A fragments from LDS (one
// ds_read_b128 per MFMA = half the LDS read rate).  No query traffic at all; for B > 128 the
// workgroups of one XCD that share a row-tile sequence hit that XCD's L2.
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

// what the K-loop needs to prefetch one tile by LDS-DMA between its MFMAs
template <int PPW>
struct QregDma {
    unsigned voff[PPW];         // per-lane byte offset of this wave's piece i inside the tile block
    const unsigned char* tile;  // wave-uniform: first byte of the tile in the row matrix
    unsigned lds_dst;           // wave-uniform: LDS byte address of this wave's piece 0 in the target buffer
    unsigned xoff;              // per-lane byte offset into the |x|^2 column (lane * 4)
    const float* xn;            // wave-uniform: |x|^2 of the tile's first row
    unsigned xn_dst;            // wave-uniform: LDS byte address of this wave's |x|^2 slot
};

#ifndef QR_KLOOP_INC
#define QR_KLOOP_INC "crag_qreg_kloop.inc"
#endif
#include QR_KLOOP_INC

constexpr int QR_NSTAGE = 3;     // LDS tile buffers: two tiles in flight behind the one being multiplied
constexpr int QR_XN_BYTES = 256;  // per wave and stage: |x|^2 of the tile's rows (DMA'd beside the tile)

template <int KS, int MODE>
__global__ __launch_bounds__(64 * QR_WAVES, 1) void k_scan_qreg(const unsigned short* __restrict__ rows,
                                                      const float* __restrict__ xnorm, long long row_begin, long long row_end,
                                                      const unsigned short* __restrict__ queries, int n_queries,
                                                      int nqg, int metric, const float* __restrict__ tau,
                                                      uint32_t* __restrict__ sample, long long sample_stride,
                                                      unsigned long long* __restrict__ cand,
                                                      uint32_t* __restrict__ cnt) {
    constexpr int C = KS * 2;                  // 16-byte chunks per row
    constexpr int D = KS * 16;                 // elements per row
    constexpr int STAGE = QR_TM * C * 16;      // bytes per LDS tile buffer
    constexpr int PIECES = QR_TM * C / 64;     // 1-KiB DMA pieces per tile
    constexpr int PPW = PIECES / QR_WAVES;     // pieces per wave
    constexpr int XN_BASE = QR_NSTAGE * STAGE; // |x|^2 slots: [stage][wave][64 floats]
    static_assert(PIECES % QR_WAVES == 0, "pieces must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // the ONLY LDS object

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
    long long t = row_begin / QR_TM + (long long)stream * 8 + xcd;  // row_begin: a multiple of the tile height

    const int q_base = qg * QR_QPB + wave * QR_QPW;
    const bool wave_active = q_base < n_queries;  // wave-uniform
    const int my_q = q_base + fr;
    const bool q_ok = my_q < n_queries;
    // stationary B operand: this lane's query, k = 16 s + 8 fh + [0,8)  (lives in AGPRs, see the K-loop)
    bf16x8_t qf[KS];
    {
        const unsigned short* qp = queries + (long long)(q_ok ? my_q : n_queries - 1) * D + 8 * fh;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = *(const bf16x8_t*)(qp + 16 * s);
    }
    float tq = -__builtin_inff();
    if (MODE == 1 && q_ok) tq = tau[my_q];

    // LDS-DMA of one tile: piece (wave + WAVES*i) of the lane-linear image; linear 16-byte position P
    // holds global chunk (row P / C, chunk (P % C) ^ (row & 15)).  Plus this wave's copy of |x|^2.
    // Every wave issues exactly PPW + 1 DMA instructions per tile (the counted waits rely on it).
    // A tile is one contiguous block of the row-major matrix, so the per-lane source is a wave-uniform
    // tile base (SGPR pair) plus a per-lane byte offset that never changes (saddr + voffset form).
    unsigned voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int P = (wave + QR_WAVES * i) * 64 + lane;
        const int r = P / C, p = P - r * C;
        voff[i] = (unsigned)(r * C + (p ^ (r & 15))) * 16u;
    }
    auto issue = [&](long long tile, int buf) {
        const long long row0 = tile * QR_TM;
        if (row0 + QR_TM <= row_end) {
            const unsigned char* tb = (const unsigned char*)rows + row0 * (long long)(D * 2);
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + voff[i]),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (wave + QR_WAVES * i) * 1024),
                                                 16, 0, 0);
        } else {  // ragged last tile: clamp rows (never read past the matrix)
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
        }
        const float* gx = xnorm + row0 + lane;  // xnorm is padded by >= 256 floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                         (__attribute__((address_space(3))) void*)(lds + XN_BASE + (buf * QR_WAVES + wave) * QR_XN_BYTES),
                                         4, 0, 0);
    };

    // fragment read offsets: row fr, chunk 2 s + fh, swizzled
    const int a_row_off = fr * C * 16;
    const int a_sw = fr & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

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
        if (nxt2 >= QR_NSTAGE) nxt2 -= QR_NSTAGE;
        // prefetch tile t + 2 steps into the buffer of tile t - step (every wave left it before the last
        // barrier): between the MFMAs of the K-loop when this wave multiplies and the tile is whole,
        // up front otherwise
        const bool dma_in_loop = more && wave_active && (nt2 + 1) * QR_TM <= row_end;
        if (more && !dma_in_loop) issue(nt2, nxt2);
        if (wave_active) {
            unsigned ad[8];  // LDS byte address of k-step j (j < 8); k-step s adds (s >> 3) * 256
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ad[jj] = lds_base + cur * STAGE + a_row_off + (((2 * jj + fh) ^ a_sw) << 4);
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (dma_in_loop) {
                QregDma<PPW> d;
#pragma unroll
                for (int i = 0; i < PPW; ++i) d.voff[i] = voff[i];
                d.tile = (const unsigned char*)rows + nt2 * (long long)(QR_TM * D * 2);
                d.lds_dst = lds_base + nxt2 * STAGE + wave * 1024;
                d.xoff = lane * 4;
                d.xn = xnorm + nt2 * QR_TM;
                d.xn_dst = lds_base + XN_BASE + (nxt2 * QR_WAVES + wave) * QR_XN_BYTES;
                QregKLoop<KS, true>::run(acc, ad, qf, d);
            } else {
                QregKLoop<KS, false>::run(acc, ad, qf);
            }
#ifdef QR_ABL_NOEPI
            asm volatile("" ::"a"(acc));
#else
            // |x|^2 of this lane's 16 rows (rows 8 g + 4 fh + e) from this wave's LDS copy; one asm
            // statement with its own wait, so no compiler-side alias wait on the pending DMA
            float4 xn[4];
            {
                const unsigned xa = lds_base + XN_BASE + (cur * QR_WAVES + wave) * QR_XN_BYTES + 16 * fh;
                asm volatile(
                    "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                    "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(xn[0]), "=&v"(xn[1]), "=&v"(xn[2]), "=&v"(xn[3])
                    : "v"(xa)
                    : "memory");
            }
            // fused epilogue: 16 rows of this lane's query
            const long long mb = t * QR_TM + 4 * fh;
            float sc[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float xv[4] = {xn[g4].x, xn[g4].y, xn[g4].z, xn[g4].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ip = acc[g4 * 4 + e];
                    sc[g4 * 4 + e] = metric == CRAG_L2 ? fmaf(-2.f, ip, xv[e]) : -ip;
                }
            }
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = mb + 8 * (r >> 2) + (r & 3);
                    if (q_ok && m < row_end) sample[m * sample_stride + my_q] = crag_fkey(sc[r]);
                }
            } else {
                // fast path: nothing in this wave's 32 x 32 block reaches its query's threshold
                float mn = sc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mn = fminf(mn, sc[r]);
                if (__any(mn <= tq)) {
                    // one returning atomic per lane (= per query) for all of its passing rows
                    int npass = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long m = mb + 8 * (r >> 2) + (r & 3);
                        npass += (sc[r] <= tq && m < row_end) ? 1 : 0;
                    }
                    if (npass) {
                        unsigned pos = atomicAdd(&cnt[CRAG_CNT(my_q)], (unsigned)npass);
                        unsigned long long* cq = cand + (long long)my_q * CRAG_CAND_CAP;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long m = mb + 8 * (r >> 2) + (r & 3);
                            if (sc[r] <= tq && m < row_end) {
                                if (pos < CRAG_CAND_CAP)
                                    cq[pos] = ((unsigned long long)crag_fkey(sc[r]) << 32) | (unsigned long long)(uint32_t)m;
                                ++pos;
                            }
                        }
                    }
                }
            }
#endif
        }
        // tile t + step must have landed in every wave's view before anyone reads it: each wave retires all
        // but the DMA instructions it issued in THIS iteration (they are its youngest), then the barrier
        if (more) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == QR_NSTAGE ? 0 : cur + 1;
    }
}

template <int KS>
int launch_ks(const MfmaArgs& a, int mode, int grid, int nqg, hipStream_t s) {
    const size_t lds = (size_t)QR_NSTAGE * QR_TM * KS * 2 * 16 + (size_t)QR_NSTAGE * QR_WAVES * QR_XN_BYTES;
    static std::atomic<uint64_t> lds_ok[2];
    crag_allow_dynamic_lds(mode == 0 ? (const void*)k_scan_qreg<KS, 0> : (const void*)k_scan_qreg<KS, 1>, lds, lds_ok[mode]);
    if (mode == 0)
        hipLaunchKernelGGL((k_scan_qreg<KS, 0>), dim3(grid), dim3(64 * QR_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                           a.row_begin, a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample,
                           a.sample_stride, a.cand, a.cnt);
    else
        hipLaunchKernelGGL((k_scan_qreg<KS, 1>), dim3(grid), dim3(64 * QR_WAVES), lds, s, (const unsigned short*)a.rows, a.xnorm,
                           a.row_begin, a.row_end, (const unsigned short*)a.queries, a.n_queries, nqg, a.metric, a.tau, a.sample,
                           a.sample_stride, a.cand, a.cnt);
    return (int)hipGetLastError();
}

}  // namespace

bool crag_qreg_supported(int dim_pad) { return dim_pad == 256 || dim_pad == 384 || dim_pad == 512 || dim_pad == 768; }

// sample layout of this kernel: sample[row * sample_stride + query]  (row-major)
int crag_launch_scan_qreg(const MfmaArgs& a, int mode, int n_cus, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.n_queries <= 0) return 0;
    if (a.row_begin % QR_TM) return -1;
    const int nqg = (a.n_queries + QR_QPB - 1) / QR_QPB;
    if (nqg > 32) return -1;
    int streams = (n_cus / 8) / nqg;  // row-tile streams per XCD
    if (streams < 1) streams = 1;
    const long long n_tiles = (a.row_end - a.row_begin + QR_TM - 1) / QR_TM;
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
