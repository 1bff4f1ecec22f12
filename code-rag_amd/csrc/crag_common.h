// Shared host/device definitions of the coderag-hip retrieval scoring library (gfx950 only).
//
// The path this library replaces: /root/reference/retriever/hybrid_search.py:42-125
// (HybridSearcher.search) -- the Chroma `col.query` k-NN (:60-64) plus the rerank (:69-125).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define CRAG_F32 0
#define CRAG_BF16 1
#define CRAG_L2 0
#define CRAG_IP 1

// status words written per query by k_finalize
#define CRAG_ST_OK 0u
#define CRAG_ST_UNCERTIFIED 1u  // exactness certificate failed -> exact fallback scan
#define CRAG_ST_OVERFLOW 2u     // candidate buffer overflowed  -> exact fallback scan

// flags of a candidate entry
#define CRAG_FLAG_CLASS_MASK 3u  // 0 neutral, 1 penalised (x0.1), 2 boosted (x1.2): hybrid_search.py:105-111
#define CRAG_FLAG_SYMBOL 4u      // lowered chunk name is a substring of the lowered query (:114)

// One candidate as it crosses GPUs (16 B): the exact float32 distance Chroma would hand to
// hybrid_search.py:71, the global row id, and the query-independent/-dependent boost bits.
struct CragEntry {
    float dist;
    uint32_t flags;
    long long id;
};

// Queries that share one row range and are scanned together (GEMV regime: up to 8).
struct GroupDesc {
    const void* rows;
    const float* xnorm;
    long long n_rows;
    long long n_sample;  // level 0: rows [0, n_sample) are scored densely
    long long n_mid;     // level 1: rows [0, n_mid) are scanned with tau0 (== n_sample: no level 1)
    int q_first, nq;
    int blk_first[3];  // per pass: 0 dense sample, 1 level-1 filter, 2 full filter
    int n_blk[3];
    // first row of each pass: no row is scanned twice when the earlier levels' survivors stay in the buffer.
    // from[0] = 0; from[1] = n_sample (k_kth seeds the buffer with the sample rows <= tau0);
    // from[2] = n_mid (level-1 candidates kept) | n_sample (no level 1) | 0 (level 1 only set tau)
    long long from[3];
};

// Everything the per-query kernels need (one per query of the batch).
struct QueryDesc {
    const void* rows;
    const float* xnorm;
    const unsigned long long* name_hash;  // may be null
    const unsigned char* path_class;      // may be null
    long long n_rows;
    long long n_sample;
    long long n_mid;
    long long id_base;
    float max_xnorm;
    int keep_mid;  // level-1 candidates stay in the buffer and the full scan starts at n_mid (no double scan)
};

// ---------------------------------------------------------------------------------------
// order-preserving float <-> uint32 key
__host__ __device__ inline uint32_t crag_fkey(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float crag_fkey_inv(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}

// geometry of the streamed row layout: C = 16-byte chunks per (padded) row
__host__ __device__ constexpr int crag_gcd(int a, int b) { return b == 0 ? a : crag_gcd(b, a % b); }
__host__ __device__ constexpr int crag_lcm(int a, int b) { return a / crag_gcd(a, b) * b; }

// candidate buffer and sort limits of the select stage
#define CRAG_CAND_CAP 16384
// per-query candidate counters sit one per 128-byte line: returning atomics on one line serialise in its L2
// channel, and with many queries appending at once (64 collections in one launch) neighbours would queue
#define CRAG_CNT_STRIDE 32
#define CRAG_CNT(q) ((long long)(q) * CRAG_CNT_STRIDE)
#define CRAG_MAX_M 512      // rescored candidates per query (2*top_k + slack) upper limit
#define CRAG_MAX_KC 400     // 2*top_k upper limit
#define CRAG_MAX_GROUP_Q 8  // queries per GEMV group

// launchers implemented in the .hip files ------------------------------------------------
struct ScanArgs {
    const GroupDesc* groups;
    int n_groups;
    const void* queries;  // [B][dim_pad], index dtype
    int dtype;
    int chunks;  // C
    int metric;
    const float* tau;         // [B]
    uint32_t* sample;         // [B][sample_stride] ordered-float keys
    long long sample_stride;
    unsigned long long* cand;  // [B][CRAG_CAND_CAP]
    uint32_t* cnt;             // [B]
};
// pass: 0 dense sample of rows [0,n_sample), 1 filter rows [0,n_mid), 2 filter all rows
int crag_launch_scan_gemv(const ScanArgs& a, int pass, int nq_template, int total_blocks, hipStream_t s);
bool crag_gemv_supported(int dtype, int chunks);

struct MfmaArgs {
    const void* rows;
    const float* xnorm;
    long long n_rows;   // rows scanned by this launch: [row_begin, row_end); row_begin is a multiple of 32
    long long row_begin, row_end;
    const void* queries;  // [Bpad][dim_pad] bf16
    int n_queries;        // real queries
    int dim_pad;
    int metric;
    const float* tau;
    uint32_t* sample;
    long long sample_stride;
    unsigned long long* cand;
    uint32_t* cnt;
    // crag_launch_scan_q16 only: SEGMENTED candidate buffers.  A query's cand_stride slots are cut into
    // n_seg segments of seg_cap, four per workgroup that scans for the query; a lane appends to its own segment
    // with a private counter (segcnt[q * n_seg + seg], loaded at kernel start, stored at its end): no atomics.
    uint32_t* segcnt;
    int n_seg, seg_cap;
    long long cand_stride;  // slots per query = n_seg * seg_cap (may exceed CRAG_CAND_CAP; the COUNT may not)
    // crag_launch_scan_q16 with several query groups (B > 256): pacing words, one 128-byte line per (xcd, stream)
    // pair = CRAG_PROGRESS_WORDS(n_cus) 32-bit words, ZERO at launch (each launch its own); null = no pacing.
    // ahead_max: low byte = tiles a workgroup may run ahead of the slowest one that reads the same tiles; next byte
    // = mask of the tile count at which wave 0 publishes and checks (0: every tile, 7: every eighth)
    uint32_t* progress;
    int ahead_max;
};
// defaults (env CRAG_AHEAD_MAX / CRAG_PACE_EVERY override them for A/B runs).  Batch-1024 top-100, last level, one box
// (profiles/r03_pacing_traffic.txt): ahead 8 reads 12.3 GB per launch (1.22 x the algorithmic 10.1 GB) whatever the
// check interval, ahead <= 6 reads 10.3 GB (1.02 x) -- the drift the XCD's 4-MiB L2 tolerates ends between 6 and 8 tiles;
// checking every 4th tile with ahead 4 costs no time against 8 / 8, checking every 8th with ahead 4 costs 3.5 %.
#define CRAG_AHEAD_MAX_DEFAULT 4
#define CRAG_PACE_EVERY_DEFAULT 4
#define CRAG_PROGRESS_WORDS(n_cus) ((long long)((n_cus) / 8 * 8) * 32)
// query-stationary MFMA scans; their samples are query-minor ([row][stride], or tile-major for crag_scan_q16.hip)
// same contract for 1024-element rows (two K-parts per tile, four LDS buffers): crag_scan_qsplit.hip
bool crag_qsplit_supported(int dim_pad);
int crag_launch_scan_qsplit(const MfmaArgs& a, int mode, int n_cus, hipStream_t s);
// bf16 rows of 256 / 384 / 512 / 768 elements, 256 queries per workgroup (four 16-query blocks per wave,
// v_mfma_f32_16x16x32_bf16), segmented candidate buffers (four segments per workgroup), tile-major sample: crag_scan_q16.hip
bool crag_q16_supported(int dim_pad);
int crag_launch_scan_q16(const MfmaArgs& a, int mode, int n_cus, hipStream_t s);
// segments per query that crag_launch_scan_q16 will use for this batch size (MfmaArgs::n_seg)
int crag_q16_segments(int n_queries, int n_cus);
// its MfmaArgs::queries are the caller's queries scaled by -2 (l2) or -1 (ip): this writes that copy ([n_queries][dim_pad] bf16)
int crag_q16_scale_queries(const void* queries, int n_queries, int dim_pad, int metric, void* out, hipStream_t s);
// bf16 rows too long for a query-stationary form (1536 / 2048 elements) and float32 rows of every length: 128 rows x
// 32 / 64 / 128 queries per workgroup, both operands through LDS; flat candidate buffers, row-major sample: crag_scan_qlong.hip
bool crag_qlong_supported(int dtype, int dim_pad);
int crag_launch_scan_qlong(const MfmaArgs& a, int dtype, int mode, int n_cus, hipStream_t s);


int crag_launch_row_norms(const void* rows, int dtype, int dim_pad, long long n, float* xnorm, float* max_out,
                          hipStream_t s);
// per query: tau = float of the M-th smallest ordered-score key of a strided list
//   element i of query q = src[q*q_stride + off0 + i*e_stride]
//   cnt == null: list length = qd[q].n_sample (dense sample); fewer than M+1 entries -> tau = +inf
//   cnt != null: list length = min(cnt[q], CRAG_CAND_CAP) (candidate buffer); fewer than M -> tau kept
//   cnt_reset != null: cnt_reset[q] = 0 afterwards
struct KthArgs {
    const uint32_t* src;
    long long q_stride, e_stride, off0;
    const QueryDesc* qd;
    const uint32_t* cnt;
    uint32_t* cnt_reset;
    int B, M;
    float* tau;
    uint32_t* segcnt;  // segmented candidate buffer (see MfmaArgs) when n_seg > 0; then cnt / cnt_reset are unused
    int n_seg, seg_cap;
    // level 0 only, may be null: append the sample rows with key <= tau to the (flat) candidate buffer, so
    // that the later passes need not scan rows [0, n_sample) again
    unsigned long long* seed_cand;
    uint32_t* seed_cnt;
    // the same into a SEGMENTED buffer (seed_n_seg > 0): seed j goes to segment j % n_seg, slot j / n_seg, and
    // seed_cnt is the segment counter array -- the scan workgroups then append behind the seeds
    int seed_n_seg, seed_seg_cap;
    long long seed_stride;
    // > 0: src is the tile-major sample of crag_launch_scan_q16 with this Bpad (q_stride, e_stride, off0 unused)
    long long tile_bpad;
};
int crag_launch_kth(const KthArgs& a, hipStream_t s);
struct FinalizeArgs {
    const QueryDesc* qd;
    const void* queries;
    int B, dtype, dim, dim_pad, metric;
    int M, kc;
    const unsigned long long* cand;
    const uint32_t* cnt;
    const float* tau;                 // [B] filter thresholds (certificate bound when the buffer is exhausted)
    int trusted;                      // 1: keys are already exact (fallback path) -> skip the certificate
    const unsigned long long* qhash;  // sorted per query
    const int* qhash_off;             // [B+1], may be null
    CragEntry* entries;               // [B][kc]
    uint32_t* status;                 // [B]
    const uint32_t* segcnt;           // segmented candidate buffer (see MfmaArgs) when n_seg > 0; then cnt is unused
    int n_seg, seg_cap;
    long long cand_stride;            // slots per query in cand (CRAG_CAND_CAP when flat)
    const int* qmap;                  // null: block b serves query b; else block b serves query qmap[b] (fallback passes)
};
int crag_launch_finalize(const FinalizeArgs& a, hipStream_t s);
int crag_launch_merge_rerank(const CragEntry* entries, int G, int B, int kc, int top_k, double symbol_boost,
                             long long* out_ids, double* out_scores, float* out_dist, int* out_counts,
                             hipStream_t s);
// fallback tier 1b: per query of the batch (a.B, a.status readable from the device) the re-scan threshold; clears the
// failed queries' candidate counters (segcnt may be null)
int crag_launch_tau_refine(const FinalizeArgs& a, float* tau, unsigned int* cnt, uint32_t* segcnt, hipStream_t s);
// fallback tier 1: every buffered candidate of the a.qmap[0 .. n_failed) queries re-scored exactly, keys rewritten in place
int crag_launch_rescore_wide(const FinalizeArgs& a, unsigned long long* cand_rw, int n_failed, hipStream_t s);
// fallback tier 2: exact keys of every row for F queries, radix select of the min(kc, n) smallest per query on the
// device (no host round trips), gathered into the queries' flat candidate buffers (cnt must be zero for them)
struct CragRadixState {
    unsigned long long prefix, mask;
    unsigned int want, pad;
};
struct ExactSelectArgs {
    const QueryDesc* qd;   // device, whole batch
    const int* qmap;       // device, [F] query numbers
    int F;
    const void* queries;   // device, whole batch, stride dim_pad
    int dtype, dim, dim_pad, metric, kc;
    long long n_max;       // most rows among the F queries' collections
    unsigned long long* keys;  // [F][key_stride]
    long long key_stride;
    CragRadixState* state;     // [F]
    unsigned int* hist;        // [F][256]
    unsigned long long* cand;
    long long cand_stride;
    unsigned int* cnt;
};
int crag_launch_exact_select(const ExactSelectArgs& a, hipStream_t s);

// Kernels that take more than 64 KiB of dynamic LDS opt in once per (kernel, device): the attribute belongs
// to the device that is current, and one process may hold collections on several devices.  `done` is the
// launcher's own bit mask of devices already set (a lost race only repeats an idempotent call).
inline void crag_allow_dynamic_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    done.fetch_or(bit, std::memory_order_relaxed);
}
