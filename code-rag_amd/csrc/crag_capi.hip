// Host side of the C ABI declared in include/coderag_hip.h (gfx950 only, no CPU fallback:
// every search runs the HIP kernels or fails with CRAG_E_HIP).
//
// Pipeline of one search (replaces hybrid_search.py:60-125 of the reference):
//   level 0         dense scores of the first n0 rows                  (scan kernel, mode 0)
//   k_kth           per query threshold tau = M-th smallest score so far; sample rows <= tau seed the buffer
//   levels 1..L     filter scans of the rows behind the previous level, fused compare-and-append of the
//                   rare s <= tau rows (mode 1), a k_kth between levels; the last level is the long one
//   k_finalize      cut to the M+1 best, exact fp64 re-score of the M best, certificate, 2*top_k entries
//   k_merge_rerank  merge shards, base score, path/symbol boost, filter, stable sort, top_k
//   (+ exact fallback scan for queries whose certificate failed or whose buffer overflowed)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/coderag_hip.h"
#include "crag_common.h"

namespace {

thread_local std::string g_err = "";

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(CRAG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                                        ":" + std::to_string(__LINE__) + ")");                            \
    } while (0)

// Waits for a stream by polling: hipStreamSynchronize parks the thread and its wake-up costs tens of
// microseconds, which is a tenth of a batch-1 search; a search call owns its host thread anyway.
inline hipError_t stream_wait(hipStream_t st) {
    for (int spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (spins > 200000) return hipStreamSynchronize(st);  // a long wait (> ~50 ms): stop burning the core
        __builtin_ia32_pause();
    }
}

// A fixed block of device memory handed out front to back (crag_search_device_noalloc: the block is the
// file-scope `scratch` array, see the end of this file).  Nothing is ever returned to it: a call carves what it
// needs and the next call starts over.
struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    Arena* arena = nullptr;  // set: carve from the arena, never hipMalloc / hipFree
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (arena) {
            const size_t want = (bytes + 255) / 256 * 256;
            if (want > arena->size - arena->used)
                return fail(CRAG_E_LIMIT, "crag_search_device_noalloc: this search needs more than the " +
                                              std::to_string(arena->size >> 20) + " MiB scratch array holds; use crag_search_device");
            p = arena->base + arena->used;
            arena->used += want;
            cap = want;
            return 0;
        }
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(CRAG_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release() {
        if (p && !arena) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Small arrays of descriptors live on the stack (no heap allocation on the search path for batches up to N;
// larger batches spill to the heap).  T is a plain struct.
template <typename T, int N>
struct SmallVec {
    T inl[N];
    std::vector<T> heap;
    size_t n = 0;
    bool on_heap = false;
    T* data() { return on_heap ? heap.data() : inl; }
    const T* data() const { return on_heap ? heap.data() : inl; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void resize(size_t k) {
        if (k > (size_t)N) {
            if (!on_heap) heap.assign(inl, inl + n);
            heap.resize(k);
            on_heap = true;
        }
        n = k;
    }
    void push_back(const T& v) {
        resize(n + 1);
        data()[n - 1] = v;
    }
    T& operator[](size_t i) { return data()[i]; }
    const T& operator[](size_t i) const { return data()[i]; }
    T* begin() { return data(); }
    T* end() { return data() + n; }
    const T* begin() const { return data(); }
    const T* end() const { return data() + n; }
    T& back() { return data()[n - 1]; }
};

struct Workspace {
    int device = 0;
    hipStream_t stream = nullptr;
    Arena* arena = nullptr;  // crag_search_device_noalloc: every buffer below is carved from it, nothing is pinned
    void use_arena(Arena* a) {
        arena = a;
        for (DevBuf* b : {&q, &qscan, &desc, &tau, &cntst, &segcnt, &sample, &cand, &entries, &qhash, &out, &keys, &misc}) b->arena = a;
    }
    DevBuf q, desc, tau, cntst, segcnt, sample, cand, entries, qhash, out, keys, misc;
    DevBuf qscan;  // the 16x16x32 scan's B operand: the queries scaled by -2 / -1 (crag_q16_scale_queries)
    // host-buffer entry points, one chunk of queries: descriptors, staged queries and symbol hashes lie back to back in
    // `pinned` and go up in ONE copy into `stage` (three small H2D copies cost a blit kernel and a gap each);
    // merged_bytes != 0 tells the chunk launcher to send that many bytes and to find its descriptors in `stage`
    DevBuf stage;
    size_t merged_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // scan timing (crag_set_profiling)
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    uint32_t* pinned_status = nullptr;  // certificate read-back (pinned: a pageable D2H copy costs a staging round trip)
    size_t pinned_status_cap = 0;
    int ensure_pinned_status(size_t n) {
        if (n <= pinned_status_cap) return 0;
        if (pinned_status) (void)hipHostFree(pinned_status);
        pinned_status = nullptr;
        pinned_status_cap = 0;
        const size_t want = n * 2 + 64;
        hipError_t e = hipHostMalloc((void**)&pinned_status, want * sizeof(uint32_t), hipHostMallocDefault);
        if (e != hipSuccess) return fail(CRAG_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        pinned_status_cap = want;
        return 0;
    }
    // results of the host-buffer entry points when they are small: the rerank kernel writes them straight into this
    // pinned, device-visible block (its own allocation: `pinned` below may be regrown by a fallback tier mid-call)
    void* pinned_out = nullptr;
    size_t pinned_out_cap = 0;
    int ensure_pinned_out(size_t bytes) {
        if (bytes <= pinned_out_cap) return 0;
        if (pinned_out) (void)hipHostFree(pinned_out);
        pinned_out = nullptr;
        pinned_out_cap = 0;
        const size_t want = bytes * 2 + 4096;
        hipError_t e = hipHostMalloc(&pinned_out, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(CRAG_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        pinned_out_cap = want;
        return 0;
    }
    int ensure_events() {
        for (auto& e : ev)
            if (!e && hipEventCreate(&e) != hipSuccess) return fail(CRAG_E_HIP, "hipEventCreate failed");
        return 0;
    }
    int ensure_pinned(size_t bytes) {
        if (bytes <= pinned_cap) return 0;
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr;
        pinned_cap = 0;
        size_t want = bytes * 2 + 4096;
        hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(CRAG_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        pinned_cap = want;
        return 0;
    }
};

std::mutex g_pool_mu;
std::vector<Workspace*> g_pool;  // free workspaces, any device

Workspace* ws_acquire(int device) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i]->device == device) {
                Workspace* w = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                return w;
            }
    }
    Workspace* w = new Workspace();
    w->device = device;
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
        delete w;
        return nullptr;
    }
    return w;
}
void ws_release(Workspace* w) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(w);
}
struct WsGuard {
    Workspace* w;
    explicit WsGuard(int device) : w(ws_acquire(device)) {}
    ~WsGuard() {
        if (w) ws_release(w);
    }
};

const int kChunkSizes[] = {32, 48, 64, 96, 128, 192, 256};
constexpr int64_t CRAG_MAX_ROWS = (1ll << 32) - 64;  // candidate keys store the row number in 32 bits

inline uint64_t fnv1a(const unsigned char* p, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h ? h : 1ull;
}

inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x40);  // NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

struct crag_index {
    int dim = 0, dim_pad = 0, dtype = 0, metric = 0, device = 0, chunks = 0;
    size_t esz = 4;
    int64_t n_rows = 0, cap_rows = 0;
    int64_t id_base = 0;
    void* rows = nullptr;
    bool owns_rows = true, attached = false;
    float* xnorm = nullptr;
    float max_xnorm = 0.f;
    bool finalized = false;
    uint64_t* name_hash = nullptr;
    uint8_t* path_class = nullptr;
    bool owns_meta = true;
    std::unordered_set<uint64_t> names;
    std::vector<uint8_t> len_present;  // index = byte length
    int max_name_len = 0;
    int force_path = 0;
    uint32_t* na_status = nullptr;  // pinned, device-visible certificates of crag_search_device_noalloc (made by finalize)
    int n_cus = 256;
    int profiling = 0;
    std::mutex prof_mu;
    double prof[4] = {0, 0, 0, 0};  // sample-scan ms, filter-scan ms, searches timed, rows the filter-pass launches scanned
    std::atomic<long long> stats[9];  // [7]: queries that needed the full exact scan (tier 2), [8]: the certified re-scan (tier 1b)
    crag_index() {
        for (auto& s : stats) s = 0;
    }
};

namespace {

int grow(crag_index* h, int64_t need_rows, bool need_hash, bool need_class) {
    if (h->attached) return fail(CRAG_E_STATE, "collection is attached to caller memory; cannot add rows");
    int64_t cap = h->cap_rows;
    bool realloc_rows = need_rows > cap;
    if (realloc_rows) {
        cap = std::max<int64_t>(need_rows, cap + cap / 2);
        cap = std::max<int64_t>(cap, 1024);
    }
    const size_t row_bytes = (size_t)h->dim_pad * h->esz;
    if (realloc_rows) {
        void* nr = nullptr;
        HIPCHK(hipMalloc(&nr, (size_t)cap * row_bytes));
        if (h->n_rows) HIPCHK(hipMemcpy(nr, h->rows, (size_t)h->n_rows * row_bytes, hipMemcpyDeviceToDevice));
        if (h->rows) (void)hipFree(h->rows);
        h->rows = nr;
    }
    if ((need_hash || h->name_hash) && (realloc_rows || !h->name_hash)) {
        uint64_t* nh = nullptr;
        HIPCHK(hipMalloc((void**)&nh, (size_t)cap * 8));
        HIPCHK(hipMemset(nh, 0, (size_t)cap * 8));
        if (h->name_hash && h->n_rows)
            HIPCHK(hipMemcpy(nh, h->name_hash, (size_t)h->n_rows * 8, hipMemcpyDeviceToDevice));
        if (h->name_hash) (void)hipFree(h->name_hash);
        h->name_hash = nh;
    }
    if ((need_class || h->path_class) && (realloc_rows || !h->path_class)) {
        uint8_t* nc = nullptr;
        HIPCHK(hipMalloc((void**)&nc, (size_t)cap));
        HIPCHK(hipMemset(nc, 0, (size_t)cap));
        if (h->path_class && h->n_rows) HIPCHK(hipMemcpy(nc, h->path_class, (size_t)h->n_rows, hipMemcpyDeviceToDevice));
        if (h->path_class) (void)hipFree(h->path_class);
        h->path_class = nc;
    }
    h->cap_rows = cap;
    return 0;
}

int add_common(crag_index* h, const void* rows, int64_t n, const uint64_t* name_hash, const uint8_t* path_class,
               hipMemcpyKind kind) {
    if (!h || n < 0 || (n > 0 && !rows)) return fail(CRAG_E_INVALID, "crag_index_add: bad arguments");
    if (n == 0) return 0;
    if (h->n_rows + n > CRAG_MAX_ROWS) return fail(CRAG_E_LIMIT, "crag_index_add: a collection holds at most 2^32 - 64 rows (candidate keys carry the row in 32 bits)");
    HIPCHK(hipSetDevice(h->device));
    int rc = grow(h, h->n_rows + n, name_hash != nullptr, path_class != nullptr);
    if (rc) return rc;
    const size_t src_pitch = (size_t)h->dim * h->esz, dst_pitch = (size_t)h->dim_pad * h->esz;
    char* dst = (char*)h->rows + (size_t)h->n_rows * dst_pitch;
    if (src_pitch == dst_pitch) {
        HIPCHK(hipMemcpy(dst, rows, (size_t)n * dst_pitch, kind));
    } else {
        HIPCHK(hipMemset(dst, 0, (size_t)n * dst_pitch));
        HIPCHK(hipMemcpy2D(dst, dst_pitch, rows, src_pitch, src_pitch, (size_t)n, kind));
    }
    if (name_hash) HIPCHK(hipMemcpy(h->name_hash + h->n_rows, name_hash, (size_t)n * 8, kind));
    if (path_class) HIPCHK(hipMemcpy(h->path_class + h->n_rows, path_class, (size_t)n, kind));
    h->n_rows += n;
    h->finalized = false;
    return 0;
}

// Threshold cascade (DESIGN.md "Selection"): level 0 = dense scores of the first n0 rows, level 1 =
// filtered scan of the first n1 rows with tau0, level 2 = filtered scan of all rows with tau1.
// tau_l = M-th smallest score of level l's rows, so at least M rows pass every filter; expected
// candidates per query: M*n1/n0 at level 1 and M*N/n1 at level 2, balanced by n1 = sqrt(n0*N).
long long level0_rows(long long N, int M) { return std::min<long long>(N, M <= 64 ? 8192 : 16384); }
// `direct_max`: expected survivors per query up to which the full scan runs straight behind the sample.  A level
// costs a launch and a selection (~30 us on the GEMV path); a survivor costs the GEMV a returning atomic that
// stalls its wave (measured 5-7 ns each with many queries appending, ~15 ns on one query's counter), so what
// decides is the survivors of the whole batch: same box, 64 collections x 156 k rows (650 survivors each) run
// 2 % faster WITH level 1, 32 x 312 k 5 % faster, one query over 1M rows 8 % faster.
long long level1_rows(long long N, long long n0, int M, double direct_max) {
    if (N <= n0) return n0;
    if ((double)M * (double)N / (double)n0 <= direct_max) return n0;  // level 2 alone is cheap enough
    long long n1 = (long long)std::sqrt((double)n0 * (double)N);
    n1 = (n1 + 127) / 128 * 128;
    return std::max(n0, std::min(n1, N));
}

// Filter levels of the multi-level cascade (segmented MFMA path): ascending row bounds, the last one == N;
// level l scans rows [bound[l-1], bound[l]) (level 0 of the cascade, the dense sample, ends at n0).  Bounds grow
// geometrically, n0 r^l, so every level is expected to add M r survivors; a level costs a launch and a
// selection (~50 us), its survivors cost the filter's slow path on the tiles that hold them.  Same-box sweeps
// (tools/sweep_tunables.sh history in DESIGN.md): at top-10 (M = 36) r between 3 and 6 is flat within 0.5 %, so
// r ~ 6 (fewest launches), and up to r = 20 it stays within 0.5 % (round 2: a level saved is paid back by the
// longer last level and its looser threshold); at top-100 (M = 225) r ~ 3 is 1.5-2 % ahead of r ~ 6 and of r ~ 2.
constexpr int CRAG_MAX_LEVELS = 8;
typedef SmallVec<long long, CRAG_MAX_LEVELS> LevelBounds;
LevelBounds cascade_bounds(long long N, long long n0, int M) {
    LevelBounds nb;
    if (N <= n0) return nb;
    const double R = (double)N / (double)n0;
    constexpr int L_MAX = CRAG_MAX_LEVELS;
    static const double r_small = [] { const char* e = getenv("CRAG_CASCADE_R10"); return e && atof(e) > 1.1 ? atof(e) : 6.0; }();
    static const double r_large = [] { const char* e = getenv("CRAG_CASCADE_R100"); return e && atof(e) > 1.1 ? atof(e) : 3.0; }();
    int L = (int)std::ceil(std::log(R) / std::log(M > 64 ? r_large : r_small));  // (the two variables: same-box sweeps only)
    L = std::max(1, std::min(L, L_MAX));
    auto est = [&](int l) { return (double)l * (double)M * std::pow(R, 1.0 / l); };
    while (L < L_MAX && est(L) > 0.75 * CRAG_CAND_CAP && est(L + 1) < est(L)) ++L;
    const double r = std::pow(R, 1.0 / L);
    double x = (double)n0;
    for (int l = 1; l < L; ++l) {
        x *= r;
        long long b = ((long long)x + 127) / 128 * 128;
        const long long prev = nb.empty() ? n0 : nb.back();
        if (b > prev && b < N) nb.push_back(b);
    }
    nb.push_back(N);
    return nb;
}

struct Plan {
    int B = 0, top_k = 0, kc = 0, M = 0;
    bool mfma = false;
    int nq_template = 1;
    int n_groups = 0;
    int blocks[3] = {0, 0, 0};
    long long sample_stride = 0;
    bool need_mid = false;
};

// GEMV work list: blocks of each group for one pass (0 dense sample, 1 level-1 filter, 2 full filter), shares of the
// chip in proportion to the rows the pass scans for the group.  Fills blk_first / n_blk of the pass; returns the grid.
int gemv_partition(SmallVec<GroupDesc, 64>& gd, int pass, int nq_template, int n_cus) {
    const int RT = 64 / nq_template;
    const long long max_waves = (long long)n_cus * 12;
    auto rows_of = [&](const GroupDesc& g) {
        if (pass == 0) return g.n_sample;
        if (pass == 1) return g.n_mid > g.n_sample ? g.n_mid - g.from[1] : 0LL;  // level 1 only where it exists
        return g.n_rows - g.from[2];
    };
    long long total_st = 0;
    for (auto& g : gd) total_st += (rows_of(g) + RT - 1) / RT;
    int blk = 0;
    for (auto& g : gd) {
        const long long stg = (rows_of(g) + RT - 1) / RT;
        g.blk_first[pass] = blk;
        if (stg == 0) {
            g.n_blk[pass] = 0;
            continue;
        }
        long long share = std::max<long long>(4, max_waves * stg / std::max<long long>(total_st, 1));
        long long waves = std::min(stg, share);
        const long long T = (stg + waves - 1) / waves;
        waves = (stg + T - 1) / T;
        g.n_blk[pass] = (int)((waves + 3) / 4);
        blk += g.n_blk[pass];
    }
    return blk;
}

// where the single-GPU entry points want the reranked results (stage 2 launched optimistically right
// behind stage 1, before the host looks at the certificates; re-launched if a fallback patched entries)
struct MergeOut {
    double symbol_boost;
    long long* ids;
    double* scores;
    float* dist;
    int* counts;
};

// CRAG_NOALLOC_MAX_B (include/coderag_hip.h): batch size limit of crag_search_device_noalloc; its scratch array is
// sized for it, see the end of this file
// descriptors to device memory as kernel arguments (no pinned staging buffer): one QueryDesc replicated B times
int put_descriptors(const QueryDesc& qd, const GroupDesc* gd, int n_groups, int B, QueryDesc* qd_dev, GroupDesc* gd_dev,
                    hipStream_t st);

// one batch on one device: queries already on the device in index dtype, row stride dim_pad
int search_entries_chunk(crag_index* const* hs, bool same_handle, const void* q_dev, int B, int top_k,
                         const uint64_t* qh_dev, const int32_t* qhoff_dev, CragEntry* entries_dev, Workspace* ws,
                         hipStream_t st, const MergeOut* mo);

// Batches of any size: the kernels take at most CRAG_MAX_CHUNK queries per launch sequence (query groups per
// launch, candidate-buffer memory), so larger batches run as consecutive chunks on the same stream.
constexpr int CRAG_MAX_CHUNK = 2048;
// host-buffer entry points: result sets up to this size are written by the GPU straight into pinned host memory
constexpr size_t CRAG_DIRECT_OUT_BYTES = 256 * 1024;
// batches from this size up take the LDS-tiled scan (crag_scan_qlong.hip) instead of GEMV groups: one group of 8 is
// the GEMV's best case, a second group reads the corpus a second time (table in search_entries_chunk_launch)
constexpr int CRAG_QLONG_FROM = 9;      // bf16 rows of 1536 / 2048 elements
constexpr int CRAG_QLONG_FROM_F32 = 9;  // float32 rows
int search_entries(crag_index* const* hs, bool same_handle, const void* q_dev, int B, int top_k,
                   const uint64_t* qh_dev, const int32_t* qhoff_dev, CragEntry* entries_dev, Workspace* ws,
                   hipStream_t st, const MergeOut* mo = nullptr) {
    for (int off = 0; off < B; off += CRAG_MAX_CHUNK) {
        const int nb = std::min(CRAG_MAX_CHUNK, B - off);
        const crag_index* h0 = hs[off];
        MergeOut m;
        if (mo) {
            m = *mo;
            m.ids += (size_t)off * top_k;
            m.scores += (size_t)off * top_k;
            m.dist += (size_t)off * top_k;
            m.counts += off;
        }
        const int rc = search_entries_chunk(hs + off, same_handle, (const char*)q_dev + (size_t)off * h0->dim_pad * h0->esz, nb, top_k,
                                            qh_dev, qhoff_dev ? qhoff_dev + off : nullptr, entries_dev + (size_t)off * 2 * top_k,
                                            ws, st, mo ? &m : nullptr);
        if (rc) return rc;
    }
    return 0;
}

int search_entries_chunk_launch(crag_index* const* hs, bool same_handle, const void* q_dev, int B, int top_k,
                                const uint64_t* qh_dev, const int32_t* qhoff_dev, CragEntry* entries_dev, Workspace* ws,
                                hipStream_t st, const MergeOut* mo);

// An error return after the first launch must not hand the workspace back to the pool (or the caller's output
// buffers back to the caller) while kernels may still be reading and writing them: wait for the stream first.
int search_entries_chunk(crag_index* const* hs, bool same_handle, const void* q_dev, int B, int top_k,
                         const uint64_t* qh_dev, const int32_t* qhoff_dev, CragEntry* entries_dev, Workspace* ws,
                         hipStream_t st, const MergeOut* mo) {
    const int rc = search_entries_chunk_launch(hs, same_handle, q_dev, B, top_k, qh_dev, qhoff_dev, entries_dev, ws, st, mo);
    if (rc) {
        const std::string keep = g_err;  // the wait must not replace the message of what went wrong
        (void)hipStreamSynchronize(st);
        g_err = keep;
    }
    return rc;
}

int search_entries_chunk_launch(crag_index* const* hs, bool same_handle, const void* q_dev, int B, int top_k,
                                const uint64_t* qh_dev, const int32_t* qhoff_dev, CragEntry* entries_dev, Workspace* ws,
                                hipStream_t st, const MergeOut* mo) {
    crag_index* h0 = hs[0];
    Plan P;
    P.B = B;
    P.top_k = top_k;
    P.kc = 2 * top_k;  // hybrid_search.py:62
    // slack rows behind the 2*top_k-th: what the certificate may lose to approximate-score misordering before a
    // query has to take the exact fallback (kc / 8 was 1.5 % ahead of kc / 4 at top-100; no fallbacks either way)
    P.M = P.kc + std::max(16, P.kc / 8);
    if (P.M > CRAG_MAX_M) P.M = CRAG_MAX_M;

    // ---- which scan: GEMV groups of up to 8 queries, or one of the MFMA scans (one collection per call only) ----
    //   bf16 up to 768 elements   query-stationary cascade (crag_scan_q16.hip)
    //   bf16 1024 elements        split-K query-stationary scan (crag_scan_qsplit.hip)
    //   bf16 1536 / 2048, float32 LDS-tiled scan (crag_scan_qlong.hip); a hand-scheduled query-stationary float32 scan
    //                             served rows up to 384 elements until round 2 -- the tiled scan in plain HIP was 3-10 %
    //                             ahead of it at every row length and batch size (tools/cmp_f32_scans.py) and it went
    // The smallest batch that takes the MFMA scan in auto mode, each from a same-box comparison (ms per batch, GEMV / MFMA):
    //   bf16 256:  5   (tools/sweep_small_batch.py, 4M rows: 4 queries 0.43 / 0.49, 6 queries 0.56 / 0.50)
    //   bf16 384:  5   (4 queries 0.59 / 0.62, 6 queries 0.83 / 0.62); 3 for both when M > 64 (more appends per GEMV pass)
    //   bf16 512:  3   (2 queries 0.70 / 0.72, 3 queries 0.74 / 0.72)
    //   bf16 768:  1 on 4M rows and more (0.99 / 0.96; 10M rows 2.34 / 2.29), else 2 (the cascade's two extra launches
    //              weigh more on short scans: the GEMV is ahead below ~3M rows; 2 queries 1.04 / 0.96)
    //   bf16 1024: 1   (nontemporal DMA; 1M / 3M / 10M rows 0.46 / 0.42, 1.13 / 1.00, 3.42 / 2.99)
    //   bf16 1536 / 2048: 9  (two GEMV groups 2.8 ms against one pass of the tiled scan 1.8 ms on 3M x 1536)
    //   float32:   9, i.e. one GEMV group at most (tools/cmp_f32_small_batch.py, 4 GB: 384 elements 8 queries 0.75 / 0.74,
    //              12 queries 1.40 / 0.74); 5 on rows up to 192 elements, where the 8-query GEMV variant is
    //              VALU-bound (6-8 queries 0.90-1.0 / 0.75)
    const int path = h0->force_path;  // 0 auto, 1 GEMV, 2 MFMA, 3 tiled MFMA (tests, bench, tools)
    const bool bf16 = h0->dtype == CRAG_BF16;
    const bool can_q16 = bf16 && crag_q16_supported(h0->dim_pad), can_qsplit = bf16 && crag_qsplit_supported(h0->dim_pad);
    const bool can_qlong = crag_qlong_supported(h0->dtype, h0->dim_pad);
    int mfma_from;
    if (can_qsplit) mfma_from = 1;
    else if (can_q16) mfma_from = h0->dim_pad >= 768 ? (h0->n_rows >= 4000000 ? 1 : 2) : (h0->dim_pad >= 512 || P.M > 64 ? 3 : 5);
    else if (bf16) mfma_from = CRAG_QLONG_FROM;
    else mfma_from = h0->dim_pad <= 192 ? 5 : CRAG_QLONG_FROM_F32;
    P.mfma = same_handle && (can_q16 || can_qsplit || can_qlong) && (path >= 2 || (path == 0 && B >= mfma_from));

    const bool qsplit = P.mfma && can_qsplit;  // 1024-element rows: flat buffers, row-major sample
    // (the LDS-tiled scan, crag_scan_qlong.hip: flat buffers and a row-major sample as well)
    // the two-block MFMA scan appends to per-workgroup segments of the candidate buffers (no atomics); the
    // one-block form (1024-element rows) and the tiled scan keep flat buffers and returning atomics.  Every
    // MFMA scan runs the multi-level cascade in which a level continues where the previous one
    // stopped.  (A one-block scan for bf16 rows up to 768 elements served two queries at top-10 until round 2;
    // the two-block cascade was within 1 % of it there, same box, and it went.)
    const bool segmented = P.mfma && can_q16;
    const bool cascade = P.mfma;
    // ---- descriptors ----
    SmallVec<QueryDesc, 64> qd;
    qd.resize((size_t)B);
    SmallVec<GroupDesc, 64> gd;
    for (int b = 0; b < B; ++b) {
        crag_index* h = hs[b];
        QueryDesc& d = qd[(size_t)b];
        d.rows = h->rows;
        d.xnorm = h->xnorm;
        d.name_hash = (const unsigned long long*)h->name_hash;
        d.path_class = h->path_class;
        d.n_rows = h->n_rows;
        d.n_sample = level0_rows(h->n_rows, P.M);
        d.n_mid = level1_rows(h->n_rows, d.n_sample, P.M, P.mfma ? CRAG_CAND_CAP / 8 : std::max(32.0, 2048.0 / B));
        d.id_base = h->id_base;
        d.max_xnorm = h->max_xnorm;
        // keep the level-1 candidates and let the full scan start at n_mid when both levels fit the buffer
        // with a 2x margin (expected survivors: M n1/n0 + M N/n1); otherwise level 1 only sets tau
        d.keep_mid = 0;
        if (d.n_mid > d.n_sample && d.n_mid < d.n_rows && d.n_mid % 128 == 0) {
            const double est = (double)P.M * ((double)d.n_mid / (double)d.n_sample + (double)d.n_rows / (double)d.n_mid);
            d.keep_mid = est <= CRAG_CAND_CAP / 2 ? 1 : 0;
        }
        if (cascade) d.keep_mid = 1;  // survivors of every level stay in the buffer
        P.sample_stride = std::max<long long>(P.sample_stride, d.n_sample);
        P.need_mid = P.need_mid || d.n_mid > d.n_sample;
    }
    P.sample_stride = (P.sample_stride + 63) / 64 * 64;
    if (!P.mfma) {
        int b = 0;
        int max_nq = 1;
        while (b < B) {
            int e = b + 1;
            while (e < B && e - b < CRAG_MAX_GROUP_Q && hs[e] == hs[b]) ++e;
            GroupDesc g;
            memset(&g, 0, sizeof(g));
            g.rows = hs[b]->rows;
            g.xnorm = hs[b]->xnorm;
            g.n_rows = hs[b]->n_rows;
            g.n_sample = qd[(size_t)b].n_sample;
            g.n_mid = qd[(size_t)b].n_mid;
            g.q_first = b;
            g.nq = e - b;
            // same collection => same decisions for the whole group
            const bool has_mid = g.n_mid > g.n_sample;
            g.from[0] = 0;
            g.from[1] = g.n_sample;
            g.from[2] = has_mid ? (qd[(size_t)b].keep_mid ? g.n_mid : 0) : g.n_sample;
            max_nq = std::max(max_nq, g.nq);
            gd.push_back(g);
            b = e;
        }
        P.nq_template = max_nq <= 1 ? 1 : (max_nq <= 2 ? 2 : (max_nq <= 4 ? 4 : 8));
        for (int pass = 0; pass < 3; ++pass) P.blocks[pass] = gemv_partition(gd, pass, P.nq_template, h0->n_cus);
    }
    P.n_groups = (int)gd.size();


    // ---- workspace ----
    const size_t desc_bytes = sizeof(QueryDesc) * (size_t)B + sizeof(GroupDesc) * gd.size();
    int rc;
    const bool noalloc = ws->arena != nullptr;  // crag_search_device_noalloc: no pinned staging, no fallback memory
    if (!noalloc && (rc = ws->ensure_pinned(desc_bytes))) return rc;
    if ((rc = ws->desc.ensure(desc_bytes))) return rc;
    if ((rc = ws->tau.ensure(sizeof(float) * (size_t)B))) return rc;
    if ((rc = ws->cntst.ensure(sizeof(uint32_t) * ((size_t)B * CRAG_CNT_STRIDE + (size_t)B)))) return rc;
    const bool query_minor = P.mfma;  // every MFMA scan writes a query-minor sample: [row][bpad] or tile-major
    const long long bpad = ((long long)B + 15) / 16 * 16;
    if ((rc = ws->sample.ensure(sizeof(uint32_t) * (size_t)std::max<long long>(B, bpad) * (size_t)P.sample_stride))) return rc;
    const bool merged = !noalloc && ws->merged_bytes != 0;
    char* desc_dev = merged ? (char*)ws->stage.p : (char*)ws->desc.p;
    const QueryDesc* qd_dev = (const QueryDesc*)desc_dev;
    const GroupDesc* gd_dev = (const GroupDesc*)(desc_dev + sizeof(QueryDesc) * (size_t)B);
    if (noalloc) {
        // one collection, so one QueryDesc for all queries and at most B / 8 groups: they travel as kernel arguments
        if ((rc = put_descriptors(qd[0], gd.data(), (int)gd.size(), B, (QueryDesc*)ws->desc.p, (GroupDesc*)gd_dev, st))) return rc;
    } else {
        memcpy(ws->pinned, qd.data(), sizeof(QueryDesc) * (size_t)B);
        if (!gd.empty()) memcpy((char*)ws->pinned + sizeof(QueryDesc) * (size_t)B, gd.data(), sizeof(GroupDesc) * gd.size());
        HIPCHK(hipMemcpyAsync(desc_dev, ws->pinned, merged ? ws->merged_bytes : desc_bytes, hipMemcpyHostToDevice, st));
    }
    uint32_t* cnt = (uint32_t*)ws->cntst.p;
    float* tau = (float*)ws->tau.p;
    uint32_t* sample = (uint32_t*)ws->sample.p;
    unsigned long long* cand = nullptr;
    // no memset of the counters: level 0's selection (k_kth) writes every query's counter -- flat or per segment --
    // when it seeds the buffers, and k_finalize writes every query's status
    const int n_seg = segmented ? crag_q16_segments(B, h0->n_cus) : 0;
    // physical room for twice CRAG_CAND_CAP per query (a segment holds >= 32): the COUNT must stay <= CAP
    const int seg_cap = segmented ? std::max(32, 2 * CRAG_CAND_CAP / n_seg) : 0;
    const long long cand_stride = segmented ? (long long)n_seg * seg_cap : (long long)CRAG_CAND_CAP;
    if ((rc = ws->cand.ensure(sizeof(unsigned long long) * (size_t)B * (size_t)cand_stride))) return rc;
    cand = (unsigned long long*)ws->cand.p;
    uint32_t* segcnt = nullptr;
    // pacing words of the two-block scan's filter levels when the batch has several query groups (B > 256): one
    // zeroed block per level behind the segment counters (MfmaArgs::progress)
    static const int ahead_max = [] {
        const char* e = getenv("CRAG_AHEAD_MAX");  // tuning / A-B runs; 0 switches the pacing off
        const char* v = getenv("CRAG_PACE_EVERY");  // 1, 2, 4, 8 ...: wave 0 publishes and checks every so many tiles
        int every = v ? atoi(v) : CRAG_PACE_EVERY_DEFAULT;
        if (every < 1 || every > 128 || (every & (every - 1))) every = CRAG_PACE_EVERY_DEFAULT;
        return ((e ? atoi(e) : CRAG_AHEAD_MAX_DEFAULT) & 0xff) | ((every - 1) << 8);  // MfmaArgs::ahead_max: tiles | mask << 8
    }();
    const bool paced = segmented && B > 256 && (ahead_max & 0xff) > 0;
    const size_t prog_words = paced ? (size_t)CRAG_PROGRESS_WORDS(h0->n_cus) : 0;
    uint32_t* progress = nullptr;
    if (segmented) {
        const size_t seg_words = ((size_t)B * (size_t)n_seg + 31) / 32 * 32;  // keeps the pacing lines 128-byte aligned
        const size_t words = seg_words + prog_words * CRAG_MAX_LEVELS;
        if ((rc = ws->segcnt.ensure(sizeof(uint32_t) * words))) return rc;
        segcnt = (uint32_t*)ws->segcnt.p;
        if (paced) progress = segcnt + seg_words;
        if (paced) HIPCHK(hipMemsetAsync(progress, 0, sizeof(uint32_t) * prog_words * CRAG_MAX_LEVELS, st));  // pacing words start at zero
    }

    // ---- passes ----
    const bool prof = h0->profiling != 0 && !noalloc;  // events are created on demand: not in the no-alloc call
    if (prof && (rc = ws->ensure_events())) return rc;
    MfmaArgs ma;
    ScanArgs ga;
    if (P.mfma) {
        ma.rows = h0->rows;
        ma.xnorm = h0->xnorm;
        ma.n_rows = h0->n_rows;
        ma.queries = q_dev;
        if (segmented || qsplit) {  // their accumulators are the scores: B operand = -2 q (l2) / -q (ip), once per search
            if ((rc = ws->qscan.ensure((size_t)B * h0->dim_pad * 2))) return rc;
            if (crag_q16_scale_queries(q_dev, B, h0->dim_pad, h0->metric, ws->qscan.p, st)) return fail(CRAG_E_HIP, "query scaling launch failed");
            ma.queries = ws->qscan.p;
        }
        ma.n_queries = B;
        ma.dim_pad = h0->dim_pad;
        ma.metric = h0->metric;
        ma.tau = tau;
        ma.sample = sample;
        ma.sample_stride = query_minor ? bpad : P.sample_stride;
        ma.cand = cand;
        ma.cnt = cnt;
        ma.segcnt = segcnt;
        ma.n_seg = n_seg;
        ma.seg_cap = seg_cap;
        ma.cand_stride = cand_stride;
        ma.row_begin = 0;
        ma.progress = nullptr;
        ma.ahead_max = ahead_max;
        h0->stats[5]++;
    } else {
        ga.groups = gd_dev;
        ga.n_groups = P.n_groups;
        ga.queries = q_dev;
        ga.dtype = h0->dtype;
        ga.chunks = h0->chunks;
        ga.metric = h0->metric;
        ga.tau = tau;
        ga.sample = sample;
        ga.sample_stride = P.sample_stride;
        ga.cand = cand;
        ga.cnt = cnt;
        h0->stats[6]++;
    }
    auto scan = [&](int pass) -> int {  // GEMV passes (the MFMA scans run the cascade below)
        if (P.blocks[pass] == 0) return 0;  // every row of every group is already behind an earlier level
        return crag_launch_scan_gemv(ga, pass, P.nq_template, P.blocks[pass], st);
    };
    if (prof) HIPCHK(hipEventRecord(ws->ev[0], st));
    double prof_rows = 0.0;  // rows the last (timed) filter launch scans
    if (cascade) {
        auto launch_level = [&](int mode) -> int {
            if (segmented) return crag_launch_scan_q16(ma, mode, h0->n_cus, st);
            if (qsplit) return crag_launch_scan_qsplit(ma, mode, h0->n_cus, st);
            return crag_launch_scan_qlong(ma, h0->dtype, mode, h0->n_cus, st);
        };
        // level 0: dense sample of rows [0, n0) -> tau0; its rows at or below tau0 seed the candidate buffers
        const long long N = h0->n_rows, n0 = qd[0].n_sample;
        ma.row_begin = 0;
        ma.row_end = n0;
        if (launch_level(0)) return fail(CRAG_E_HIP, "sample scan launch failed");
        KthArgs k;
        memset(&k, 0, sizeof(k));
        k.src = sample;
        if (segmented) {
            k.tile_bpad = bpad;  // the two-block scan writes a tile-major sample
        } else {
            k.q_stride = 1;      // the one-block scans write it row-major: [row][bpad]
            k.e_stride = bpad;
        }
        k.qd = qd_dev;
        k.B = B;
        k.M = P.M;
        k.tau = tau;
        k.seed_cand = cand;
        k.seed_cnt = segmented ? segcnt : cnt;
        k.seed_n_seg = n_seg;
        k.seed_seg_cap = seg_cap;
        k.seed_stride = cand_stride;
        if (crag_launch_kth(k, st)) return fail(CRAG_E_HIP, "kth (level 0) launch failed");
        // filter levels [n0, n1), [n1, n2), ..., [.., N): each tightens tau from everything found so far
        const LevelBounds nb = cascade_bounds(N, n0, P.M);
        long long from = n0;
        if (nb.empty() && prof) {
            HIPCHK(hipEventRecord(ws->ev[1], st));
            HIPCHK(hipEventRecord(ws->ev[2], st));
            HIPCHK(hipEventRecord(ws->ev[3], st));
        }
        for (size_t l = 0; l < nb.size(); ++l) {
            const bool last = l + 1 == nb.size();
            if (last && prof) {
                HIPCHK(hipEventRecord(ws->ev[1], st));
                HIPCHK(hipEventRecord(ws->ev[2], st));
            }
            ma.row_begin = from;
            ma.row_end = nb[l];
            ma.progress = paced ? progress + l * prog_words : nullptr;
            if (launch_level(1)) return fail(CRAG_E_HIP, "filter scan launch failed");
            if (last) {
                if (prof) HIPCHK(hipEventRecord(ws->ev[3], st));
                prof_rows = (double)(nb[l] - from);
            } else {
                KthArgs k1;
                memset(&k1, 0, sizeof(k1));
                k1.src = (const uint32_t*)cand;
                k1.q_stride = 2LL * cand_stride;
                k1.e_stride = 2;
                k1.off0 = 1;  // high word of the 64-bit key = ordered score
                k1.qd = qd_dev;
                k1.B = B;
                k1.M = P.M;
                k1.tau = tau;
                if (segmented) {
                    k1.segcnt = segcnt;
                    k1.n_seg = n_seg;
                    k1.seg_cap = seg_cap;
                } else {
                    k1.cnt = cnt;  // flat buffer: everything found so far, nothing reset (keep_mid)
                }
                if (crag_launch_kth(k1, st)) return fail(CRAG_E_HIP, "kth (filter level) launch failed");
            }
            from = nb[l];
        }
    } else {
        // GEMV groups: level 0 = dense sample -> tau0, its rows at or below tau0 seed the buffers
        if (scan(0)) return fail(CRAG_E_HIP, "sample scan launch failed");
        {
            KthArgs k;
            memset(&k, 0, sizeof(k));
            k.src = sample;
            k.q_stride = P.sample_stride;
            k.e_stride = 1;
            k.qd = qd_dev;
            k.B = B;
            k.M = P.M;
            k.tau = tau;
            k.seed_cand = cand;  // later passes continue behind the sample rows
            k.seed_cnt = cnt;
            if (crag_launch_kth(k, st)) return fail(CRAG_E_HIP, "kth (level 0) launch failed");
        }
        // level 1: filtered scan of rows [n0, n1) -> tau1 (tightens tau0; never loosens it)
        if (P.need_mid) {
            if (scan(1)) return fail(CRAG_E_HIP, "level-1 scan launch failed");
            KthArgs k;
            memset(&k, 0, sizeof(k));
            k.src = (const uint32_t*)cand;
            k.q_stride = 2LL * cand_stride;
            k.e_stride = 2;
            k.off0 = 1;  // high word of the 64-bit key = ordered score
            k.qd = qd_dev;
            k.cnt = cnt;
            k.cnt_reset = cnt;
            k.B = B;
            k.M = P.M;
            k.tau = tau;
            if (crag_launch_kth(k, st)) return fail(CRAG_E_HIP, "kth (level 1) launch failed");
        }
        if (prof) HIPCHK(hipEventRecord(ws->ev[1], st));
        // level 2: the full scan
        if (prof) HIPCHK(hipEventRecord(ws->ev[2], st));
        if (scan(2)) return fail(CRAG_E_HIP, "full scan launch failed");
        if (prof) HIPCHK(hipEventRecord(ws->ev[3], st));
        for (const auto& g : gd) prof_rows += (double)(g.n_rows - g.from[2]);
    }
    FinalizeArgs f;
    f.qd = qd_dev;
    f.queries = q_dev;
    f.B = B;
    f.dtype = h0->dtype;
    f.dim = h0->dim;
    f.dim_pad = h0->dim_pad;
    f.metric = h0->metric;
    f.M = P.M;
    f.kc = P.kc;
    f.cand = (const unsigned long long*)ws->cand.p;
    f.cnt = cnt;
    f.tau = (const float*)ws->tau.p;
    f.trusted = 0;
    f.qhash = (const unsigned long long*)qh_dev;
    f.qhash_off = (const int*)qhoff_dev;
    f.entries = entries_dev;
    // the certificates go straight to pinned host memory (device-visible): no copy-engine trip after the last kernel
    uint32_t* hst;
    if (noalloc) {
        // the handle's own pinned block (crag_index_finalize made it: nothing is allocated here, and no pageable D2H
        // copy, which the runtime would stage and wait for, follows the last kernel)
        if (!h0->na_status) return fail(CRAG_E_STATE, "crag_search_device_noalloc: the collection was not finalized");
        hst = h0->na_status;
        f.status = hst;
    } else {
        if ((rc = ws->ensure_pinned_status((size_t)B))) return rc;
        hst = ws->pinned_status;
        f.status = hst;
    }
    f.segcnt = segcnt;
    f.n_seg = n_seg;
    f.seg_cap = seg_cap;
    f.cand_stride = cand_stride;
    f.qmap = nullptr;
    if ((rc = crag_launch_finalize(f, st))) return fail(CRAG_E_HIP, "finalize launch failed");
    auto merge = [&]() -> int {
        if (!mo) return 0;
        return crag_launch_merge_rerank(entries_dev, 1, B, P.kc, top_k, mo->symbol_boost, mo->ids, mo->scores, mo->dist,
                                        mo->counts, st);
    };
    if (merge()) return fail(CRAG_E_HIP, "merge_rerank launch failed");
    bool patched = false;

    // ---- verify: every query must carry an exactness certificate ----
    HIPCHK(stream_wait(st));
    if (prof) {
        float ms0 = 0.f, ms1 = 0.f;
        HIPCHK(hipEventElapsedTime(&ms0, ws->ev[0], ws->ev[1]));
        HIPCHK(hipEventElapsedTime(&ms1, ws->ev[2], ws->ev[3]));
        std::lock_guard<std::mutex> lk(h0->prof_mu);
        h0->prof[0] += ms0;
        h0->prof[1] += ms1;
        h0->prof[2] += 1.0;
        h0->prof[3] += prof_rows;
    }
    h0->stats[0]++;
    h0->stats[1] += B;
    // ---- fallback for the queries that came back without a certificate ----
    SmallVec<int, 64> failed;
    for (int b = 0; b < B; ++b) {
        if (hst[(size_t)b] == CRAG_ST_OK) continue;
        h0->stats[2]++;
        if (hst[(size_t)b] == CRAG_ST_UNCERTIFIED) h0->stats[3]++;
        if (hst[(size_t)b] == CRAG_ST_OVERFLOW) h0->stats[4]++;
        failed.push_back(b);
    }
    if (!failed.empty()) {
        if (noalloc)  // the exact scan needs 8 bytes per row of the collection: not in a fixed scratch array
            return fail(CRAG_E_INEXACT, "crag_search_device_noalloc: a query failed its exactness certificate and needs the exact "
                                        "fallback scan (8 bytes of workspace per row); call crag_search_device for this batch");
        // device list of the failed queries (tier 1 takes the uncertified ones first, tier 2 whatever is left)
        const size_t nf = failed.size();
        if ((rc = ws->ensure_pinned(sizeof(int) * nf))) return rc;
        if ((rc = ws->misc.ensure(sizeof(int) * nf + 256))) return rc;
        int* qmap_dev = (int*)ws->misc.p;
        auto upload_qmap = [&](const SmallVec<int, 64>& v) -> int {
            memcpy(ws->pinned, v.data(), sizeof(int) * v.size());
            HIPCHK(hipMemcpyAsync(qmap_dev, ws->pinned, sizeof(int) * v.size(), hipMemcpyHostToDevice, st));
            return 0;
        };
        // tier 1: the buffer of an uncertified query holds EVERY row with approximate score <= tau; re-score all of
        // it exactly (not just the M best) and certify against tau alone.  Covers duplicated chunks and near-ties
        // beyond the slack at the cost of a few thousand row reads per query.
        SmallVec<int, 64> t1;
        for (int b : failed)
            if (hst[(size_t)b] == CRAG_ST_UNCERTIFIED) t1.push_back(b);
        if (!t1.empty()) {
            if ((rc = upload_qmap(t1))) return rc;
            FinalizeArgs g = f;
            g.qmap = qmap_dev;
            g.B = (int)t1.size();
            g.trusted = 2;
            if (crag_launch_rescore_wide(g, (unsigned long long*)ws->cand.p, g.B, st)) return fail(CRAG_E_HIP, "wide re-score launch failed");
            if ((rc = crag_launch_finalize(g, st))) return fail(CRAG_E_HIP, "tier-1 finalize launch failed");
            HIPCHK(stream_wait(st));
            patched = true;
        }
        // tier 1b: certified re-scan.  The kc exact distances a failed query already has bound its true kc-th nearest
        // from above; one more pass over ALL rows with a threshold derived from that bound collects every row that
        // can enter or tie into the result, and re-scoring all of them exactly settles it.  One scan for the whole
        // batch (the queries that were fine append nothing), whatever the number of ties -- unless more than the
        // buffer holds (16384) lie within the float32 error of the boundary, which is left to tier 2.
        SmallVec<int, 64> t1b;
        for (int b : failed)
            if (hst[(size_t)b] != CRAG_ST_OK) t1b.push_back(b);
        if (!t1b.empty()) {
            FinalizeArgs r = f;  // a.status: the pinned certificates, readable from the device
            if (crag_launch_tau_refine(r, tau, cnt, segcnt, st)) return fail(CRAG_E_HIP, "tau refine launch failed");
            if (P.mfma) {
                ma.row_begin = 0;
                ma.row_end = h0->n_rows;
                ma.progress = nullptr;
                if (segmented ? crag_launch_scan_q16(ma, 1, h0->n_cus, st)
                    : qsplit  ? crag_launch_scan_qsplit(ma, 1, h0->n_cus, st)
                              : crag_launch_scan_qlong(ma, h0->dtype, 1, h0->n_cus, st))
                    return fail(CRAG_E_HIP, "re-scan launch failed");
            } else {
                // the same groups over all of their rows: descriptors with from[2] = 0 behind the first set
                SmallVec<GroupDesc, 64> gd2;
                for (const auto& g : gd) {
                    gd2.push_back(g);
                    gd2.back().from[2] = 0;
                }
                const int blocks2 = gemv_partition(gd2, 2, P.nq_template, h0->n_cus);
                const size_t bytes2 = sizeof(GroupDesc) * gd2.size();
                if ((rc = ws->ensure_pinned(bytes2))) return rc;
                if ((rc = ws->keys.ensure(bytes2))) return rc;
                memcpy(ws->pinned, gd2.data(), bytes2);
                HIPCHK(hipMemcpyAsync(ws->keys.p, ws->pinned, bytes2, hipMemcpyHostToDevice, st));
                ScanArgs ga2 = ga;
                ga2.groups = (const GroupDesc*)ws->keys.p;
                if (crag_launch_scan_gemv(ga2, 2, P.nq_template, blocks2, st)) return fail(CRAG_E_HIP, "re-scan launch failed");
            }
            HIPCHK(hipStreamSynchronize(st));  // the pinned staging (descriptors, then the query list) is reused below
            if ((rc = upload_qmap(t1b))) return rc;
            FinalizeArgs g = f;
            g.qmap = qmap_dev;
            g.B = (int)t1b.size();
            g.trusted = 2;
            if (crag_launch_rescore_wide(g, (unsigned long long*)ws->cand.p, g.B, st)) return fail(CRAG_E_HIP, "wide re-score launch failed");
            if ((rc = crag_launch_finalize(g, st))) return fail(CRAG_E_HIP, "tier-1b finalize launch failed");
            HIPCHK(stream_wait(st));
            h0->stats[8] += (long long)t1b.size();
            patched = true;
        }
        // tier 2: canonical distance of every row + radix select on the device, batched (chunks bounded by 2 GiB of keys)
        SmallVec<int, 64> t2;
        long long n_max = 0;
        for (int b : failed)
            if (hst[(size_t)b] != CRAG_ST_OK) {
                t2.push_back(b);
                n_max = std::max<long long>(n_max, hs[b]->n_rows);
            }
        h0->stats[7] += (long long)t2.size();
        if (!t2.empty()) {
            const long long key_stride = (std::max<long long>(n_max, 1) + 255) / 256 * 256;
            const size_t per_chunk = (size_t)std::max<long long>(1, std::min<long long>((long long)t2.size(), (2ll << 30) / (8 * key_stride)));
            if ((rc = ws->keys.ensure(sizeof(unsigned long long) * per_chunk * (size_t)key_stride))) return rc;
            const size_t qmap_bytes = (sizeof(int) * t2.size() + 255) / 256 * 256;
            if ((rc = ws->misc.ensure(qmap_bytes + per_chunk * (256 * sizeof(unsigned int) + sizeof(CragRadixState))))) return rc;
            qmap_dev = (int*)ws->misc.p;
            if ((rc = upload_qmap(t2))) return rc;
            for (size_t c0 = 0; c0 < t2.size(); c0 += per_chunk) {
                ExactSelectArgs e;
                e.qd = qd_dev;
                e.qmap = qmap_dev + c0;
                e.F = (int)std::min(per_chunk, t2.size() - c0);
                e.queries = q_dev;
                e.dtype = h0->dtype;
                e.dim = h0->dim;
                e.dim_pad = h0->dim_pad;
                e.metric = h0->metric;
                e.kc = P.kc;
                e.n_max = n_max;
                e.keys = (unsigned long long*)ws->keys.p;
                e.key_stride = key_stride;
                e.hist = (unsigned int*)((char*)ws->misc.p + qmap_bytes);
                e.state = (CragRadixState*)(e.hist + per_chunk * 256);
                e.cand = (unsigned long long*)ws->cand.p;
                e.cand_stride = cand_stride;
                e.cnt = cnt;
                if (crag_launch_exact_select(e, st)) return fail(CRAG_E_HIP, "exact fallback launch failed");
                FinalizeArgs g = f;
                g.qmap = e.qmap;
                g.B = e.F;
                g.M = std::max(P.kc, 1);
                g.trusted = 1;
                g.segcnt = nullptr;  // the gather fills the flat view of each query's slots
                g.n_seg = 0;
                g.seg_cap = 0;
                if ((rc = crag_launch_finalize(g, st))) return fail(CRAG_E_HIP, "fallback finalize launch failed");
            }
            HIPCHK(stream_wait(st));
            for (int b : t2)
                if (hst[(size_t)b] != CRAG_ST_OK) return fail(CRAG_E_HIP, "exact fallback did not produce a certified result");
            patched = true;
        }
    }
    if (patched && merge()) return fail(CRAG_E_HIP, "merge_rerank launch failed");
    return 0;
}

int check_search_args(crag_index* const* hs, int B, int top_k) {
    if (!hs || B <= 0 || top_k <= 0) return fail(CRAG_E_INVALID, "search: B and top_k must be positive");
    if (top_k > CRAG_MAX_TOP_K) return fail(CRAG_E_LIMIT, "search: top_k exceeds CRAG_MAX_TOP_K");
    crag_index* h0 = hs[0];
    for (int b = 0; b < B; ++b) {
        crag_index* h = hs[b];
        if (!h) return fail(CRAG_E_INVALID, "search: null collection handle");
        if (!h->finalized) return fail(CRAG_E_STATE, "search: collection not finalized");
        if (h->dim != h0->dim || h->dtype != h0->dtype || h->metric != h0->metric || h->device != h0->device)
            return fail(CRAG_E_INVALID, "search_multi: collections differ in dim/dtype/metric/device");
    }
    return 0;
}

// host queries (float32) -> device, index dtype, padded stride.  Returns device pointer in ws->q.
// queries (float32, stride dim) -> index dtype, stride dim_pad, zero-padded, at pinned + pinned_off; copy: also send
// them to ws->q (otherwise they travel with the merged copy of the chunk launcher)
int upload_queries(crag_index* h, const float* queries, int B, Workspace* ws, hipStream_t st, size_t pinned_off, bool copy) {
    const size_t bytes = (size_t)B * h->dim_pad * h->esz;
    int rc;
    if (copy && (rc = ws->q.ensure(bytes))) return rc;
    char* stage = (char*)ws->pinned + pinned_off;
    memset(stage, 0, bytes);
    for (int b = 0; b < B; ++b) {
        const float* src = queries + (size_t)b * h->dim;
        if (h->dtype == CRAG_F32) {
            memcpy(stage + (size_t)b * h->dim_pad * 4, src, (size_t)h->dim * 4);
        } else {
            uint16_t* dst = (uint16_t*)(stage + (size_t)b * h->dim_pad * 2);
            for (int d = 0; d < h->dim; ++d) dst[d] = f32_to_bf16(src[d]);
        }
    }
    if (copy) HIPCHK(hipMemcpyAsync(ws->q.p, stage, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

int symbol_hashes_host(const crag_index* h, const unsigned char* q, int64_t len, std::vector<uint64_t>& out) {
    out.clear();
    if (h->names.empty() || len <= 0) return 0;
    const int maxlen = h->max_name_len;
    for (int64_t i = 0; i < len; ++i) {
        uint64_t hash = 1469598103934665603ull;
        const int64_t lim = std::min<int64_t>(maxlen, len - i);
        for (int64_t l = 1; l <= lim; ++l) {
            hash ^= q[i + l - 1];
            hash *= 1099511628211ull;
            if (h->len_present[(size_t)l]) {
                const uint64_t hh = hash ? hash : 1ull;
                if (h->names.count(hh)) out.push_back(hh);
            }
        }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return 0;
}

// the symbol-boost side of a query is either its lowered text (qtext/qtext_off: hashed here) or the sorted hash list a
// caller already holds (in_hashes/in_hoff, what crag_symbol_hashes returns -- a wrapper that verifies the hashes against
// the names as strings has them anyway and need not have the text hashed a second time)
int search_host(crag_index* const* hs, bool same, const float* queries, int B, int top_k, double symbol_boost,
                const char* qtext, const int64_t* qtext_off, const uint64_t* in_hashes, const int32_t* in_hoff,
                int64_t* out_ids, double* out_scores, float* out_dist, int32_t* out_counts) {
    int rc = check_search_args(hs, B, top_k);
    if (rc) return rc;
    if (!queries || !out_ids || !out_scores || !out_dist || !out_counts)
        return fail(CRAG_E_INVALID, "search: null buffer");
    crag_index* h0 = hs[0];
    HIPCHK(hipSetDevice(h0->device));
    WsGuard guard(h0->device);
    Workspace* ws = guard.w;
    if (!ws) return fail(CRAG_E_HIP, "could not create a HIP stream");
    hipStream_t st = ws->stream;
    const int kc = 2 * top_k;

    // symbol hashes per query (R3)
    std::vector<uint64_t> all_h;
    std::vector<int32_t> off((size_t)B + 1, 0);
    bool any_hash = false;
    if (qtext && qtext_off) {
        std::vector<uint64_t> one;
        for (int b = 0; b < B; ++b) {
            symbol_hashes_host(hs[b], (const unsigned char*)qtext + qtext_off[b], qtext_off[b + 1] - qtext_off[b], one);
            all_h.insert(all_h.end(), one.begin(), one.end());
            off[(size_t)b + 1] = (int32_t)all_h.size();
        }
        any_hash = true;
    } else if (in_hashes && in_hoff) {
        if (in_hoff[0] != 0) return fail(CRAG_E_INVALID, "search: hash offsets must start at 0");
        for (int b = 0; b < B; ++b)
            if (in_hoff[b + 1] < in_hoff[b]) return fail(CRAG_E_INVALID, "search: hash offsets must not decrease");
        all_h.assign(in_hashes, in_hashes + in_hoff[B]);
        off.assign(in_hoff, in_hoff + B + 1);
        any_hash = true;
    }
    const size_t qbytes = (size_t)B * h0->dim_pad * h0->esz;
    const size_t desc_reserve = (sizeof(QueryDesc) + sizeof(GroupDesc)) * (size_t)B;
    const size_t hash_bytes = all_h.size() * 8 + off.size() * 4 + 16;
    const size_t out_bytes = (size_t)B * top_k * (8 + 8 + 4) + (size_t)B * 4;
    if ((rc = ws->ensure_pinned(desc_reserve + qbytes + hash_bytes + out_bytes + 64))) return rc;
    // pinned layout: [desc | queries | hashes+offsets | outputs]
    const size_t q_off = (desc_reserve + 15) / 16 * 16;
    const size_t h_off = (q_off + qbytes + 15) / 16 * 16;
    const size_t o_off = (h_off + hash_bytes + 15) / 16 * 16;
    // one chunk of queries: everything the kernels read from the host goes up in one copy (see Workspace::stage)
    const bool merged = B <= CRAG_MAX_CHUNK;
    if (merged && (rc = ws->stage.ensure(o_off))) return rc;
    if ((rc = upload_queries(h0, queries, B, ws, st, q_off, !merged))) return rc;
    const void* q_dev = merged ? (const void*)((char*)ws->stage.p + q_off) : ws->q.p;
    size_t merged_bytes = q_off + qbytes;
    const uint64_t* qh_dev = nullptr;
    const int32_t* qo_dev = nullptr;
    if (any_hash) {
        char* stage = (char*)ws->pinned + h_off;
        memcpy(stage, off.data(), off.size() * 4);
        const size_t hoff = (off.size() * 4 + 7) / 8 * 8;
        if (!all_h.empty()) memcpy(stage + hoff, all_h.data(), all_h.size() * 8);
        char* hash_dev;
        if (merged) {
            hash_dev = (char*)ws->stage.p + h_off;
            merged_bytes = h_off + hoff + all_h.size() * 8;
        } else {
            if ((rc = ws->qhash.ensure(hash_bytes))) return rc;
            HIPCHK(hipMemcpyAsync(ws->qhash.p, stage, hoff + all_h.size() * 8, hipMemcpyHostToDevice, st));
            hash_dev = (char*)ws->qhash.p;
        }
        qo_dev = (const int32_t*)hash_dev;
        qh_dev = (const uint64_t*)(hash_dev + hoff);
    }
    if ((rc = ws->entries.ensure(sizeof(CragEntry) * (size_t)B * kc))) return rc;
    // Small result sets (a /search request is 10 x 20 bytes) are written by the rerank kernel straight into pinned host
    // memory: the call already waits for the stream by polling (the certificates), so nothing is left to do after it --
    // a D2H copy plus a parked hipStreamSynchronize behind the search cost 18-20 us per call (tools/latency_small.py).
    const bool direct = out_bytes <= CRAG_DIRECT_OUT_BYTES;
    if (direct) {
        if ((rc = ws->ensure_pinned_out(out_bytes))) return rc;
    } else if ((rc = ws->out.ensure(out_bytes))) {
        return rc;
    }
    char* o = direct ? (char*)ws->pinned_out : (char*)ws->out.p;
    MergeOut mo;
    mo.symbol_boost = symbol_boost;
    mo.ids = (long long*)o;
    mo.scores = (double*)(o + (size_t)B * top_k * 8);
    mo.dist = (float*)(o + (size_t)B * top_k * 16);
    mo.counts = (int*)(o + (size_t)B * top_k * 20);
    ws->merged_bytes = merged ? merged_bytes : 0;
    rc = search_entries(hs, same, q_dev, B, top_k, qh_dev, qo_dev, (CragEntry*)ws->entries.p, ws, st, &mo);
    ws->merged_bytes = 0;
    if (rc) return rc;
    char* ho = direct ? o : (char*)ws->pinned + o_off;
    if (!direct) HIPCHK(hipMemcpyAsync(ho, o, out_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(stream_wait(st));  // direct: only non-trivial after a fallback re-ran the rerank
    memcpy(out_ids, ho, (size_t)B * top_k * 8);
    memcpy(out_scores, ho + (size_t)B * top_k * 8, (size_t)B * top_k * 8);
    memcpy(out_dist, ho + (size_t)B * top_k * 16, (size_t)B * top_k * 4);
    memcpy(out_counts, ho + (size_t)B * top_k * 20, (size_t)B * 4);
    return 0;
}

// device queries [B][dim] in index dtype -> pointer with stride dim_pad (copy only when padding is needed)
int stage_device_queries(crag_index* h, const void* queries_dev, int B, Workspace* ws, hipStream_t st, const void** out) {
    if (h->dim == h->dim_pad) {
        *out = queries_dev;
        return 0;
    }
    const size_t bytes = (size_t)B * h->dim_pad * h->esz;
    int rc;
    if ((rc = ws->q.ensure(bytes))) return rc;
    HIPCHK(hipMemsetAsync(ws->q.p, 0, bytes, st));
    HIPCHK(hipMemcpy2DAsync(ws->q.p, (size_t)h->dim_pad * h->esz, queries_dev, (size_t)h->dim * h->esz,
                            (size_t)h->dim * h->esz, (size_t)B, hipMemcpyDeviceToDevice, st));
    *out = ws->q.p;
    return 0;
}

}  // namespace

// ---- crag_search_device_noalloc: the scratch array and its per-device address ----------------------------
// Everything crag_search_device takes from hipMalloc'ed workspace buffers (staged queries, descriptors, tau,
// candidate counters and buffers, the dense sample, the 2*top_k entries) is carved from this array instead.
// Sized for CRAG_NOALLOC_MAX_B = 64 queries at CRAG_MAX_TOP_K on a 256-CU device: 256 KiB queries + 16 KiB
// descriptors + counters + 128 KiB segment counters + 4 MiB sample + 16 MiB segmented candidate slots + 400 KiB
// entries = 21 MiB; a search that needs more is refused with CRAG_E_LIMIT.  Every GPU the library runs on
// holds its own copy until the process ends.
#define CRAG_NOALLOC_SCRATCH_BYTES (24u << 20)
static __device__ __align__(256) unsigned char scratch[CRAG_NOALLOC_SCRATCH_BYTES];

namespace {

// In host code `scratch` is a stand-in for the symbol, not a device address: the address of a device's copy
// comes from hipGetSymbolAddress, once per device with that device current, and only that pointer is used.
std::mutex g_scratch_mu;
void* g_scratch_ptr[64] = {};
int scratch_of_device(int device, void** out) {
    if (device < 0 || device >= 64) return fail(CRAG_E_INVALID, "crag_search_device_noalloc: device index out of range");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        HIPCHK(hipSetDevice(device));
        void* ptr = nullptr;
        HIPCHK(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(CRAG_E_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

struct DescBlob {
    QueryDesc qd;
    GroupDesc gd[CRAG_NOALLOC_MAX_B / CRAG_MAX_GROUP_Q];
};
__global__ __launch_bounds__(64) void k_put_descriptors(DescBlob blob, int B, int n_groups, QueryDesc* __restrict__ qd_out,
                                                        GroupDesc* __restrict__ gd_out) {
    for (int i = threadIdx.x; i < B; i += 64) qd_out[i] = blob.qd;
    for (int i = threadIdx.x; i < n_groups; i += 64) gd_out[i] = blob.gd[i];
}

int put_descriptors(const QueryDesc& qd, const GroupDesc* gd, int n_groups, int B, QueryDesc* qd_dev, GroupDesc* gd_dev,
                    hipStream_t st) {
    if (B > CRAG_NOALLOC_MAX_B || n_groups > CRAG_NOALLOC_MAX_B / CRAG_MAX_GROUP_Q)
        return fail(CRAG_E_LIMIT, "crag_search_device_noalloc: too many queries for the descriptor block");
    DescBlob blob;
    memset(&blob, 0, sizeof(blob));
    blob.qd = qd;
    for (int i = 0; i < n_groups; ++i) blob.gd[i] = gd[i];
    hipLaunchKernelGGL(k_put_descriptors, dim3(1), dim3(64), 0, st, blob, B, n_groups, qd_dev, gd_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int crag_version(void) { return 100; }
const char* crag_last_error(void) { return g_err.c_str(); }
int crag_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(CRAG_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int crag_index_create(int dim, int dtype, int metric, int device, crag_index** out) {
    if (!out) return fail(CRAG_E_INVALID, "crag_index_create: out is NULL");
    *out = nullptr;
    if (dim <= 0 || (dtype != CRAG_F32 && dtype != CRAG_BF16) || (metric != CRAG_L2 && metric != CRAG_IP))
        return fail(CRAG_E_INVALID, "crag_index_create: bad dim/dtype/metric");
    const size_t esz = dtype == CRAG_F32 ? 4 : 2;
    const size_t chunks_needed = ((size_t)dim * esz + 15) / 16;
    int chunks = 0;
    for (int c : kChunkSizes)
        if ((size_t)c >= chunks_needed) {
            chunks = c;
            break;
        }
    if (!chunks) return fail(CRAG_E_LIMIT, "crag_index_create: rows longer than 4096 bytes are not supported");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CRAG_E_INVALID, "crag_index_create: no such device");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(CRAG_E_HIP, std::string("crag_index_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    crag_index* h = new crag_index();
    h->dim = dim;
    h->dtype = dtype;
    h->metric = metric;
    h->device = device;
    h->esz = esz;
    h->chunks = chunks;
    h->dim_pad = (int)((size_t)chunks * 16 / esz);
    h->len_present.assign(65536 + 1, 0);
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = h;
    return 0;
}

int crag_index_destroy(crag_index* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    if (h->owns_rows && h->rows) (void)hipFree(h->rows);
    if (h->xnorm) (void)hipFree(h->xnorm);
    if (h->na_status) (void)hipHostFree(h->na_status);
    if (h->owns_meta) {
        if (h->name_hash) (void)hipFree(h->name_hash);
        if (h->path_class) (void)hipFree(h->path_class);
    }
    delete h;
    return 0;
}

int64_t crag_index_size(const crag_index* h) { return h ? h->n_rows : 0; }
int crag_index_dim_pad(const crag_index* h) { return h ? h->dim_pad : 0; }
int crag_index_set_id_base(crag_index* h, int64_t id_base) {
    if (!h) return fail(CRAG_E_INVALID, "null handle");
    h->id_base = id_base;
    return 0;
}

int crag_index_set_symbols(crag_index* h, const uint64_t* hashes, const uint16_t* lens, int64_t n) {
    if (!h || n < 0 || (n > 0 && (!hashes || !lens))) return fail(CRAG_E_INVALID, "crag_index_set_symbols: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        if (hashes[i] == 0 || lens[i] == 0) continue;
        h->names.insert(hashes[i]);
        h->len_present[lens[i]] = 1;
        h->max_name_len = std::max<int>(h->max_name_len, lens[i]);
    }
    return 0;
}

int crag_index_add(crag_index* h, const void* rows, int64_t n, const uint64_t* name_hash, const uint16_t* name_len,
                   const uint8_t* path_class) {
    int rc = add_common(h, rows, n, name_hash, path_class, hipMemcpyHostToDevice);
    if (rc) return rc;
    if (name_hash && name_len) return crag_index_set_symbols(h, name_hash, name_len, n);
    return 0;
}

int crag_index_add_device(crag_index* h, const void* rows_dev, int64_t n, const uint64_t* name_hash_dev,
                          const uint8_t* path_class_dev) {
    return add_common(h, rows_dev, n, name_hash_dev, path_class_dev, hipMemcpyDeviceToDevice);
}

int crag_index_attach_device(crag_index* h, const void* rows_dev, int64_t n, const uint64_t* name_hash_dev,
                             const uint8_t* path_class_dev) {
    if (!h || !rows_dev || n < 0) return fail(CRAG_E_INVALID, "crag_index_attach_device: bad arguments");
    if (h->n_rows != 0) return fail(CRAG_E_STATE, "crag_index_attach_device: collection already holds rows");
    if (n > CRAG_MAX_ROWS) return fail(CRAG_E_LIMIT, "crag_index_attach_device: a collection holds at most 2^32 - 64 rows");
    if (h->dim != h->dim_pad) return fail(CRAG_E_INVALID, "crag_index_attach_device: dim must equal crag_index_dim_pad");
    if (((uintptr_t)rows_dev) & 15) return fail(CRAG_E_INVALID, "crag_index_attach_device: rows must be 16-byte aligned");
    h->rows = const_cast<void*>(rows_dev);
    h->owns_rows = false;
    h->attached = true;
    h->name_hash = const_cast<uint64_t*>(name_hash_dev);
    h->path_class = const_cast<uint8_t*>(path_class_dev);
    h->owns_meta = false;
    h->n_rows = n;
    h->cap_rows = n;
    h->finalized = false;
    return 0;
}

int crag_index_get_rows(const crag_index* h, int64_t first, int64_t n, void* out_rows) {
    if (!h || first < 0 || n < 0 || first + n > h->n_rows || (n > 0 && !out_rows))
        return fail(CRAG_E_INVALID, "crag_index_get_rows: bad arguments");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    const size_t src_pitch = (size_t)h->dim_pad * h->esz, dst_pitch = (size_t)h->dim * h->esz;
    HIPCHK(hipMemcpy2D(out_rows, dst_pitch, (const char*)h->rows + (size_t)first * src_pitch, src_pitch, dst_pitch, (size_t)n,
                       hipMemcpyDeviceToHost));
    return 0;
}

int crag_index_finalize(crag_index* h) {
    if (!h) return fail(CRAG_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->xnorm) (void)hipFree(h->xnorm);
    h->xnorm = nullptr;
    h->max_xnorm = 0.f;
    // padded by 256 floats: the MFMA epilogue reads |x|^2 for whole 128-row tiles
    const size_t nx = (size_t)std::max<int64_t>(h->n_rows, 1) + 256;
    HIPCHK(hipMalloc((void**)&h->xnorm, sizeof(float) * (nx + 4)));
    HIPCHK(hipMemset(h->xnorm, 0, sizeof(float) * (nx + 4)));
    float* mx = h->xnorm + nx;
    if (h->n_rows) {
        int rc = crag_launch_row_norms(h->rows, h->dtype, h->dim_pad, h->n_rows, h->xnorm, mx, nullptr);
        if (rc) return fail(CRAG_E_HIP, "row_norms launch failed");
    }
    HIPCHK(hipMemcpy(&h->max_xnorm, mx, 4, hipMemcpyDeviceToHost));
    if (!(h->max_xnorm <= 3.4028234663852886e38f)) {
        // the error bound of the certificates is built on max |x|^2: a collection with such a row cannot be searched exactly
        h->max_xnorm = 0.f;
        return fail(CRAG_E_INVALID, "a row holds a NaN or an infinite value (or its squared norm overflows float32)");
    }
    // certificates of crag_search_device_noalloc: that entry point may not allocate, so its pinned block is made here
    if (!h->na_status) HIPCHK(hipHostMalloc((void**)&h->na_status, sizeof(uint32_t) * CRAG_NOALLOC_MAX_B, hipHostMallocDefault));
    h->finalized = true;
    return 0;
}

uint64_t crag_hash_name(const char* utf8, int64_t len) {
    if (!utf8 || len <= 0) return 0;
    return fnv1a((const unsigned char*)utf8, len);
}
void crag_hash_names(const char* buf, const int64_t* off, int64_t n, uint64_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = crag_hash_name(buf + off[i], off[i + 1] - off[i]);
}

int crag_symbol_hashes(const crag_index* h, const char* q_lower_utf8, int64_t len, uint64_t* out, int cap) {
    if (!h || (len > 0 && !q_lower_utf8) || (cap > 0 && !out)) return fail(CRAG_E_INVALID, "crag_symbol_hashes: bad arguments");
    std::vector<uint64_t> v;
    symbol_hashes_host(h, (const unsigned char*)q_lower_utf8, len, v);
    const int n = (int)std::min<size_t>(v.size(), (size_t)std::max(cap, 0));
    for (int i = 0; i < n; ++i) out[i] = v[(size_t)i];
    return n;
}

int crag_search(crag_index* h, const float* queries, int B, int top_k, double symbol_boost, const char* qtext,
                const int64_t* qtext_off, int64_t* out_ids, double* out_scores, float* out_dist,
                int32_t* out_counts) {
    if (!h || B <= 0) return fail(CRAG_E_INVALID, "crag_search: bad arguments");
    std::vector<crag_index*> hs((size_t)B, h);
    return search_host(hs.data(), true, queries, B, top_k, symbol_boost, qtext, qtext_off, nullptr, nullptr, out_ids, out_scores, out_dist,
                       out_counts);
}

int crag_search_multi(crag_index* const* hs, const float* queries, int B, int top_k, double symbol_boost,
                      const char* qtext, const int64_t* qtext_off, int64_t* out_ids, double* out_scores,
                      float* out_dist, int32_t* out_counts) {
    if (!hs || B <= 0) return fail(CRAG_E_INVALID, "crag_search_multi: bad arguments");
    bool same = true;
    for (int b = 1; b < B; ++b) same = same && hs[b] == hs[0];
    return search_host(hs, same, queries, B, top_k, symbol_boost, qtext, qtext_off, nullptr, nullptr, out_ids, out_scores, out_dist,
                       out_counts);
}

int crag_search_multi_hashed(crag_index* const* hs, const float* queries, int B, int top_k, double symbol_boost,
                             const uint64_t* q_hashes, const int32_t* q_hash_off, int64_t* out_ids, double* out_scores,
                             float* out_dist, int32_t* out_counts) {
    if (!hs || B <= 0) return fail(CRAG_E_INVALID, "crag_search_multi_hashed: bad arguments");
    bool same = true;
    for (int b = 1; b < B; ++b) same = same && hs[b] == hs[0];
    return search_host(hs, same, queries, B, top_k, symbol_boost, nullptr, nullptr, q_hashes, q_hash_off, out_ids, out_scores, out_dist,
                       out_counts);
}

int crag_search_local_device(crag_index* h, const void* queries_dev, int B, int top_k, const uint64_t* q_hashes_dev,
                             const int32_t* q_hash_off_dev, void* entries_dev, void* stream) {
    if (!h || !queries_dev || !entries_dev) return fail(CRAG_E_INVALID, "crag_search_local_device: bad arguments");
    std::vector<crag_index*> hs((size_t)std::max(B, 1), h);
    int rc = check_search_args(hs.data(), B, top_k);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    WsGuard guard(h->device);
    if (!guard.w) return fail(CRAG_E_HIP, "could not create a HIP stream");
    hipStream_t st = (hipStream_t)stream;
    const void* q = nullptr;
    if ((rc = stage_device_queries(h, queries_dev, B, guard.w, st, &q))) return rc;
    return search_entries(hs.data(), true, q, B, top_k, q_hashes_dev, q_hash_off_dev, (CragEntry*)entries_dev, guard.w, st);
}

int crag_merge_rerank_device(int device, const void* entries_dev, int G, int B, int top_k, double symbol_boost,
                             int64_t* out_ids_dev, double* out_scores_dev, float* out_dist_dev,
                             int32_t* out_counts_dev, void* stream) {
    if (!entries_dev || G <= 0 || B <= 0 || top_k <= 0 || top_k > CRAG_MAX_TOP_K)
        return fail(CRAG_E_INVALID, "crag_merge_rerank_device: bad arguments");
    HIPCHK(hipSetDevice(device));
    int rc = crag_launch_merge_rerank((const CragEntry*)entries_dev, G, B, 2 * top_k, top_k, symbol_boost,
                                      (long long*)out_ids_dev, out_scores_dev, out_dist_dev, out_counts_dev,
                                      (hipStream_t)stream);
    if (rc == -2) return fail(CRAG_E_LIMIT, "crag_merge_rerank_device: G * 2*top_k too large for one workgroup");
    if (rc) return fail(CRAG_E_HIP, "merge_rerank launch failed");
    return 0;
}

int crag_search_device(crag_index* h, const void* queries_dev, int B, int top_k, double symbol_boost,
                       const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, int64_t* out_ids_dev,
                       double* out_scores_dev, float* out_dist_dev, int32_t* out_counts_dev, void* stream) {
    if (!h || !queries_dev || !out_ids_dev || !out_scores_dev || !out_dist_dev || !out_counts_dev)
        return fail(CRAG_E_INVALID, "crag_search_device: bad arguments");
    std::vector<crag_index*> hs((size_t)std::max(B, 1), h);
    int rc = check_search_args(hs.data(), B, top_k);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    WsGuard guard(h->device);
    if (!guard.w) return fail(CRAG_E_HIP, "could not create a HIP stream");
    Workspace* ws = guard.w;
    hipStream_t st = (hipStream_t)stream;
    const void* q = nullptr;
    if ((rc = stage_device_queries(h, queries_dev, B, ws, st, &q))) return rc;
    if ((rc = ws->entries.ensure(sizeof(CragEntry) * (size_t)B * 2 * top_k))) return rc;
    MergeOut mo;
    mo.symbol_boost = symbol_boost;
    mo.ids = (long long*)out_ids_dev;
    mo.scores = out_scores_dev;
    mo.dist = out_dist_dev;
    mo.counts = out_counts_dev;
    rc = search_entries(hs.data(), true, q, B, top_k, q_hashes_dev, q_hash_off_dev, (CragEntry*)ws->entries.p, ws, st, &mo);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));  // only non-trivial after a fallback re-merge
    return 0;
}

int crag_search_device_noalloc(crag_index* h, const void* queries_dev, int B, int top_k, double symbol_boost,
                               const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, int64_t* out_ids_dev,
                               double* out_scores_dev, float* out_dist_dev, int32_t* out_counts_dev, void* stream) {
    if (!h || !queries_dev || !out_ids_dev || !out_scores_dev || !out_dist_dev || !out_counts_dev)
        return fail(CRAG_E_INVALID, "crag_search_device_noalloc: bad arguments");
    if (B > CRAG_NOALLOC_MAX_B)
        return fail(CRAG_E_LIMIT, "crag_search_device_noalloc: at most 64 queries per call (CRAG_NOALLOC_MAX_B); use crag_search_device");
    crag_index* hs[CRAG_NOALLOC_MAX_B];
    for (int b = 0; b < CRAG_NOALLOC_MAX_B; ++b) hs[b] = h;
    int rc = check_search_args(hs, B, top_k);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    void* ptr = nullptr;
    if ((rc = scratch_of_device(h->device, &ptr))) return rc;
    Arena arena;
    arena.base = (char*)ptr;
    arena.size = CRAG_NOALLOC_SCRATCH_BYTES;
    Workspace ws;  // on the stack: no pool, no stream of its own, no pinned host memory
    ws.device = h->device;
    ws.use_arena(&arena);
    hipStream_t st = (hipStream_t)stream;
    const void* q = nullptr;
    if ((rc = stage_device_queries(h, queries_dev, B, &ws, st, &q))) return rc;
    if ((rc = ws.entries.ensure(sizeof(CragEntry) * (size_t)B * 2 * top_k))) return rc;
    MergeOut mo;
    mo.symbol_boost = symbol_boost;
    mo.ids = (long long*)out_ids_dev;
    mo.scores = out_scores_dev;
    mo.dist = out_dist_dev;
    mo.counts = out_counts_dev;
    return search_entries(hs, true, q, B, top_k, q_hashes_dev, q_hash_off_dev, (CragEntry*)ws.entries.p, &ws, st, &mo);
}

int crag_search_multi_device(crag_index* const* hs, const void* queries_dev, int B, int top_k, double symbol_boost,
                             const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, int64_t* out_ids_dev,
                             double* out_scores_dev, float* out_dist_dev, int32_t* out_counts_dev, void* stream) {
    if (!hs || !queries_dev || !out_ids_dev || !out_scores_dev || !out_dist_dev || !out_counts_dev)
        return fail(CRAG_E_INVALID, "crag_search_multi_device: bad arguments");
    int rc = check_search_args(hs, B, top_k);
    if (rc) return rc;
    if (B <= 0) return 0;
    crag_index* h = hs[0];
    bool same = true;
    for (int b = 1; b < B; ++b) same = same && hs[b] == h;
    HIPCHK(hipSetDevice(h->device));
    WsGuard guard(h->device);
    if (!guard.w) return fail(CRAG_E_HIP, "could not create a HIP stream");
    Workspace* ws = guard.w;
    hipStream_t st = (hipStream_t)stream;
    const void* q = nullptr;
    if ((rc = stage_device_queries(h, queries_dev, B, ws, st, &q))) return rc;
    if ((rc = ws->entries.ensure(sizeof(CragEntry) * (size_t)B * 2 * top_k))) return rc;
    MergeOut mo;
    mo.symbol_boost = symbol_boost;
    mo.ids = (long long*)out_ids_dev;
    mo.scores = out_scores_dev;
    mo.dist = out_dist_dev;
    mo.counts = out_counts_dev;
    rc = search_entries(hs, same, q, B, top_k, q_hashes_dev, q_hash_off_dev, (CragEntry*)ws->entries.p, ws, st, &mo);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));  // only non-trivial after a fallback re-merge
    return 0;
}

int crag_stats(crag_index* h, int64_t* out9, int reset) {
    if (!h || !out9) return fail(CRAG_E_INVALID, "crag_stats: bad arguments");
    for (int i = 0; i < 9; ++i) {
        out9[i] = h->stats[i];
        if (reset) h->stats[i] = 0;
    }
    return 0;
}

int crag_debug_dense_scores(crag_index* h, const void* queries_dev, int B, int path, float* out_scores_dev,
                            void* stream) {
    if (!h || !queries_dev || !out_scores_dev || B <= 0) return fail(CRAG_E_INVALID, "crag_debug_dense_scores: bad arguments");
    if (!h->finalized) return fail(CRAG_E_STATE, "collection not finalized");
    HIPCHK(hipSetDevice(h->device));
    WsGuard guard(h->device);
    if (!guard.w) return fail(CRAG_E_HIP, "could not create a HIP stream");
    Workspace* ws = guard.w;
    hipStream_t st = (hipStream_t)stream;
    const void* q = nullptr;
    int rc;
    if ((rc = stage_device_queries(h, queries_dev, B, ws, st, &q))) return rc;
    const long long n = h->n_rows;
    const long long stride = n;
    bool rowmajor = false, tiled = false;
    long long bpad = B;
    if ((rc = ws->sample.ensure(sizeof(uint32_t) * (size_t)B * (size_t)std::max<long long>(stride, 1)))) return rc;
    if (path >= 2) {
        const bool d_bf16 = h->dtype == CRAG_BF16;
        MfmaArgs a;
        memset(&a, 0, sizeof(a));
        a.rows = h->rows;
        a.xnorm = h->xnorm;
        a.n_rows = n;
        a.row_begin = 0;
        a.row_end = n;
        a.queries = q;
        a.n_queries = B;
        a.dim_pad = h->dim_pad;
        a.metric = h->metric;
        a.sample = (uint32_t*)ws->sample.p;
        a.sample_stride = stride;
        const bool d_qsplit = d_bf16 && crag_qsplit_supported(h->dim_pad);
        const bool d_q16 = d_bf16 && crag_q16_supported(h->dim_pad);
        const bool d_qlong = crag_qlong_supported(h->dtype, h->dim_pad);
        if (!d_qsplit && !d_q16 && !d_qlong) return fail(CRAG_E_INVALID, "mfma path: rows of this length have no MFMA scan");
        rowmajor = true;
        bpad = ((long long)B + 15) / 16 * 16;
        tiled = d_q16;  // crag_launch_scan_q16's tile-major sample
        if (d_q16 || d_qsplit) {
            if ((rc = ws->qscan.ensure((size_t)B * h->dim_pad * 2))) return rc;
            if (crag_q16_scale_queries(q, B, h->dim_pad, h->metric, ws->qscan.p, st)) return fail(CRAG_E_HIP, "query scaling launch failed");
            a.queries = ws->qscan.p;
        }
        if ((rc = ws->sample.ensure(sizeof(uint32_t) * (size_t)bpad * (size_t)((std::max<long long>(n, 1) + 31) / 32 * 32)))) return rc;
        a.sample = (uint32_t*)ws->sample.p;
        a.sample_stride = bpad;
        if (d_qsplit   ? crag_launch_scan_qsplit(a, 0, h->n_cus, st)
            : d_qlong  ? crag_launch_scan_qlong(a, h->dtype, 0, h->n_cus, st)
                       : crag_launch_scan_q16(a, 0, h->n_cus, st))
            return fail(CRAG_E_HIP, "mfma scan launch failed");
    } else {
        std::vector<GroupDesc> gd;
        int blk = 0;
        for (int b = 0; b < B; b += CRAG_MAX_GROUP_Q) {
            GroupDesc g;
            memset(&g, 0, sizeof(g));
            g.rows = h->rows;
            g.xnorm = h->xnorm;
            g.n_rows = n;
            g.n_sample = n;
            g.q_first = b;
            g.nq = std::min(CRAG_MAX_GROUP_Q, B - b);
            g.blk_first[0] = blk;
            g.n_blk[0] = (int)std::min<long long>(512, (n + 255) / 256);
            blk += g.n_blk[0];
            gd.push_back(g);
        }
        const int nqt = B <= 1 ? 1 : (B <= 2 ? 2 : (B <= 4 ? 4 : 8));
        if ((rc = ws->desc.ensure(sizeof(GroupDesc) * gd.size()))) return rc;
        HIPCHK(hipMemcpyAsync(ws->desc.p, gd.data(), sizeof(GroupDesc) * gd.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        ScanArgs a;
        memset(&a, 0, sizeof(a));
        a.groups = (const GroupDesc*)ws->desc.p;
        a.n_groups = (int)gd.size();
        a.queries = q;
        a.dtype = h->dtype;
        a.chunks = h->chunks;
        a.metric = h->metric;
        a.sample = (uint32_t*)ws->sample.p;
        a.sample_stride = stride;
        if (crag_launch_scan_gemv(a, 0, nqt, blk, st)) return fail(CRAG_E_HIP, "gemv launch failed");
    }
    HIPCHK(hipStreamSynchronize(st));
    // keys -> floats on the host side of the boundary is the caller's job; here we decode on the host
    std::vector<uint32_t> keys((size_t)(rowmajor ? bpad : B) * (size_t)(tiled ? (n + 31) / 32 * 32 : n));
    HIPCHK(hipMemcpy(keys.data(), ws->sample.p, keys.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> sc((size_t)B * (size_t)n);
    for (int b = 0; b < B; ++b)
        for (long long i = 0; i < n; ++i)
            sc[(size_t)b * n + i] = crag_fkey_inv(tiled      ? keys[((size_t)(i >> 5) * bpad + b) * 32 + (i & 31)]
                                                  : rowmajor ? keys[(size_t)i * bpad + b]
                                                             : keys[(size_t)b * n + i]);
    HIPCHK(hipMemcpy(out_scores_dev, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

int crag_set_profiling(crag_index* h, int on) {
    if (!h) return fail(CRAG_E_INVALID, "crag_set_profiling: null handle");
    h->profiling = on ? 1 : 0;
    return 0;
}

int crag_get_profile(crag_index* h, double* out4, int reset) {
    if (!h || !out4) return fail(CRAG_E_INVALID, "crag_get_profile: bad arguments");
    std::lock_guard<std::mutex> lk(h->prof_mu);
    for (int i = 0; i < 4; ++i) {
        out4[i] = h->prof[i];
        if (reset) h->prof[i] = 0.0;
    }
    return 0;
}

int crag_set_path(crag_index* h, int path) {
    if (!h || path < 0 || path > 3) return fail(CRAG_E_INVALID, "crag_set_path: bad arguments");
    h->force_path = path;
    return 0;
}

}  // extern "C"
