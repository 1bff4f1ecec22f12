/*
 * coderag_hip.h -- C ABI of the MI355X (gfx950) retrieval scoring library.
 *
 * This is the boundary a maintainer of Achernar-Eridani/code-rag binds (ctypes, see
 * INTEGRATION.md) to replace what retriever/hybrid_search.py does today through ChromaDB:
 *
 *   reference interface                                   entry point here
 *   ---------------------------------------------------   ---------------------------------
 *   chromadb.PersistentClient / create_collection          crag_index_create / _destroy
 *     (indexer/embed_ingest.py:38-54, space = l2)
 *   Collection.add(ids, documents, metadatas)              crag_index_add / _add_device /
 *     (indexer/embed_ingest.py:93)                         _attach_device, crag_index_finalize
 *   Collection.count() (indexer/embed_ingest.py:107)       crag_index_size
 *   Collection.get(include=["embeddings"]) (Chroma API;    crag_index_get_rows
 *     the persistence behind data/chroma_db)
 *   Collection.query(query_texts=[q], n_results=2*top_k)   crag_search (host buffers),
 *     + base score + path/symbol boost + filter + sort     crag_search_device (device buffers)
 *     + [:top_k]  (retriever/hybrid_search.py:60-125)
 *   `name in q_lower` (retriever/hybrid_search.py:114)     crag_hash_name(s), crag_symbol_hashes
 *   one collection per workspace, queried concurrently     crag_search_multi, crag_search_multi_hashed
 *     (api/main.py:77-82, retriever/hybrid_search.py:28-39) (host buffers), crag_search_multi_device
 *                                                          (device buffers)
 *   row-sharded corpus over the GPUs of one node           crag_search_local_device +
 *     (no reference counterpart; north_star)               crag_merge_rerank_device
 *
 * Conventions: every function returns 0 on success or a negative CRAG_E_* code; no C++
 * exception crosses this boundary; the caller allocates every in/out buffer; the library owns
 * only device memory behind a handle.  crag_search* are re-entrant on one handle (a scratch
 * workspace and a HIP stream are taken from a per-device pool for the duration of the call);
 * add/finalize/destroy must not overlap anything else on that handle.
 * Row ids are positions: row i of a collection has id  id_base + i.
 */
#ifndef CODERAG_HIP_H
#define CODERAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRAG_DTYPE_F32 0
#define CRAG_DTYPE_BF16 1
#define CRAG_METRIC_L2 0 /* squared L2: Chroma's default space, what the reference uses */
#define CRAG_METRIC_IP 1 /* Chroma "ip": 1 - <q,x> */

#define CRAG_OK 0
#define CRAG_E_INVALID (-1)   /* bad argument */
#define CRAG_E_HIP (-2)       /* a HIP call failed; see crag_last_error */
#define CRAG_E_STATE (-3)     /* e.g. search before finalize, add after attach */
#define CRAG_E_LIMIT (-4)     /* top_k / dim beyond what the kernels support */
#define CRAG_E_NOMEM (-5)
#define CRAG_E_INEXACT (-6)   /* crag_search_device_noalloc only: a query needs the exact fallback scan */

#define CRAG_MAX_TOP_K 200    /* 2*top_k candidates <= 400 */

typedef struct crag_index crag_index;

/* library version (major*10000 + minor*100 + patch) */
int crag_version(void);
/* message of the last failure on this thread (never NULL) */
const char* crag_last_error(void);
/* number of visible HIP devices, or a negative error */
int crag_device_count(void);

/* ---- collection lifetime ------------------------------------------------------------ */
int crag_index_create(int dim, int dtype, int metric, int device, crag_index** out);
int crag_index_destroy(crag_index* h);
/* rows held */
int64_t crag_index_size(const crag_index* h);
/* elements per stored row after padding to the kernels' row geometry (>= dim) */
int crag_index_dim_pad(const crag_index* h);
/* id of row 0 (default 0); a shard of a row-partitioned corpus sets its global offset here */
int crag_index_set_id_base(crag_index* h, int64_t id_base);

/*
 * Append n rows from HOST memory.  rows: [n][dim] in the index dtype (bf16 = uint16 bit
 * patterns).  name_hash[n] (crag_hash_name of the lowered chunk name, 0 = no name),
 * name_len[n] (UTF-8 byte length of the lowered name) and path_class[n] (0 neutral,
 * 1 penalised, 2 boosted; hybrid_search.py:105-111) may each be NULL.
 */
int crag_index_add(crag_index* h, const void* rows, int64_t n, const uint64_t* name_hash,
                   const uint16_t* name_len, const uint8_t* path_class);
/* Same from DEVICE memory (copied device to device).  name_hash / path_class may be NULL. */
int crag_index_add_device(crag_index* h, const void* rows_dev, int64_t n, const uint64_t* name_hash_dev,
                          const uint8_t* path_class_dev);
/*
 * Zero-copy: adopt caller-owned DEVICE arrays as the whole collection (they must outlive the
 * handle).  Requires dim == crag_index_dim_pad and 16-byte aligned rows.
 */
int crag_index_attach_device(crag_index* h, const void* rows_dev, int64_t n, const uint64_t* name_hash_dev,
                             const uint8_t* path_class_dev);
/*
 * The set of distinct (name hash, byte length) pairs of the collection, for collections whose
 * per-row hashes were produced on the device.  crag_index_add with name_len builds it itself.
 */
int crag_index_set_symbols(crag_index* h, const uint64_t* hashes, const uint16_t* lens, int64_t n);
/*
 * Copy rows [first, first + n) back to HOST memory as [n][dim] in the index dtype (bf16 = uint16 bit patterns):
 * what persistence writes to disk, so that no second copy of the embeddings has to live in host memory.
 */
int crag_index_get_rows(const crag_index* h, int64_t first, int64_t n, void* out_rows);
/* Row norms + bounds; required after the last add and before the first search.  CRAG_E_INVALID when a row holds a NaN
 * or an infinite value (the exactness certificates are built on max |x|^2; Chroma's behaviour on such rows is not
 * defined either).  A QUERY with a non-finite component is not an error: it gets no results (count 0). */
int crag_index_finalize(crag_index* h);

/* ---- symbol hashing (R3) -------------------------------------------------------------- */
/* FNV-1a 64 of the bytes, never 0.  Pass the UTF-8 of str(name).lower(). */
uint64_t crag_hash_name(const char* utf8, int64_t len);
void crag_hash_names(const char* buf, const int64_t* off /* [n+1] */, int64_t n, uint64_t* out);
/*
 * Hashes of every byte window of the lowered query that equals some chunk name of the
 * collection (exactly the names for which `name in q_lower` holds, up to 64-bit collisions,
 * which the caller rules out by comparing strings).  Returns the count (<= cap), sorted
 * ascending, or a negative error.
 */
int crag_symbol_hashes(const crag_index* h, const char* q_lower_utf8, int64_t len, uint64_t* out, int cap);

/* ---- search --------------------------------------------------------------------------- */
/*
 * HybridSearcher.search for a batch of B queries against ONE collection, host buffers.
 *   queries      [B][dim] float32 (cast to the index dtype, round-to-nearest-even, like the
 *                oracle does)
 *   qtext/qtext_off  concatenated UTF-8 of query.lower() per query and B+1 offsets, or NULL
 *                (no symbol boost)
 *   out_ids      [B][top_k] int64 row ids, -1 beyond out_counts[b]
 *   out_scores   [B][top_k] float64 reranked score (hybrid_search.py:117)
 *   out_dist     [B][top_k] float32 distance (hybrid_search.py:71)
 *   out_counts   [B] results kept (<= top_k; fewer after the > 0.01 filter or on tiny collections)
 */
int crag_search(crag_index* h, const float* queries, int B, int top_k, double symbol_boost, const char* qtext,
                const int64_t* qtext_off, int64_t* out_ids, double* out_scores, float* out_dist,
                int32_t* out_counts);
/*
 * The same for queries that each address their own collection (workspace isolation):
 * hs[b] is the collection of query b; all must share dim, dtype, metric and device.
 * Consecutive queries on the same collection share one pass over its rows.
 */
int crag_search_multi(crag_index* const* hs, const float* queries, int B, int top_k, double symbol_boost,
                      const char* qtext, const int64_t* qtext_off, int64_t* out_ids, double* out_scores,
                      float* out_dist, int32_t* out_counts);
/*
 * crag_search_multi (and, with B copies of one handle, crag_search) for a caller that already holds each query's
 * symbol hashes -- q_hashes: the sorted lists crag_symbol_hashes returns, concatenated; q_hash_off: B+1 offsets
 * starting at 0; both NULL = no symbol boost.  The Python wrapper verifies every matched hash against the chunk names
 * as strings (`name in q_lower`, hybrid_search.py:114, must not depend on a 64-bit collision), so it has the hashes
 * before it searches; this entry point saves hashing each query text a second time.
 */
int crag_search_multi_hashed(crag_index* const* hs, const float* queries, int B, int top_k, double symbol_boost,
                             const uint64_t* q_hashes, const int32_t* q_hash_off, int64_t* out_ids,
                             double* out_scores, float* out_dist, int32_t* out_counts);
/*
 * Device-buffer form: queries_dev [B][dim] ALREADY in the index dtype; q_hashes_dev /
 * q_hash_off_dev (sorted symbol hashes per query, B+1 offsets) may be NULL; outputs are device
 * arrays shaped as above.  Work is enqueued on `stream` (a hipStream_t, NULL = default stream);
 * the call returns after the results are complete and verified (it synchronises the stream,
 * because a search that fails its exactness certificate is re-run through the exact scan).
 */
int crag_search_device(crag_index* h, const void* queries_dev, int B, int top_k, double symbol_boost,
                       const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, int64_t* out_ids_dev,
                       double* out_scores_dev, float* out_dist_dev, int32_t* out_counts_dev, void* stream);

/*
 * crag_search_device for callers that cannot allocate (code under HIP graph capture elsewhere in the process,
 * tight serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments, same
 * results, bit for bit.  It calls no hipMalloc / hipFree / hipHostMalloc and creates no stream or event: every
 * workspace buffer is carved from one 24 MiB `static __device__` array inside the library (each GPU the library
 * runs on holds its own copy until the process ends).
 *   limits   B <= CRAG_NOALLOC_MAX_B (64), top_k <= CRAG_MAX_TOP_K; beyond them CRAG_E_LIMIT and nothing is launched.
 *            (The array is sized for those limits on a 256-CU device.  Should a search still need more than it holds,
 *            the call returns CRAG_E_LIMIT as well, but by then it may have enqueued small copies on `stream` and it
 *            waits for the stream before it returns.)  The certificates come back through a small pinned block that
 *            crag_index_finalize made for the handle: no device-to-host copy follows the last kernel.
 *   exactness  a query that fails its exactness certificate or overflows its candidate buffer cannot be re-run
 *            here (the exact fallback scan needs 8 bytes of workspace per row): the call returns CRAG_E_INEXACT and
 *            the caller repeats the batch through crag_search_device.  The outputs are not valid in that case.
 *   sharing  all calls on one GPU use that GPU's one array, which keeps the last call's contents (the call clears
 *            the counters it needs at zero itself): two calls that overlap on one GPU race -- serialise them.
 * Like crag_search_device it returns after the results are complete and verified (it waits for `stream`).
 */
#define CRAG_NOALLOC_MAX_B 64
int crag_search_device_noalloc(crag_index* h, const void* queries_dev, int B, int top_k, double symbol_boost,
                               const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, int64_t* out_ids_dev,
                               double* out_scores_dev, float* out_dist_dev, int32_t* out_counts_dev, void* stream);

/*
 * The same for ONE QUERY PER COLLECTION (query b against hs[b]; a handle may repeat): the device-buffer
 * form of crag_search_multi -- one HybridSearcher serving many `code_chunks__<workspace>` collections
 * (api/main.py:77-82) with the queries already embedded on the GPU.  All handles share dim, dtype, metric
 * and device.
 */
int crag_search_multi_device(crag_index* const* hs, const void* queries_dev, int B, int top_k, double symbol_boost,
                             const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, int64_t* out_ids_dev,
                             double* out_scores_dev, float* out_dist_dev, int32_t* out_counts_dev, void* stream);

/* ---- row-sharded corpus: one process per GPU ------------------------------------------ */
/* bytes of one candidate entry as exchanged between ranks (float dist, u32 flags, i64 id) */
#define CRAG_ENTRY_BYTES 16
/*
 * Stage 1 on every rank: this shard's exact 2*top_k nearest per query, ordered by
 * (float32 distance, id), into entries_dev [B][2*top_k] entries.  Verified like
 * crag_search_device.  The caller all-gathers the entries (RCCL) ...
 */
int crag_search_local_device(crag_index* h, const void* queries_dev, int B, int top_k,
                             const uint64_t* q_hashes_dev, const int32_t* q_hash_off_dev, void* entries_dev,
                             void* stream);
/*
 * ... and stage 2 merges G shards' entries [G][B][2*top_k] by (distance, id) and applies the
 * rerank of hybrid_search.py:82-125.  Asynchronous on `stream`; `device` selects the GPU.
 */
int crag_merge_rerank_device(int device, const void* entries_dev, int G, int B, int top_k, double symbol_boost,
                             int64_t* out_ids_dev, double* out_scores_dev, float* out_dist_dev,
                             int32_t* out_counts_dev, void* stream);

/* ---- introspection (tests, bench) ------------------------------------------------------ */
/* counters since the last reset: [0] searches, [1] queries, [2] fallback queries (came back from the cascade without
 * an exactness certificate), of which [3] uncertified and [4] overflowed, [5] mfma-path batches, [6] gemv-path
 * batches, [7] fallback queries that needed the full exact scan, [8] fallback queries that went through the certified
 * re-scan (the rest of [2] were settled by re-scoring their whole candidate buffer) */
int crag_stats(crag_index* h, int64_t* out9, int reset);
/* force a path for tests/bench: 0 auto, 1 gemv groups, 2 mfma (the query-stationary scan where the row length has
 * one), 3 the LDS-tiled mfma scan (bf16 rows of 1536 / 2048 elements, every float32 row length; elsewhere as 2) */
int crag_set_path(crag_index* h, int path);
/*
 * Kernel timing for bench.py's roofline line: when on, every search brackets its two scan
 * launches with HIP events on the stream they are launched on.  crag_get_profile returns
 * [0] total ms of the sample + level-1 passes, [1] total ms of the filter-pass scans (the
 * HBM/MFMA-bound kernel), [2] searches timed, [3] total rows those filter-pass launches scanned
 * (a collection's rows minus the level-1 prefix whose candidates were kept).
 */
int crag_set_profiling(crag_index* h, int on);
/*
 * Test hook: the scan kernels' approximate score (|x|^2 - 2<q,x> for l2, -<q,x> for ip) of EVERY
 * row, out_scores_dev [B][rows] float32, through path 1 (gemv) or 2 (mfma).  Tests use it to
 * check the float32 error bound the exactness certificate relies on.  Slow; not a product path.
 */
int crag_debug_dense_scores(crag_index* h, const void* queries_dev, int B, int path, float* out_scores_dev,
                            void* stream);
int crag_get_profile(crag_index* h, double* out4, int reset);

#ifdef __cplusplus
}
#endif
#endif
