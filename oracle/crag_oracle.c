/*
 * CPU oracle, C part  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load the library
 * built from this file (oracle/_build/libcrag_oracle.so).  The product never links it.
 *
 * It restates, in plain C, the exact brute-force form of the vector query that
 * /root/reference/retriever/hybrid_search.py:60-64 delegates to chromadb==1.3.0
 * (requirements.txt:12, not vendored, not installed): squared-L2 (Chroma's default space,
 * /root/reference/indexer/embed_ingest.py:49-53 sets no hnsw:space) or "ip" (1 - dot),
 * n_results nearest rows ascending by distance.  PARITY UNPINNED at that boundary: the
 * reference holds no golden distances or id lists.  The arithmetic is the canonical order
 * documented in oracle/oracle.py (8 strided fp64 partial sums, fixed combine tree, result
 * rounded to float32, ties broken by row id); oracle.py's numpy version is the second,
 * independent statement of the same thing and tests/ check the two agree bit for bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC crag_oracle.c -o _build/libcrag_oracle.so
 *        (-ffp-contract=off is REQUIRED: no FMA contraction.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_F32 0
#define ORACLE_BF16 1
#define ORACLE_L2 0
#define ORACLE_IP 1

static inline double load_elem(const void* base, int dtype, size_t idx) {
    if (dtype == ORACLE_F32) return (double)((const float*)base)[idx];
    uint32_t u = ((uint32_t)((const uint16_t*)base)[idx]) << 16;
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
}

/* float32 canonical distance of one query against one row */
static float canon_dist(const void* rows, const void* q, int dtype, int metric, size_t row, int dim) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t off = row * (size_t)dim;
    for (int d = 0; d < dim; ++d) {
        double a = load_elem(q, dtype, (size_t)d);
        double b = load_elem(rows, dtype, off + (size_t)d);
        double p;
        if (metric == ORACLE_L2) {
            double t = a - b;
            p = t * t;
        } else {
            p = a * b;
        }
        acc[d & 7] = acc[d & 7] + p;
    }
    double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (metric == ORACLE_IP) s = 1.0 - s;
    return (float)s;
}

/* out[b*n + i] = canonical float32 distance of query b to row i */
int oracle_distances(const void* rows, int64_t n, int dim, int dtype, int metric, const void* queries,
                     int nq, float* out) {
    size_t esz = dtype == ORACLE_F32 ? 4 : 2;
    for (int b = 0; b < nq; ++b) {
        const void* q = (const char*)queries + (size_t)b * dim * esz;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) out[(size_t)b * n + i] = canon_dist(rows, q, dtype, metric, (size_t)i, dim);
    }
    return 0;
}

typedef struct {
    float d;
    int64_t id;
} cand_t;

static inline int cand_less(cand_t a, cand_t b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }

/* bounded insertion into an ascending list of at most k entries */
static void push_sorted(cand_t* list, int* cnt, int k, cand_t c) {
    if (*cnt == k && !cand_less(c, list[k - 1])) return;
    int pos = *cnt < k ? *cnt : k - 1;
    while (pos > 0 && cand_less(c, list[pos - 1])) {
        list[pos] = list[pos - 1];
        --pos;
    }
    list[pos] = c;
    if (*cnt < k) ++*cnt;
}

/*
 * For each query: the k nearest rows by (float32 distance, id) ascending.
 * out_ids[b*k + j], out_dist[b*k + j]; out_cnt[b] = min(k, n).  Unused slots: id -1, dist +inf.
 * Threads: OpenMP over row ranges, partial lists merged in thread order (the merge is
 * order independent because (d, id) is a strict total order).
 */
int oracle_knn(const void* rows, int64_t n, int dim, int dtype, int metric, const void* queries, int nq, int k,
               int64_t id_base, int64_t* out_ids, float* out_dist, int32_t* out_cnt) {
    size_t esz = dtype == ORACLE_F32 ? 4 : 2;
    if (k <= 0) return -1;
    for (int b = 0; b < nq; ++b) {
        const void* q = (const char*)queries + (size_t)b * dim * esz;
        cand_t* best = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
        int cnt = 0;
#pragma omp parallel
        {
            cand_t* loc = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
            int lc = 0;
#pragma omp for schedule(static) nowait
            for (int64_t i = 0; i < n; ++i) {
                cand_t c = {canon_dist(rows, q, dtype, metric, (size_t)i, dim), i};
                push_sorted(loc, &lc, k, c);
            }
#pragma omp critical
            {
                for (int j = 0; j < lc; ++j) push_sorted(best, &cnt, k, loc[j]);
            }
            free(loc);
        }
        for (int j = 0; j < k; ++j) {
            if (j < cnt) {
                out_ids[(size_t)b * k + j] = best[j].id + id_base;
                out_dist[(size_t)b * k + j] = best[j].d;
            } else {
                out_ids[(size_t)b * k + j] = -1;
                out_dist[(size_t)b * k + j] = INFINITY;
            }
        }
        out_cnt[b] = cnt;
        free(best);
    }
    return 0;
}

/*
 * R1..R5 of hybrid_search.py:82-125 on numeric columns: for one query's candidate list
 * (already ascending by distance), classes 0 neutral / 1 penalised (x0.1) / 2 boosted (x1.2),
 * match flag => x symbol_boost; keep score > 0.01; stable sort descending; first top_k.
 * Returns the number kept; out_pos are positions into the candidate list.
 */
int oracle_rerank(const float* dist, const uint8_t* cls, const uint8_t* match, int n_cand, int top_k,
                  double symbol_boost, int32_t* out_pos, double* out_score) {
    static const double factor[3] = {1.0, 0.1, 1.2};
    int m = 0;
    int32_t* pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_cand > 0 ? n_cand : 1));
    double* sc = (double*)malloc(sizeof(double) * (size_t)(n_cand > 0 ? n_cand : 1));
    for (int i = 0; i < n_cand; ++i) {
        double base = 1.0 / (1.0 + (double)dist[i]);
        double boost = 1.0;
        boost = boost * factor[cls[i] <= 2 ? cls[i] : 0];
        if (match[i]) boost = boost * symbol_boost;
        double s = base * boost;
        if (s > 0.01) {
            /* stable insertion by descending score */
            int p = m;
            while (p > 0 && sc[p - 1] < s) {
                sc[p] = sc[p - 1];
                pos[p] = pos[p - 1];
                --p;
            }
            sc[p] = s;
            pos[p] = i;
            ++m;
        }
    }
    int keep = m < top_k ? m : top_k;
    for (int i = 0; i < keep; ++i) {
        out_pos[i] = pos[i];
        out_score[i] = sc[i];
    }
    free(pos);
    free(sc);
    return keep;
}

/*
 * Throughput-oriented CPU baseline for bench.py (cpu_baseline.kind = "port"): the same scan
 * in plain float32 (one rounding per multiply-add as the compiler emits it, no canonical
 * order), followed by the same (distance, id) selection.  Not used for parity.
 */
int oracle_knn_fast_f32(const void* rows, int64_t n, int dim, int dtype, const void* queries, int nq, int k,
                        int64_t* out_ids, float* out_dist) {
    size_t esz = dtype == ORACLE_F32 ? 4 : 2;
    for (int b = 0; b < nq; ++b) {
        const void* q = (const char*)queries + (size_t)b * dim * esz;
        float* qf = (float*)malloc(sizeof(float) * (size_t)dim);
        for (int d = 0; d < dim; ++d) qf[d] = (float)load_elem(q, dtype, (size_t)d);
        cand_t* best = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
        int cnt = 0;
#pragma omp parallel
        {
            cand_t* loc = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
            int lc = 0;
#pragma omp for schedule(static) nowait
            for (int64_t i = 0; i < n; ++i) {
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (dtype == ORACLE_F32) {
                    const float* x = (const float*)rows + (size_t)i * dim;
                    for (int d = 0; d < dim; ++d) {
                        float t = qf[d] - x[d];
                        acc[d & 7] += t * t;
                    }
                } else {
                    const uint16_t* x = (const uint16_t*)rows + (size_t)i * dim;
                    for (int d = 0; d < dim; ++d) {
                        uint32_t u = ((uint32_t)x[d]) << 16;
                        float xf;
                        memcpy(&xf, &u, 4);
                        float t = qf[d] - xf;
                        acc[d & 7] += t * t;
                    }
                }
                float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
                cand_t c = {s, i};
                push_sorted(loc, &lc, k, c);
            }
#pragma omp critical
            {
                for (int j = 0; j < lc; ++j) push_sorted(best, &cnt, k, loc[j]);
            }
            free(loc);
        }
        for (int j = 0; j < k; ++j) {
            out_ids[(size_t)b * k + j] = j < cnt ? best[j].id : -1;
            out_dist[(size_t)b * k + j] = j < cnt ? best[j].d : INFINITY;
        }
        free(best);
        free(qf);
    }
    return 0;
}
