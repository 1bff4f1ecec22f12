/* HNSW restatement  --  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * What /root/reference/retriever/hybrid_search.py:60-64 really runs on the CPU is not a brute-force scan but Chroma's
 * approximate index: chromadb==1.3.0 over chroma-hnswlib==0.7.3 (requirements.txt:11-12), neither of which is in
 * /root/reference, this image or the GPU box.  This file restates hnswlib's PUBLISHED algorithm (Malkov & Yashunin,
 * "Efficient and robust approximate nearest neighbor search using Hierarchical Navigable Small World graphs";
 * hnswlib's hnswalg.h: addPoint / searchBaseLayer / getNeighborsByHeuristic2 / mutuallyConnectNewElement / searchKnn)
 * with the parameters Chroma documents as its defaults for a collection created without metadata
 * (indexer/embed_ingest.py:49-53): space l2 (squared L2), M = 16 (32 links on level 0), ef_construction = 100,
 * ef_search = 100 -- so that bench.py can put an approximate-index CPU leg and a recall@k figure beside the GPU path
 * ("recall@k vs HNSW").
 *
 * PARITY UNPINNED: the reference holds no fixture for this step and hnswlib cannot be run here.  Level draws use this
 * file's own generator (hnswlib: std::mt19937 seeded 100), the build inserts from several threads (as hnswlib's
 * ParallelFor does, so the graph depends on timing), and float32 sums run in a different order -- the index is
 * statistically the same kind of graph, not the same graph.  Use its recall and queries/s as that: estimates.
 *
 * Only tests/ and bench.py's cpu_baseline leg load this.
 *
 *   gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC hnsw_oracle.c -o _build/libhnsw_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
    int n, d, M, M0, efc;
    const float* x;      /* [n][d], borrowed: the caller keeps it alive */
    int* links0;         /* [n][1 + M0]: count, neighbours (level 0) */
    int** linksU;        /* [n]: NULL or [level][1 + M] for levels 1..level */
    int* level;          /* [n] */
    unsigned char* lock; /* [n] byte spin locks: a link list is read and written under its node's lock */
    int enter, maxlevel;
    omp_lock_t top;      /* entry point / maxlevel */
    double mult;
} hnsw_t;

typedef struct { float d; int id; } cand_t;

/* ---- binary heaps of (distance, id): max-heap (worst on top) and min-heap (closest on top) ---------------------- */
typedef struct { cand_t* a; int n, cap; } heap_t;
static void heap_init(heap_t* h, int cap) { h->a = (cand_t*)malloc(sizeof(cand_t) * (size_t)cap); h->n = 0; h->cap = cap; }
static void heap_grow(heap_t* h) { h->cap *= 2; h->a = (cand_t*)realloc(h->a, sizeof(cand_t) * (size_t)h->cap); }
static inline int before(cand_t u, cand_t v, int maxheap) { return maxheap ? u.d > v.d : u.d < v.d; }
static void heap_push(heap_t* h, cand_t c, int maxheap) {
    if (h->n == h->cap) heap_grow(h);
    int i = h->n++;
    while (i > 0) {
        const int p = (i - 1) >> 1;
        if (!before(c, h->a[p], maxheap)) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = c;
}
static cand_t heap_pop(heap_t* h, int maxheap) {
    const cand_t top = h->a[0], last = h->a[--h->n];
    int i = 0;
    for (;;) {
        int c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && before(h->a[c + 1], h->a[c], maxheap)) ++c;
        if (!before(h->a[c], last, maxheap)) break;
        h->a[i] = h->a[c];
        i = c;
    }
    if (h->n) h->a[i] = last;
    return top;
}

static inline float l2sqr(const float* a, const float* b, int d) {
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < d; ++i) {
        const float t = a[i] - b[i];
        s += t * t;
    }
    return s;
}

static inline void node_lock(hnsw_t* h, int i) { while (__atomic_test_and_set(&h->lock[i], __ATOMIC_ACQUIRE)) { } }
static inline void node_unlock(hnsw_t* h, int i) { __atomic_clear(&h->lock[i], __ATOMIC_RELEASE); }
static inline int* list_of(hnsw_t* h, int i, int lev) { return lev == 0 ? h->links0 + (size_t)i * (1 + h->M0) : h->linksU[i] + (size_t)(lev - 1) * (1 + h->M); }
/* copy of a node's neighbours at a level (taken under its lock) */
static inline int neighbours(hnsw_t* h, int i, int lev, int* out) {
    node_lock(h, i);
    const int* l = list_of(h, i, lev);
    const int c = l[0];
    memcpy(out, l + 1, sizeof(int) * (size_t)c);
    node_unlock(h, i);
    return c;
}

/* per-thread scratch */
typedef struct { unsigned* tag; unsigned epoch; heap_t cand, top, tmp; int* nb; cand_t* sel; } scratch_t;
static void scratch_init(scratch_t* s, int n, int M0, int ef) {
    s->tag = (unsigned*)calloc((size_t)n, sizeof(unsigned));
    s->epoch = 0;
    heap_init(&s->cand, 4 * ef + 64);
    heap_init(&s->top, ef + 64);
    heap_init(&s->tmp, ef + 64);
    s->nb = (int*)malloc(sizeof(int) * (size_t)(M0 + 1));
    s->sel = (cand_t*)malloc(sizeof(cand_t) * (size_t)(ef + M0 + 64));
}
static void scratch_free(scratch_t* s) { free(s->tag); free(s->cand.a); free(s->top.a); free(s->tmp.a); free(s->nb); free(s->sel); }

/* hnswlib searchBaseLayer / searchBaseLayerST: the ef closest to q reachable from ep on one level, left in s->top
   (max-heap).  The loop ends when the closest open candidate is farther than the worst of a FULL result set. */
static void search_layer(hnsw_t* h, scratch_t* s, const float* q, int ep, float ep_d, int lev, int ef) {
    if (++s->epoch == 0) { memset(s->tag, 0, sizeof(unsigned) * (size_t)h->n); s->epoch = 1; }
    s->cand.n = s->top.n = 0;
    cand_t e = {ep_d, ep};
    heap_push(&s->cand, e, 0);
    heap_push(&s->top, e, 1);
    s->tag[ep] = s->epoch;
    float bound = ep_d;
    while (s->cand.n) {
        const cand_t c = s->cand.a[0];
        if (c.d > bound && s->top.n == ef) break;
        heap_pop(&s->cand, 0);
        const int cnt = neighbours(h, c.id, lev, s->nb);
        for (int j = 0; j < cnt; ++j) {
            const int v = s->nb[j];
            if (s->tag[v] == s->epoch) continue;
            s->tag[v] = s->epoch;
            const float dv = l2sqr(q, h->x + (size_t)v * h->d, h->d);
            if (s->top.n < ef || dv < bound) {
                cand_t nv = {dv, v};
                heap_push(&s->cand, nv, 0);
                heap_push(&s->top, nv, 1);
                if (s->top.n > ef) heap_pop(&s->top, 1);
                bound = s->top.a[0].d;
            }
        }
    }
}

/* hnswlib getNeighborsByHeuristic2: from candidates (distance to the base point in .d), closest first, keep one only
   if it is closer to the base point than to every neighbour kept so far; at most M.  In: c[0..n) any order.  Out: the
   kept ones in c[0..ret), closest first. */
static int cmp_cand(const void* a, const void* b) {
    const cand_t *u = (const cand_t*)a, *v = (const cand_t*)b;
    return u->d < v->d ? -1 : (u->d > v->d ? 1 : (u->id < v->id ? -1 : u->id > v->id));
}
static int select_heuristic(hnsw_t* h, cand_t* c, int n, int M) {
    if (n < M) { qsort(c, (size_t)n, sizeof(cand_t), cmp_cand); return n; }
    qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);
    int kept = 0;
    for (int i = 0; i < n && kept < M; ++i) {
        int good = 1;
        const float* xi = h->x + (size_t)c[i].id * h->d;
        for (int r = 0; r < kept; ++r) {
            if (l2sqr(xi, h->x + (size_t)c[r].id * h->d, h->d) < c[i].d) { good = 0; break; }
        }
        if (good) c[kept++] = c[i];
    }
    return kept;
}

/* hnswlib mutuallyConnectNewElement: the new node takes the heuristic's choice (at most M); every chosen neighbour
   gets the back link, re-running the heuristic over its own list when that is full.  Returns the closest chosen one. */
static int connect_new(hnsw_t* h, scratch_t* s, int cur, int lev) {
    const int Mmax = lev ? h->M : h->M0;
    int n = 0;
    while (s->top.n) s->sel[n++] = heap_pop(&s->top, 1);
    n = select_heuristic(h, s->sel, n, h->M);
    node_lock(h, cur);
    int* l = list_of(h, cur, lev);
    l[0] = n;
    for (int i = 0; i < n; ++i) l[1 + i] = s->sel[i].id;
    node_unlock(h, cur);
    const int closest = n ? s->sel[0].id : -1;
    cand_t* tmp = (cand_t*)malloc(sizeof(cand_t) * (size_t)(Mmax + 1));
    for (int i = 0; i < n; ++i) {
        const int v = s->sel[i].id;
        node_lock(h, v);
        int* lv = list_of(h, v, lev);
        const int cnt = lv[0];
        int present = 0;
        for (int j = 0; j < cnt; ++j) present |= lv[1 + j] == cur;
        if (!present) {
            if (cnt < Mmax) {
                lv[1 + cnt] = cur;
                lv[0] = cnt + 1;
            } else {
                const float* xv = h->x + (size_t)v * h->d;
                tmp[0].d = s->sel[i].d;  /* distance(cur, v) */
                tmp[0].id = cur;
                for (int j = 0; j < cnt; ++j) {
                    tmp[1 + j].id = lv[1 + j];
                    tmp[1 + j].d = l2sqr(xv, h->x + (size_t)lv[1 + j] * h->d, h->d);
                }
                const int k = select_heuristic(h, tmp, cnt + 1, Mmax);
                lv[0] = k;
                for (int j = 0; j < k; ++j) lv[1 + j] = tmp[j].id;
            }
        }
        node_unlock(h, v);
    }
    free(tmp);
    return closest;
}

static int greedy_descent(hnsw_t* h, scratch_t* s, const float* q, int ep, float* ep_d, int from_level, int to_level) {
    float cur_d = *ep_d;
    for (int lev = from_level; lev > to_level; --lev) {
        int changed = 1;
        while (changed) {
            changed = 0;
            const int cnt = neighbours(h, ep, lev, s->nb);
            for (int j = 0; j < cnt; ++j) {
                const float dv = l2sqr(q, h->x + (size_t)s->nb[j] * h->d, h->d);
                if (dv < cur_d) { cur_d = dv; ep = s->nb[j]; changed = 1; }
            }
        }
    }
    *ep_d = cur_d;
    return ep;
}

static void add_point(hnsw_t* h, scratch_t* s, int cur) {
    const int lev = h->level[cur];
    omp_set_lock(&h->top);
    int ep = h->enter;
    const int maxlevel = h->maxlevel;
    const int holds_top = lev > maxlevel;  /* a new top level: keep everyone else out, as hnswlib does */
    if (!holds_top) omp_unset_lock(&h->top);
    const float* q = h->x + (size_t)cur * h->d;
    float ep_d = l2sqr(q, h->x + (size_t)ep * h->d, h->d);
    if (lev < maxlevel) ep = greedy_descent(h, s, q, ep, &ep_d, maxlevel, lev);
    for (int l = lev < maxlevel ? lev : maxlevel; l >= 0; --l) {
        search_layer(h, s, q, ep, ep_d, l, h->efc);
        const int nxt = connect_new(h, s, cur, l);
        if (nxt >= 0) { ep = nxt; ep_d = l2sqr(q, h->x + (size_t)ep * h->d, h->d); }
    }
    if (holds_top) {
        h->enter = cur;
        h->maxlevel = lev;
        omp_unset_lock(&h->top);
    }
}

/* splitmix64: the level draws (hnswlib: -log(U) * 1 / log(M) from std::mt19937) */
static inline uint64_t mix(uint64_t* st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void* hnsw_build(const float* x, int n, int d, int M, int ef_construction, uint64_t seed, int threads) {
    if (n <= 0 || d <= 0 || M < 2) return NULL;
    hnsw_t* h = (hnsw_t*)calloc(1, sizeof(hnsw_t));
    h->n = n; h->d = d; h->M = M; h->M0 = 2 * M; h->efc = ef_construction > M ? ef_construction : M; h->x = x;
    h->mult = 1.0 / log((double)M);
    h->links0 = (int*)calloc((size_t)n * (size_t)(1 + h->M0), sizeof(int));
    h->linksU = (int**)calloc((size_t)n, sizeof(int*));
    h->level = (int*)malloc(sizeof(int) * (size_t)n);
    h->lock = (unsigned char*)calloc((size_t)n, 1);
    omp_init_lock(&h->top);
    uint64_t st = seed;
    for (int i = 0; i < n; ++i) {
        const double u = ((double)(mix(&st) >> 11) + 1.0) / 9007199254740993.0;  /* (0, 1) */
        const int lev = (int)(-log(u) * h->mult);
        h->level[i] = lev;
        if (lev > 0) h->linksU[i] = (int*)calloc((size_t)lev * (size_t)(1 + M), sizeof(int));
    }
    h->enter = 0;
    h->maxlevel = h->level[0];
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        scratch_t s;
        scratch_init(&s, n, h->M0, h->efc);
#pragma omp for schedule(dynamic, 32)
        for (int i = 1; i < n; ++i) add_point(h, &s, i);
        scratch_free(&s);
    }
    return h;
}

/* hnswlib searchKnn for nq queries (OpenMP over the queries): ids [nq][k] (-1 padded) and squared-L2 distances, closest
   first, ties by id; ef = max(ef_search, k) */
int hnsw_search(void* hv, const float* q, int nq, int k, int ef_search, int threads, int64_t* ids, float* dist) {
    hnsw_t* h = (hnsw_t*)hv;
    if (!h || k <= 0) return -1;
    const int ef = ef_search > k ? ef_search : k;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        scratch_t s;
        scratch_init(&s, h->n, h->M0, ef);
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < nq; ++b) {
            const float* qb = q + (size_t)b * h->d;
            int ep = h->enter;
            float ep_d = l2sqr(qb, h->x + (size_t)ep * h->d, h->d);
            ep = greedy_descent(h, &s, qb, ep, &ep_d, h->maxlevel, 0);
            search_layer(h, &s, qb, ep, ep_d, 0, ef);
            int n = 0;
            while (s.top.n) s.sel[n++] = heap_pop(&s.top, 1);
            qsort(s.sel, (size_t)n, sizeof(cand_t), cmp_cand);
            for (int j = 0; j < k; ++j) {
                ids[(size_t)b * k + j] = j < n ? s.sel[j].id : -1;
                dist[(size_t)b * k + j] = j < n ? s.sel[j].d : INFINITY;
            }
        }
        scratch_free(&s);
    }
    return 0;
}

/* mean out-degree on level 0 and the top level (sanity figures for the tests) */
void hnsw_stats(void* hv, double* mean_degree0, int* maxlevel) {
    hnsw_t* h = (hnsw_t*)hv;
    double s = 0.0;
    for (int i = 0; i < h->n; ++i) s += h->links0[(size_t)i * (1 + h->M0)];
    *mean_degree0 = s / h->n;
    *maxlevel = h->maxlevel;
}

void hnsw_free(void* hv) {
    hnsw_t* h = (hnsw_t*)hv;
    if (!h) return;
    for (int i = 0; i < h->n; ++i) free(h->linksU[i]);
    free(h->linksU); free(h->links0); free(h->level); free(h->lock);
    omp_destroy_lock(&h->top);
    free(h);
}
