// K3 + R1..R5: threshold selection, candidate selection, exact re-scoring, merge and rerank.
//
//   k_row_norms     |x|^2 per row (fp32) + max, once per collection
//   k_tau_select    per query: M-th smallest score of the dense sample -> filter threshold tau
//   k_finalize      per query: sort the filtered candidates by (score, row), re-score the M best in
//                   the canonical fp64 order (bit-identical to oracle/oracle.py), certify that no
//                   row outside the re-scored set can enter the 2*top_k nearest, emit CragEntry[kc]
//   k_merge_rerank  per query: merge G shards' entries by (float32 distance, id), then
//                   hybrid_search.py:82-125 verbatim in fp64 (base score, path/symbol boost,
//                   strict > 0.01 filter, stable descending sort, truncate to top_k)
//   k_rescore_wide                          fallback tier 1: a query whose certificate failed gets EVERY buffered
//                                           candidate re-scored exactly (duplicated chunks, near-ties beyond the slack)
//   k_exact_keys_b / k_hist64_b / k_radix_pick / k_gather_le_b
//                                           fallback tier 2: canonical distance of every row + radix select, batched
//                                           over the failed queries, no host round trip between the passes
#include "crag_common.h"

namespace {

__device__ __forceinline__ double load_elem_f64(const void* p, int dtype, long long idx) {
    if (dtype == CRAG_F32) return (double)((const float*)p)[idx];
    uint32_t u = ((uint32_t)((const unsigned short*)p)[idx]) << 16;
    return (double)__builtin_bit_cast(float, u);
}

// ----------------------------------------------------------------------------------------
// row norms: one wave per row, 16-byte coalesced loads, fp64 accumulation
__global__ __launch_bounds__(256) void k_row_norms(const uint4* __restrict__ rows, int dtype, int chunks,
                                                   long long n, float* __restrict__ xnorm,
                                                   unsigned int* __restrict__ max_bits) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    float wmax = 0.f;
    for (long long r = wave0; r < n; r += n_waves) {
        double acc = 0.0;
        for (int c = lane; c < chunks; c += 64) {
            const uint4 v = rows[r * chunks + c];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (dtype == CRAG_F32) {
                    const double f = (double)__builtin_bit_cast(float, w[k]);
                    acc += f * f;
                } else {
                    const double lo = (double)__builtin_bit_cast(float, w[k] << 16);
                    const double hi = (double)__builtin_bit_cast(float, w[k] & 0xffff0000u);
                    acc += lo * lo;
                    acc += hi * hi;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        const float f = (float)acc;
        if (lane == 0) xnorm[r] = f;
        // a NaN would slip through fmaxf: a row that is not a vector of finite numbers reads as an infinite norm
        wmax = fmaxf(wmax, f == f ? f : __builtin_inff());
    }
    if (lane == 0) atomicMax(max_bits, __builtin_bit_cast(unsigned int, wmax));
}

// ----------------------------------------------------------------------------------------
// Block-wide selection (1024 threads) over n 32-bit keys read through key(i).  Returns, to every thread,
// a key T with at least want + 1 keys <= T: the exact want-th smallest (0-based, want < n) unless the
// histogram bin that holds it is crowded (> KTH_LIST keys, e.g. all keys equal), in which case T is that
// bin's upper edge.  Selection, not sorting: min/max, a 2048-bin histogram of (key - min) >> shift, a
// block scan to the bin that holds the wanted rank, then the exact rank inside that bin (sparse when the
// wanted key sits in the tail of the score distribution, as it does here).
// scratch: KTH_SCRATCH 32-bit words of LDS; the caller's LDS writes of the keys must be fenced by the caller.
// 2048 bins: k_kth then needs 64 KiB of keys + 12.2 KiB of scratch = under half a CU's LDS, so two of its
// 1024-thread blocks share a CU (a batch of 1024 queries runs its selections in two rounds instead of four)
constexpr int KTH_BINS = 2048;
constexpr int KTH_BITS = 11;
constexpr int KTH_LIST = 1024;
constexpr int KTH_SCRATCH = KTH_BINS + KTH_LIST + 48 + 4;
template <typename KeyFn>
__device__ __forceinline__ uint32_t block_select(KeyFn key, int n, uint32_t want, uint32_t* scratch) {
    uint32_t* hist = scratch;              // [KTH_BINS]
    uint32_t* list = hist + KTH_BINS;      // [KTH_LIST]
    uint32_t* wsum = list + KTH_LIST;      // [48] wave totals, wave minima, wave maxima
    uint32_t* ctl = wsum + 48;             // [4]: bin, rank inside the bin, list length, result
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t mn = 0xffffffffu, mx = 0u;
    for (int i = tid; i < n; i += 1024) {
        const uint32_t k = key(i);
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
    for (int i = tid; i < KTH_BINS; i += 1024) hist[i] = 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o1 = __shfl_xor(mn, off), o2 = __shfl_xor(mx, off);
        mn = o1 < mn ? o1 : mn;
        mx = o2 > mx ? o2 : mx;
    }
    if (lane == 0) {
        wsum[16 + wv] = mn;
        wsum[32 + wv] = mx;
    }
    if (tid == 0) ctl[2] = 0;
    __syncthreads();
    mn = wsum[16];
    mx = wsum[32];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
        mn = wsum[16 + w] < mn ? wsum[16 + w] : mn;
        mx = wsum[32 + w] > mx ? wsum[32 + w] : mx;
    }
    const uint32_t range = mx - mn;
    const int bits = range ? 32 - __builtin_clz(range) : 0;
    const int shift = bits > KTH_BITS ? bits - KTH_BITS : 0;  // (key - mn) >> shift < KTH_BINS
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[(key(i) - mn) >> shift], 1u);
    __syncthreads();
    // exclusive scan over the bins: thread t owns bins 2t and 2t+1
    const uint32_t h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    const uint32_t mine = h0 + h1;
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    const uint32_t exc = base + inc - mine;
    const uint32_t M = want + 1;
    if (exc < M && M <= exc + mine) {  // exactly one thread: the wanted key is in one of its bins
        uint32_t e = exc;
        int bsel = 2 * tid;
        if (e + h0 < M) { e += h0; ++bsel; }
        ctl[0] = (uint32_t)bsel;
        ctl[1] = M - e - 1;  // rank of the wanted key inside bin bsel
    }
    __syncthreads();
    const uint32_t bsel = ctl[0], rank = ctl[1];
    for (int i = tid; i < n; i += 1024) {
        const uint32_t k = key(i);
        if (((k - mn) >> shift) == bsel) {
            const uint32_t p = atomicAdd(&ctl[2], 1u);
            if (p < KTH_LIST) list[p] = k;
        }
    }
    __syncthreads();
    const uint32_t L = ctl[2];
    if (L > KTH_LIST) {  // crowded bin: its upper edge is a valid (looser) answer
        if (tid == 0) {
            const unsigned long long edge = (unsigned long long)mn + (((unsigned long long)bsel + 1) << shift) - 1;
            ctl[3] = edge > mx ? mx : (uint32_t)edge;
        }
    } else if (tid < (int)L) {
        const uint32_t k = list[tid];
        uint32_t lt = 0, eq = 0;
        for (uint32_t j = 0; j < L; ++j) {
            const uint32_t o = list[j];
            lt += o < k ? 1u : 0u;
            eq += o == k ? 1u : 0u;
        }
        if (lt <= rank && rank < lt + eq) ctl[3] = k;  // duplicates write the same value
    }
    __syncthreads();
    const uint32_t r = ctl[3];
    __syncthreads();  // scratch may be reused at once
    return r;
}

// Segmented candidate buffer (MfmaArgs in crag_common.h): exclusive prefix of the per-segment counts of one
// query into segoff[0 .. n_seg] (LDS), returns the total; *overflow if a segment ran over.  n_seg <= 1024.
__device__ __forceinline__ int block_seg_offsets(const uint32_t* __restrict__ segcnt_q, int n_seg, int seg_cap,
                                                 uint32_t* segoff, uint32_t* wsum, bool* overflow) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t c = tid < n_seg ? segcnt_q[tid] : 0u;
    const bool ov = c > (uint32_t)seg_cap;
    c = ov ? (uint32_t)seg_cap : c;
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    const int any_ov = __syncthreads_or(ov ? 1 : 0);
    uint32_t base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    if (tid < n_seg) segoff[tid] = base + inc - c;
    if (tid == n_seg - 1) segoff[n_seg] = base + inc;
    __syncthreads();
    *overflow = any_ov != 0;
    return (int)segoff[n_seg];
}
// slot of the i-th gathered candidate: segment by binary search in segoff, then its position inside
__device__ __forceinline__ int seg_slot(const uint32_t* segoff, int n_seg, int seg_cap, int i) {
    int lo = 0, hi = n_seg;  // segoff[lo] <= i < segoff[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (segoff[mid] <= (uint32_t)i) lo = mid; else hi = mid;
    }
    return lo * seg_cap + (i - (int)segoff[lo]);
}

// level 0 seeds the candidate buffer with the sample rows at or below tau (KthArgs::seed_*): seed number j
__device__ __forceinline__ void kth_seed(const KthArgs& a, int q, uint32_t j, uint32_t key, uint32_t row) {
    const unsigned long long e = ((unsigned long long)key << 32) | row;
    if (a.seed_n_seg > 0) {
        const uint32_t sg = j % (uint32_t)a.seed_n_seg, slot = j / (uint32_t)a.seed_n_seg;
        if (slot < (uint32_t)a.seed_seg_cap) a.seed_cand[(long long)q * a.seed_stride + (long long)sg * a.seed_seg_cap + slot] = e;
    } else if (j < CRAG_CAND_CAP) {
        a.seed_cand[(long long)q * CRAG_CAND_CAP + j] = e;
    }
}
// ... and publishes how many there are (all threads of the block call this)
__device__ __forceinline__ void kth_seed_counts(const KthArgs& a, int q, uint32_t total) {
    const uint32_t tid = threadIdx.x;
    if (a.seed_n_seg > 0) {
        const uint32_t ns = (uint32_t)a.seed_n_seg;
        for (uint32_t sg = tid; sg < ns; sg += 1024) a.seed_cnt[(long long)q * ns + sg] = total > sg ? (total - sg + ns - 1) / ns : 0u;
    } else if (tid == 0) {
        a.seed_cnt[CRAG_CNT(q)] = total;
    }
}

// per query: tau = score of the M-th smallest ordered key of a strided list (KthArgs in crag_common.h);
// any tau >= the true M-th smallest keeps the cascade exact (see block_select for when it is looser)
__global__ __launch_bounds__(1024) void k_kth(KthArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* keys = (uint32_t*)smem;                        // [CRAG_CAND_CAP]
    uint32_t* scratch = keys + CRAG_CAND_CAP;                // [KTH_SCRATCH]
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const bool seg = a.n_seg > 0;
    const bool from_cand = seg || a.cnt != nullptr;
    long long n;
    if (seg) {
        bool ov;  // an overflowed segment still holds seg_cap rows <= the old tau
        n = block_seg_offsets(a.segcnt + (long long)q * a.n_seg, a.n_seg, a.seg_cap, scratch, scratch + KTH_BINS + KTH_LIST, &ov);
        if (n > CRAG_CAND_CAP) n = CRAG_CAND_CAP;  // the first CAP gathered rows are all <= the old tau
        if (!a.qd[q].keep_mid && tid < a.n_seg) a.segcnt[(long long)q * a.n_seg + tid] = 0;
    } else if (from_cand) {
        const unsigned int c = a.cnt[CRAG_CNT(q)];
        n = c < CRAG_CAND_CAP ? c : CRAG_CAND_CAP;  // an overflowed buffer still holds CAP rows <= the old tau
    } else {
        n = a.qd[q].n_sample;
    }
    __syncthreads();
    // reset only where a level 1 really ran and its survivors are NOT kept (the full scan then rescans from row 0).
    // A query whose collection has no level 1 (n_mid == n_sample) shares this launch with queries that have one
    // (crag_search_multi over collections of different sizes): its buffer holds the level-0 seeds, and its full
    // scan starts behind the sample rows, so the seeds must stay.
    if (!seg && a.cnt_reset && tid == 0 && a.qd[q].n_mid > a.qd[q].n_sample && !a.qd[q].keep_mid) a.cnt_reset[CRAG_CNT(q)] = 0;
    const bool tiled = a.tile_bpad > 0;
    const uint32_t* __restrict__ src = tiled ? a.src : a.src + (long long)q * a.q_stride + a.off0;
    auto src_at = [&](int i) -> uint32_t {
        return tiled ? src[((long long)(i >> 5) * a.tile_bpad + q) * 32 + (i & 31)] : src[(long long)i * a.e_stride];
    };
    if (n < (long long)a.M || (!from_cand && n == (long long)a.M)) {
        if (!from_cand && tid == 0) a.tau[q] = __builtin_inff();  // every row is a candidate
        if (!from_cand && a.seed_cand) {                           // ... and every sample row is seeded
            for (int i = tid; i < (int)n; i += 1024)
                kth_seed(a, q, (uint32_t)i, src_at(i), (uint32_t)i);
            __syncthreads();
            kth_seed_counts(a, q, (uint32_t)n);
        }
        return;                                                    // from_cand: keep the looser tau
    }
    for (int i = tid; i < (int)n; i += 1024) {
        const int e = seg ? seg_slot(scratch, a.n_seg, a.seg_cap, i) : i;
        keys[i] = src_at(e);
    }
    __syncthreads();
    const uint32_t t = block_select([&](int i) { return keys[i]; }, (int)n, (uint32_t)a.M - 1u, scratch);
    if (tid == 0) a.tau[q] = crag_fkey_inv(t);
    if (!from_cand && a.seed_cand) {  // sample row i IS row i of the collection; n <= 16384 = CRAG_CAND_CAP
        uint32_t* seed_total = scratch;  // block_select is done with its scratch
        if (tid == 0) *seed_total = 0;
        __syncthreads();
        for (int i = tid; i < (int)n; i += 1024) {
            const uint32_t k = keys[i];
            if (k <= t) kth_seed(a, q, atomicAdd(seed_total, 1u), k, (uint32_t)i);
        }
        __syncthreads();
        kth_seed_counts(a, q, *seed_total);
    }
}

// ----------------------------------------------------------------------------------------
// bitonic sort of P (power of two) 64-bit keys in LDS, ascending, by all threads of the block
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* s, int P) {
    const int T = blockDim.x;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (P >> 1); i += T) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b = a | j;
                const bool up = (a & k) == 0;
                const unsigned long long x = s[a], y = s[b];
                if ((x > y) == up) {
                    s[a] = y;
                    s[b] = x;
                }
            }
            __syncthreads();
        }
    }
}

// canonical fp64 partial sum of lane j (elements d = j, j+8, ...), oracle/oracle.py order.  Eight elements
// per step are loaded before the (sequential, order-defining) additions, so the loads overlap.
template <int DT>
__device__ __forceinline__ double canon_lane_t(const void* row, const void* q, int dim, int j, int metric, long long row_off) {
    double acc = 0.0;
    for (int d0 = j; d0 < dim; d0 += 64) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int d = d0 + 8 * u;
            av[u] = d < dim ? load_elem_f64(q, DT, d) : 0.0;
            bv[u] = d < dim ? load_elem_f64(row, DT, row_off + d) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (d0 + 8 * u < dim) {
                double p;
                if (metric == CRAG_L2) {
                    const double t = __dsub_rn(av[u], bv[u]);
                    p = __dmul_rn(t, t);
                } else {
                    p = __dmul_rn(av[u], bv[u]);
                }
                acc = __dadd_rn(acc, p);
            }
        }
    }
    return acc;
}
__device__ __forceinline__ double canon_lane(const void* row, const void* q, int dtype, int dim, int j, int metric,
                                             long long row_off) {
    return dtype == CRAG_F32 ? canon_lane_t<CRAG_F32>(row, q, dim, j, metric, row_off)
                             : canon_lane_t<CRAG_BF16>(row, q, dim, j, metric, row_off);
}

__device__ __forceinline__ double combine8(double acc) {
    // ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)); IEEE addition commutes, so all 8 lanes agree
    acc = __dadd_rn(acc, __shfl_xor(acc, 1));
    acc = __dadd_rn(acc, __shfl_xor(acc, 2));
    acc = __dadd_rn(acc, __shfl_xor(acc, 4));
    return acc;
}

constexpr int FIN_GATHER = 2048;  // candidates that survive the selection cut (m + 1 plus score ties), else sort all
__global__ __launch_bounds__(1024) void k_finalize(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* skeys = (unsigned long long*)smem;                            // [CRAG_CAND_CAP]
    unsigned long long* ekeys = skeys + CRAG_CAND_CAP;                                // [CRAG_MAX_M]
    double* red = (double*)(ekeys + CRAG_MAX_M);                                      // [16]
    uint32_t* sel_scratch = (uint32_t*)(red + 16);                                    // [KTH_SCRATCH]
    uint32_t* sel_count = sel_scratch + KTH_SCRATCH;                                  // [4]
    uint32_t* seg_off = sel_count + 4;                                                // [1024 + 1] segment offsets
    const int q = a.qmap ? a.qmap[blockIdx.x] : blockIdx.x;  // fallback passes run on a subset of the batch
    const QueryDesc D = a.qd[q];
    const int tid = threadIdx.x;
    unsigned int status = CRAG_ST_OK;
    const bool seg = a.n_seg > 0;
    int c;
    if (seg) {
        bool ov;
        c = block_seg_offsets(a.segcnt + (long long)q * a.n_seg, a.n_seg, a.seg_cap, seg_off, sel_scratch, &ov);
        if (ov || c > CRAG_CAND_CAP) status = CRAG_ST_OVERFLOW;
        if (c > CRAG_CAND_CAP) c = CRAG_CAND_CAP;
    } else {
        const unsigned int raw_cnt = a.cnt[CRAG_CNT(q)];
        if (raw_cnt > CRAG_CAND_CAP) status = CRAG_ST_OVERFLOW;
        c = raw_cnt < CRAG_CAND_CAP ? (int)raw_cnt : CRAG_CAND_CAP;
    }
    const int m = c < a.M ? c : a.M;  // candidates re-scored exactly
    const unsigned long long* __restrict__ cq = a.cand + (long long)q * a.cand_stride;
    auto cand_at = [&](int i) { return cq[seg ? seg_slot(seg_off, a.n_seg, a.seg_cap, i) : i]; };
    // skeys[0 .. m] = the m + 1 smallest candidates in (score, row) order (skeys[m], the best candidate NOT
    // re-scored, bounds the certificate).  Small buffers are sorted whole; big ones are first cut down to the
    // candidates whose score is <= the (m + 1)-th smallest score (selection), and only those are sorted.
    int c_sorted = c;
    if (c > 256) {
        for (int i = tid; i < c; i += 1024) skeys[i] = cand_at(i);
        __syncthreads();
        const uint32_t want = (uint32_t)(m < c ? m : c - 1);
        const uint32_t t = block_select([&](int i) { return (uint32_t)(skeys[i] >> 32); }, c, want, sel_scratch);
        // compact in place is not possible without a second buffer: gather into the tail region of skeys
        unsigned long long* gathered = skeys + CRAG_CAND_CAP - FIN_GATHER;
        if (tid == 0) *sel_count = 0;
        __syncthreads();
        if (c <= CRAG_CAND_CAP - FIN_GATHER) {
            for (int i = tid; i < c; i += 1024) {
                const unsigned long long k = skeys[i];
                if ((uint32_t)(k >> 32) <= t) {
                    const uint32_t p = atomicAdd(sel_count, 1u);
                    if (p < FIN_GATHER) gathered[p] = k;
                }
            }
        }
        __syncthreads();
        const uint32_t L = *sel_count;
        __syncthreads();
        if (c <= CRAG_CAND_CAP - FIN_GATHER && L <= FIN_GATHER) {  // L >= want + 1 by construction
            for (int i = tid; i < (int)L; i += 1024) skeys[i] = gathered[i];  // FIN_GATHER <= CAP / 2: no overlap
            c_sorted = (int)L;
            __syncthreads();
        }
    } else {
        for (int i = tid; i < c; i += 1024) skeys[i] = cand_at(i);
    }
    __syncthreads();
    if (c_sorted <= 1024) {
        // rank by counting (keys are unique: the row id is part of the key): every thread reads the same LDS
        // word at a time (broadcast), two barriers in all -- a bitonic network of this size costs 21-55
        unsigned long long k = ~0ull;
        int rank = 0;
        if (tid < c_sorted) {
            k = skeys[tid];
            for (int j = 0; j < c_sorted; ++j) rank += skeys[j] < k ? 1 : 0;
        }
        __syncthreads();
        if (tid < c_sorted) skeys[rank] = k;
        __syncthreads();
    } else {
        int P = 2048;
        while (P < c_sorted) P <<= 1;
        for (int i = c_sorted + tid; i < P; i += 1024) skeys[i] = ~0ull;
        __syncthreads();
        bitonic_sort_u64(skeys, P);
    }
    const size_t esz = a.dtype == CRAG_F32 ? 4 : 2;
    const void* qv = (const char*)a.queries + (size_t)q * a.dim_pad * esz;

    // |q|^2 in fp64 (error bound only)
    {
        double part = 0.0;
        for (int d = tid; d < a.dim; d += blockDim.x) {
            const double v = load_elem_f64(qv, a.dtype, d);
            part += v * v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
            red[0] = s;
        }
        __syncthreads();
    }
    const double qn = red[0];
    if (!(qn <= 1.7976931348623157e308)) {
        // a query with a NaN or an infinite component (or a norm beyond float64) has no nearest neighbours: no results,
        // nothing for the fallback tiers to settle.  (The reference cannot get here: its queries come out of the embedder.)
        CragEntry* o = a.entries + (long long)q * a.kc;
        for (int i = tid; i < a.kc; i += blockDim.x) {
            CragEntry e;
            e.dist = __builtin_inff();
            e.flags = 0;
            e.id = 0x7fffffffffffffffll;
            o[i] = e;
        }
        if (tid == 0) a.status[q] = CRAG_ST_OK;
        return;
    }

    // exact re-scoring: 8 lanes per candidate.  The query and the candidates' rows are first staged in LDS with
    // coalesced 16-byte loads, all in flight at once (the sorted keys occupy skeys[0 .. m]; the rest of that
    // 128-KiB region is free now), so the order-defining add chains read LDS instead of chasing 2-byte loads
    {
        unsigned char* stage_q = (unsigned char*)(skeys + 1024);  // m + 1 <= CRAG_MAX_M + 1 <= 1024 keys stay live
        unsigned char* stage_r = stage_q + 4096;                  // dim_pad * esz <= 4096
        const int row_bytes = a.dim_pad * (int)esz, chunks = row_bytes >> 4;
        int rpr = (int)((sizeof(unsigned long long) * (CRAG_CAND_CAP - 1024) - 4096) / (size_t)row_bytes);
        if (rpr > 128) rpr = 128;  // 1024 threads / 8 lanes
        for (int c = tid; c < chunks; c += 1024) ((uint4*)stage_q)[c] = ((const uint4*)qv)[c];
        for (int base = 0; base < m; base += rpr) {
            const int nr = m - base < rpr ? m - base : rpr;
            __syncthreads();  // the previous round's readers are done
            // four loads in flight per thread, through the GLOBAL address space: D.rows comes out of the descriptor,
            // which makes it a generic pointer to the compiler, and FLAT loads are waited for one at a time
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
            const __attribute__((address_space(1))) u32x4_t* grows = (const __attribute__((address_space(1))) u32x4_t*)D.rows;
            const int total = nr * chunks;
            for (int idx0 = tid; idx0 < total; idx0 += 4 * 1024) {
                u32x4_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = idx0 + u * 1024;
                    if (idx < total) {
                        const int r = idx / chunks, c = idx - r * chunks;
                        const uint32_t row = (uint32_t)(skeys[base + r] & 0xffffffffull);
                        v[u] = grows[(long long)row * chunks + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = idx0 + u * 1024;
                    if (idx < total) ((u32x4_t*)stage_r)[idx] = v[u];
                }
            }
            __syncthreads();
            const int r = tid >> 3, j = tid & 7;
            double acc = 0.0;
            if (r < nr) acc = canon_lane(stage_r, stage_q, a.dtype, a.dim, j, a.metric, (long long)r * a.dim_pad);
            acc = combine8(acc);
            if (r < nr && j == 0) {
                const double dist = a.metric == CRAG_L2 ? acc : __dsub_rn(1.0, acc);
                ekeys[base + r] = ((unsigned long long)crag_fkey((float)dist) << 32) | (uint32_t)(skeys[base + r] & 0xffffffffull);
            }
        }
    }
    __syncthreads();

    // rank the m exact keys by counting (keys are unique: the row id is part of the key)
    const int kc_valid = m < a.kc ? m : a.kc;
    CragEntry* out = a.entries + (long long)q * a.kc;
    __shared__ unsigned int kth_bits;
    for (int i = tid; i < m; i += blockDim.x) {
        const unsigned long long k = ekeys[i];
        int rank = 0;
        for (int t = 0; t < m; ++t) rank += ekeys[t] < k;
        if (rank < a.kc) {
            const uint32_t row = (uint32_t)(k & 0xffffffffull);
            uint32_t flags = 0;
            if (D.path_class) flags |= (uint32_t)((const __attribute__((address_space(1))) unsigned char*)D.path_class)[row] & CRAG_FLAG_CLASS_MASK;
            if (D.name_hash && a.qhash_off) {
                const unsigned long long h = ((const __attribute__((address_space(1))) unsigned long long*)D.name_hash)[row];
                int lo = a.qhash_off[q], hi = a.qhash_off[q + 1];
                if (h != 0ull) {
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const unsigned long long v = a.qhash[mid];
                        if (v == h) {
                            flags |= CRAG_FLAG_SYMBOL;
                            break;
                        }
                        if (v < h) lo = mid + 1; else hi = mid;
                    }
                }
            }
            CragEntry e;
            e.dist = crag_fkey_inv((uint32_t)(k >> 32));
            e.flags = flags;
            e.id = D.id_base + (long long)row;
            out[rank] = e;
            if (rank == kc_valid - 1) kth_bits = (uint32_t)(k >> 32);
        }
    }
    for (int i = kc_valid + tid; i < a.kc; i += blockDim.x) {
        CragEntry e;
        e.dist = __builtin_inff();
        e.flags = 0;
        e.id = 0x7fffffffffffffffll;
        out[i] = e;
    }
    __syncthreads();

    // exactness certificate.  Every row that was NOT re-scored has approximate score >= s_M: the
    // buffer is sorted, so the remaining candidates start at skeys[m], and rows outside the buffer
    // have s > tau >= every buffered score (s_M = tau when the whole buffer was re-scored).  Its exact distance is therefore at least
    // s_M + offset - eps.  If the kc-th exact float32 distance is strictly below the float32
    // rounding of that bound, no such row can enter (or tie into) the kc nearest.
    if (tid == 0) {
        // trusted == 1: the keys are exact and the buffer provably holds the kc nearest (tier 2): nothing to certify.
        // trusted == 2: the keys are exact but the buffer is still only "every row with approximate score <= tau"
        // (tier 1): the order inside the buffer is exact, what remains to certify is the rows OUTSIDE it.
        if (a.trusted != 1 && status == CRAG_ST_OK && (long long)(a.trusted == 2 ? c : m) < D.n_rows && kc_valid > 0) {
            // tightest known lower bound on the approximate score of every row not re-scored
            const double sM = (m < c && a.trusted == 0) ? (double)crag_fkey_inv((uint32_t)(skeys[m] >> 32)) : (double)a.tau[q];
            const double u = 5.9604644775390625e-08;  // 2^-24
            const double X = (double)D.max_xnorm;
            const double qx = sqrt(qn * X);
            double eps, lb;
            if (a.metric == CRAG_L2) {
                // every partial sum of a scan is bounded by |x|^2 + 2 |q||x|: the 16x16x32 scan starts its accumulators
                // at |x|^2 (crag_scan_q16.hip); the others add |x|^2 last, which this bound covers as well
                eps = 2.0 * u * ((double)(a.dim_pad + 4) * (2.0 * qx + X) + 3.0 * X + 1e-30);
                lb = sM + qn - eps - 1e-12 * (qn + X);
            } else {
                eps = 2.0 * u * ((double)(a.dim_pad + 4) * qx + 1e-30);
                lb = 1.0 + sM - eps - 1e-12 * (1.0 + qx);
            }
            float lbf = (float)lb;
            if ((double)lbf > lb) lbf = nextafterf(lbf, -__builtin_inff());
            const float dk = crag_fkey_inv(kth_bits);
            if (!(dk < lbf)) status = CRAG_ST_UNCERTIFIED;
        }
        a.status[q] = status;
    }
}

// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_merge_rerank(const CragEntry* __restrict__ entries, int G, int B, int kc,
                                                      int top_k, double symbol_boost, long long* __restrict__ out_ids,
                                                      double* __restrict__ out_scores, float* __restrict__ out_dist,
                                                      int* __restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = G * kc;
    long long* ids = (long long*)smem;             // [n]
    double* score = (double*)(ids + n);            // [kc]
    uint32_t* fk = (uint32_t*)(score + kc);        // [n]
    uint32_t* fl = fk + n;                         // [n]
    int* sel = (int*)(fl + n);                     // [kc]   position in [0,n) of the j-th nearest
    int* keep = sel + kc;                          // [kc]
    __shared__ int n_valid, n_keep;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) {
        n_valid = 0;
        n_keep = 0;
    }
    for (int i = tid; i < n; i += blockDim.x) {
        const int g = i / kc, j = i - g * kc;
        const CragEntry e = entries[((long long)g * B + q) * kc + j];
        // an entry with a NaN distance (only a caller of crag_merge_rerank_device can hand one in) counts as padding: with
        // its own key it could rank ahead of real entries or behind padding and leave a hole in sel[]
        const bool nan = e.dist != e.dist;
        ids[i] = nan ? 0x7fffffffffffffffll : e.id;
        fk[i] = nan ? 0xffffffffu : crag_fkey(e.dist);
        fl[i] = e.flags;
    }
    __syncthreads();
    // the kc nearest by (float32 distance, id): rank by counting
    for (int i = tid; i < n; i += blockDim.x) {
        const long long id = ids[i];
        if (id == 0x7fffffffffffffffll) continue;
        const uint32_t k = fk[i];
        int rank = 0;
        for (int t = 0; t < n; ++t) {
            const uint32_t kt = fk[t];
            rank += (kt < k) || (kt == k && ids[t] < id);
        }
        if (rank < kc) {
            sel[rank] = i;
            atomicAdd(&n_valid, 1);
        }
    }
    __syncthreads();
    const int nv = n_valid;  // = min(kc, number of real entries)
    // R1..R4: hybrid_search.py:82,99-121
    for (int j = tid; j < nv; j += blockDim.x) {
        const int i = sel[j];
        const double dist = (double)crag_fkey_inv(fk[i]);
        const double base = __ddiv_rn(1.0, __dadd_rn(1.0, dist));
        const uint32_t f = fl[i];
        const uint32_t cls = f & CRAG_FLAG_CLASS_MASK;
        double boost = cls == 1u ? 0.1 : (cls == 2u ? 1.2 : 1.0);
        if (f & CRAG_FLAG_SYMBOL) boost = __dmul_rn(boost, symbol_boost);
        const double s = __dmul_rn(base, boost);
        score[j] = s;
        keep[j] = s > 0.01;
    }
    __syncthreads();
    // R5: stable descending sort + truncate, hybrid_search.py:124-125
    for (int j = tid; j < nv; j += blockDim.x) {
        if (!keep[j]) continue;
        const double s = score[j];
        int rank = 0;
        for (int t = 0; t < nv; ++t) rank += keep[t] && (score[t] > s || (score[t] == s && t < j));
        atomicAdd(&n_keep, 1);
        if (rank < top_k) {
            const int i = sel[j];
            out_ids[(long long)q * top_k + rank] = ids[i];
            out_scores[(long long)q * top_k + rank] = s;
            out_dist[(long long)q * top_k + rank] = crag_fkey_inv(fk[i]);
        }
    }
    __syncthreads();
    const int nk = n_keep < top_k ? n_keep : top_k;
    for (int j = nk + tid; j < top_k; j += blockDim.x) {
        out_ids[(long long)q * top_k + j] = -1;
        out_scores[(long long)q * top_k + j] = 0.0;
        out_dist[(long long)q * top_k + j] = __builtin_inff();
    }
    if (tid == 0) out_counts[q] = nk;
}

// ----------------------------------------------------------------------------------------
// Fallback tier 1.  One workgroup per failed query: every buffered candidate (all rows whose approximate score is
// <= tau) gets its canonical float32 distance, 8 lanes per candidate, and its key is rewritten IN PLACE as
// (exact distance key << 32 | row).  k_finalize then runs on these queries with trusted = 2.
__global__ __launch_bounds__(1024) void k_rescore_wide(FinalizeArgs a, unsigned long long* __restrict__ cand_rw) {
    __shared__ uint32_t seg_off[1024 + 1];
    __shared__ uint32_t wsum[16];
    const int q = a.qmap[blockIdx.x];
    const QueryDesc D = a.qd[q];
    const int tid = threadIdx.x;
    const bool seg = a.n_seg > 0;
    int c;
    if (seg) {
        bool ov;
        c = block_seg_offsets(a.segcnt + (long long)q * a.n_seg, a.n_seg, a.seg_cap, seg_off, wsum, &ov);
    } else {
        c = (int)a.cnt[CRAG_CNT(q)];
    }
    if (c > CRAG_CAND_CAP) c = CRAG_CAND_CAP;  // (an overflowed query goes to tier 2 anyway)
    unsigned long long* __restrict__ cq = cand_rw + (long long)q * a.cand_stride;
    const size_t esz = a.dtype == CRAG_F32 ? 4 : 2;
    const void* qv = (const char*)a.queries + (size_t)q * a.dim_pad * esz;
    const int j = tid & 7;
    for (int i0 = 0; i0 < c; i0 += 128) {  // the trip count is block-uniform: combine8 shuffles among all lanes
        const int i = i0 + (tid >> 3);
        int slot = 0;
        uint32_t row = 0;
        double acc = 0.0;
        if (i < c) {
            slot = seg ? seg_slot(seg_off, a.n_seg, a.seg_cap, i) : i;
            row = (uint32_t)(cq[slot] & 0xffffffffull);
            acc = canon_lane(D.rows, qv, a.dtype, a.dim, j, a.metric, (long long)row * a.dim_pad);
        }
        acc = combine8(acc);
        if (i < c && j == 0) {
            const double dist = a.metric == CRAG_L2 ? acc : __dsub_rn(1.0, acc);
            cq[slot] = ((unsigned long long)crag_fkey((float)dist) << 32) | row;
        }
    }
}

// ----------------------------------------------------------------------------------------
// Fallback tier 1b: thresholds for a CERTIFIED RE-SCAN.  A failed query still has kc exact distances in its entries
// (of a subset of the rows), so d0 = the largest of them bounds the true kc-th nearest distance from above.  Every row
// that can enter or tie into the result has exact distance <= d0, hence approximate score <= d0 - |q|^2 + eps; the
// re-scan collects every row at or below tau' = that bound plus a few float32 steps of margin, and re-scoring ALL of
// them exactly settles the query (k_finalize with trusted = 2 certifies against tau').  Queries that were fine get
// tau = -inf: the re-scan appends nothing for them.  One block per query of the batch; clears the failed queries'
// candidate counters.
__global__ __launch_bounds__(256) void k_tau_refine(FinalizeArgs a, float* __restrict__ tau, unsigned int* __restrict__ cnt,
                                                    uint32_t* __restrict__ segcnt) {
    __shared__ double red[4];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    if (a.status[q] == CRAG_ST_OK) {
        if (tid == 0) tau[q] = -__builtin_inff();
        return;
    }
    const QueryDesc D = a.qd[q];
    const size_t esz = a.dtype == CRAG_F32 ? 4 : 2;
    const void* qv = (const char*)a.queries + (size_t)q * a.dim_pad * esz;
    double part = 0.0;
    for (int d = tid; d < a.dim; d += 256) {
        const double v = load_elem_f64(qv, a.dtype, d);
        part += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const double qn = red[0] + red[1] + red[2] + red[3];
        float d0 = a.entries[(long long)q * a.kc + a.kc - 1].dist;  // +inf when fewer than kc rows were found: no bound
        for (int i = 0; i < 4; ++i) d0 = nextafterf(d0, __builtin_inff());  // a float32 tie with d0 must be collected too
        const double u = 5.9604644775390625e-08;  // 2^-24
        const double X = (double)D.max_xnorm;
        const double qx = sqrt(qn * X);
        double t;
        if (a.metric == CRAG_L2) {
            const double eps = 2.0 * u * ((double)(a.dim_pad + 4) * (2.0 * qx + X) + 3.0 * X + 1e-30);
            t = (double)d0 - qn + 1.01 * eps + 2e-12 * (qn + X);
        } else {
            const double eps = 2.0 * u * ((double)(a.dim_pad + 4) * qx + 1e-30);
            t = (double)d0 - 1.0 + 1.01 * eps + 2e-12 * (1.0 + qx);
        }
        float tf = (float)t;
        if ((double)tf < t) tf = nextafterf(tf, __builtin_inff());
        tau[q] = tf;
        cnt[CRAG_CNT(q)] = 0;
    }
    if (segcnt)
        for (int i = tid; i < a.n_seg; i += 256) segcnt[(long long)q * a.n_seg + i] = 0;
}

// ----------------------------------------------------------------------------------------
// Fallback tier 2, batched over F failed queries (f = blockIdx.x % F runs fastest, so the workgroups that read one
// block of rows for different queries are dispatched together and share it through the L2).
// keys[f][row] = (canonical float32 distance key << 32 | row), one thread per (row, query).
__global__ __launch_bounds__(256) void k_exact_keys_b(const QueryDesc* __restrict__ qd, const int* __restrict__ qmap, int F,
                                                      const void* __restrict__ queries, int dtype, int dim, int dim_pad,
                                                      int metric, unsigned long long* __restrict__ keys, long long key_stride) {
    const int f = blockIdx.x % F;
    const long long r = (long long)(blockIdx.x / F) * blockDim.x + threadIdx.x;
    const int q = qmap[f];
    const long long n = qd[q].n_rows;
    if (r >= n) return;
    const void* __restrict__ rows = qd[q].rows;
    const size_t esz = dtype == CRAG_F32 ? 4 : 2;
    const void* query = (const char*)queries + (size_t)q * dim_pad * esz;  // staged: stride dim_pad, zero-padded
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // eight elements per step through 16-byte loads (rows and staged queries are zero-padded to dim_pad, a multiple of
    // eight elements, and a padded element adds +0.0, which changes no partial sum)
    // (global address space spelled out: a pointer read from the descriptor is generic, and FLAT loads serialise)
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const __attribute__((address_space(1))) u32x4_t* __restrict__ rv =
        (const __attribute__((address_space(1))) u32x4_t*)((const char*)rows + (size_t)r * dim_pad * esz);
    const uint4* __restrict__ qv4 = (const uint4*)query;
    for (int d0 = 0; d0 < dim; d0 += 8) {
        double x[8], y[8];
        if (dtype == CRAG_F32) {
            const uint4 a0 = qv4[d0 >> 2], a1 = qv4[(d0 >> 2) + 1];
            const u32x4_t b0 = rv[d0 >> 2], b1 = rv[(d0 >> 2) + 1];
            const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = (double)__builtin_bit_cast(float, aw[j]);
                y[j] = (double)__builtin_bit_cast(float, bw[j]);
            }
        } else {
            const uint4 a0 = qv4[d0 >> 3];
            const u32x4_t b0 = rv[d0 >> 3];
            const uint32_t aw[4] = {a0.x, a0.y, a0.z, a0.w}, bw[4] = {b0.x, b0.y, b0.z, b0.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[2 * j] = (double)__builtin_bit_cast(float, aw[j] << 16);
                x[2 * j + 1] = (double)__builtin_bit_cast(float, aw[j] & 0xffff0000u);
                y[2 * j] = (double)__builtin_bit_cast(float, bw[j] << 16);
                y[2 * j + 1] = (double)__builtin_bit_cast(float, bw[j] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double p;
            if (metric == CRAG_L2) {
                const double t = __dsub_rn(x[j], y[j]);
                p = __dmul_rn(t, t);
            } else {
                p = __dmul_rn(x[j], y[j]);
            }
            acc[j] = __dadd_rn(acc[j], p);
        }
    }
    double s = __dadd_rn(__dadd_rn(__dadd_rn(acc[0], acc[1]), __dadd_rn(acc[2], acc[3])),
                         __dadd_rn(__dadd_rn(acc[4], acc[5]), __dadd_rn(acc[6], acc[7])));
    if (metric == CRAG_IP) s = __dsub_rn(1.0, s);
    keys[(long long)f * key_stride + r] = ((unsigned long long)crag_fkey((float)s) << 32) | (unsigned long long)(uint32_t)r;
}

// radix select of the want-th smallest 64-bit key per query, 8 bits per pass, state on the device
__global__ __launch_bounds__(256) void k_radix_init(const QueryDesc* __restrict__ qd, const int* __restrict__ qmap, int kc,
                                                    CragRadixState* __restrict__ st, unsigned int* __restrict__ hist,
                                                    unsigned int* __restrict__ cnt) {
    const int f = blockIdx.x;
    hist[f * 256 + threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        cnt[CRAG_CNT(qmap[f])] = 0;  // the gather refills this query's flat buffer from slot 0
        const long long n = qd[qmap[f]].n_rows;
        st[f].prefix = 0;
        st[f].mask = 0;
        st[f].want = (unsigned int)(n < kc ? n : kc);
    }
}

__global__ __launch_bounds__(256) void k_hist64_b(const unsigned long long* __restrict__ keys, long long key_stride,
                                                  const QueryDesc* __restrict__ qd, const int* __restrict__ qmap,
                                                  const CragRadixState* __restrict__ st, int shift, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    const int f = blockIdx.y;
    const long long n = qd[qmap[f]].n_rows;
    const unsigned long long prefix = st[f].prefix, mask = st[f].mask;
    const unsigned long long* __restrict__ k0 = keys + (long long)f * key_stride;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long k = k0[i];
        if ((k & mask) == prefix) atomicAdd(&h[(unsigned int)(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[f * 256 + threadIdx.x], h[threadIdx.x]);
}

// one block per query: the bin that holds the want-th smallest matching key joins the prefix; hist is cleared for the next pass
__global__ __launch_bounds__(256) void k_radix_pick(CragRadixState* __restrict__ st, int shift, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    const int f = blockIdx.x;
    h[threadIdx.x] = hist[f * 256 + threadIdx.x];
    hist[f * 256 + threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int want = st[f].want, cum = 0;
        if (want > 0) {
            int bin = 255;
            for (int i = 0; i < 256; ++i) {
                if (cum + h[i] >= want) {
                    bin = i;
                    break;
                }
                cum += h[i];
            }
            st[f].want = want - cum;
            st[f].prefix |= (unsigned long long)bin << shift;
            st[f].mask |= 0xffull << shift;
        }
    }
}

// the keys <= the selected threshold (exactly min(kc, n) of them: keys are unique) into the query's flat buffer
__global__ __launch_bounds__(256) void k_gather_le_b(const unsigned long long* __restrict__ keys, long long key_stride,
                                                     const QueryDesc* __restrict__ qd, const int* __restrict__ qmap,
                                                     const CragRadixState* __restrict__ st, unsigned long long* __restrict__ cand,
                                                     long long cand_stride, unsigned int* __restrict__ cnt) {
    const int f = blockIdx.y;
    const int q = qmap[f];
    const long long n = qd[q].n_rows;
    if (st[f].mask == 0) return;  // an empty collection: nothing selected
    const unsigned long long thr = st[f].prefix;
    const unsigned long long* __restrict__ k0 = keys + (long long)f * key_stride;
    unsigned long long* __restrict__ out = cand + (long long)q * cand_stride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long k = k0[i];
        if (k <= thr) {
            const unsigned int pos = atomicAdd(&cnt[CRAG_CNT(q)], 1u);
            if (pos < CRAG_CAND_CAP) out[pos] = k;
        }
    }
}

}  // namespace

int crag_launch_row_norms(const void* rows, int dtype, int dim_pad, long long n, float* xnorm, float* max_out,
                          hipStream_t s) {
    if (n <= 0) return 0;
    const int chunks = dim_pad * (dtype == CRAG_F32 ? 4 : 2) / 16;
    long long blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_row_norms, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)rows, dtype, chunks, n,
                       xnorm, (unsigned int*)max_out);
    return (int)hipGetLastError();
}

int crag_launch_kth(const KthArgs& a, hipStream_t s) {
    const size_t lds = sizeof(uint32_t) * (CRAG_CAND_CAP + KTH_SCRATCH);
    static std::atomic<uint64_t> lds_ok;
    crag_allow_dynamic_lds((const void*)k_kth, lds, lds_ok);
    hipLaunchKernelGGL(k_kth, dim3(a.B), dim3(1024), lds, s, a);
    return (int)hipGetLastError();
}

int crag_launch_finalize(const FinalizeArgs& a, hipStream_t s) {
    const size_t lds = sizeof(unsigned long long) * (CRAG_CAND_CAP + CRAG_MAX_M) + sizeof(double) * 16 +
                       sizeof(uint32_t) * (KTH_SCRATCH + 4 + 1024 + 4);
    static std::atomic<uint64_t> lds_ok;
    crag_allow_dynamic_lds((const void*)k_finalize, lds, lds_ok);
    hipLaunchKernelGGL(k_finalize, dim3(a.B), dim3(1024), lds, s, a);
    return (int)hipGetLastError();
}

int crag_launch_merge_rerank(const CragEntry* entries, int G, int B, int kc, int top_k, double symbol_boost,
                             long long* out_ids, double* out_scores, float* out_dist, int* out_counts,
                             hipStream_t s) {
    const int n = G * kc;
    const size_t lds = (size_t)n * (8 + 4 + 4) + (size_t)kc * (8 + 4 + 4);
    if (lds > 60 * 1024) return -2;
    hipLaunchKernelGGL(k_merge_rerank, dim3(B), dim3(256), lds, s, entries, G, B, kc, top_k, symbol_boost, out_ids,
                       out_scores, out_dist, out_counts);
    return (int)hipGetLastError();
}

int crag_launch_tau_refine(const FinalizeArgs& a, float* tau, unsigned int* cnt, uint32_t* segcnt, hipStream_t s) {
    hipLaunchKernelGGL(k_tau_refine, dim3(a.B), dim3(256), 0, s, a, tau, cnt, segcnt);
    return (int)hipGetLastError();
}

int crag_launch_rescore_wide(const FinalizeArgs& a, unsigned long long* cand_rw, int n_failed, hipStream_t s) {
    hipLaunchKernelGGL(k_rescore_wide, dim3(n_failed), dim3(1024), 0, s, a, cand_rw);
    return (int)hipGetLastError();
}

int crag_launch_exact_select(const ExactSelectArgs& a, hipStream_t s) {
    if (a.F <= 0) return 0;
    if (a.n_max <= 0) {  // empty collections only: still write the counters the tier-2 finalize reads (k_radix_init is
        // the one writer of the failed queries' flat counters; level 0's k_kth wrote the segment counters only)
        hipLaunchKernelGGL(k_radix_init, dim3(a.F), dim3(256), 0, s, a.qd, a.qmap, a.kc, a.state, a.hist, a.cnt);
        return (int)hipGetLastError();
    }
    const long long row_blocks = (a.n_max + 255) / 256;
    if (row_blocks * a.F > 0x7fffffffll) return -2;
    hipLaunchKernelGGL(k_exact_keys_b, dim3((unsigned)(row_blocks * a.F)), dim3(256), 0, s, a.qd, a.qmap, a.F, a.queries, a.dtype,
                       a.dim, a.dim_pad, a.metric, a.keys, a.key_stride);
    hipLaunchKernelGGL(k_radix_init, dim3(a.F), dim3(256), 0, s, a.qd, a.qmap, a.kc, a.state, a.hist, a.cnt);
    long long blocks = row_blocks < 1024 ? row_blocks : 1024;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_hist64_b, dim3((unsigned)blocks, a.F), dim3(256), 0, s, a.keys, a.key_stride, a.qd, a.qmap, a.state, shift, a.hist);
        hipLaunchKernelGGL(k_radix_pick, dim3(a.F), dim3(256), 0, s, a.state, shift, a.hist);
    }
    hipLaunchKernelGGL(k_gather_le_b, dim3((unsigned)blocks, a.F), dim3(256), 0, s, a.keys, a.key_stride, a.qd, a.qmap, a.state, a.cand,
                       a.cand_stride, a.cnt);
    return (int)hipGetLastError();
}
