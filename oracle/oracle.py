# -*- coding: utf-8 -*-
"""
CPU oracle for the retrieval scoring path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (code-rag_amd/) never imports it and fails loudly when its HIP
library is missing.

What is restated, and from where (paths relative to /root/reference):

  * rerank / filter / sort / truncate  -- retriever/hybrid_search.py:69-125, line by line
    (`rerank_candidates` below).  Pinned by the known-answer cases derivable from that
    source; see tests/test_oracle_known_answers.py.
  * collection naming                  -- api/main.py:77-82 (`collection_name_for`).
  * /search response shape + filters   -- api/main.py:151-164,191-216 (`search_response`).
  * search_by_symbol                   -- retriever/hybrid_search.py:128-146.
  * the vector query itself (hybrid_search.py:60-64) lives in chromadb==1.3.0 /
    chroma-hnswlib==0.7.3 (requirements.txt:11-12), which is NOT vendored in the reference
    and NOT installed here.  Its published behaviour is restated as an EXACT brute-force
    scan: space "l2" (the collection is created with no hnsw:space, embed_ingest.py:49-53,
    and Chroma's default space is squared L2), results ascending by distance,
    n_results = 2*top_k (hybrid_search.py:62).  The reference holds no golden vector, test
    or fixture that pins a distance or an ordered id list at that boundary, so

        PARITY UNPINNED at the Chroma boundary (distance numerics, HNSW recall).

    Everything after that boundary (the ~55 lines of Python rerank) is pinned as above.

Canonical arithmetic (what "exact" means for both this oracle and the GPU path):

    inputs (fp32, or bf16 upcast exactly) are promoted to fp64;
    eight strided partial sums  acc[d & 7] += t*t   with t = q[d]-x[d]   (l2)
                                acc[d & 7] += q[d]*x[d]                   (ip)
    each product rounded to fp64, then each add rounded to fp64 (no FMA contraction);
    s = ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7));
    l2: dist = s            ip: dist = 1.0 - s            (Chroma's "ip" space)
    the distance handed on is float32(dist) -- Chroma returns float32 distances -- and
    candidates are ordered by (float32 distance ascending, row id ascending).
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

LANES = 8  # strided partial sums of the canonical accumulation order


# --------------------------------------------------------------------------------------
# dtype helpers (bf16 is carried as uint16 bit patterns on the host)
# --------------------------------------------------------------------------------------
def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit pattern (uint16), round-to-nearest-even.  NaN stays NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    nan = np.isnan(a)
    out = rounded.astype(np.uint16)
    if nan.any():
        out = np.where(nan, ((u >> np.uint32(16)) | np.uint32(0x40)).astype(np.uint16), out)
    return out


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def as_f64(rows: np.ndarray) -> np.ndarray:
    """fp32 array or bf16 bit-pattern (uint16) array -> exact fp64 values."""
    if rows.dtype == np.uint16:
        return bf16_bits_to_f32(rows).astype(np.float64)
    if rows.dtype == np.float32:
        return rows.astype(np.float64)
    raise TypeError(f"rows must be float32 or uint16 (bf16 bits), got {rows.dtype}")


# --------------------------------------------------------------------------------------
# Q1 (external, restated): exact distances in the canonical arithmetic
# --------------------------------------------------------------------------------------
def canonical_distances(q: np.ndarray, rows: np.ndarray, metric: str = "l2") -> np.ndarray:
    """float32 distances of one query against every row, canonical order (module docstring)."""
    q64 = as_f64(np.asarray(q).reshape(-1))
    x64 = as_f64(rows)
    n, d = x64.shape
    assert q64.shape[0] == d, (q64.shape, x64.shape)
    acc = np.zeros((n, LANES), dtype=np.float64)
    full = (d // LANES) * LANES
    for s in range(0, full, LANES):
        if metric == "l2":
            t = q64[s:s + LANES][None, :] - x64[:, s:s + LANES]
            acc += t * t
        else:
            acc += q64[s:s + LANES][None, :] * x64[:, s:s + LANES]
    for dd in range(full, d):  # ragged tail: lane = d & 7
        if metric == "l2":
            t = q64[dd] - x64[:, dd]
            acc[:, dd & 7] += t * t
        else:
            acc[:, dd & 7] += q64[dd] * x64[:, dd]
    s = ((acc[:, 0] + acc[:, 1]) + (acc[:, 2] + acc[:, 3])) + ((acc[:, 4] + acc[:, 5]) + (acc[:, 6] + acc[:, 7]))
    if metric == "l2":
        dist = s
    elif metric == "ip":
        dist = 1.0 - s
    else:
        raise ValueError(f"unknown metric {metric!r}")
    return dist.astype(np.float32)


def knn_candidates(q: np.ndarray, rows: np.ndarray, n_results: int, metric: str = "l2",
                   id_base: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """The n_results nearest rows, ordered by (float32 distance, id) ascending.

    Restates what hybrid_search.py:60-64 asks of `col.query(..., n_results=top_k*2)`:
    ids and distances ascending by distance, fewer than n_results when the collection is
    smaller.  Returns (ids int64, distances float32).
    """
    d32 = canonical_distances(q, rows, metric)
    n = d32.shape[0]
    k = min(int(n_results), n)
    if k <= 0:
        return np.zeros(0, np.int64), np.zeros(0, np.float32)
    order = np.lexsort((np.arange(n, dtype=np.int64), d32))[:k]  # last key is primary
    return order.astype(np.int64) + int(id_base), d32[order]


# --------------------------------------------------------------------------------------
# R2 / R3 metadata derivation (hybrid_search.py:98-115)
# --------------------------------------------------------------------------------------
PATH_NEUTRAL, PATH_PENALISED, PATH_BOOSTED = 0, 1, 2
_PATH_FACTOR = {PATH_NEUTRAL: 1.0, PATH_PENALISED: 0.1, PATH_BOOSTED: 1.2}


def path_class(path: Any) -> int:
    """if/elif chain of hybrid_search.py:99-111 as a per-chunk class (query independent)."""
    p = str(path).lower()
    if "/test/" in p or "/tests/" in p or "/spec/" in p or "/__tests__/" in p:
        return PATH_PENALISED
    elif p.endswith(".d.ts"):
        return PATH_PENALISED
    elif "/src/" in p or "/lib/" in p:
        return PATH_BOOSTED
    return PATH_NEUTRAL


def symbol_matches(name: Any, query: str) -> bool:
    """hybrid_search.py:93,98,114:  name = str(name).lower(); name and name in query.lower()."""
    n = str(name).lower()
    return bool(n) and (n in query.lower())


# --------------------------------------------------------------------------------------
# R1..R5 + O1: hybrid_search.py:69-125 restated line by line
# --------------------------------------------------------------------------------------
def rerank_candidates(ids: Sequence[Any], dists: Sequence[float], metas: Sequence[Optional[dict]],
                      docs: Optional[Sequence[Optional[str]]], query: str, top_k: int = 10,
                      symbol_boost: float = 2.0) -> List[Dict[str, Any]]:
    """Takes what col.query returned for one query and produces search()'s return value."""
    if len(ids) == 0:                                   # :66-67
        return []
    if docs is None:
        docs = [None] * len(ids)                        # :73
    candidates = []
    for i, cid in enumerate(ids):                       # :76
        dist = float(dists[i])                          # Chroma hands back float32-origin floats
        meta = metas[i] or {}                           # :78
        doc_text = docs[i]
        base_score = 1.0 / (1.0 + dist)                 # :82
        candidates.append({
            "id": cid,
            "score": base_score,
            "metadata": meta,
            "text_full": doc_text,
            "text_preview": (doc_text or "")[:80].replace("\n", " "),   # :89
        })
    q_lower = query.lower()                             # :93
    final_list = []
    for cand in candidates:
        m = cand["metadata"]
        name = str(m.get("name", "")).lower()           # :98
        path = str(m.get("path", "")).lower()           # :99
        boost = 1.0
        if "/test/" in path or "/tests/" in path or "/spec/" in path or "/__tests__/" in path:
            boost *= 0.1                                # :105-106
        elif path.endswith(".d.ts"):
            boost *= 0.1                                # :107-108
        elif "/src/" in path or "/lib/" in path:
            boost *= 1.2                                # :110-111
        if name and name in q_lower:
            boost *= symbol_boost                       # :114-115
        cand["score"] *= boost                          # :117
        if cand["score"] > 0.01:                        # :120 (strict)
            final_list.append(cand)
    final_list.sort(key=lambda x: x["score"], reverse=True)   # :124 (stable)
    return final_list[:top_k]                           # :125


def rerank_numeric(dists: np.ndarray, classes: np.ndarray, matches: np.ndarray, top_k: int,
                   symbol_boost: float) -> Tuple[np.ndarray, np.ndarray]:
    """The same R1..R5 on numeric columns (what the GPU kernel sees): returns
    (positions into the candidate list, fp64 scores), already filtered, sorted, truncated."""
    out: List[Tuple[float, int]] = []
    for i in range(len(dists)):
        base = 1.0 / (1.0 + float(np.float32(dists[i])))
        boost = 1.0
        boost *= _PATH_FACTOR[int(classes[i])]
        if bool(matches[i]):
            boost *= float(symbol_boost)
        score = base * boost
        if score > 0.01:
            out.append((score, i))
    out.sort(key=lambda t: t[0], reverse=True)
    out = out[:top_k]
    return (np.array([i for _, i in out], dtype=np.int64),
            np.array([s for s, _ in out], dtype=np.float64))


# --------------------------------------------------------------------------------------
# S2: HybridSearcher.search() end to end on in-memory collections
# --------------------------------------------------------------------------------------
def oracle_search(query_vec: np.ndarray, query: str, rows: Optional[np.ndarray],
                  ids: Sequence[str], metas: Sequence[Optional[dict]],
                  docs: Optional[Sequence[Optional[str]]] = None, top_k: int = 10,
                  symbol_boost: float = 2.0, include_documents: bool = False,
                  metric: str = "l2") -> List[Dict[str, Any]]:
    """search() for one collection held as arrays.  rows=None models a missing collection
    (hybrid_search.py:53-56) and returns []."""
    if rows is None or len(rows) == 0:
        return []
    pos, d32 = knn_candidates(query_vec, rows, top_k * 2, metric)     # :60-64
    c_ids = [ids[int(p)] for p in pos]
    c_metas = [metas[int(p)] for p in pos]
    c_docs = [docs[int(p)] for p in pos] if (include_documents and docs is not None) else None
    return rerank_candidates(c_ids, [float(x) for x in d32], c_metas, c_docs, query, top_k, symbol_boost)


def oracle_search_numeric(query_vec: np.ndarray, rows: np.ndarray, classes: np.ndarray,
                          matches: np.ndarray, top_k: int, symbol_boost: float,
                          metric: str = "l2", id_base: int = 0):
    """Numeric-column form used against the C ABI: returns (ids int64, scores f64, dists f32)."""
    pos, d32 = knn_candidates(query_vec, rows, top_k * 2, metric)
    sel, scores = rerank_numeric(d32, classes[pos], matches[pos], top_k, symbol_boost)
    return pos[sel] + int(id_base), scores, d32[sel]


def collection_name_for(workspace_id: Optional[str]) -> str:
    """api/main.py:77-82."""
    raw = (workspace_id or "").strip()
    if not raw or raw == "default":
        return "code_chunks"
    safe_id = re.sub(r"[^a-zA-Z0-9_-]", "_", raw)[:64] or "default"
    return "code_chunks" if safe_id == "default" else f"code_chunks__{safe_id}"


def search_response(query: str, results: List[Dict[str, Any]], path_prefix: Optional[str] = None,
                    kind: Optional[str] = None, min_score: Optional[float] = None) -> Dict[str, Any]:
    """api/main.py:191-216: map to SearchResult, then the three post-filters, then total."""
    out = []
    for r in results:
        m = r["metadata"]
        out.append({
            "id": r["id"],
            "score": float(r["score"]),
            "name": str(m.get("name", "")),
            "kind": str(m.get("kind", "")),
            "path": str(m.get("path", "")),
            "start_line": int(m.get("start_line", 0) or 0),
            "end_line": int(m.get("end_line", 0) or 0),
            "text_preview": str(r.get("text_preview", "")),
        })
    if path_prefix:
        pref = path_prefix.replace("\\", "/")
        out = [x for x in out if x["path"].replace("\\", "/").startswith(pref)]
    if kind:
        out = [x for x in out if (x["kind"] or "").lower() == kind.lower()]
    if min_score is not None:
        out = [x for x in out if x["score"] >= float(min_score)]
    return {"query": query, "total": len(out), "results": out}


def search_by_symbol(symbol: str, ids: Sequence[str], metas: Sequence[Optional[dict]],
                     docs: Sequence[Optional[str]], top_k: int = 10):
    """hybrid_search.py:133-143, the exact-match branch: first top_k chunks whose metadata
    name equals `symbol` (case sensitive), score fixed at 1.0.  Returns None when nothing
    matches, in which case the caller falls back to search() (:146)."""
    out = []
    for i, m in enumerate(metas):
        if (m or {}).get("name") == symbol:
            out.append({"id": ids[i], "metadata": m, "text_full": docs[i], "score": 1.0})
            if len(out) >= top_k:
                break
    return out or None


# --------------------------------------------------------------------------------------
# synthetic data, SURVEY.md section 8(d)
# --------------------------------------------------------------------------------------
def synth_unit_rows(n: int, d: int, seed: int, dtype: str = "f32") -> np.ndarray:
    """standard normal fp32 [n,d], each row / its L2 norm, cast to dtype ('f32' | 'bf16')."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    if dtype == "f32":
        return x
    if dtype == "bf16":
        return f32_to_bf16_bits(x)
    raise ValueError(dtype)
