# -*- coding: utf-8 -*-
"""Retrieval metrics of the reference's mini eval, for the /search half only.

/root/reference/eval/run_eval.py posts every query of a JSONL file ({"id", "query", "expect": {"hit": ["path:name", ...]}},
e.g. eval/qa_min.jsonl) to /search, keys each result as "path:name" (:66-67), and reports Hit@K (any gold key among the
results, :69), MRR@K (reciprocal rank of the first gold key, :28-33,71) and the P50 / P95 latency with a linear
percentile (:19-26).  Here the same numbers come from any callable that answers like the /search route -- the in-process
route (`search_route`), or an HTTP client -- so that a machine which has the lodash checkout and an embedder can
compare against the reference's committed eval/results/report.json (hit@5 1.0, mrr@5 0.833).  The /explain half of that
script measures the LLM and is out of scope.

    python -m coderag_hip.eval --data eval/qa_min.jsonl --db data/coderag_db --workspace <id> --top_k 5
"""
from __future__ import annotations

import json
import time
from typing import Any, Callable, Dict, Iterable, List, Sequence


def percentile(values: Sequence[float], p: float) -> float:
    """linear interpolation between closest ranks (run_eval.py:19-26)"""
    if not values:
        return 0.0
    a = sorted(values)
    k = (len(a) - 1) * p
    f = int(k)
    c = min(f + 1, len(a) - 1)
    if f == c:
        return float(a[f])
    return float(a[f] + (a[c] - a[f]) * (k - f))


def result_keys(results: Iterable[Dict[str, Any]]) -> List[str]:
    """the eval's identity of a search result: "path:name" (run_eval.py:67)"""
    return [f'{it.get("path", "")}:{it.get("name", "")}' for it in results]


def hit_at_k(ranked_keys: Sequence[str], gold: Iterable[str]) -> float:
    gold = set(gold)
    return 1.0 if any(k in gold for k in ranked_keys) else 0.0


def mrr_at_k(ranked_keys: Sequence[str], gold: Iterable[str]) -> float:
    gold = set(gold)
    for rank, k in enumerate(ranked_keys, start=1):
        if k in gold:
            return 1.0 / rank
    return 0.0


def evaluate_search(search_fn: Callable[[str, int, float], Dict[str, Any]], qa_items: Iterable[Dict[str, Any]],
                    top_k: int = 5, symbol_boost: float = 2.0) -> Dict[str, Any]:
    """search_fn(query, top_k, symbol_boost) -> a /search response ({"results": [{path, name, ...}]}).
    Returns the "search" block of the reference's report.json plus the per-query rows."""
    hits, mrrs, lat, rows = [], [], [], []
    for ex in qa_items:
        gold = (ex.get("expect") or {}).get("hit", [])
        t0 = time.perf_counter()
        try:
            resp = search_fn(ex["query"], top_k, symbol_boost)
        except Exception as e:  # run_eval.py:54-56: an error counts as an empty result list
            resp = {"results": [], "error": f"{type(e).__name__}: {e}"}
        lat.append((time.perf_counter() - t0) * 1e3)
        keys = result_keys(resp.get("results", []))
        hits.append(hit_at_k(keys, gold))
        mrrs.append(mrr_at_k(keys, gold))
        rows.append({"id": ex.get("id"), "hit": hits[-1], "rr": mrrs[-1], "keys": keys})
    n = len(hits)
    return {"n": n, "search": {"top_k": top_k, "hit@k": sum(hits) / n if n else 0.0, "mrr@k": sum(mrrs) / n if n else 0.0,
                                "latency_ms": {"p50": percentile(lat, 0.50), "p95": percentile(lat, 0.95)}},
            "queries": rows}


def main(argv=None):
    import argparse
    from .api import search_route
    from .hybrid_search import HybridSearcher
    from .store import iter_jsonl
    ap = argparse.ArgumentParser(description="Hit@K / MRR@K of the GPU retrieval path on a qa JSONL file")
    ap.add_argument("--data", default="eval/qa_min.jsonl")
    ap.add_argument("--db", default="data/chroma_db")
    ap.add_argument("--workspace", default=None)
    ap.add_argument("--top_k", type=int, default=5)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    hs = HybridSearcher(db_path=args.db, device=args.device)
    rep = evaluate_search(lambda q, k, sb: search_route(hs, q, top_k=k, symbol_boost=sb, x_workspace_id=args.workspace),
                          list(iter_jsonl(args.data)), top_k=args.top_k)
    print(json.dumps(rep, ensure_ascii=False, indent=2))


if __name__ == "__main__":
    main()
