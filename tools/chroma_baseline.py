"""The reference's own k-NN step, timed on the host cores beside the GPU path -- used by
bench.py only when `chromadb` is importable on the box (it is not in this image, so this file has never
run here; bench.py reports any failure as text instead of raising).

The reference's call sequence (indexer/embed_ingest.py:38-54, 93; retriever/hybrid_search.py:60-64), with the
embeddings passed in instead of computed by Chroma's ONNX embedder: client -> create_collection with no
`hnsw:space` (default squared L2) -> add(ids, embeddings) in batches -> query(query_embeddings, n_results=2*top_k).
Returns queries/s of the query step and recall@2k of its HNSW answer against the exact ids."""
import time

import numpy as np


def run(rows_f32: np.ndarray, queries_f32: np.ndarray, top_k: int, exact_ids: np.ndarray, build_budget_s: float = 60.0):
    import chromadb
    n, _ = rows_f32.shape
    client = chromadb.EphemeralClient() if hasattr(chromadb, "EphemeralClient") else chromadb.Client()
    name = "code_chunks__bench"
    try:
        client.delete_collection(name)
    except Exception:
        pass
    col = client.create_collection(name=name)
    t0 = time.perf_counter()
    added = 0
    step = 5000
    while added < n and time.perf_counter() - t0 < build_budget_s:
        hi = min(n, added + step)
        col.add(ids=[str(i) for i in range(added, hi)], embeddings=rows_f32[added:hi].tolist())
        added = hi
    build_s = time.perf_counter() - t0
    kc = 2 * top_k
    nq = len(queries_f32)
    col.query(query_embeddings=queries_f32[:1].tolist(), n_results=min(kc, added), include=["distances"])  # warm
    t0 = time.perf_counter()
    res = col.query(query_embeddings=queries_f32.tolist(), n_results=min(kc, added), include=["distances"])
    dt = time.perf_counter() - t0
    # recall against the exact answer restricted to the rows that were added
    hits = total = 0
    for b in range(nq):
        want = [int(i) for i in exact_ids[b] if 0 <= int(i) < added][:kc]
        got = {int(s) for s in res["ids"][b]}
        hits += sum(1 for i in want if i in got)
        total += len(want)
    return {"rows_indexed": int(added), "build_s": round(build_s, 2), "queries": int(nq), "queries_per_s": round(nq / dt, 3),
            "recall_at_2k_vs_exact": round(hits / max(total, 1), 4),
            "note": "HNSW defaults (M=16, ef_construction=100, ef_search=100), squared L2, embeddings passed in"}
