"""End-to-end latency of HybridSearcher.search() -- the reference's own call -- with a constant-time embedder,
on a collection of the size the reference indexes, split into its host parts (cProfile) and the library call.

    python tools/latency_searcher.py [rows]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import numpy as np
import coderag_hip as ch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
d = 384
rng = np.random.default_rng(1)
x = rng.standard_normal((n, d), dtype=np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
qv = rng.standard_normal((64, d), dtype=np.float32)
qv /= np.linalg.norm(qv, axis=1, keepdims=True)
texts = [f"where is function sym{i % 50} defined and how does it call helper{i}" for i in range(64)]
table = {t: qv[i:i + 1] for i, t in enumerate(texts)}
hs = ch.HybridSearcher(db_path="/nonexistent", embed_fn=lambda ts: np.concatenate([table[t] for t in ts]))
col = hs.create_collection("code_chunks", d)
metas = [{"path": f"pkg/src/m{i % 97}.js" if i % 3 else f"pkg/test/m{i}.js", "name": f"sym{i % 400}", "kind": "function",
          "start_line": i, "end_line": i + 5} for i in range(n)]
col.add([f"id{i:08d}" for i in range(n)], x, metas, documents=[f"function sym{i % 400}() {{}}\n// body {i}" for i in range(n)])
for t in texts[:8]:
    hs.search(t)
N = 400
t0 = time.perf_counter()
for i in range(N):
    hs.search(texts[i % 64])
dt = (time.perf_counter() - t0) / N
print(f"HybridSearcher.search(), {n} rows x {d} float32, top-10: {dt * 1e6:.1f} us per call", flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    hs.search(texts[i % 64])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
