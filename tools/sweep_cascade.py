#!/usr/bin/env python3
"""Whole-step rate of the bench corpus (10M x 768 bf16) for one (batch, top_k) list: run once per value of a tunable
that the library reads from the environment at its first search (CRAG_CASCADE_R10 / CRAG_CASCADE_R100 / CRAG_AHEAD_MAX
/ CRAG_PACE_EVERY), all values inside one gpurun call = same box.   python tools/sweep_cascade.py 256:10,1024:100"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import coderag_hip as ch  # noqa: E402

cases = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[1] if len(sys.argv) > 1 else "256:10,1024:100").split(",")]
n, d = int(os.environ.get("SWEEP_ROWS", "10000000")), 768
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1234)
rows = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
for lo in range(0, n, 1_000_000):
    m = min(1_000_000, n - lo)
    x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
    rows[lo:lo + m] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
del x
q = torch.randn((2048, d), generator=torch.Generator(device=dev).manual_seed(1235), device=dev, dtype=torch.float32)
q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
col = ch.GpuCollection(d, "bf16", "l2")
col.attach_device(rows)
col.finalize()
out = {}
for B, top_k in cases:
    qb = q[:B].contiguous()
    res = col.search_device(qb, top_k, 2.0)
    best = []
    steps = 8 if B >= 1024 else 20
    for r in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            col.search_device(qb, top_k, 2.0, out=res)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps * 1e3)
    best.sort()
    out[f"{B}:{top_k}"] = {"ms_med": round(best[2], 4), "ms_min": round(best[0], 4), "qps": round(B / best[2] * 1e3, 1)}
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("CRAG_")}, "fallback": col.stats()["fallback"], **out}), flush=True)
