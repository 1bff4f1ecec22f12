#!/usr/bin/env python3
"""Same-process, interleaved A/B of the two bf16 matrix shapes of the query-stationary scan (cdna_hip_programming.md
section 5.4 rules 24 and 28): k_scan_qreg2 (v_mfma_f32_32x32x16_bf16) against k_scan_q16 (v_mfma_f32_16x16x32_bf16)
on the bench corpus (10M x 768 bf16 unit-norm random rows), whole steps and the last cascade level by HIP events.

    python tools/ab_shapes.py [rows] [rounds]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import coderag_hip as ch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = int(os.environ.get("AB_DIM", "768"))
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
cases = [(1024, 100), (1024, 10), (256, 10), (64, 10), (8, 10), (2048, 10)]
if os.environ.get("AB_CASES"):
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["AB_CASES"].split(",")]
out = []
for B, top_k in cases:
    qb = q[:B].contiguous()
    res = {}
    for shape in ("mfma32", "mfma16"):
        col.set_path(shape)
        res[shape] = col.search_device(qb, top_k, 2.0)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(res["mfma32"], res["mfma16"]))
    t = {"mfma32": [], "mfma16": []}
    k = {"mfma32": [], "mfma16": []}
    steps = 8 if B >= 1024 else 20
    for r in range(rounds):
        for shape in ("mfma32", "mfma16") if r % 2 == 0 else ("mfma16", "mfma32"):
            col.set_path(shape)
            col.search_device(qb, top_k, 2.0, out=res[shape])
            col.profile(reset=True)
            col.set_profiling(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                col.search_device(qb, top_k, 2.0, out=res[shape])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            col.set_profiling(False)
            pr = col.profile(reset=True)
            t[shape].append(dt * 1e3)
            k[shape].append(pr["filter_scan_ms"] / max(pr["searches"], 1))
            rows_last = pr["filter_scan_rows"] / max(pr["searches"], 1)
    med = lambda v: sorted(v)[len(v) // 2]
    line = {"B": B, "top_k": top_k, "rows": n, "dim": d, "results_equal": same, "last_level_rows": int(rows_last)}
    for shape in ("mfma32", "mfma16"):
        ms, kms = med(t[shape]), med(k[shape])
        line[shape] = {"step_ms_med": round(ms, 4), "step_ms_min": round(min(t[shape]), 4), "qps": round(B / ms * 1e3, 1),
                       "last_level_ms_med": round(kms, 4), "last_level_ms_min": round(min(k[shape]), 4),
                       "last_level_TF": round(2.0 * B * rows_last * d / kms / 1e9, 1),
                       "last_level_GBps": round(rows_last * (2 * d + 4) / kms / 1e6, 1)}
    line["step_speedup_16_over_32"] = round(med(t["mfma32"]) / med(t["mfma16"]), 4)
    line["last_level_speedup_16_over_32"] = round(med(k["mfma32"]) / med(k["mfma16"]), 4)
    print(json.dumps(line), flush=True)
    out.append(line)
print("fallbacks:", col.stats()["fallback"])
