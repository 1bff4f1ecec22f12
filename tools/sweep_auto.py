"""Whole-call throughput of the AUTO dispatch over (dtype, row length, batch size): one line per point with the
HBM- and MFMA-equivalent rates of the whole step, to spot a regime that sits far below its neighbours
(round 2: 1536-element rows with one query ran at 4.0 TB/s while 768 ran at 6.9 -- serialised loads).

    python tools/sweep_auto.py [bytes of corpus, default 4e9]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import torch
import coderag_hip as ch

dev = torch.device("cuda:0")
corpus_bytes = float(sys.argv[1]) if len(sys.argv) > 1 else 4e9
g = torch.Generator(device=dev).manual_seed(1)
for name, dims in (("f32", (128, 384, 768)),) if os.environ.get("SWEEP_F32") else (("f32", (384, 512, 768, 1024)), ("bf16", (1536, 2048))) if os.environ.get("SWEEP_LONG") else (("bf16", (256, 384, 512, 768, 1024, 1536, 2048)), ("f32", (128, 192, 256, 384, 512, 768, 1024))):
    tdt = torch.bfloat16 if name == "bf16" else torch.float32
    esz = 2 if name == "bf16" else 4
    for d in dims:
        n = int(corpus_bytes / (d * esz)) // 1000 * 1000
        rows = torch.empty((n, d), dtype=tdt, device=dev)
        step = 500_000
        for lo in range(0, n, step):
            m = min(step, n - lo)
            x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
            rows[lo:lo + m] = (x / x.norm(dim=1, keepdim=True)).to(tdt)
        q = torch.randn((512, d), generator=g, device=dev, dtype=torch.float32)
        q = (q / q.norm(dim=1, keepdim=True)).to(tdt)
        col = ch.GpuCollection(d, name, "l2", device=0)
        col.attach_device(rows)
        col.finalize()
        line = []
        for B in ((32, 64, 96, 128, 130, 160, 192, 200, 256, 260, 300, 384) if os.environ.get("SWEEP_F32") else (1, 2, 4, 8, 16, 32, 64, 128, 256)):
            qb = q[:B].contiguous()
            for _ in range(3):
                res = col.search_device(qb, 10, 2.0)
            torch.cuda.synchronize()
            reps = 10 if B <= 64 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                col.search_device(qb, 10, 2.0, out=res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            line.append(f"B{B}: {dt * 1e3:6.2f}ms {n * d * esz / dt / 1e12:4.2f}TB/s {2.0 * B * n * d / dt / 1e12:6.1f}TF")
        st = col.stats()
        print(f"{name} d={d:4d} n={n:9d} | " + " | ".join(line) + f" | fallbacks {st['fallback']}", flush=True)
        col.close()
        del rows, col
        torch.cuda.empty_cache()
