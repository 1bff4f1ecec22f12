"""Which scan serves small bf16 batches best: GEMV (B <= 8 per group) or the one-block MFMA kernel?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import torch
import coderag_hip as ch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
n, d = 4_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 768
name = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tdt = torch.bfloat16 if name == "bf16" else torch.float32
esz = 2 if name == "bf16" else 4
rows = torch.empty((n, d), dtype=tdt, device=dev)
for lo in range(0, n, 1_000_000):
    x = torch.randn((1_000_000, d), generator=g, device=dev, dtype=torch.float32)
    rows[lo:lo + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(tdt)
q = torch.randn((64, d), generator=g, device=dev, dtype=torch.float32)
q = (q / q.norm(dim=1, keepdim=True)).to(tdt)
col = ch.GpuCollection(d, name, "l2", device=0)
col.attach_device(rows)
col.finalize()
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    for path in ("gemv", "mfma"):
        col.set_path(path)
        qb = q[:B].contiguous()
        for _ in range(3):
            res = col.search_device(qb, 10, 2.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            col.search_device(qb, 10, 2.0, out=res)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"{name} d={d} B={B:3d} {path}: {dt * 1e3:.3f} ms/batch  {B / dt:9.1f} q/s  {n * d * esz / dt / 1e12:.2f} TB/s-equivalent", flush=True)
