"""Batch-1 latency profile driver (run under rocprofv3 --kernel-trace): configs[1] shape
(1M x 384 fp32) and the C3 corpus shape at 2M rows (bf16 x 768), 30 single-query searches each."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import torch
import coderag_hip as ch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def unit(nr, dd, dtype):
    x = torch.randn((nr, dd), generator=g, device=dev, dtype=torch.float32)
    return (x / x.norm(dim=1, keepdim=True)).to(dtype).contiguous()


for (n, d, dt, name) in ((1_000_000, 384, torch.float32, "f32"), (2_000_000, 768, torch.bfloat16, "bf16")):
    col = ch.GpuCollection(d, name, "l2", device=0)
    rows = unit(n, d, dt)
    col.attach_device(rows)
    col.finalize()
    q = unit(64, d, dt)
    qs = [q[i:i + 1] for i in range(64)]
    for i in range(3):
        res = col.search_device(qs[i], 10, 2.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        col.search_device(qs[i], 10, 2.0, out=res)
    torch.cuda.synchronize()
    print(f"{n} x {d} {name}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per batch-1 search", flush=True)
    col.close()
    del rows
