import os, sys, time
sys.path.insert(0, "/root/repo/code-rag_amd")
import torch
import coderag_hip as ch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for d in (128, 192, 256, 384, 768):
    n = int(4e9 / (d * 4)) // 1000 * 1000
    rows = torch.empty((n, d), dtype=torch.float32, device=dev)
    for lo in range(0, n, 500_000):
        m = min(500_000, n - lo)
        x = torch.randn((m, d), generator=g, device=dev)
        rows[lo:lo + m] = x / x.norm(dim=1, keepdim=True)
    q = torch.randn((64, d), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    col = ch.GpuCollection(d, "f32", "l2", device=0)
    col.attach_device(rows); col.finalize()
    for B in (1, 2, 3, 4, 6, 8, 12, 16):
        line = []
        for path in ("gemv", "tiled"):
            col.set_path(path)
            qb = q[:B].contiguous()
            for _ in range(2):
                res = col.search_device(qb, 10, 2.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                col.search_device(qb, 10, 2.0, out=res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            line.append(f"{path} {dt*1e3:6.3f} ms")
        print(f"f32 d={d} B={B}: " + " | ".join(line), flush=True)
    col.close(); del rows, col; torch.cuda.empty_cache()
