"""bf16 rows: forced "mfma" (query-stationary scans) against forced "tiled" (k_scan_qlong).  The tiled scan only takes
bf16 rows of 1536 / 2048 elements in the library as committed; the comparison on shorter rows was run with an experiment
build (crag_qlong_supported() true for every bf16 row length of a multiple of 128 elements, and the forced "tiled" path
switching the query-stationary scans off).  Result in DESIGN.md section 5: the query-stationary scans win everywhere."""
import os, sys, time
sys.path.insert(0, "/root/repo/code-rag_amd")
import torch
import coderag_hip as ch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for d in (256, 384, 512, 768, 1024):
    n = int(6e9 / (d * 2)) // 1000 * 1000
    rows = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
    for lo in range(0, n, 500_000):
        m = min(500_000, n - lo)
        x = torch.randn((m, d), generator=g, device=dev)
        rows[lo:lo + m] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    q = torch.randn((1024, d), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
    col = ch.GpuCollection(d, "bf16", "l2", device=0)
    col.attach_device(rows); col.finalize()
    for B in (8, 32, 64, 128, 256, 1024):
        line = []
        for path in ("mfma", "tiled"):
            col.set_path(path)
            qb = q[:B].contiguous()
            for _ in range(2):
                res = col.search_device(qb, 10, 2.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                col.search_device(qb, 10, 2.0, out=res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            line.append(f"{path} {dt*1e3:7.3f} ms {n*d*2/dt/1e12:5.2f} TB/s {2.0*B*n*d/dt/1e12:7.1f} TF")
        print(f"bf16 d={d} n={n} B={B}: " + " | ".join(line), flush=True)
    col.close(); del rows, col; torch.cuda.empty_cache()
