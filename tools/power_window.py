"""Is the batch-256 step bound by a power budget that integrates over milliseconds?  (diagnostic, GPU box)

Runs the bench's default workload (10M x 768 bf16, batch 256, top-10) back to back and with an idle gap of G microseconds
between the steps, and prints the HIP-event duration of the last cascade level and of the whole search for each G.  If
the last level gets faster as the gap grows, time saved in the low-power parts of a step (launch prologues, selections)
is paid back by a lower clock in the high-power part -- which is what same-box A/Bs of such savings showed.

    python tools/power_window.py [gap_us ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
    sys.path.insert(0, p)
import torch
import coderag_hip as ch

gaps = [int(a) for a in sys.argv[1:]] or [0, 200, 500, 1000, 3000, 0]
n, d, B = 10_000_000, 768, 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1234)
rows = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
for lo in range(0, n, 1_000_000):
    x = torch.randn((1_000_000, d), generator=g, device=dev, dtype=torch.float32)
    rows[lo:lo + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
q = torch.randn((B, d), generator=g, device=dev, dtype=torch.float32)
q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
col = ch.GpuCollection(d, "bf16", "l2", device=0)
col.attach_device(rows, None, None)
col.finalize()
for _ in range(5):
    col.search_device(q, 10, 2.0)
for gap in gaps:
    col.profile(reset=True)
    col.set_profiling(True)
    torch.cuda.synchronize()
    t_busy = 0.0
    for _ in range(40):
        t0 = time.perf_counter()
        col.search_device(q, 10, 2.0)
        torch.cuda.synchronize()
        t_busy += time.perf_counter() - t0
        if gap:
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < gap * 1e-6:
                pass
    col.set_profiling(False)
    pr = col.profile(reset=True)
    print(f"gap {gap:5d} us: last level {pr['filter_scan_ms'] / pr['searches']:.4f} ms   search (host clock) {t_busy / 40 * 1e3:.4f} ms", flush=True)
