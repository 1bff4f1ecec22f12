"""Whole-job queries/s of one collection served by several host threads at once (each with its own stream and
result tensors), the reference's threadpool model (api/main.py:166-167):

    python tools/throughput_threads.py [rows] [dim] [batch] [top_k]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import torch
import coderag_hip as ch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
top_k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
rows = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
for lo in range(0, n, 1_000_000):
    m = min(1_000_000, n - lo)
    x = torch.randn((m, d), generator=g, device=dev)
    rows[lo:lo + m] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
q = torch.randn((16, B, d), generator=g, device=dev)
q = (q / q.norm(dim=2, keepdim=True)).to(torch.bfloat16).contiguous()
col = ch.GpuCollection(d, "bf16", "l2", device=0)
col.attach_device(rows)
col.finalize()
col.search_device(q[0], top_k, 2.0)


def run(n_thr, per_thr):
    def client(k, barrier):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            mine = col.search_device(q[k % 16], top_k, 2.0)
            s.synchronize()
            barrier.wait()
            for i in range(per_thr):
                col.search_device(q[(k + i) % 16], top_k, 2.0, out=mine)
            s.synchronize()
    bar = threading.Barrier(n_thr + 1)
    thr = [threading.Thread(target=client, args=(k, bar)) for k in range(n_thr)]
    for t in thr:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in thr:
        t.join()
    return n_thr * per_thr * B / (time.perf_counter() - t0)


for n_thr in (1, 2, 3, 4):
    print(f"{n} x {d} bf16, batch {B}, top-{top_k}, {n_thr} client threads: {run(n_thr, 40):10.1f} q/s", flush=True)
