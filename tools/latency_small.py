"""Batch-1 latency on collections of the size the reference itself indexes (10^2 - 10^5 chunks, 384-d float32):
device-buffer entry point, host-buffer entry point, and HybridSearcher.search() with a constant embedding.

    python tools/latency_small.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import numpy as np
import torch
import coderag_hip as ch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
d = 384
for n in (300, 3000, 10_000, 30_000, 100_000, 300_000):
    x = torch.randn((n, d), generator=g, device=dev)
    rows = (x / x.norm(dim=1, keepdim=True)).contiguous()
    q = torch.randn((64, d), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    col = ch.GpuCollection(d, "f32", "l2", device=0)
    col.attach_device(rows)
    col.finalize()
    qs = [q[i:i + 1] for i in range(64)]
    qh = [t.cpu().numpy() for t in qs]
    res = col.search_device(qs[0], 10, 2.0)
    N = 300

    def loop(fn):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / N * 1e6

    t_dev = loop(lambda i: col.search_device(qs[i % 64], 10, 2.0, out=res))
    t_host = loop(lambda i: col.search(qh[i % 64], 10, 2.0))
    import ctypes as C
    lib = ch.load()
    st = torch.cuda.current_stream(0).cuda_stream
    args = [(col.handle, C.c_void_p(qs[i].data_ptr()), 1, 10, C.c_double(2.0), None, None, C.c_void_p(res[0].data_ptr()),
             C.c_void_p(res[1].data_ptr()), C.c_void_p(res[2].data_ptr()), C.c_void_p(res[3].data_ptr()), C.c_void_p(st)) for i in range(64)]
    t_na = loop(lambda i: lib.crag_search_device_noalloc(*args[i % 64]))
    print(f"n={n:7d}: device buffers {t_dev:7.1f} us   host buffers {t_host:7.1f} us   no-alloc entry point {t_na:7.1f} us", flush=True)
    col.close()
