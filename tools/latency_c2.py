"""Where a batch-1 search over configs[1] (1M x 384 fp32) spends its time outside the kernels: the Python
wrapper against the bare C call, event profiling on and off."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import torch
import coderag_hip as ch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((1_000_000, 384), generator=g, device=dev)
rows = (x / x.norm(dim=1, keepdim=True)).contiguous()
q = torch.randn((64, 384), generator=g, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
col = ch.GpuCollection(384, "f32", "l2", device=0)
col.attach_device(rows)
col.finalize()
qs = [q[i:i + 1] for i in range(64)]
res = col.search_device(qs[0], 10, 2.0)
N = 200


def loop(fn):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


if len(sys.argv) > 1 and sys.argv[1] == "quick":  # under rocprofv3 --kernel-trace: a few searches, then out
    col.set_profiling(False)
    for i in range(20):
        col.search_device(qs[i], 10, 2.0, out=res)
    torch.cuda.synchronize()
    sys.exit(0)
for prof in (True, False):
    col.set_profiling(prof)
    print(f"python wrapper, profiling={prof}: {loop(lambda i: col.search_device(qs[i % 64], 10, 2.0, out=res)):.1f} us", flush=True)
lib = ch.load()
st = torch.cuda.current_stream(0).cuda_stream
args = [(col._h, qs[i].data_ptr(), 1, 10, C.c_double(2.0), None, None, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(),
         res[3].data_ptr(), st) for i in range(64)]
print(f"bare C call: {loop(lambda i: lib.crag_search_device(*args[i % 64])):.1f} us", flush=True)

# the reference's serving model: several threadpool threads on the one handle (api/main.py:166-167)
import threading


def threaded(n_thr, per_thr=200):
    def client(k, barrier):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            mine = col.search_device(qs[k], 10, 2.0)
            s.synchronize()
            barrier.wait()
            for i in range(per_thr):
                col.search_device(qs[(k * 16 + i) % 64], 10, 2.0, out=mine)
            s.synchronize()
    bar = threading.Barrier(n_thr + 1)
    thr = [threading.Thread(target=client, args=(k, bar)) for k in range(n_thr)]
    for t in thr:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in thr:
        t.join()
    return n_thr * per_thr / (time.perf_counter() - t0)


col.set_profiling(False)
for n_thr in (1, 2, 3, 4, 6, 8, 16):
    print(f"{n_thr:2d} client threads: {threaded(n_thr):8.1f} q/s", flush=True)
