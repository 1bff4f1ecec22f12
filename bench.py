#!/usr/bin/env python3
"""bench.py -- queries/sec of the retrieval scoring path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N child ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric "queries/sec @ top-10, 10M x 768 bf16 corpus, 1-8 GPUs"):
every GPU holds a 10M x 768 bf16 shard of unit-normalised synthetic rows (the default, weak scaling:
N GPUs = N x 10M rows; `--scaling strong` splits configs[3]'s fixed 100M rows over the N GPUs and
reports merged answers per second), the query batch is 256 (configs[3]) at top-10.  One step = one batch through the whole hot path: sample scan, tau
select, filtered scan with the fused top-k filter, exact re-score + certificate, RCCL
all-gather of the per-shard candidates (N > 1), merge + symbol/path rerank.  Inputs (corpus,
queries, metadata columns, symbol hashes) are resident in HBM before the timed region.

One JSON line on rank 0.  `roofline` is for the dominant kernel (the filter-pass scan), timed
with HIP events on its own stream inside the library (crag_set_profiling).  `cpu_baseline` is
the oracle's C port timed on the host cores on a bounded sample (rank 0, N=1 only).
`regimes` adds the two other regimes north_star names (batch-1 GEMV, batch-1024 top-100 MFMA),
measured on the same corpus after the timed region; they are not part of `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def pmc_entry(n, d, B):
    """The committed PMC pass of this same command for the workload rows|dim|batch (rocprofv3 cannot wrap
    itself; profiles/pmc_traffic.json says how each was taken), else {}."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(f"{n}|{d}|{B}", {})
    except OSError:
        return {}


def pmc_traffic(n, d, B):
    """HBM bytes per launch of the dominant kernel from that pass, else None."""
    return pmc_entry(n, d, B).get("bytes")


def scan_kernel_name(d, B, path, n=10_000_000, top_k=10):
    """The scan the library's auto dispatch takes for bf16 rows of d elements, n rows and B queries of one collection
    (crag_capi.hip: search_entries_chunk_launch)."""
    if path == "gemv":
        return "k_scan_gemv"
    if d > 1024:
        return "k_scan_qlong" if (B >= 9 or path == "mfma") else "k_scan_gemv"
    if d > 768:
        return "k_scan_qsplit"
    m = 2 * top_k + max(16, 2 * top_k // 8)
    first = (1 if n >= 4_000_000 else 2) if d >= 768 else (3 if d >= 512 or m > 64 else 5)
    return "k_scan_q16" if (B >= first or path == "mfma") else "k_scan_gemv"


def kernel_sources_sha():
    """sha256 over the scan/select kernel sources the PMC passes were taken on (profiles/pmc_traffic.json records
    it per entry): a committed traffic figure is only reported while the kernels it measured are the ones built."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "code-rag_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".inc", ".h")):
            with open(os.path.join(src, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic_checked(n, d, B):
    """(bytes or None, provenance string): None with the reason when there is no entry or the kernels changed."""
    e = pmc_entry(n, d, B)
    if not e.get("bytes"):
        return None, "no committed rocprofv3 --pmc FETCH_SIZE pass for this workload (rocprofv3 cannot wrap itself)"
    now = kernel_sources_sha()
    if e.get("kernel_sources_sha") != now:
        return None, (f"stale: {e.get('source')} was taken on kernel sources {e.get('kernel_sources_sha')}, "
                      f"the built library is {now}; re-run tools/profile_regimes.sh")
    return e["bytes"], f"{e.get('source')} (kernel sources {now}, git {e.get('git', '?')})"


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N child processes (one rank per GPU) with the
    torch.distributed environment and wait for them.  This parent never touches the GPU (it imports neither
    torch nor the library), so no process that initialised HIP is ever replaced or forked."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad:  # a rank died: the others would wait in a collective for ever
                rc = bad[0]
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --rows-per-gpu rows on every GPU; strong: --total-rows (configs[3]: 100M) split over the GPUs")
    ap.add_argument("--rows-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--total-rows", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline and the extra regimes")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--path", choices=("auto", "gemv", "mfma"), default="auto", help="force a scan kernel family (A/B runs)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (gloo, host stand-ins for the two GPU stages): no GPU work, no valid number")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    from coderag_hip.dist import ShardedSearcher, shard_range

    if args.dry_run:
        return dry_run(args, world, rank, np, torch, dist, ShardedSearcher, shard_range)
    import coderag_hip as ch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    d, B, top_k = args.dim, args.batch, args.top_k
    strong = args.scaling == "strong"
    if strong:
        lo, hi = shard_range(args.total_rows, rank, world)
        n, id_base, total_rows = hi - lo, lo, args.total_rows
    else:
        n, id_base, total_rows = args.rows_per_gpu, rank * args.rows_per_gpu, world * args.rows_per_gpu

    # ---- synthetic shard, generated on the device in chunks -----------
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    rows = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
    step_rows = 1_000_000
    for lo_ in range(0, n, step_rows):
        m = min(step_rows, n - lo_)
        x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
        rows[lo_:lo_ + m] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    del x
    gq = torch.Generator(device=dev).manual_seed(1235)  # same queries on every rank
    q = torch.randn((max(B, 1024), d), generator=gq, device=dev, dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    # metadata columns (configs[4] mix): path class {70% neutral, 25% boosted, 5% penalised},
    # names Zipf-drawn from a 50k-symbol dictionary; each query mentions 2 dictionary symbols
    n_sym = 50_000
    sym_hash, sym_len = ch.hash_names([f"sym{i}" for i in range(n_sym)])
    sym_hash_t = torch.from_numpy(sym_hash.view(np.int64)).to(dev)
    gz = torch.Generator(device=dev).manual_seed(99 + rank)
    name_hash = torch.empty((n,), dtype=torch.int64, device=dev)
    path_class = torch.empty((n,), dtype=torch.uint8, device=dev)
    for lo_ in range(0, n, 10_000_000):  # in chunks: the temporaries of 100M rows at once would be GBs
        m = min(10_000_000, n - lo_)
        zipf = (torch.rand((m,), generator=gz, device=dev).pow(3.0) * n_sym).long().clamp_(max=n_sym - 1)
        name_hash[lo_:lo_ + m] = sym_hash_t[zipf]
        u = torch.rand((m,), generator=gz, device=dev)
        path_class[lo_:lo_ + m] = torch.where(u < 0.70, 0, torch.where(u < 0.95, 2, 1)).to(torch.uint8)
    del zipf, u
    rng = np.random.default_rng(7)
    qsyms = np.sort(sym_hash[rng.integers(0, n_sym, size=(q.shape[0], 2))], axis=1)
    q_hashes = torch.from_numpy(qsyms.reshape(-1).view(np.int64).copy()).to(dev)
    q_hash_off = torch.arange(0, 2 * q.shape[0] + 1, 2, dtype=torch.int32, device=dev)

    col = ch.GpuCollection(d, "bf16", "l2", device=local_rank)
    col.attach_device(rows, name_hash, path_class)
    col.set_symbols(sym_hash, sym_len)
    col.set_id_base(id_base)
    col.finalize()
    col.set_path(args.path)
    sharded = ShardedSearcher(col, rank, world) if world > 1 else None
    qb = q[:B].contiguous()

    def step():
        if sharded is not None:
            return sharded.search_device(qb, top_k, 2.0, q_hashes, q_hash_off)
        return col.search_device(qb, top_k, 2.0, q_hashes, q_hash_off)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    col.stats(reset=True)
    col.profile(reset=True)
    col.set_profiling(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    dt = time.perf_counter() - t0
    col.set_profiling(False)
    prof = col.profile(reset=True)
    st = col.stats(reset=True)
    multi = None
    if world > 1:
        # what the first hardware run of the scaling tier needs to be diagnosable: every rank's own step time (min / max
        # over ranks), the world size the process group really has, and the exchange step alone (a few extra steps
        # with device waits around the all-gather, outside the timed region)
        own = torch.tensor([dt], dtype=torch.float64, device=dev)
        tmin, tmax = own.clone(), own.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        fb = torch.tensor([st["fallback"]], dtype=torch.int64, device=dev)
        dist.all_reduce(fb)
        st["fallback"] = int(fb.item())
        sharded.timing = {}
        for _ in range(5):
            step()
        ag = torch.tensor([sharded.timing["allgather_s"] / max(sharded.timing["calls"], 1)], dtype=torch.float64, device=dev)
        sharded.timing = None
        dist.all_reduce(ag, op=dist.ReduceOp.MAX)
        multi = {"world_size_seen": dist.get_world_size(), "rank_step_ms_min": round(float(tmin.item()) / args.steps * 1e3, 4),
                 "rank_step_ms_max": round(float(tmax.item()) / args.steps * 1e3, 4),
                 "allgather_us_max_over_ranks": round(float(ag.item()) * 1e6, 1),
                 "allgather_bytes_per_rank": B * 2 * top_k * 16, "backend": dist.get_backend()}
    ms_per_step = dt / args.steps * 1e3
    merged_qps = B * args.steps / dt      # answers per second over the whole (sharded) corpus
    # weak scaling: the unit of work is one query scored against one --rows-per-gpu shard (rows shard,
    # queries replicate; every rank scans ITS rows for all B queries), so a step processes world x B units.
    # strong scaling: the corpus is fixed (configs[3]), the unit is one merged answer.
    qps = merged_qps if strong else world * merged_qps

    # ---- roofline of the dominant kernel (filter-pass scan = last cascade level), this rank's shard -----------
    scan_ms = prof["filter_scan_ms"] / max(prof["searches"], 1)
    scan_rows = prof["filter_scan_rows"] / max(prof["searches"], 1)   # rows behind the earlier levels
    row_bytes = d * 2 + 4                   # the row once + its |x|^2
    alg_bytes = int(scan_rows * row_bytes)
    flops = 2.0 * B * scan_rows * d
    achieved_gbs = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    hbm_bound = B < 312                     # 2*B/2 flop per byte vs the ~312 flop/B ridge
    traffic, traffic_src = (pmc_traffic_checked(n, d, B) if top_k == (100 if B == 1024 else 10) and (B != 1 or args.path == "gemv") else
                            (None, "the committed passes are top-10 runs (top-100 at batch 1024; batch 1 with --path gemv)"))
    step_s = ms_per_step * 1e-3
    prof_ms = pmc_entry(n, d, B).get("kernel_ms_kernel_trace") if traffic else None  # only while the sources match
    roofline = {
        "kernel": scan_kernel_name(d, B, args.path, n, top_k) + " (last cascade level: every row behind the earlier levels)",
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": round(achieved_gbs if hbm_bound else flops / (scan_ms * 1e-3) / 1e12, 2),
        "peak": HBM_PEAK_GBS if hbm_bound else MFMA_BF16_PEAK_TF,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": round((achieved_gbs / HBM_PEAK_GBS) if hbm_bound else (flops / (scan_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF), 4),
        "traffic": traffic,
        "traffic_source": traffic_src,
        "kernel_ms": round(scan_ms, 4),
        # the same launch under rocprofv3 --kernel-trace (committed profile set; a profiled launch runs a few per cent
        # longer) and the fraction that follows from it: frac above is from this run's own HIP events
        "kernel_ms_profiled": prof_ms,
        "frac_profiled": (round((alg_bytes / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hbm_bound else
                                (flops / (prof_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF), 4) if prof_ms else None),
        "algorithmic_bytes": alg_bytes,
        "rows_scanned": int(scan_rows),
        "mfma_tflops": round(flops / (scan_ms * 1e-3) / 1e12, 2) if scan_ms > 0 else 0.0,
        "mfma_frac": round(flops / (scan_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4) if scan_ms > 0 else 0.0,
        # the WHOLE step on this rank's shard: every level, every selection, finalize, (all-gather,) merge
        "step": {"algorithmic_bytes": int(n * row_bytes), "flops": 2.0 * B * n * d, "ms": round(ms_per_step, 4),
                 "hbm_GBps": round(n * row_bytes / step_s / 1e9, 1), "hbm_frac": round(n * row_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4),
                 "mfma_TFLOPs": round(2.0 * B * n * d / step_s / 1e12, 2),
                 "mfma_frac": round(2.0 * B * n * d / step_s / 1e12 / MFMA_BF16_PEAK_TF, 4)},
    }

    shard_txt = (f"{total_rows} x {d} bf16 unit-norm rows split over {world} GPU(s) ({n} on rank 0)" if strong else
                 f"{n} x {d} bf16 unit-norm rows per GPU ({total_rows} in total)")
    line = {
        "metric": ("queries/sec @ top-10, 100M x 768 bf16 corpus row-sharded over the GPUs (configs[3])" if strong else
                   "queries/sec @ top-10, 10M x 768 bf16 corpus"),
        "value": round(qps, 2),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"{shard_txt}, batch-{B} queries, top-{top_k} (2*top_k={2 * top_k} candidates), squared-L2, "
                               "symbolBoost=2.0 rerank" + (", one RCCL all-gather of candidates per batch" if world > 1 else ", single MI355X"),
                   "rows_per_gpu": n, "total_rows": total_rows, "dim": d, "batch": B, "top_k": top_k, "parallelism": f"rows/{world}"},
        "roofline": roofline,
        "exact_fallback_queries": st["fallback"],
        "merged_queries_per_s": round(merged_qps, 2),
    }
    if multi:
        line["multi_gpu"] = multi
    line["config"]["unit"] = (
        f"one merged answer over the whole {total_rows}-row corpus; value = batch x steps / time" if strong else
        f"one query scored against one {n}-row shard (each rank scans its own rows for all {B} queries: weak scaling); "
        f"value = n_gpus x batch x steps / time; merged answers over the whole {total_rows}-row corpus: {merged_qps:.2f} queries/s")

    if rank == 0 and world == 1 and not args.no_extras:
        line["regimes"] = [{"regime": "configs[0]: lodash via chunker.py + embed_ingest.py into Chroma, eval/qa_min.jsonl through "
                                      "HybridSearcher top-k=8 on CPU (reference path)",
                            "skipped": "needs the lodash checkout, the MiniLM ONNX model, chromadb and tree-sitter; none is in "
                                       "/root/reference, this image or the GPU box, and there is no network (BASELINE.md 3.4)"}]
        line["regimes"] += extra_regimes(col, q, n, d, torch)
        line["cpu_baseline"] = cpu_baseline(rows, q, n, d, top_k, args.cpu_sample_rows, torch, np)
        line["cpu_baseline"]["hnsw"] = hnsw_leg(ch, rows, q, n, d, top_k, local_rank, torch, np)
        col.close()
        del col, rows, name_hash, path_class
        torch.cuda.empty_cache()
        line["regimes"] += config_regimes(ch, torch, np, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def dry_run(args, world, rank, np, torch, dist, ShardedSearcher, shard_range):
    """CPU rehearsal of `--gpus N`: the launcher, the rendezvous, ShardedSearcher and its all-gather with host
    stand-ins for the two GPU stages (synthetic sorted candidates per shard; (distance, id) merge in numpy)."""
    entry = np.dtype([("dist", "<f4"), ("flags", "<u4"), ("id", "<i8")])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus
    B, top_k, kc = args.batch, args.top_k, 2 * args.top_k
    total = args.total_rows if args.scaling == "strong" else world * args.rows_per_gpu
    lo, hi = shard_range(total, rank, world)
    rng = np.random.default_rng(100 + rank)

    def stage1(queries, top_k_, entries, qh, qo):
        e = np.zeros((B, kc), entry)
        e["dist"] = np.sort(rng.random((B, kc), dtype=np.float32), axis=1)
        e["id"] = np.sort(rng.integers(lo, hi, size=(B, kc)), axis=1)
        entries.copy_(torch.from_numpy(e.view(np.uint8).reshape(B, kc, 16)))

    def stage2(gathered, G, b, top_k_, symbol_boost):
        g = gathered.numpy().reshape(G, b, kc * 16).view(entry)
        ids = np.empty((b, top_k_), np.int64)
        for i in range(b):
            a = np.concatenate([g[r, i] for r in range(G)])
            ids[i] = a["id"][np.lexsort((a["id"], a["dist"]))[:top_k_]]
        return ids

    s = ShardedSearcher(None, rank, world, stage1=stage1, stage2=stage2)
    q = torch.zeros((B, args.dim))
    for _ in range(args.warmup):
        s.search_device(q, top_k)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = s.search_device(q, top_k)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = bool(ids.min() >= 0 and ids.max() < total)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run of the multi-rank plumbing (no GPU work: not a measurement)", "dry_run": True,
                          "value": None, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 4), "scaling": args.scaling, "merged_ids_in_range": ok,
                          "data": "synthetic"}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def extra_regimes(col, q, n, d, torch):
    """batch-1 (HBM-bound GEMV) and batch-1024 top-100 (MFMA-bound GEMM) on the same corpus."""
    out = []
    for name, B, top_k, path, steps in (("batch-1 top-10 GEMV", 1, 10, "gemv", 20), ("batch-8 top-10 GEMV", 8, 10, "gemv", 10),
                                        ("batch-8 top-10 MFMA cascade (what auto picks)", 8, 10, "mfma", 10),
                                        ("batch-1024 top-100 MFMA", 1024, 100, "mfma", 5),
                                        ("batch-1024 top-10 MFMA", 1024, 10, "mfma", 5)):
        col.set_path(path)
        qb = q[:B].contiguous()
        for _ in range(2):
            res = col.search_device(qb, top_k, 2.0)
        col.profile(reset=True)
        col.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            col.search_device(qb, top_k, 2.0, out=res)  # a serving loop reuses its result tensors
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        col.set_profiling(False)
        prof = col.profile(reset=True)
        ms = prof["filter_scan_ms"] / max(prof["searches"], 1)
        sr = prof["filter_scan_rows"] / max(prof["searches"], 1)
        gbs = sr * (d * 2 + 4) / (ms * 1e-3) / 1e9
        tf = 2.0 * B * sr * d / (ms * 1e-3) / 1e12
        out.append({"regime": name, "queries_per_s": round(B * steps / dt, 2), "ms_per_batch": round(dt / steps * 1e3, 4),
                    "scan_kernel_ms": round(ms, 4), "hbm_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    "mfma_TFLOPs": round(tf, 2), "mfma_frac": round(tf / MFMA_BF16_PEAK_TF, 4)})
        if top_k == (100 if B == 1024 else 10) and path == ("gemv" if B == 1 else "mfma"):
            byts, _src = pmc_traffic_checked(n, d, B)   # None when the committed pass measured other kernel sources
            if byts:
                out[-1]["pmc_hbm_bytes_per_launch"] = byts  # rocprofv3 FETCH_SIZE pass, gfx950 correction applied
                out[-1]["algorithmic_bytes_per_launch"] = int(sr * (d * 2 + 4))
                if pmc_entry(n, d, B).get("mfma_busy_frac"):
                    out[-1]["pmc_mfma_busy_frac"] = pmc_entry(n, d, B)["mfma_busy_frac"]  # matrix pipe busy cycles / (SIMDs x kernel cycles)
    col.set_path("auto")
    out.append(adversarial_regime(q, n, d, torch))
    return out


def adversarial_regime(q, n, d, torch):
    """The fallback under adversarial input: every vector of the corpus exists 3000 times and the 256 queries are
    corpus vectors, so each query's 2*top_k nearest are decided by row id among 3000 exact ties and no certificate of
    the first pass can hold: the WHOLE batch falls back.  (The certified re-scan settles it with one more pass over the
    corpus; only more ties than a candidate buffer holds, 16384 per query, would need the full exact scan.)"""
    import coderag_hip as ch
    dev = q.device
    nb = (n + 2999) // 3000
    g = torch.Generator(device=dev).manual_seed(77)
    base = torch.randn((nb, d), generator=g, device=dev, dtype=torch.float32)
    base = (base / base.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    rows = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
    for lo in range(0, n, 1_000_000):
        m = min(1_000_000, n - lo)
        rows[lo:lo + m] = base[(torch.arange(lo, lo + m, device=dev) % nb)]
    col = ch.GpuCollection(d, "bf16", "l2", device=dev.index)
    col.attach_device(rows)
    col.finalize()
    qa = base[:256].contiguous()
    res = col.search_device(qa, 10, 2.0)
    col.stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 2
    for _ in range(steps):
        col.search_device(qa, 10, 2.0, out=res)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = col.stats(reset=True)
    ids = res[0].cpu()
    ok = bool((ids[:, 0] == torch.arange(256)).all() and (res[2].cpu() == 0).all())  # the first copies, distance 0
    col.close()
    return {"regime": f"adversarial ties: {nb} vectors x 3000 copies = {n} x {d} bf16, batch-256 top-10, every query falls back",
            "queries_per_s": round(256 / dt, 2), "ms_per_batch": round(dt * 1e3, 2), "fallback_queries_per_batch": st["fallback"] // steps,
            "settled_by_the_certified_rescan_per_batch": st["rescan"] // steps, "exact_scan_queries_per_batch": st["exact_scan"] // steps,
            "answers_are_the_first_copies_at_distance_0": ok}


def config_regimes(ch, torch, np, dev):
    """BASELINE.json configs[1] (1M x 384 fp32, batch-1, top-10) and configs[4] (10M x 1024 bf16 in 64
    workspace-isolated collections with symbol/path metadata, one query per collection per batch)."""
    out = []
    g = torch.Generator(device=dev).manual_seed(4321)

    def unit(nr, dd, dtype):
        x = torch.randn((nr, dd), generator=g, device=dev, dtype=torch.float32)
        return (x / x.norm(dim=1, keepdim=True)).to(dtype).contiguous()

    # ---- configs[1]
    n2, d2 = 1_000_000, 384
    c2 = ch.GpuCollection(d2, "f32", "l2", device=dev.index)
    r2 = unit(n2, d2, torch.float32)
    c2.attach_device(r2)
    c2.finalize()
    q2 = unit(64, d2, torch.float32)
    qs = [q2[i:i + 1] for i in range(64)]
    for i in range(3):
        res = c2.search_device(qs[i], 10, 2.0)
    # latency loop without the event timing (four event records per search cost ~15 us of a 0.34 ms search:
    # tools/latency_c2.py), then the same loop with it for the scan kernel's duration
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        c2.search_device(qs[i], 10, 2.0, out=res)  # a serving loop reuses its result tensors
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c2.profile(reset=True)
    c2.set_profiling(True)
    for i in range(20):
        c2.search_device(qs[i], 10, 2.0, out=res)
    torch.cuda.synchronize()
    pr = c2.profile(reset=True)
    ms = pr["filter_scan_ms"] / max(pr["searches"], 1)
    gbs = pr["filter_scan_rows"] / max(pr["searches"], 1) * (d2 * 4 + 4) / (ms * 1e-3) / 1e9
    out.append({"regime": "configs[1]: 1M x 384 fp32, batch-1 top-10 GEMV", "queries_per_s": round(50 / dt, 2),
                "ms_per_batch": round(dt / 50 * 1e3, 4), "scan_kernel_ms": round(ms, 4), "hbm_GBps": round(gbs, 1),
                "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
    # the same collection served the way the reference serves it: /search is a sync route on FastAPI's
    # threadpool (api/main.py:166-167), so several host threads call one searcher at once.  Four client
    # threads, each with its own stream and result tensors; a thread's launch and wait gaps hide behind the
    # other threads' scans (crag_search* is re-entrant on one handle)
    import threading
    n_thr, per_thr = 4, 200  # 200 each: with 50 the start-up of the threads was 3 % of the timed region
    c2.set_profiling(False)

    def client(k, barrier):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            mine = c2.search_device(qs[k], 10, 2.0)
            s.synchronize()
            barrier.wait()
            for i in range(per_thr):
                c2.search_device(qs[(k * 16 + i) % 64], 10, 2.0, out=mine)
            s.synchronize()

    bar = threading.Barrier(n_thr + 1)
    thr = [threading.Thread(target=client, args=(k, bar)) for k in range(n_thr)]
    for t in thr:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in thr:
        t.join()
    dt_thr = time.perf_counter() - t0
    out[-1]["queries_per_s_with_4_client_threads"] = round(n_thr * per_thr / dt_thr, 2)
    c2.close()
    del r2
    # ---- configs[4]: 64 collections x 156,250 rows x 1024 bf16 (10M rows in total)
    ncol, nper, d5 = 64, 156_250, 1024
    n_sym = 50_000
    sym_hash, sym_len = ch.hash_names([f"sym{i}" for i in range(n_sym)])
    sym_hash_t = torch.from_numpy(sym_hash.view(np.int64)).to(dev)
    cols, keep = [], []
    for c in range(ncol):
        rr = unit(nper, d5, torch.bfloat16)
        zipf = (torch.rand((nper,), generator=g, device=dev).pow(3.0) * n_sym).long().clamp_(max=n_sym - 1)
        nh = sym_hash_t[zipf].contiguous()
        u = torch.rand((nper,), generator=g, device=dev)
        pc = torch.where(u < 0.70, 0, torch.where(u < 0.95, 2, 1)).to(torch.uint8).contiguous()
        col = ch.GpuCollection(d5, "bf16", "l2", device=dev.index)
        col.attach_device(rr, nh, pc)
        col.set_symbols(sym_hash, sym_len)
        col.finalize()
        cols.append(col)
        keep.append((rr, nh, pc))
    rng = np.random.default_rng(5)
    q5 = unit(ncol, d5, torch.float32).cpu().numpy()
    texts = [f"where is sym{int(rng.integers(0, n_sym))} used together with sym{int(rng.integers(0, n_sym))}?" for _ in range(ncol)]
    for _ in range(2):
        ch.search_multi(cols, q5, 8, 2.0, texts)
    t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        ch.search_multi(cols, q5, 8, 2.0, texts)
    dt_host = time.perf_counter() - t0
    # the same with queries and their symbol hashes resident in HBM (device-buffer entry point)
    hashes = [cols[c].symbol_hashes(texts[c]) for c in range(ncol)]
    off = np.zeros(ncol + 1, np.int32)
    np.cumsum([len(h) for h in hashes], out=off[1:])
    qh = torch.from_numpy(np.concatenate(hashes).view(np.int64)).to(dev)
    qo = torch.from_numpy(off).to(dev)
    q5d = torch.from_numpy(q5).to(dev).to(torch.bfloat16).contiguous()
    for _ in range(2):
        res = ch.search_multi_device(cols, q5d, 8, 2.0, qh, qo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ch.search_multi_device(cols, q5d, 8, 2.0, qh, qo, out=res)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    byts = ncol * nper * (d5 * 2 + 4)
    out.append({"regime": "configs[4]: 64 collections x 156250 x 1024 bf16, one query each, top-8, symbolBoost rerank "
                          "(one grouped GEMV launch; inputs resident in HBM)", "queries_per_s": round(ncol * steps / dt, 2),
                "ms_per_batch": round(dt / steps * 1e3, 4), "hbm_GBps": round(byts / (dt / steps) / 1e9, 1),
                "hbm_frac": round(byts / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
                "host_buffer_entry_point_queries_per_s": round(ncol * steps / dt_host, 2)})
    for col in cols:
        col.close()
    del cols, keep
    torch.cuda.empty_cache()
    # ---- configs[3] at one GPU: the whole 100M x 768 bf16 corpus (153.6 GB) resident on a single MI355X
    n3, d3, b3 = 100_000_000, 768, 256
    if torch.cuda.mem_get_info()[0] > 175 * 2**30:
        r3 = torch.empty((n3, d3), dtype=torch.bfloat16, device=dev)
        for lo in range(0, n3, 1_000_000):
            x = torch.randn((1_000_000, d3), generator=g, device=dev, dtype=torch.float32)
            r3[lo:lo + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
        del x
        c3 = ch.GpuCollection(d3, "bf16", "l2", device=dev.index)
        c3.attach_device(r3)
        c3.finalize()
        q3 = unit(b3, d3, torch.bfloat16)
        for _ in range(2):
            res = c3.search_device(q3, 10, 2.0)
        c3.profile(reset=True)
        c3.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 5
        for _ in range(steps):
            c3.search_device(q3, 10, 2.0, out=res)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pr = c3.profile(reset=True)
        ms = pr["filter_scan_ms"] / max(pr["searches"], 1)
        sr = pr["filter_scan_rows"] / max(pr["searches"], 1)
        gbs = sr * (d3 * 2 + 4) / (ms * 1e-3) / 1e9
        tf = 2.0 * b3 * sr * d3 / (ms * 1e-3) / 1e12
        out.append({"regime": "configs[3] at one GPU: 100M x 768 bf16 (153.6 GB resident), batch-256 top-10",
                    "queries_per_s": round(b3 * steps / dt, 2), "ms_per_batch": round(dt / steps * 1e3, 4),
                    "scan_kernel_ms": round(ms, 4), "hbm_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    "mfma_TFLOPs": round(tf, 2), "mfma_frac": round(tf / MFMA_BF16_PEAK_TF, 4)})
        c3.close()
        del r3
    return out


def cpu_baseline(rows, q, n, d, top_k, sample_rows, torch, np):
    """Two CPU legs on the host cores, each on the first `sample_rows` rows (the scan is linear in rows, so q/s for
    the full corpus is the sample rate times sample_rows/n, stated in `sample`):
      port  the oracle's C port (oracle/crag_oracle.c: oracle_knn_fast_f32, OpenMP over rows): float32 brute force +
            (dist, id) top-2k;
      blas  SURVEY.md section 8d line (1): numpy / BLAS float32 brute force -- |x|^2 - 2 Q X^T in row blocks (sgemm),
            argpartition to 2*top_k per block, merge by (dist, id) -- then the oracle's rerank (R1-R5) per query.
    `value` is the faster of the two."""
    from oracle import coracle
    coracle.build()
    m = min(sample_rows, n)
    host_rows = rows[:m].view(torch.int16).cpu().numpy().view(np.uint16)
    nq = 256
    host_q = q[:nq].view(torch.int16).cpu().numpy().view(np.uint16)
    kc = 2 * top_k
    coracle.knn_fast_f32(host_rows[:10000], host_q[:1], kc)  # warm the library
    t0 = time.perf_counter()
    coracle.knn_fast_f32(host_rows, host_q, kc)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    port = {"value": round(nq / dt * m / n, 4), "seconds": round(dt, 2), "cores": cores,
            "what": "oracle C port (float32 brute force + (dist,id) top-%d, OpenMP)" % kc}
    # ---- numpy / BLAS leg (inputs converted to float32 before the clock starts: "resident" as on the GPU side)
    blas_threads = cores
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([int(p.get("num_threads", 1)) for p in threadpool_info() if p.get("user_api") == "blas"] or [cores])
    except Exception:  # noqa: BLE001
        pass
    X = (host_rows.astype(np.uint32) << 16).view(np.float32)
    Q = (host_q.astype(np.uint32) << 16).view(np.float32)
    xn = np.einsum("ij,ij->i", X, X)
    (Q[:8] @ X[:4096].T).sum()  # warm the BLAS threads
    blk = 131072
    t0 = time.perf_counter()
    best_d = np.full((nq, 0), np.inf, np.float32)
    best_i = np.zeros((nq, 0), np.int64)
    for lo in range(0, m, blk):
        hi = min(m, lo + blk)
        S = Q @ X[lo:hi].T
        S *= -2.0
        S += xn[lo:hi][None, :]
        k = min(kc, hi - lo)
        part = np.argpartition(S, k - 1, axis=1)[:, :k]
        best_d = np.concatenate([best_d, np.take_along_axis(S, part, axis=1)], axis=1)
        best_i = np.concatenate([best_i, part + lo], axis=1)
        if best_d.shape[1] > 8 * kc:
            keep = np.argpartition(best_d, kc - 1, axis=1)[:, :kc]
            best_d, best_i = np.take_along_axis(best_d, keep, axis=1), np.take_along_axis(best_i, keep, axis=1)
    zeros_c = np.zeros(m, np.uint8)
    for b in range(nq):
        order = np.lexsort((best_i[b], best_d[b]))[:kc]
        dist = best_d[b][order] + float(np.dot(Q[b], Q[b]))  # squared L2 = |x|^2 - 2 q.x + |q|^2
        coracle.rerank(dist, zeros_c[best_i[b][order]], np.zeros(len(order), np.uint8), top_k, 2.0)
    dt_blas = time.perf_counter() - t0
    blas = {"value": round(nq / dt_blas * m / n, 4), "seconds": round(dt_blas, 2), "cores": blas_threads,
            "what": "numpy/BLAS float32: |x|^2 - 2 Q X^T in blocks of %d rows (sgemm), argpartition to %d, (dist,id) merge, "
                    "oracle rerank" % (blk, kc), "GFLOPs": round(2.0 * nq * m * d / dt_blas / 1e9, 1)}
    del X, S
    fast = blas if blas["value"] >= port["value"] else port
    import importlib.util
    chroma = ("unavailable: chromadb not installed on this box (BASELINE.md 3.3), so the reference's Chroma+HybridSearcher "
              "CPU path and recall@k against its HNSW result cannot be measured; the exact search has recall 1 by construction. "
              "cpu_baseline.hnsw holds the same measurement against a restatement of that index (oracle/hnsw_oracle.c, parity unpinned)")
    if importlib.util.find_spec("chromadb"):
        # the reference's own call sequence on a bounded slice (tools/chroma_baseline.py); never fatal to the bench
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import chroma_baseline
            sub = min(m, 200_000)
            xr = rows[:sub].float().cpu().numpy()
            xq = q[:32].float().cpu().numpy()
            d2 = (xr * xr).sum(1)[None, :] - 2.0 * (xq @ xr.T)
            exact = np.argsort(d2, axis=1, kind="stable")[:, :2 * top_k]
            chroma = chroma_baseline.run(xr, xq, top_k, exact)
        except Exception as e:  # noqa: BLE001
            chroma = f"chromadb importable, but its run failed: {type(e).__name__}: {e}"
    return {"reference_chroma_path": chroma, "value": fast["value"], "unit": "queries/s", "cores": fast["cores"], "kind": "port",
            "host_cpus_visible": os.cpu_count(), "port": port, "blas": blas,
            "sample": f"{nq} queries x first {m} of {n} rows: C port {dt:.2f} s on {cores} OpenMP threads, numpy/BLAS "
                      f"{dt_blas:.2f} s on {blas_threads} BLAS threads; value = the faster leg's sample q/s x {m}/{n}"}


def hnsw_leg(ch, rows, q, n, d, top_k, device, torch, np, sample_rows=100_000, nq=256):
    """The approximate-index CPU leg: what hybrid_search.py:60-64 runs is Chroma's HNSW index, not a brute-force scan.
    oracle/hnsw_oracle.c restates hnswlib's published algorithm with Chroma's documented defaults (l2, M = 16,
    ef_construction = 100, ef_search = 100; PARITY UNPINNED: chromadb / chroma-hnswlib are not installable here).  On the
    first `sample_rows` rows: build (all host threads), `nq` queries -> 2*top_k candidates -> the oracle's rerank (R1-R5),
    timed; recall of its candidates against the exact 2*top_k, and of its final top_k against the GPU path's answer on
    the same rows (the north_star's "recall@k against the reference's HNSW result").  An HNSW query costs ~log(rows), so
    its q/s on the sample is an UPPER bound for the full corpus.  Isotropic synthetic rows are the hardest case for a
    graph index (real embeddings cluster), so the same measurement is repeated on a clustered synthetic corpus of the
    same shape (1000 centres, rows = centre + 0.5 x unit noise, renormalised, bf16)."""
    from oracle import coracle, hnsw
    m = min(sample_rows, n)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    kc = 2 * top_k
    ef = hnsw.CHROMA_DEFAULTS["ef_search"]

    def one(rows_dev, q_dev):
        X = rows_dev.float().cpu().numpy()
        Q = q_dev.float().cpu().numpy()
        t0 = time.perf_counter()
        ix = hnsw.HnswIndex(X, threads=cores, M=hnsw.CHROMA_DEFAULTS["M"], ef_construction=hnsw.CHROMA_DEFAULTS["ef_construction"])
        t_build = time.perf_counter() - t0
        ix.query(Q[:8], kc, threads=cores)  # warm
        zeros_c = np.zeros(len(X), np.uint8)
        t0 = time.perf_counter()
        ids, dist = ix.query(Q, kc, ef_search=ef, threads=cores)
        final = []
        for b in range(len(Q)):
            ok = ids[b] >= 0
            order, _ = coracle.rerank(dist[b][ok], zeros_c[ids[b][ok]], np.zeros(int(ok.sum()), np.uint8), top_k, 2.0)
            final.append(ids[b][ok][order])
        t_query = time.perf_counter() - t0
        t0 = time.perf_counter()
        ix.query(Q[:64], kc, ef_search=ef, threads=1)
        t_one = time.perf_counter() - t0
        stats = ix.stats()
        ix.close()
        # exact candidates of the same rows (float32 BLAS) and the GPU path's final answer on them
        d2 = (X * X).sum(1)[None, :] - 2.0 * (Q @ X.T)
        exact = np.argsort(d2, axis=1, kind="stable")[:, :kc]
        rec_c = float(np.mean([len(set(ids[b].tolist()) & set(exact[b].tolist())) / kc for b in range(len(Q))]))
        sub = ch.GpuCollection(d, "bf16", "l2", device=device)
        sub.attach_device(rows_dev, None, None)
        sub.finalize()
        g_ids, _, _, g_cn = sub.search_device(q_dev.contiguous(), top_k, 2.0)
        torch.cuda.synchronize()
        g_ids, g_cn = g_ids.cpu().numpy(), g_cn.cpu().numpy()
        sub.close()
        rec_f = float(np.mean([len(set(final[b].tolist()) & set(g_ids[b, :g_cn[b]].tolist())) / max(int(g_cn[b]), 1)
                               for b in range(len(Q))]))
        return {"rows": len(X), "build_seconds": round(t_build, 2), "queries_per_s": round(len(Q) / t_query, 1),
                "queries_per_s_one_thread": round(64 / t_one, 1),
                "recall_of_its_%d_candidates_vs_exact" % kc: round(rec_c, 4),
                "recall_at_%d_of_its_answer_vs_the_gpu_path" % top_k: round(rec_f, 4),
                "mean_degree_level0": round(stats["mean_degree_level0"], 2), "max_level": stats["max_level"]}

    out = {"what": "HNSW restatement (oracle/hnsw_oracle.c: hnswlib's published algorithm, Chroma defaults l2 / M=16 / "
                   "ef_construction=100 / ef_search=100; parity unpinned) + the oracle's rerank, on the first %d of %d rows; "
                   "%d threads for the build and for the queries" % (m, n, cores), "threads": cores}
    out.update(one(rows[:m], q[:nq]))
    dev = rows.device
    g = torch.Generator(device=dev).manual_seed(4321)
    centres = torch.randn((1000, d), generator=g, device=dev)
    centres /= centres.norm(dim=1, keepdim=True)

    def draw(k):
        noise = torch.randn((k, d), generator=g, device=dev)
        x = centres[torch.randint(0, 1000, (k,), generator=g, device=dev)] + 0.5 * noise / noise.norm(dim=1, keepdim=True)
        return (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)

    out["clustered_corpus_same_shape"] = one(draw(m), draw(nq))
    out["note"] = ("q/s on the sample is an upper bound for the full corpus (cost ~ log rows); isotropic rows (the bench corpus) are the "
                   "hardest case for a graph index: its recall there is a lower bound for real embeddings, which cluster")
    return out


if __name__ == "__main__":
    main()
