"""Filter the rocprofv3 output of tools/profile_regimes.sh (gpurun_out/<set>/) down to this library's kernels, write the
summaries the docs cite into profiles/<set>_*, and refresh profiles/pmc_traffic.json (what bench.py reports as
roofline.traffic while the kernel sources are the ones profiled):  python tools/profile_summarise.py r02a"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02a"
src = os.path.join("gpurun_out", tag)
git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
sha = kernel_sources_sha()


def ours(name):
    return "anonymous namespace)::k_" in name and "at::native" not in name


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


def filt(inp, outp, key):
    rows = list(csv.DictReader(open(inp)))
    keep = [r for r in rows if ours(r[key])]
    with open(outp, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
    return keep


def scan_counters(rows):
    per = collections.defaultdict(dict)
    for r in rows:
        if "k_scan_" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    return per


summary = {}
traffic = {"_doc": "HBM-side bytes per launch of the dominant scan kernel (last cascade level) from rocprofv3 --pmc FETCH_SIZE, taken in a "
                   "pass of its own on the same bench.py command, gfx950 correction applied: KiB x 1024 x 2 (MI355X_MICROARCH.md, HBM "
                   "section). Key: rows|dim|batch. bench.py reports an entry as roofline.traffic only while kernel_sources_sha equals "
                   "the sha of the sources it runs (bench.kernel_sources_sha)."}
for name, B in (("b256", 256), ("b1", 1), ("b8", 8), ("b1024", 1024)):
    st = filt(find(f"kt_{name}/**/{name}_kernel_stats.csv"), f"profiles/{tag}_{name}_kernel_stats.csv", "Name")
    tr = list(csv.DictReader(open(find(f"kt_{name}/**/{name}_kernel_trace.csv"))))
    pm = filt(find(f"pmc_{name}/**/{name}_counter_collection.csv"), f"profiles/{tag}_{name}_pmc.csv", "Kernel_Name")
    scans = [r for r in tr if "k_scan_" in r["Kernel_Name"]]
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in scans)
    longest = [d for d in dur if d > 0.5 * dur[-1]]  # the last cascade level of every step
    best = max(scan_counters(pm).values(), key=lambda c: c.get("FETCH_SIZE", 0))
    kname = max(scans, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))["Kernel_Name"]
    s = {"kernel": kname, "last_level_launches": len(longest), "last_level_avg_ms": sum(longest) / len(longest) / 1e6,
         "hbm_bytes_per_launch": best["FETCH_SIZE"] * 1024 * 2}  # KiB, and the gfx950 factor 2 (MI355X_MICROARCH.md)
    entry = {"kernel": kname.split("::")[-1] + " (last cascade level)", "bytes": int(s["hbm_bytes_per_launch"]),
             "fetch_size_kib": best["FETCH_SIZE"], "kernel_ms_kernel_trace": round(s["last_level_avg_ms"], 4),
             "source": f"profiles/{tag}_{name}_pmc.csv, profiles/{tag}_{name}_kernel_stats.csv (tools/profile_regimes.sh, tools/profile_summarise.py)",
             "kernel_sources_sha": sha, "git": git}
    if name == "b1024":
        pm2 = filt(find("pmc_b1024_mfma/**/b1024_counter_collection.csv"), f"profiles/{tag}_b1024_mfma_pmc.csv", "Kernel_Name")
        b2 = max(scan_counters(pm2).values(), key=lambda c: c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
        s["mfma_busy_frac"] = b2["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * b2["GRBM_GUI_ACTIVE"] / 8)
        s["mfma_counters_of_one_last_level_launch"] = b2
        entry.update({"mfma_busy_frac": round(s["mfma_busy_frac"], 4), "SQ_VALU_MFMA_BUSY_CYCLES": b2["SQ_VALU_MFMA_BUSY_CYCLES"],
                      "SQ_INSTS_VALU_MFMA_MOPS_BF16": b2.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"), "GRBM_GUI_ACTIVE": b2["GRBM_GUI_ACTIVE"],
                      "_how": "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) for the last cascade level; "
                              "MOPS x 512 = flop of that launch = 2 x 1024 x 768 x its rows; top-100 run"})
    summary[name] = s
    traffic[f"10000000|768|{B}"] = entry
with open(f"{src}/bench_full.log") as f:
    line = [l for l in f if l.startswith("{")][-1]
open(f"profiles/{tag}_bench_full_line.json", "w").write(line)
# the same launches timed by the bench's own HIP events WITHOUT the profiler, same lease: a launch under rocprofv3
# runs a few per cent longer, so roofline.frac (events) and roofline.frac_profiled (kernel trace) are both on record
full = json.loads(line)
unprof = {"b256": full["roofline"]["kernel_ms"]}
for r in full.get("regimes", []):
    for name, key in (("b1", "batch-1 top-10 GEMV"), ("b8", "batch-8 top-10 MFMA"), ("b1024", "batch-1024 top-100")):
        if r.get("regime", "").startswith(key):
            unprof[name] = r["scan_kernel_ms"]
for name, ms in unprof.items():
    summary[name]["last_level_ms_hip_events_unprofiled_same_lease"] = ms
    summary[name]["profiled_over_unprofiled"] = round(summary[name]["last_level_avg_ms"] / ms, 4)
    traffic[f"10000000|768|{ {'b256': 256, 'b1': 1, 'b8': 8, 'b1024': 1024}[name] }"]["kernel_ms_hip_events_unprofiled"] = ms
# batch-1024: where a wave's cycles go
try:
    pm3 = filt(find("pmc_b1024_sq/**/b1024_counter_collection.csv"), f"profiles/{tag}_b1024_sq_pmc.csv", "Kernel_Name")
    b3 = max(scan_counters(pm3).values(), key=lambda c: c.get("SQ_WAVE_CYCLES", 0))
    wc = b3["SQ_WAVE_CYCLES"]
    summary["b1024"]["wave_cycle_shares"] = {k: round(b3[k] / wc, 4) for k in b3 if k != "SQ_WAVE_CYCLES" and k != "SQ_BUSY_CYCLES"}
    summary["b1024"]["clock_GHz_gui_active"] = round(summary["b1024"]["mfma_counters_of_one_last_level_launch"]["GRBM_GUI_ACTIVE"] / 8 /
                                                      (summary["b1024"]["last_level_avg_ms"] * 1e-3) / 1e9, 3)
except SystemExit:
    pass
# configs[3] at one GPU against the 10M-row shard: duration per row, traffic per row, clock
try:
    tr = list(csv.DictReader(open(find("kt_b256_100m/**/b256_100m_kernel_trace.csv"))))
    filt(find("kt_b256_100m/**/b256_100m_kernel_stats.csv"), f"profiles/{tag}_b256_100m_kernel_stats.csv", "Name")
    scans = [r for r in tr if "k_scan_" in r["Kernel_Name"]]
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in scans)
    longest = [d for d in dur if d > 0.5 * dur[-1]]
    pm = filt(find("pmc_b256_100m/**/b256_100m_counter_collection.csv"), f"profiles/{tag}_b256_100m_pmc.csv", "Kernel_Name")
    big = max(scan_counters(pm).values(), key=lambda c: c.get("FETCH_SIZE", 0))
    pmc = filt(find("pmc_b256_clk/**/b256_clk_counter_collection.csv"), f"profiles/{tag}_b256_clk_pmc.csv", "Kernel_Name")
    small = max(scan_counters(pmc).values(), key=lambda c: c.get("GRBM_GUI_ACTIVE", 0))
    ms100 = sum(longest) / len(longest) / 1e6
    summary["b256_100m"] = {"last_level_avg_ms": ms100, "hbm_bytes_per_launch": big["FETCH_SIZE"] * 1024 * 2,
                            "clock_GHz_gui_active": round(big["GRBM_GUI_ACTIVE"] / 8 / (ms100 * 1e-3) / 1e9, 3),
                            "clock_GHz_gui_active_10m_rows": round(small["GRBM_GUI_ACTIVE"] / 8 / (summary["b256"]["last_level_avg_ms"] * 1e-3) / 1e9, 3)}
except SystemExit:
    pass
try:
    open(f"profiles/{tag}_c2_timeline.txt", "w").write(open(f"{src}/c2_timeline.txt").read())
except OSError:
    pass
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1)
json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
