"""Filter the rocprofv3 output of tools/profile_regimes.sh (gpurun_out/<set>/) down to this library's kernels
and write the summaries the docs cite into profiles/<set>_*:  python tools/profile_summarise.py r01k"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01k"
src = os.path.join("gpurun_out", tag)


def ours(name):
    return "anonymous namespace)::k_" in name and "at::native" not in name


def filt(inp, outp, key):
    rows = list(csv.DictReader(open(inp)))
    keep = [r for r in rows if ours(r[key])]
    with open(outp, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
    return keep


summary = {}
for name in ("b256", "b1", "b1024"):
    st = filt(f"{src}/kt_{name}/{name}_kernel_stats.csv", f"profiles/{tag}_{name}_kernel_stats.csv", "Name")
    tr = list(csv.DictReader(open(f"{src}/kt_{name}/{name}_kernel_trace.csv")))
    pm = filt(f"{src}/pmc_{name}/{name}_counter_collection.csv", f"profiles/{tag}_{name}_pmc.csv", "Kernel_Name")
    scans = [r for r in tr if "k_scan_" in r["Kernel_Name"]]
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in scans)
    longest = [d for d in dur if d > 0.5 * dur[-1]]  # the last cascade level of every step
    per = collections.defaultdict(dict)
    for r in pm:
        if "k_scan_" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    first = next(iter(per.values()))
    key = "FETCH_SIZE" if "FETCH_SIZE" in first else "SQ_VALU_MFMA_BUSY_CYCLES"
    best = max(per.values(), key=lambda c: c.get(key, 0))
    s = {"last_level_launches": len(longest), "last_level_avg_ms": sum(longest) / len(longest) / 1e6, "counters_of_one_last_level_launch": best}
    if key == "FETCH_SIZE":
        s["hbm_bytes_per_launch"] = best["FETCH_SIZE"] * 1024 * 2  # KiB, and the gfx950 factor 2 (MI355X_MICROARCH.md)
    else:
        s["mfma_busy_frac"] = best["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * best["GRBM_GUI_ACTIVE"] / 8)
    summary[name] = s
with open(f"{src}/bench_full.log") as f:
    line = [l for l in f if l.startswith("{")][-1]
open(f"profiles/{tag}_bench_full_line.json", "w").write(line)
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
