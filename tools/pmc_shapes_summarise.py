"""Summarise tools/pmc_shapes.sh: per matrix shape the last cascade level's duration (kernel trace), the SQ wait / issue
breakdown as fractions of SQ_WAVE_CYCLES, matrix-pipe busy fraction, clock and HBM-side traffic.
    python tools/pmc_shapes_summarise.py <tag>   (reads gpurun_out/<tag>, writes gpurun_out/<tag>/summary.json)"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)


def counters(d):
    f = glob.glob(os.path.join(src, d, "**", "p_counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_scan_q" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            per[r["Dispatch_Id"]]["_kernel"] = r["Kernel_Name"]
    return per


def last_level(per, key):
    """the last cascade level = the dispatch with the largest value of `key`"""
    return max(per.values(), key=lambda c: c.get(key, 0))


out = {}
for shape in ("32", "16"):
    tr = glob.glob(os.path.join(src, f"kt_{shape}", "**", "p_kernel_trace.csv"), recursive=True)[0]
    scans = [r for r in csv.DictReader(open(tr)) if "k_scan_q" in r["Kernel_Name"]]
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in scans)
    longest = [d for d in dur if d > 0.5 * dur[-1]]
    ms = sum(longest) / len(longest) / 1e6
    sq = last_level(counters(f"sq_{shape}"), "SQ_WAVE_CYCLES")
    mf = last_level(counters(f"mfma_{shape}"), "SQ_VALU_MFMA_BUSY_CYCLES")
    fe = last_level(counters(f"fetch_{shape}"), "FETCH_SIZE")
    wc = sq["SQ_WAVE_CYCLES"]
    flop = mf["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    s = {"kernel": sq["_kernel"].split("::")[-1][:40], "last_level_ms_kernel_trace": round(ms, 4), "launches": len(longest),
         "flop": flop, "TFLOPs": round(flop / ms / 1e9, 1), "frac_of_2500": round(flop / ms / 1e9 / 2500, 4),
         "wave_cycles_quads": wc,
         "wait_any_frac": round(sq["SQ_WAIT_ANY"] / wc, 4), "wait_inst_any_frac": round(sq["SQ_WAIT_INST_ANY"] / wc, 4),
         "active_inst_any_frac": round(sq["SQ_ACTIVE_INST_ANY"] / wc, 4), "active_inst_valu_frac": round(sq["SQ_ACTIVE_INST_VALU"] / wc, 4),
         "active_inst_lds_frac": round(sq["SQ_ACTIVE_INST_LDS"] / wc, 4), "active_inst_misc_frac": round(sq["SQ_ACTIVE_INST_MISC"] / wc, 4),
         "mfma_busy_frac": round(mf["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * mf["GRBM_GUI_ACTIVE"] / 8), 4),
         "clock_GHz_gui_active": round(mf["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9, 3),
         "hbm_bytes": fe["FETCH_SIZE"] * 1024 * 2, "raw_sq": {k: v for k, v in sq.items() if k != "_kernel"},
         "raw_mfma": {k: v for k, v in mf.items() if k != "_kernel"}}
    out[shape] = s
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if not kk.startswith("raw")} for k, v in out.items()}, indent=1))
