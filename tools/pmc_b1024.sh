#!/bin/bash
# FETCH_SIZE of the batch-1024 top-100 last level for pacing settings "ahead[:every]" (same box): tools/pmc_b1024.sh "0 8:8 4:4"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for vv in $1; do
  v=${vv%%:*}; e=${vv##*:}; [ "$e" = "$vv" ] && e=8
  O=gpurun_out/pmc_pace_${v}_$e; rm -rf $O; mkdir -p $O
  CRAG_AHEAD_MAX=$v CRAG_PACE_EVERY=$e timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O -o p --output-format csv -- python3 bench.py --batch 1024 --top-k 100 --no-extras --steps 3 --warmup 1 > $O/log.txt 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("$O/**/p_counter_collection.csv", recursive=True)[0]
best = {}
for r in csv.DictReader(open(f)):
    if "k_scan_q16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        best[r["Dispatch_Id"]] = float(r["Counter_Value"])
top = sorted(best.values())[-3:]
print("CRAG_AHEAD_MAX=$v CRAG_PACE_EVERY=$e last-level FETCH_SIZE x2 (GB):", [round(x * 1024 * 2 / 1e9, 2) for x in top])
PY
done
