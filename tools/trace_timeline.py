"""Print the kernel timeline (duration and gap to the previous kernel) of the last `n` dispatches of a
rocprofv3 --kernel-trace CSV:  python tools/trace_timeline.py <kernel_trace.csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{name:44s} grid {r.get('Grid_Size', r.get('Grid_Size_X', '')):>9s}  dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us")
    prev = e
