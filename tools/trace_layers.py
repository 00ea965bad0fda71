"""Per-position kernel durations and gaps from a rocprofv3 kernel trace of bench.py (one step = a fixed
sequence of launches): usage  trace_layers.py <dir> <launches per step>"""
import csv, glob, os, sys
from collections import defaultdict
d, per = sys.argv[1], int(sys.argv[2])
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("ahf", "gauss", "nsf", "rnvp", "affine", "linear"))]
rows = rows[-per * 15:]  # the last 15 steps
dur, gap = defaultdict(list), defaultdict(list)
for i, r in enumerate(rows):
    p = i % per
    dur[p].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if i:
        gap[p].append(int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
for p in range(per):
    name = rows[p]["Kernel_Name"][:48]
    print(f"{p:2d} {name:<48s} dur {sum(dur[p]) / len(dur[p]) / 1e3:8.1f} us   gap before {sum(gap[p]) / max(len(gap[p]), 1) / 1e3:7.1f} us")
tot = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / (len(rows) / per)
print(f"step period {tot / 1e3:.1f} us")
