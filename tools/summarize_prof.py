"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True))


for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  {:<70s} calls {:>5s} avg_ns {:>12s} total_ns {:>14s} pct {:>6s}".format(
            r.get("Name", "")[:70], r.get("Calls", ""), r.get("AverageNs", ""), r.get("TotalDurationNs", ""),
            r.get("Percentage", "")))

for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")[:60]
            acc[k][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", "0") or 0))
            meta[k] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("LDS_Block_Size", ""),
                       r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        print(f"== {sub}: per-dispatch mean of each counter")
        for k, cs in acc.items():
            if "ahf" not in k and "gauss" not in k and "nsf" not in k and "rnvp" not in k:
                continue
            print(f"  {k}  vgpr/lds/grid/wg={meta[k]}")
            for c, v in cs.items():
                print(f"      {c:<28s} n={len(v):<4d} mean={sum(v) / len(v):.6g}")
