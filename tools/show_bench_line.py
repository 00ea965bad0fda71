"""Digest of one bench.py JSON line: value, ms per step, roofline fraction and kernel time, then the secondary block.
usage: show_bench_line.py <file with the line last>   (or /dev/stdin)"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"])
for k, v in d.get("secondary", {}).items():
    print(k, v.get("ms_per_step"), v.get("value"))
