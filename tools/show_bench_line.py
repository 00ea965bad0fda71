import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"])
for k,v in d["secondary"].items(): print(k, v.get("ms_per_step"), v.get("value"))
