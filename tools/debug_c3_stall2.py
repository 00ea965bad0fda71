import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for i in range(6):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--workload", "c3", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(i, round(d["value"] / 1e6, 1), round(d["ms_per_step"], 3), d["roofline"]["per_layer_us"])
