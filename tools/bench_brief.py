"""One bench.py line, briefly: value, ms/step, the dominant kernel's average launch time.  usage: python tools/bench_brief.py [bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1]); r = d["roofline"]
print(f"{d['config']['workload'][:40]:40s} value {d['value']:.4g} {d['unit']}  {d['ms_per_step']:.4f} ms/step  kernel {r.get('avg_kernel_us', 0):.1f} us  "
      f"frac {r['frac']:.3f} phys {r.get('frac_physical')}")
