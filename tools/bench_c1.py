"""Config 1 latency: 9 x AffineHalfFlow d=2 on 4096 half-moon points (generic kernel), eager vs HIP graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch_mnf_amd as amd
from helpers import g1_layers, t
fx = dict(np.load(os.path.join(ROOT, "tests/golden/g1_c1_stack_trained.npz")))
flows = []
for spec in g1_layers(fx):
    f = amd.AffineHalfFlow(2, spec["parity"]); f.load_state_dict(spec["params"]); flows.append(f)
model = amd.NormalizingFlowModel(amd.StandardNormal(2), flows).to("cuda")
x = t(fx["x"]).cuda()
def timed(fn, n=200):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    te = timed(lambda: model.log_prob(x, return_sum=True))
    lp, tot = model.log_prob(x, return_sum=True)
replay = model.graphed_log_prob(x)
tg = timed(lambda: replay(x))
ref = float(fx["mean_log_prob"]); got = float(tot) / x.shape[0]
print(f"C1 (9xAHF d=2, 4096 rows): eager {te*1e6:.0f} us/pass = {4096/te:.3e} samples/s; graph {tg*1e6:.0f} us/pass = {4096/tg:.3e} samples/s; "
      f"mean log-prob {got:.6f} vs reference {ref:.6f} (rel {abs(got-ref)/abs(ref):.1e})")
