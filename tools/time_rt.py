"""Time per row of the run-time-shaped kernels (force_generic = 2), forward and forward + backward, for a few shapes.
usage: python3 tools/time_rt.py [rows] [only the cases whose name contains this]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import torch_mnf_amd as amd

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
DEV = "cuda"


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e6 / ROWS


CASES = [
    ("ahf d=64 (24,24)", lambda: amd.AffineHalfFlow(64, False, h_sizes=(24, 24)), 64, lambda m, x: m.inverse(x)),
    ("ahf d=64 (64,64,64)", lambda: amd.AffineHalfFlow(64, False, h_sizes=(64, 64, 64)), 64, lambda m, x: m.inverse(x)),
    ("ahf d=256 (32,32,32)", lambda: amd.AffineHalfFlow(256, False, h_sizes=(32, 32, 32)), 256, lambda m, x: m.inverse(x)),
    ("ahf d=512 (24,24,24)", lambda: amd.AffineHalfFlow(512, False, h_sizes=(24, 24, 24)), 512, lambda m, x: m.inverse(x)),
    ("ahf d=512 (64,64,64)", lambda: amd.AffineHalfFlow(512, False, h_sizes=(64, 64, 64)), 512, lambda m, x: m.inverse(x)),
    ("nsf d=32 K=8 n_h=16", lambda: amd.NSF_CL(32, K=8, B=3, n_h=16), 32, lambda m, x: m.inverse(x)),
    ("nsf d=32 K=8 n_h=32", lambda: amd.NSF_CL(32, K=8, B=3, n_h=32), 32, lambda m, x: m.inverse(x)),
    ("nsf d=128 K=8 n_h=8", lambda: amd.NSF_CL(128, K=8, B=3, n_h=8), 128, lambda m, x: m.inverse(x)),
    ("rnvp d=800 h=50", lambda: amd.RNVP(800, h_sizes=(50,)), 800, lambda m, x: m.forward(x, seed=3)),
    ("rnvp d=800 h=100", lambda: amd.RNVP(800, h_sizes=(100,)), 800, lambda m, x: m.forward(x, seed=3)),
    ("rnvp d=2048 h=100", lambda: amd.RNVP(2048, h_sizes=(100,)), 2048, lambda m, x: m.forward(x, seed=3)),
]
for name, make, dim, call in CASES:
    if ONLY not in name:
        continue
    layer = make().to(DEV)
    layer.force_generic = 2
    x = torch.randn(ROWS, dim, device=DEV)
    with torch.no_grad():
        tf = timed(lambda: call(layer, x))
    kf = amd.last_kernel()
    xg = x.clone().requires_grad_(True)

    def train():
        xg.grad = None
        y, ld = call(layer, xg)
        (y.sum() + ld.sum()).backward()

    tb = timed(train)
    print(f"{name:24s} forward {tf:7.3f} ns/row ({kf}); forward + backward {tb:7.3f} ns/row ({amd.last_kernel()})")
