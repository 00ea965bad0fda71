"""RNVP forward / backward at a handful of rows: the few-rows kernels (mnf_rnvp_few.hip) against the streaming ones
(`MNF_RNVP_FEW=0`), per kernel by HIP events.  `python3 tools/time_rnvp_few.py`"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    for few in ("1", "0"):
        env = dict(os.environ, MNF_RNVP_FEW=few)
        print(f"MNF_RNVP_FEW={few}", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
    sys.exit(0)

import torch
import torch_mnf_amd as amd

dev = "cuda"
for dim, rows in ((800, 1), (800, 2), (50, 1), (20, 1)):
    f = amd.RNVP(dim, h_sizes=(50,)).to(dev)
    z = torch.randn(rows, dim, device=dev, requires_grad=True)
    gx, gl = torch.randn(rows, dim, device=dev), torch.randn(rows, device=dev)
    x, ld = f.forward(z, seed=5)

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            fn()
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) / n * 1e3

    bwd = timed(lambda: torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True))
    with torch.no_grad():
        fwd = timed(lambda: f.forward(z, seed=5))
    print(f"  d={dim} rows={rows}: forward {fwd:.1f} us, backward {bwd:.1f} us (host launch overhead included)", flush=True)
