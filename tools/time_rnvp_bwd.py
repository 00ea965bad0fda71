"""RNVP forward + backward timings (`python3 tools/time_rnvp_bwd.py [rows] [dim] [hid]`): the matrix-core gradient
kernels (mnf_rnvp_bwd_mfma: launches A and B) against the generic kernel (MNF_RNVP_BWD_GENERIC=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 800
hid = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = "cuda"
torch.manual_seed(0)
f = amd.RNVP(dim, h_sizes=(hid,)).to(dev)
z = torch.randn(rows, dim, device=dev, requires_grad=True)
w = torch.randn(rows, dim, device=dev) / rows


def step():
    x, ld = f.forward(z, seed=7)
    loss = (x * w).sum() + ld.mean()
    loss.backward()


for generic in (False, True):
    amd._dispatch.RNVP_BWD_GENERIC = generic
    if generic and rows > 70000:
        continue
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        step()
    t1.record()
    torch.cuda.synchronize()
    print(f"rows {rows} dim {dim} hid {hid} {'generic' if generic else 'mfma'}: {t0.elapsed_time(t1) / n:.3f} ms per "
          f"forward + loss + backward")
