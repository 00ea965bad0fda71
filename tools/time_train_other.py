"""Training-step time of the layers whose gradient kernels are the generic ones: the C3 spline stack and
MNFLinear(800, 50) forward + kl_div.  `python3 tools/time_train_other.py [rows_c3] [rows_c5]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import torch_mnf_amd as amd
from torch_mnf_amd import synthetic as recipes

dev = torch.device("cuda", 0)


def timed(step, steps=5):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
model, layers = bench.build_c3(dev)
x = torch.randn(rows, 32, device=dev)
with torch.no_grad():
    t_fwd = timed(lambda: model.log_prob(x))
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def step():
    opt.zero_grad()
    (-model.log_prob(x).mean()).backward()
    opt.step()


print(f"C3 stack rows {rows}: log_prob {t_fwd:.3f} ms, training step {timed(step):.3f} ms")

rows5 = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
layer = amd.MNFLinear(800, 50).to(dev)
xin = torch.randn(rows5, 800, device=dev)
with torch.no_grad():
    t_fwd = timed(lambda: layer(xin))
opt5 = torch.optim.Adam(layer.parameters(), lr=1e-4)


def step5():
    opt5.zero_grad()
    out = layer(xin)
    (out.square().mean() + layer.kl_div()).backward()
    opt5.step()


print(f"MNFLinear(800,50) rows {rows5}: forward {t_fwd:.3f} ms, training step {timed(step5):.3f} ms")
