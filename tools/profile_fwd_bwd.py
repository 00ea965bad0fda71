"""Which launches one forward + backward call of a layer consists of (torch profiler, GPU time per kernel): the harness of
tools/coverage_map.py -- forward with the autograd link, sum() of both outputs, backward -- around one layer.
usage: python3 tools/profile_fwd_bwd.py [ahf|nsf|rnvp] [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import torch_mnf_amd as amd

kind = sys.argv[1] if len(sys.argv) > 1 else "ahf"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
if kind == "ahf":
    dim, layer, call = 64, amd.AffineHalfFlow(64, False, h_sizes=(64, 64, 64)), (lambda m, x: m.inverse(x))
elif kind == "nsf":
    dim, layer, call = 32, amd.NSF_CL(32, K=8, B=3, n_h=32), (lambda m, x: m.inverse(x))
else:
    dim, layer, call = 800, amd.RNVP(800, h_sizes=(100,)), (lambda m, x: m.forward(x, seed=3))
layer.to("cuda")
x = torch.randn(rows, dim, device="cuda", requires_grad=True)


def step():
    x.grad = None
    y, ld = call(layer, x)
    (y.sum() + ld.sum()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
