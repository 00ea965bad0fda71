"""Which kernels a replay of the captured MNF-LeNet step spends its time in (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn
from torch.profiler import profile, ProfilerActivity
import torch_mnf_amd as amd

dev = "cuda"
torch.manual_seed(0)
net = nn.Sequential(amd.MNFConv2d(1, 20, 5), nn.ReLU(), nn.MaxPool2d(2), amd.MNFConv2d(20, 50, 5), nn.ReLU(),
                    nn.MaxPool2d(2), nn.Flatten(), amd.MNFLinear(800, 50), nn.ReLU(), amd.MNFLinear(50, 10),
                    nn.LogSoftmax(dim=-1)).to(dev)
opt = amd.FusedAdam(amd.FlatParameters(net), lr=1e-3, capturable=True)
x = torch.rand(128, 1, 28, 28, device=dev)
y = torch.randint(0, 10, (128,), device=dev)


def loss_fn(xb, yb):
    kl = sum(m.kl_div() for m in net if hasattr(m, "kl_div"))
    return nn.functional.nll_loss(net(xb), yb) + kl / 60000


import time
if len(sys.argv) > 1:  # eager steps first, as tools/time_lenet_train_graphed.py does
    for _ in range(int(sys.argv[1])):
        opt.zero_grad(); l = loss_fn(x, y); l.backward(); opt.step()
    del l
step = amd.GraphedStep(opt, loss_fn, (x, y), model=net)
for n in (3, 10, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step(x, y)
    torch.cuda.synchronize()
    print(f"{n} replays: {(time.perf_counter() - t0) / n * 1e3:.2f} ms each")
t0 = time.perf_counter()
for _ in range(20):
    step(x, y); torch.cuda.synchronize()
print(f"20 synchronised replays: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms each")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    step(x, y)
torch.cuda.synchronize()
print(f"50 replays again: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms each")
f = opt.flat
print("loss", float(step.loss), "params finite", bool(torch.isfinite(f.data).all()), "max|p|", float(f.data.abs().max()),
      "grads finite", bool(torch.isfinite(f.grad).all()), "max|g|", float(f.grad.abs().max()))
for name, p_ in net.named_parameters():
    if not torch.isfinite(p_.grad).all() or float(p_.grad.abs().max()) > 1e3 or float(p_.abs().max()) > 50:
        print("   ", name, "max|p|", float(p_.abs().max()), "max|g|", float(p_.grad.abs().max()))
if os.environ.get("NOPROF"): sys.exit(0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step(x, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="device_time_total", row_limit=14, max_name_column_width=70))
