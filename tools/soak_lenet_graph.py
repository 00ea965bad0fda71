"""Soak: the captured MNF-LeNet training step replayed thousands of times -- no fault, finite parameters, a replay time
that does not drift (the two hipGraph findings of profiles/r3/DESIGN_round3.md 3.6 both showed up only after tens to hundreds of replays).
`python3 tools/soak_lenet_graph.py [replays]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn
import torch_mnf_amd as amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = "cuda"
torch.manual_seed(0)
net = amd.MNFLeNet().to(dev)
opt = amd.FusedAdam(amd.FlatParameters(net), lr=1e-3, capturable=True)
protos = torch.rand(10, 1, 28, 28, device=dev)


def batch():
    y = torch.randint(0, 10, (128,), device=dev)
    return (protos[y] + 0.3 * torch.randn(128, 1, 28, 28, device=dev)).clamp(0, 1), y


x, y = batch()
step = amd.GraphedStep(opt, lambda xb, yb: nn.functional.nll_loss(net(xb), yb) + net.kl_div() / 60000, (x, y), model=net)
times, first = [], None
for block in range(n // 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        loss = step(*batch())
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) / 100 * 1e3)
    if first is None:
        first = float(loss)
flat = opt.flat
ok = bool(torch.isfinite(flat.data).all()) and bool(torch.isfinite(flat.grad).all())
with torch.no_grad():
    xv, yv = batch()
    acc = float((net(xv).argmax(1) == yv).float().mean())
print(f"{n} replays: ms per replay (with the batch draw) first block {times[0]:.2f}, last {times[-1]:.2f}, max {max(times):.2f}; "
      f"loss {first:.3f} -> {float(loss):.3f}; batch accuracy {acc:.2f}; parameters and gradients finite: {ok}")
sys.exit(0 if ok and max(times) < 3 * times[0] and acc > 0.8 else 1)
