"""Where the launches of one MNF-LeNet training step come from: kernels per segment (forward, kl_div, the two backward
halves, optimiser) and the aten ops that launch them.  `python3 tools/lenet_launch_census.py [batch]`."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn
from torch.profiler import profile, ProfilerActivity
import torch_mnf_amd as amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
torch.manual_seed(0)
net = nn.Sequential(amd.MNFConv2d(1, 20, 5), nn.ReLU(), nn.MaxPool2d(2), amd.MNFConv2d(20, 50, 5), nn.ReLU(),
                    nn.MaxPool2d(2), nn.Flatten(), amd.MNFLinear(800, 50), nn.ReLU(), amd.MNFLinear(50, 10),
                    nn.LogSoftmax(dim=-1)).to(dev)
opt = amd.FusedAdam(amd.FlatParameters(net), lr=1e-3, capturable=True)
x = torch.rand(batch, 1, 28, 28, device=dev)
y = torch.randint(0, 10, (batch,), device=dev)


def census(label, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    kernels = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ops = collections.Counter()
    for e in prof.key_averages():
        if e.device_time_total > 0 and e.self_device_time_total > 0 and e.device_type == torch.autograd.DeviceType.CPU:
            ops[e.key] += e.count
    print(f"--- {label}: {len(kernels)} device events, {sum(k.device_time for k in kernels) / 1e3:.2f} ms of device time")
    print("    " + ", ".join(f"{k} x{v}" for k, v in ops.most_common(30)))
    return out


for _ in range(3):
    opt.zero_grad()
    (nn.functional.nll_loss(net(x), y) + sum(m.kl_div() for m in net if hasattr(m, "kl_div")) / 60000).backward()
    opt.step()
opt.zero_grad()
out = census("forward net(x)", lambda: net(x))
nll = nn.functional.nll_loss(out, y)
for i, m in enumerate(net):
    if hasattr(m, "kl_div"):
        census(f"kl_div of layer {i} ({type(m).__name__})", m.kl_div)
kl = census("kl_div, all four layers", lambda: sum(m.kl_div() for m in net if hasattr(m, "kl_div")))
census("backward of the likelihood term", lambda: nll.backward())
census("backward of the KL term", lambda: (kl / 60000).backward())
census("optimiser step + zero_grad", lambda: (opt.step(), opt.zero_grad()))
