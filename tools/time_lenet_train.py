"""One MNF-LeNet training step (the reference's MNIST example: NLL + KL / dataset size, Adam) on synthetic images:
`python3 tools/time_lenet_train.py [batch] [steps]`.  Under rocprofv3 --kernel-trace --stats it shows which kernels
the step spends its time in."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn
import torch_mnf_amd as amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda"
torch.manual_seed(0)
net = nn.Sequential(amd.MNFConv2d(1, 20, 5), nn.ReLU(), nn.MaxPool2d(2), amd.MNFConv2d(20, 50, 5), nn.ReLU(),
                    nn.MaxPool2d(2), nn.Flatten(), amd.MNFLinear(800, 50), nn.ReLU(), amd.MNFLinear(50, 10),
                    nn.LogSoftmax(dim=-1)).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
x = torch.rand(batch, 1, 28, 28, device=dev)
y = torch.randint(0, 10, (batch,), device=dev)


def step():
    opt.zero_grad()
    nll = nn.functional.nll_loss(net(x), y)
    kl = sum(m.kl_div() for m in net if hasattr(m, "kl_div"))
    loss = nll + kl / 60000
    loss.backward()
    opt.step()
    return loss


for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = step()
torch.cuda.synchronize()
print(f"MNF-LeNet training step, batch {batch}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms, loss {loss.item():.4f}")
