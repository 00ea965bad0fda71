"""MNF-LeNet training step, eager vs replayed from a hipGraph (GraphedStep with torch.optim.Adam(capturable=True)):
`python3 tools/time_lenet_train_graphed.py [batch] [steps] [torch|fused]` (fused: FlatParameters + FusedAdam, one
optimiser launch and one memset for zero_grad)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn
import torch_mnf_amd as amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = "cuda"
torch.manual_seed(0)
net = nn.Sequential(amd.MNFConv2d(1, 20, 5), nn.ReLU(), nn.MaxPool2d(2), amd.MNFConv2d(20, 50, 5), nn.ReLU(),
                    nn.MaxPool2d(2), nn.Flatten(), amd.MNFLinear(800, 50), nn.ReLU(), amd.MNFLinear(50, 10),
                    nn.LogSoftmax(dim=-1)).to(dev)
which = sys.argv[3] if len(sys.argv) > 3 else "fused"
if which == "fused":
    opt = amd.FusedAdam(amd.FlatParameters(net), lr=1e-3, capturable=True)
else:
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
x = torch.rand(batch, 1, 28, 28, device=dev)
y = torch.randint(0, 10, (batch,), device=dev)


def loss_fn(xb, yb):
    kl = sum(m.kl_div() for m in net if hasattr(m, "kl_div"))
    return nn.functional.nll_loss(net(xb), yb) + kl / 60000


def eager():
    opt.zero_grad()
    loss = loss_fn(x, y)
    loss.backward(); opt.step()
    return loss


for _ in range(3): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = eager()
torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / steps
del loss  # (a live loss keeps the eager steps' AccumulateGrad nodes -- bound to the default stream -- alive into the capture)
step = amd.GraphedStep(opt, loss_fn, (x, y), model=net)
for _ in range(3): step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = step(x, y)
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / steps
print(f"MNF-LeNet training step, batch {batch}, {which} Adam: eager {t_eager * 1e3:.2f} ms, hipGraph replay {t_graph * 1e3:.2f} ms "
      f"({t_eager / t_graph:.1f} x), loss {float(loss):.4f}")
