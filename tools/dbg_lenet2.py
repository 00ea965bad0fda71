import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch import nn
import torch_mnf_amd as amd
batch, steps = 128, int(sys.argv[1])
dev = "cuda"
torch.manual_seed(0)
net = nn.Sequential(amd.MNFConv2d(1, 20, 5), nn.ReLU(), nn.MaxPool2d(2), amd.MNFConv2d(20, 50, 5), nn.ReLU(),
                    nn.MaxPool2d(2), nn.Flatten(), amd.MNFLinear(800, 50), nn.ReLU(), amd.MNFLinear(50, 10),
                    nn.LogSoftmax(dim=-1)).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
x = torch.rand(batch, 1, 28, 28, device=dev)
y = torch.randint(0, 10, (batch,), device=dev)
def loss_fn(xb, yb):
    kl = sum(m.kl_div() for m in net if hasattr(m, "kl_div"))
    return nn.functional.nll_loss(net(xb), yb) + kl / 60000
def eager():
    opt.zero_grad(); loss = loss_fn(x, y); loss.backward(); opt.step(); return loss
for i in range(steps):
    loss = eager()
    if os.environ.get("SYNC_EACH"): torch.cuda.synchronize(); print("eager", i, float(loss), flush=True)
torch.cuda.synchronize(); print("eager done", float(loss), flush=True)
del loss
step = amd.GraphedStep(opt, loss_fn, (x, y), model=net)
torch.cuda.synchronize(); print("captured", flush=True)
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for i in range(n_rep):
    loss = step(x, y)
    if os.environ.get("SYNC_REPLAY"):
        torch.cuda.synchronize()
        print("replay", i, float(loss), max(float(p.detach().abs().max()) for p in net.parameters()), flush=True)
torch.cuda.synchronize(); print("replays done", float(loss), flush=True)
