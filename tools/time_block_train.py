"""One training step of 3 x [ActNormFlow, Glow, NSF_CL] at a given dim (c3t's model at d = 32): ms per step.
usage: python3 tools/time_block_train.py [dim] [rows]   (under rocprofv3 --kernel-trace --stats for the kernel split)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19
torch.manual_seed(0)
layers = []
for _ in range(3):
    layers += [amd.ActNormFlow(dim), amd.Glow(dim), amd.NSF_CL(dim, K=8, B=3, n_h=8)]
model = amd.NormalizingFlowModel(amd.StandardNormal(dim), layers).to("cuda")
x = torch.randn(rows, dim, device="cuda")
with torch.no_grad():
    model.log_prob(x)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def step():
    opt.zero_grad()
    loss = -model.log_prob(x).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    loss = step()
t1.record()
torch.cuda.synchronize()
print(f"d={dim} rows={rows}: {t0.elapsed_time(t1) / 20:.3f} ms per step, loss {float(loss):.4f}")
