"""Smoke run of one NSF_CL shape on the GPU: forward only, then forward + backward (python tools/try_nsf_shapes.py dim K n_h)."""
import sys, torch
sys.path.insert(0, ".")
import torch_mnf_amd as amd
dim, K, n_h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h).to("cuda")
x = torch.randn(777, dim, device="cuda")
with torch.no_grad():
    y, ld = f.inverse(x)
torch.cuda.synchronize()
print(dim, K, n_h, "forward ok", float(y.abs().sum()), flush=True)
x.requires_grad_(True)
y, ld = f.inverse(x)
(y.sum() + ld.sum()).backward()
torch.cuda.synchronize()
print(dim, K, n_h, "backward ok", float(x.grad.abs().sum()), flush=True)
