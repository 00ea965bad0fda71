"""One RNVP layer's backward pass only, N times (`python3 tools/time_rnvp_bwd_only.py [rows] [n]`): for profilers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = amd.RNVP(800, h_sizes=(50,)).to("cuda")
z = torch.randn(rows, 800, device="cuda", requires_grad=True)
gx = torch.randn(rows, 800, device="cuda") / rows
gl = torch.full((rows,), 1.0 / rows, device="cuda")
x, ld = f.forward(z, seed=7)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
t0.record()
for _ in range(n):
    torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
t1.record(); torch.cuda.synchronize()
print(f"rows {rows}: backward {t0.elapsed_time(t1) / n:.3f} ms")
