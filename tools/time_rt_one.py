"""one run-time-shaped AffineHalfFlow shape, a few launches (for rocprofv3): time_rt_one.py dim h1,h2,.. [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_mnf_amd as amd
dim = int(sys.argv[1]); hs = tuple(int(v) for v in sys.argv[2].split(",")); rows = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs).to("cuda"); f.force_generic = 2
x = torch.randn(rows, dim, device="cuda")
with torch.no_grad():
    for _ in range(5):
        f.forward(x)
torch.cuda.synchronize()
print(amd.last_kernel())
