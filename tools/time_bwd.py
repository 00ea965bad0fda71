"""A few training steps of the 9 x AffineHalfFlow d = 64 stack (FlatParameters + FusedAdam) for kernel-level
profiling: `rocprofv3 --kernel-trace --stats -- python3 tools/time_bwd.py [rows] [steps]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch_mnf_amd import synthetic as recipes
import torch_mnf_amd as amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dim = 64
flows = []
for i, sd in enumerate(recipes.c2_stack_params(dim)):
    f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-4)
x = torch.randn(rows, dim, device="cuda")


def step():
    opt.zero_grad()
    loss = -model.log_prob(x).mean()
    loss.backward(); opt.step()


for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"rows {rows}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per training step")
