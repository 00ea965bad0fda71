"""A few training steps of a 9 x AffineHalfFlow stack at a given width and hidden width (FlatParameters + FusedAdam), for
kernel-level A/Bs: `rocprofv3 --kernel-trace --stats -- python3 tools/time_bwd_hid.py [dim] [hid] [rows] [steps]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch_mnf_amd import synthetic as recipes
import torch_mnf_amd as amd

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hid = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
flows = []
for i in range(9):
    sd = recipes.affine_half_params(1000 + i, dim, h_sizes=(hid, hid, hid), s_last_gain=2.0)
    f = amd.AffineHalfFlow(dim, parity=bool(i % 2), h_sizes=(hid, hid, hid)); f.load_state_dict(sd); flows.append(f)
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
print(f"dim {dim} hid {hid} rows {rows}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per training step")
