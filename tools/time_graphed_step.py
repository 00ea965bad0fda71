"""Eager vs hipGraph-replayed training step (FlatParameters + FusedAdam) of a 9 x AffineHalfFlow stack:
`python3 tools/time_graphed_step.py [dim] [rows] [steps]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch_mnf_amd import synthetic as recipes
import torch_mnf_amd as amd

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200


def build():
    flows = []
    for i, sd in enumerate(recipes.c2_stack_params(dim)):
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
    return model, amd.FusedAdam(amd.FlatParameters(model), lr=1e-3, capturable=True)


x = torch.randn(rows, dim, device="cuda")
model, opt = build()


def eager():
    opt.zero_grad()
    loss = -model.log_prob(x).mean()
    loss.backward(); opt.step()
    return loss


for _ in range(5): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = eager()
torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / steps
model, opt = build()
step = amd.GraphedStep(opt, lambda b: -model.log_prob(b).mean(), x)
for _ in range(5): step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = step(x)
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / steps
print(f"9 x AffineHalfFlow d={dim}, {rows} rows: Adam step eager {t_eager * 1e6:.1f} us, hipGraph replay {t_graph * 1e6:.1f} us "
      f"({t_eager / t_graph:.1f} x), loss {float(loss):.5f}")
