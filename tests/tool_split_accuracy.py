"""Normwise distance of one AffineHalfFlow layer / the 9-layer C2 stack from float64, split kernel vs the reference's
own fp32 arithmetic (`[MNF_LIB_PATH=...] python3 tools/split_accuracy.py`): the accuracy side of arithmetic experiments
on the headline kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import torch_mnf_amd as amd
from torch_mnf_amd import synthetic as recipes
from oracle import flow_oracle as O
from helpers import normwise_err

print("library:", amd.library_path())
for dim in (32, 64, 256):
    sd = recipes.affine_half_params(300 + dim, dim, s_last_gain=2.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    x = recipes.gaussian(301 + dim, 2048, dim)
    for inverse in (False, True):
        y64, ld64 = O.affine_half(x.double(), sd64, True, inverse)
        ry, rld = O.affine_half(x, sd, True, inverse)
        f = amd.AffineHalfFlow(dim, True); f.load_state_dict(sd); f.to("cuda")
        with torch.no_grad():
            y, ld = f.forward(x.cuda(), inverse=inverse)
        print(f"d={dim} inverse={inverse}: y: split kernel {normwise_err(y.cpu().numpy(), y64.numpy()):.2e} from float64, "
              f"reference fp32 {normwise_err(ry.numpy(), y64.numpy()):.2e}; log_det: {normwise_err(ld.cpu().numpy(), ld64.numpy()):.2e} / "
              f"{normwise_err(rld.numpy(), ld64.numpy()):.2e}")
# the C2 stack: 9 layers, mean log-prob and last z
dim = 64
sds = recipes.c2_stack_params(dim)
layers = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd} for i, sd in enumerate(sds)]
layers64 = [{"kind": "affine_half", "parity": bool(i % 2), "params": {k: v.double() for k, v in sd.items()}} for i, sd in enumerate(sds)]
x = recipes.gaussian(77, 8192, dim)
zs64, ld64 = O.flow_stack(x.double(), layers64, inverse=True)
zs32, ld32 = O.flow_stack(x, layers, inverse=True)
flows = []
for i, sd in enumerate(sds):
    f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
with torch.no_grad():
    zs, ld = model.inverse(x.cuda())
print(f"C2 stack z_last: kernel {normwise_err(zs[-1].cpu().numpy(), zs64[-1].numpy()):.2e}, reference fp32 {normwise_err(zs32[-1].numpy(), zs64[-1].numpy()):.2e}; "
      f"log_det: kernel {normwise_err(ld.cpu().numpy(), ld64.numpy()):.2e}, reference fp32 {normwise_err(ld32.numpy(), ld64.numpy()):.2e}")
