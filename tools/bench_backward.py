"""Training-step timing for the AffineHalfFlow stack (forward + backward through the HIP autograd path):
torch.optim.Adam over the 144 parameter tensors vs train.FlatParameters + train.FusedAdam (one buffer, one launch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch_mnf_amd import synthetic as recipes
import torch_mnf_amd as amd


def build(dim):
    flows = []
    for i, sd in enumerate(recipes.c2_stack_params(dim)):
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
    return amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")


for dim, rows in ((2, 128), (2, 4096), (64, 4096), (64, 1 << 16), (64, 1 << 18), (64, 1 << 20)):
    x = torch.randn(rows, dim, device="cuda")
    with torch.no_grad():
        m0 = build(dim)
        for _ in range(3): m0.log_prob(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m0.log_prob(x)
        torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 10
    res = []
    for mode in ("torch.optim.Adam", "FlatParameters+FusedAdam"):
        model = build(dim)
        if mode.startswith("Flat"):
            opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-4)
        else:
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        def step():
            opt.zero_grad()
            loss = -model.log_prob(x).mean()
            loss.backward(); opt.step()
            return loss
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): step()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / n)
    print(f"9xAHF d={dim} rows={rows}: inference pass {fwd*1e3:.2f} ms | training step: torch Adam {res[0]*1e3:.2f} ms, "
          f"flat + fused Adam {res[1]*1e3:.2f} ms = {res[1]/fwd:.1f} x the inference pass -> {rows/res[1]:.3e} samples/s")
