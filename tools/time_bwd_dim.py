"""Training-step time of a 9 x AffineHalfFlow stack at width `dim`, on the gradient kernel the library picks and on
the generic one: `python3 tools/time_bwd_dim.py [dim] [rows] [steps]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch_mnf_amd import synthetic as recipes
import torch_mnf_amd as amd

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
x = torch.randn(rows, dim, device="cuda")
for generic in (False, True):
    flows = []
    for i, sd in enumerate(recipes.c2_stack_params(dim)):
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd)
        if generic: f._bwd_index = lambda device: None  # (the forward pass keeps its MFMA kernel)
        flows.append(f)
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
    opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-4)

    def step():
        opt.zero_grad()
        loss = -model.log_prob(x).mean()
        loss.backward(); opt.step()
        return loss

    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): loss = step()
    torch.cuda.synchronize()
    print(f"dim {dim} rows {rows} {'generic' if generic else 'default'} gradient kernel: "
          f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms per training step, loss {loss.item():.6f}")
