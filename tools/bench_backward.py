"""Training-step timing for the AffineHalfFlow stack (forward + backward through the HIP autograd path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, recipes
import torch_mnf_amd as amd
for dim, rows in ((2, 128), (2, 4096), (64, 4096), (64, 1 << 16), (64, 1 << 18), (64, 1 << 20)):
    flows = []
    for i, sd in enumerate(recipes.c2_stack_params(dim)):
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    x = torch.randn(rows, dim, device="cuda")
    def step():
        loss = -model.log_prob(x).mean()
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"9xAHF d={dim} rows={rows}: training step {dt*1e3:.2f} ms -> {rows/dt:.3e} samples/s")
