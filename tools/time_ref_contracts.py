"""Step time of the reference's own training tests (tests/test_flows.py: 128 half-moon points, Adam) on the HIP modules:
`python3 tools/time_ref_contracts.py` -- under rocprofv3 --kernel-trace --stats it shows the kernels behind them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

dev = "cuda"
torch.manual_seed(0)
x = torch.randn(128, 2, device=dev) * 0.7


def model_of(name):
    if name == "rnvp":
        flows = [amd.AffineHalfFlow(dim=2, parity=i % 2 == 0) for i in range(2)]
    elif name == "half_moons":
        flows = [amd.AffineHalfFlow(dim=2, parity=bool(i % 2)) for i in range(9)]
    elif name == "nsfar":
        flows = [amd.NSF_AR(dim=2, K=8, B=3, n_h=16) for _ in range(2)]
    elif name == "nsfcl":
        flows = [amd.NSF_CL(dim=2, K=8, B=3, n_h=16) for _ in range(2)]
    else:
        flows = [amd.Glow(dim=2) for _ in range(2)]
    # (validate_args=False: the sample check is a host synchronisation, which a hipGraph capture cannot record)
    base = torch.distributions.MultivariateNormal(torch.zeros(2, device=dev), torch.eye(2, device=dev), validate_args=False)
    return amd.NormalizingFlowModel(base, flows).to(dev)


for name in ("half_moons", "rnvp", "glow", "nsfcl", "nsfar"):
    model = model_of(name)
    adam = torch.optim.Adam(model.parameters())

    def step():
        adam.zero_grad()
        zs, ld = model.inverse(x)
        loss = -(model.base.log_prob(zs[-1]) + ld).sum()
        loss.backward(); adam.step()

    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 50
    # the same step replayed from a hipGraph (torch.optim.Adam(capturable=True))
    model = model_of(name)
    adam = torch.optim.Adam(model.parameters(), capturable=True)

    def loss_fn(xb):
        zs, ld = model.inverse(xb)
        return -(model.base.log_prob(zs[-1]) + ld).sum()

    graphed = amd.GraphedStep(adam, loss_fn, x, model=model)
    for _ in range(5): graphed(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): graphed(x)
    torch.cuda.synchronize()
    print(f"{name:10s} training step on 128 rows: eager {t_eager * 1e6:.0f} us, hipGraph {(time.perf_counter() - t0) / 50 * 1e6:.0f} us")
