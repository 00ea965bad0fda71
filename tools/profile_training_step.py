"""Where a training step of the 9-layer AffineHalfFlow stack spends its host time: phase timings
(forward / backward / optimizer, each synchronised) and a cProfile of the whole step."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, recipes
import torch_mnf_amd as amd

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
flows = []
for i, sd in enumerate(recipes.c2_stack_params(dim)):
    f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
x = torch.randn(rows, dim, device="cuda")
sync = torch.cuda.synchronize

def phases(n=20):
    t = [0.0] * 4
    for _ in range(n):
        sync(); a = time.perf_counter()
        loss = -model.log_prob(x).mean(); sync(); b = time.perf_counter()
        opt.zero_grad(); sync(); c = time.perf_counter()
        loss.backward(); sync(); d = time.perf_counter()
        opt.step(); sync(); e = time.perf_counter()
        for k, v in enumerate((b - a, c - b, d - c, e - d)): t[k] += v / n
    return t

def step():
    loss = -model.log_prob(x).mean()
    opt.zero_grad(); loss.backward(); opt.step()

for _ in range(3): step()
f, z, b, o = phases()
print(f"d={dim} rows={rows}: forward {f*1e3:.2f} ms, zero_grad {z*1e3:.2f}, backward {b*1e3:.2f}, optimizer {o*1e3:.2f}")
sync(); t0 = time.perf_counter()
for _ in range(20): step()
sync(); print(f"unsynchronised step: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
sync(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
