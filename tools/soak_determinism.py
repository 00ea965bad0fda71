"""Soak test: every configuration's pass repeated many times must reproduce its first result bit for bit
(per-row outputs; the fp64 row sum is an atomic reduction and may differ in the last bits).  Catches races in the
double-buffered LDS images / LDS-DMA hand-over that a single parity run could miss."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, recipes
import torch_mnf_amd as amd

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
torch.manual_seed(0)


def ahf_model(dim):
    flows = []
    for i, sd in enumerate(recipes.c2_stack_params(dim)):
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
    return amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(dev)


def check(name, fn, n):
    with torch.no_grad():
        ref = [t.clone() for t in fn()]
        bad = 0
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
            for a, b in zip(out, ref):
                if not torch.equal(a, b):
                    bad += 1
        torch.cuda.synchronize()
    print(f"{name}: {n} repeats, {bad} mismatching tensors, {(time.perf_counter() - t0) / n * 1e3:.3f} ms per pass")
    return bad


total = 0
for dim, rows in ((64, 1 << 20), (256, 1 << 19), (2, 4096), (50, 100000), (128, 1 << 18)):
    model = ahf_model(dim)
    x = torch.randn(rows, dim, device=dev)
    def run(model=model, x=x):
        zs, ld = model.inverse(x)
        return [zs[-1], zs[4], ld, model.log_prob(x)]
    total += check(f"9xAHF d={dim} rows={rows}", run, iters if dim != 256 else iters // 2)
layer = amd.MNFLinear(800, 50).to(dev)
eps = torch.randn(256000, 800, device=dev)
masks = [f.mask_for(11 + i, 256000) for i, f in enumerate(layer.flow_q.flows)]
total += check("MNFLinear(800,50).sample_z", lambda: list(layer.sample_z(256000, eps=eps, masks=masks)), iters // 3)
layer2 = amd.MNFLinear(50, 10).to(dev)
eps2 = torch.randn(256000, 50, device=dev)
masks2 = [f.mask_for(21 + i, 256000) for i, f in enumerate(layer2.flow_q.flows)]
total += check("MNFLinear(50,10).sample_z", lambda: list(layer2.sample_z(256000, eps=eps2, masks=masks2)), iters)
print("TOTAL mismatches:", total)

# two streams at once: workgroups of different launches share CUs, which shifts every timing the double-buffered
# LDS images and their LDS-DMA hand-over depend on
model_a, model_b = ahf_model(64), ahf_model(256)
xa, xb = torch.randn(1 << 19, 64, device=dev), torch.randn(1 << 17, 256, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    ref_a, ref_b = model_a.log_prob(xa).clone(), model_b.log_prob(xb).clone()
    torch.cuda.synchronize()
    bad2 = 0
    for _ in range(iters // 2):
        with torch.cuda.stream(sa):
            la = model_a.log_prob(xa)
        with torch.cuda.stream(sb):
            lb = model_b.log_prob(xb)
        torch.cuda.synchronize()
        bad2 += int(not torch.equal(la, ref_a)) + int(not torch.equal(lb, ref_b))
print(f"two concurrent streams (d=64 and d=256 stacks): {iters // 2} rounds, {bad2} mismatching tensors")
sys.exit(1 if (total + bad2) else 0)
