import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
dev = torch.device("cuda", 0)
model, layers = bench.build_c3(dev)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(1 << 20, 32, device=dev, generator=gen)
print("x finite:", bool(torch.isfinite(x).all()), float(x.abs().max()))
with torch.no_grad():
    for _ in range(10):
        zs, ld = model.inverse(x)
    torch.cuda.synchronize()
    for i, z in enumerate(zs):
        print(i, "finite", bool(torch.isfinite(z).all()), "absmax %.3g" % float(z.abs().max()), "n_subnormal", int(((z != 0) & (z.abs() < 1.2e-38)).sum()),
              "frac |z|>3: %.4f" % float((z.abs() > 3).float().mean()))
    model.layer_events = []
    for s in range(20):
        model.log_prob(x, return_sum=True)
    torch.cuda.synchronize()
    ev = model.layer_events; model.layer_events = None
    ms = [a.elapsed_time(b) for a, b in ev]
    print("layer 3 per step (us):", [round(v * 1e3) for v in ms[3::9]])
    print("layer 0 per step (us):", [round(v * 1e3) for v in ms[0::9]])
