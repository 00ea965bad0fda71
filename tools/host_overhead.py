"""How much of a C2 step is host (Python + launch) time?  Enqueue time vs GPU time vs graph replay."""
import gc, sys, time
import torch
sys.path.insert(0, ".")
import torch_mnf_amd as amd

dim, rows = 64, 1 << 20
flows = [amd.AffineHalfFlow(dim, bool(i % 2)) for i in range(9)]
model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
x = torch.randn(rows, dim, device="cuda")
with torch.no_grad():
    for _ in range(30):
        model.log_prob(x, return_sum=True)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    for name, fn in (("eager", lambda: model.log_prob(x, return_sum=True)),):
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: enqueue {1e3 * (t1 - t0) / 100:.3f} ms/step, total {1e3 * (t2 - t0) / 100:.3f} ms/step")
    replay = model.graphed_log_prob(x)
    for _ in range(5):
        replay(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        replay.graph.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {1e3 * (time.perf_counter() - t0) / 100:.3f} ms/step")
    # tiny batch: pure host cost per step
    xs = x[:16].clone()
    for _ in range(10):
        model.log_prob(xs, return_sum=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        model.log_prob(xs, return_sum=True)
    torch.cuda.synchronize()
    print(f"16-row batch (host bound): {1e3 * (time.perf_counter() - t0) / 200:.3f} ms/step")
