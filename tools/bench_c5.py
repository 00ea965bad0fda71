"""Config 5 timing: MNFLinear(800,50).sample_z(R) and MNFLinear(50,10).sample_z(R), R = 512*500."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import recipes
import torch_mnf_amd as amd

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
dev = "cuda"
for n_in, n_out in ((800, 50), (50, 10)):
    layer = amd.MNFLinear(n_in, n_out)
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(800 + i, n_in, 50))
    layer.to(dev)
    with torch.no_grad():
        for _ in range(2):
            z, ld = layer.sample_z(R)
        torch.cuda.synchronize()
        # kernel-only timing of one RNVP layer with a fixed mask
        f = layer.flow_q.flows[0]
        zin = torch.randn(R, n_in, device=dev)
        mask = torch.bernoulli(0.5 * torch.ones_like(zin))
        ldb = torch.zeros(R, device=dev)
        f._run(zin, False, ldb, mask)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f._run(zin, False, ldb, mask)
        e1.record(); torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            z, ld = layer.sample_z(R)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    flops = 2 * (n_in * 50 + 2 * 50 * n_in) * R
    byts = (8 * n_in + 8 + 4 * n_in) * R
    print(f"MNFLinear({n_in},{n_out}).sample_z({R}): {dt*1e3:.2f} ms -> {R/dt:.3e} rows/s | one RNVP layer kernel "
          f"{k_ms:.3f} ms = {flops/k_ms/1e9:.1f} TFLOP/s, {byts/k_ms/1e6:.0f} GB/s (float mask)")
