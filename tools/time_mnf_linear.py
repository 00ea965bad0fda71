"""MNFLinear(800, 50).forward behind sample_z on 256,000 rows: the one-launch HIP path vs the reference's composition on
stock PyTorch-ROCm (same z), HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
for n_in, n_out in ((800, 50), (50, 10)):
    layer = amd.MNFLinear(n_in, n_out).to("cuda")
    x = torch.rand(R, n_in, device="cuda")
    z = 1.0 + 0.1 * torch.randn(R, n_in, device="cuda")
    real = layer.sample_z
    layer.sample_z = lambda n: (z, None)

    def stock():
        mean = (x * z) @ layer.W_mean.T + layer.b_mean
        var = x.pow(2) @ layer.W_log_var.exp().T + layer.b_log_var.exp()
        return mean + var.sqrt() * torch.randn_like(var)

    def timed(fn, n=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    with torch.no_grad():
        t_hip = timed(lambda: layer.forward(x))
        t_stock = timed(stock)
        layer.sample_z = real
        t_full = timed(lambda: layer.forward(x))
    byts = (8 * n_in + 4 * n_out) * R
    print(f"MNFLinear({n_in},{n_out}) rows {R}: forward kernel {t_hip:.1f} us = {byts / t_hip / 1e6:.2f} TB/s of x+z+out "
          f"({byts / t_hip / 8e6:.3f} of 8 TB/s); stock composition {t_stock:.1f} us; whole forward incl. sample_z {t_full:.1f} us")
