"""C5's second workload: MNFLinear(50, 10).sample_z(256000) (two RNVP(50, (50,)) flows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
for n_in, n_out in ((50, 10), (800, 50)):
    torch.manual_seed(0)
    layer = amd.MNFLinear(n_in, n_out).to("cuda")
    with torch.no_grad():
        for _ in range(3):
            layer.sample_z(rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            layer.sample_z(rows)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    algo = 2 * (8 * n_in + 8) * rows
    print(f"MNFLinear({n_in},{n_out}).sample_z({rows}): {dt*1e3:.3f} ms -> {rows/dt:.3e} rows/s, {algo/dt/1e12:.2f} TB/s algorithmic")
