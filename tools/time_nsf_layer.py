"""One NSF_CL layer at 2^20 rows, forward + backward, over a few (dim, K, n_h) shapes (which kernel family ran is
printed).  usage: [MNF_LIB_PATH=...] python3 tools/time_nsf_layer.py"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_mnf_amd as amd

warnings.simplefilter("ignore")
lib = os.path.basename(os.environ.get("MNF_LIB_PATH", "default"))
for dim, K, n_h in ((32, 8, 16), (32, 8, 8), (16, 8, 16), (32, 10, 8), (32, 10, 16), (32, 5, 16)):
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h).to("cuda")
    x = torch.randn(1 << 20, dim, device="cuda", requires_grad=True)

    def both():
        f.zero_grad()
        y, ld = f.inverse(x)
        (y.sum() + ld.sum()).backward()

    both()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); both(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(f"NSF_CL dim {dim} K {K} n_h {n_h}: forward + backward {best:.3f} ms at 2^20 rows ({amd.last_kernel()}) lib={lib}")
