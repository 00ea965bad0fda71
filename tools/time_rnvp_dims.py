"""RNVP.forward (seeded mask, 256,000 rows) over the dims the streaming split kernel serves: us per call.
usage: [MNF_LIB_PATH=...] python3 tools/time_rnvp_dims.py [dims...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

dims = [int(a) for a in sys.argv[1:]] or [50, 64, 96, 112, 200, 400, 799]
rows = 256000
for dim in dims:
    f = amd.RNVP(dim, h_sizes=(50,)).to("cuda")
    z = torch.randn(rows, dim, device="cuda")
    with torch.no_grad():
        for _ in range(5):
            f.forward(z, seed=5)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(30):
            f.forward(z, seed=5)
        t1.record()
        torch.cuda.synchronize()
    print(f"d={dim:4d}: {t0.elapsed_time(t1) / 30 * 1e3:8.1f} us")
