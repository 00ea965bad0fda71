"""Kernel time of one seeded RNVP(800, (50,)) launch on 256,000 rows (HIP events, mean of N after warm-up), for the
library named by MNF_LIB_PATH.  usage: [MNF_LIB_PATH=...] python tools/time_rnvp.py [rows] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch_mnf_amd import synthetic as recipes
import torch_mnf_amd as amd

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = amd.RNVP(800, h_sizes=(50,))
f.load_state_dict(recipes.rnvp_params(800, 800, 50))
f.to("cuda")
z = torch.randn(R, 800, device="cuda")
ld = torch.zeros(R, device="cuda")
with torch.no_grad():
    for _ in range(40):
        f._run(z, False, ld, seed=7)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        f._run(z, False, ld, seed=7)
        ev[i + 1].record()
    torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N))
algo = (8 * 800 + 8) * R
print(f"{os.path.basename(os.environ.get('MNF_LIB_PATH', 'libmnf_hip.so')):28s} rows {R}: median {ts[N // 2]:7.1f} us  min {ts[0]:7.1f} us "
      f"-> {algo / ts[N // 2] / 1e6:6.2f} TB/s algorithmic = {algo / ts[N // 2] / 8e6:.3f} of 8 TB/s")
