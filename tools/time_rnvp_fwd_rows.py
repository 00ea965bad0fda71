"""RNVP forward at small and medium row counts: the few-rows kernel launched as a grid (a workgroup per two rows) against
the streaming matrix-core kernels -- where the crossover is.  `python3 tools/time_rnvp_fwd_rows.py`"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    for name, env in (("few-rows grid", {"MNF_RNVP_FEW_FWD_ROWS": "1000000"}), ("streaming", {"MNF_RNVP_FEW": "0"})):
        print(name, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), check=True)
    sys.exit(0)

import torch
import torch_mnf_amd as amd

dev = "cuda"
for dim in (800, 50):
    f = amd.RNVP(dim, h_sizes=(50,)).to(dev)
    out = []
    for rows in (2, 16, 128, 512, 1024, 2048, 4096, 16384):
        z = torch.randn(rows, dim, device=dev)
        with torch.no_grad():
            for _ in range(5):
                f.forward(z, seed=5)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(30):
                f.forward(z, seed=5)
            t1.record(); torch.cuda.synchronize()
        out.append(f"{rows}: {t0.elapsed_time(t1) / 30 * 1e3:.0f}")
    print(f"  d={dim} forward us by rows  " + "  ".join(out), flush=True)
