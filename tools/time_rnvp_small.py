"""RNVP forward / backward latency at small row counts, per kernel choice (`python3 tools/time_rnvp_small.py`):
where the matrix-core gradient path and the register-resident forward kernel start to pay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

dev = "cuda"
for dim in (800, 50):
    f = amd.RNVP(dim, h_sizes=(50,)).to(dev)
    for rows in (1, 128, 512, 2048, 8192, 32768):
        z = torch.randn(rows, dim, device=dev, requires_grad=True)
        gx = torch.randn(rows, dim, device=dev)
        gl = torch.randn(rows, device=dev)
        res = {}
        for name, generic in (("mfma", False), ("generic", True)):
            amd._dispatch.RNVP_BWD_GENERIC = generic
            amd._dispatch.RNVP_BWD_MFMA_MIN_ROWS = 0; amd._dispatch.RNVP_BWD_MFMA_MIN_DIM = 0
            x, ld = f.forward(z, seed=5)
            for _ in range(3):
                torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
            t1.record(); torch.cuda.synchronize()
            res[name] = t0.elapsed_time(t1) / 20 * 1e3
        with torch.no_grad():
            for env in ("1", "0"):
                os.environ["MNF_RNVP_RESIDENT"] = env
                for _ in range(3): f.forward(z, seed=5)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(20): f.forward(z, seed=5)
                t1.record(); torch.cuda.synchronize()
                res["fwd_resident" if env == "1" else "fwd_streaming"] = t0.elapsed_time(t1) / 20 * 1e3
        print(f"d={dim} rows={rows}: " + "  ".join(f"{k} {v:.0f} us" for k, v in res.items()), flush=True)
