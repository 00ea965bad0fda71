"""RNVP gradients at small and medium row counts: the few-rows kernel launched as a grid (mnf_rnvp_bwd_few) against the
matrix-core / generic gradient kernels -- where the crossover is.  `python3 tools/time_rnvp_bwd_rows.py`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MNF_RNVP_FEW_BWD_ROWS", "1000000")
import torch
import torch_mnf_amd as amd
import torch_mnf_amd.flows as fl

dev = "cuda"
for dim in (800, 50, 20):
    f = amd.RNVP(dim, h_sizes=(50,)).to(dev)
    for explicit in (True, False):
        out = []
        for rows in (4, 16, 128, 512, 1024, 2048, 4096):
            z = torch.randn(rows, dim, device=dev, requires_grad=True)
            gx, gl = torch.randn(rows, dim, device=dev), torch.randn(rows, device=dev)
            mask = (torch.rand(rows, dim, device=dev) < 0.5).float() if explicit else None
            x, ld = f.forward(z, mask=mask, seed=None if explicit else 5)
            res = []
            for off in (False, True):
                fl._dispatch.RNVP_BWD_FEW_GRID_OFF = off
                for _ in range(3):
                    torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(20):
                    torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
                t1.record(); torch.cuda.synchronize()
                res.append(t0.elapsed_time(t1) / 20 * 1e3)
            out.append(f"{rows}: {res[0]:.0f}/{res[1]:.0f}")
        print(f"d={dim} {'explicit' if explicit else 'in-kernel'} mask, backward us (grid / streaming) by rows  " + "  ".join(out), flush=True)
