"""NSF_CL at dims whose halves are not whole float4 groups (the reference's dim = 2): the padded twin on the matrix-core
kernels against the any-shape kernels, forward (no_grad) and forward + backward, over row counts -- where
_dispatch.NSF_PAD_MIN_ROWS comes from.  usage: python3 tools/time_nsf_padded_twin.py [dim K n_h]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd
import torch_mnf_amd.flows as fl

warnings.simplefilter("ignore")
dim, K, n_h = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 8, 16)


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3


for rows in (1024, 4096, 8192, 16384, 65536, 262144, 1 << 20):
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h).to("cuda")
    x = torch.randn(rows, dim, device="cuda")
    xg = x.clone().requires_grad_(True)
    out = []
    for min_rows in (0, 1 << 40):
        fl._dispatch.NSF_PAD_MIN_ROWS = min_rows

        def fwd():
            with torch.no_grad():
                f.inverse(x)

        def both():
            f.zero_grad()
            y, ld = f.inverse(xg)
            (y.sum() + ld.sum()).backward()

        t_f = timed(fwd); k_f = amd.last_kernel()
        t_b = timed(both); k_b = amd.last_kernel()
        out.append(f"{k_f} {t_f:8.1f} us | {k_b} {t_b:8.1f} us")
    print(f"dim {dim} K {K} n_h {n_h} rows {rows:8d}:  twin {out[0]}    any-shape {out[1]}")
