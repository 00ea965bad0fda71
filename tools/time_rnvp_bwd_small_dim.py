"""RNVP forward + backward at narrow dims (50 / 64 / 100) over row counts: which gradient kernel family is faster where
(torch_mnf_amd._dispatch.rnvp_bwd_small's thresholds).  usage: [MNF_RNVP_BWD_MFMA_MIN_DIM=49] python3 tools/time_rnvp_bwd_small_dim.py"""
import sys, os, torch, warnings
sys.path.insert(0, os.getcwd())
import torch_mnf_amd as amd
warnings.simplefilter("ignore")
for dim in (50, 64, 100):
    for rows in (4096, 16384, 32768, 65536, 262144):
        f = amd.RNVP(dim, h_sizes=(50,)).to("cuda")
        x = torch.randn(rows, dim, device="cuda", requires_grad=True)
        def both():
            f.zero_grad(); y, ld = f.forward(x, seed=3); (y.sum() + ld.sum()).backward()
        both(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); both(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
        print(f"dim {dim} rows {rows}: {best*1e3:.1f} us fwd+bwd  ({amd.last_kernel()})  MIN_DIM={os.environ.get('MNF_RNVP_BWD_MFMA_MIN_DIM','128')}")
