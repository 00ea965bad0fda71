"""time per row of the run-time-shaped AffineHalfFlow kernel on a few shapes (tools/try_rt.py without the parity part)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_mnf_amd as amd
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
for dim, hs in [(64, (24, 24)), (64, (64, 64, 64)), (512, (24, 24, 24)), (512, (64, 64, 64)), (256, (32, 32, 32))]:
    f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs).to("cuda"); f.force_generic = 2
    x = torch.randn(ROWS, dim, device="cuda")
    with torch.no_grad():
        t = timed(lambda: f.forward(x))
    print(f"d={dim} h={hs}: {amd.last_kernel()} {t * 1e6 / ROWS:.3f} ns/row")
