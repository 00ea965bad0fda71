"""the run-time-shaped AffineHalfFlow gradient kernel alone (C ABI call, HIP events): with and without parameter sums"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch_mnf_amd as amd
from torch_mnf_amd import _lib
lib = _lib.load()
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
SHAPES = [(64, (24, 24)), (64, (24, 24, 24)), (64, (64, 64, 64)), (512, (24, 24, 24)), (256, (32, 32, 32))]
if len(sys.argv) > 3:
    SHAPES = [(int(sys.argv[2]), tuple(int(v) for v in sys.argv[3].split(",")))]
for dim, hs in SHAPES:
    f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs).to("cuda")
    flat = torch.cat([p.detach().reshape(-1) for p in f.parameters()])
    x = torch.randn(ROWS, dim, device="cuda"); gy = torch.randn(ROWS, dim, device="cuda") / ROWS; gl = torch.randn(ROWS, device="cuda") / ROWS
    gx = torch.empty_like(x); gf = torch.zeros_like(flat); sc = torch.ones(1, device="cuda") * 262144.0
    hid = _lib.int_array(hs)
    def run(with_params):
        rc = lib.mnf_affine_half_bwd_rt(x.data_ptr(), None, gy.data_ptr(), gl.data_ptr(), gx.data_ptr(), gf.data_ptr() if with_params else None,
                                        flat.data_ptr(), sc.data_ptr(), ROWS, dim, 0, 0, len(hs), hid, 1, 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    t1, t0 = timed(lambda: run(True)), timed(lambda: run(False))
    print(f"d={dim} h={hs}: bwd_rt {t1 * 1e6 / ROWS:.3f} ns/row; grad_x only {t0 * 1e6 / ROWS:.3f} ns/row")
