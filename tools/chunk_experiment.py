"""Experiment: traverse the 9-layer inverse pass chunk-by-chunk (all layers over a row chunk, then
the next chunk) so that a layer reads what the previous one just wrote out of the Infinity Cache."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, recipes
import torch_mnf_amd as amd
from torch_mnf_amd import _lib
from torch_mnf_amd.flows import _stream

dim, rows = 64, 1 << 20
dev = "cuda"
flows = []
for i, sd in enumerate(recipes.c2_stack_params(dim)):
    f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f.to(dev))
x = torch.randn(rows, dim, device=dev)
bufs = [x] + [torch.empty_like(x) for _ in flows]
ld = torch.zeros(rows, device=dev)
lib = _lib.load()
packed = [f._packed(torch.device(dev, 0)) for f in flows]
splits = [f._split_image(torch.device(dev, 0)) for f in flows]

def run(chunk):
    ld.zero_()
    order = list(reversed(range(len(flows))))
    for r0 in range(0, rows, chunk):
        n = min(chunk, rows - r0)
        for li, fi in enumerate(order):
            f = flows[fi]; flat, image = packed[fi]
            src, dst = bufs[li], bufs[li + 1]
            rc = lib.mnf_affine_half(src.data_ptr() + r0 * dim * 4, dst.data_ptr() + r0 * dim * 4, ld.data_ptr() + r0 * 4, 1,
                                     flat.data_ptr(), image.data_ptr(), splits[fi].data_ptr(), n, dim, int(f.parity), 1, 3, f._hid, 1, 1, 0, _stream())
            assert rc == 0

def timed(chunk, n=10):
    run(chunk); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run(chunk)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

run(rows); ref = bufs[-1].clone(); ref_ld = ld.clone()
for rnd in range(4):
    for chunk in (rows, 1 << 19, 1 << 18, 1 << 17, 1 << 16):
        t = timed(chunk, 60)
        run(chunk)
        ok = torch.equal(bufs[-1], ref) and torch.equal(ld, ref_ld)
        print(f"chunk {chunk:8d} rows ({chunk*dim*4/2**20:6.1f} MiB/tensor): {t*1e3:.3f} ms per 9-layer pass -> {rows/t/1e6:.1f} M samples/s  same={ok}")
