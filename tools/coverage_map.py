"""Shape -> kernel -> speed map of the three coupling layers (verdict r4 item 6): which kernel family a layer call lands
on (torch_mnf_amd.last_kernel()) and what a row costs there, forward (no_grad) and forward + backward, over the shapes a
user of the reference might pick.  The specialised kernels are narrow templates; this table is where their edges are.

usage: python3 tools/coverage_map.py [rows] > profiles/r5/coverage_map.txt"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import torch_mnf_amd as amd  # noqa: E402

warnings.simplefilter("ignore")
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
DEV = "cuda"


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e6 / ROWS  # ns per row


def probe(layer, dim, call):
    """(forward kernel, ns/row, gradient kernel, fwd+bwd ns/row) on the default path, and the same two times with the layer
    forced onto the any-shape kernels when the default path is not one of them (the cliff AT this shape)."""
    fast = probe_once(layer, dim, call)
    gen = (None, None)
    if "generic" not in fast[0] or "generic" not in fast[2]:
        layer.force_generic = True
        g = probe_once(layer, dim, call)
        layer.force_generic = False
        gen = (g[1], g[3])
    return (*fast, *gen)


def probe_once(layer, dim, call):
    layer = layer.to(DEV)
    x = torch.randn(ROWS, dim, device=DEV)

    def fwd():
        with torch.no_grad():
            call(layer, x)

    t_f = timed(fwd)
    k_f = amd.last_kernel()
    xg = x.clone().requires_grad_(True)
    k_b = [None]

    def both():
        layer.zero_grad()
        y, ld = call(layer, xg)
        (y.sum() + ld.sum()).backward()
        k_b[0] = amd.last_kernel()

    t_b = timed(both)
    return k_f, t_f, k_b[0], t_b


def main():
    torch.manual_seed(0)
    print(f"# rows per call: {ROWS}; ns per row = best of 3 launches (HIP events); 'fwd+bwd' = forward with the autograd link, "
          "sum() of both outputs, backward")
    print("# kernel = torch_mnf_amd.last_kernel() after the call; 'generic' = the same call with force_generic (the any-shape "
          "kernels), x = generic / default")
    rows = []
    for dim in (2, 8, 32, 64, 128, 256, 512):
        for hs in ((24, 24, 24), (16, 16, 16), (32, 32, 32), (64, 64, 64), (24, 24)):
            f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs)
            rows.append(("AffineHalfFlow", f"dim={dim} hidden={hs}", dim, *probe(f, dim, lambda m, x: m.inverse(x))))
    for dim in (2, 8, 16, 24, 32, 48, 64, 128):
        for K in (5, 8, 10):
            for n_h in (8, 16, 32):
                f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
                rows.append(("NSF_CL", f"dim={dim} K={K} n_h={n_h}", dim, *probe(f, dim, lambda m, x: m.inverse(x))))
    for dim in (50, 128, 784, 800, 1024, 2048):
        for h in (50, 30, 64, 100):
            f = amd.RNVP(dim, h_sizes=(h,))
            rows.append(("RNVP", f"dim={dim} hidden=({h},)", dim, *probe(f, dim, lambda m, x: m.forward(x, seed=3))))
    print(f"{'layer':15s} {'shape':30s} {'forward kernel':16s} {'ns/row':>8s} {'generic':>8s} {'x':>6s}   {'gradient kernel':20s} "
          f"{'fwd+bwd':>8s} {'generic':>8s} {'x':>6s}")
    cliffs = []
    for layer, shape, dim, kf, tf, kb, tb, gf, gb in rows:
        xf = f"{gf / tf:6.1f}" if gf else "     -"
        xb = f"{gb / tb:6.1f}" if gb else "     -"
        gfs = f"{gf:8.2f}" if gf else "       -"
        gbs = f"{gb:8.2f}" if gb else "       -"
        print(f"{layer:15s} {shape:30s} {kf:16s} {tf:8.2f} {gfs} {xf}   {kb:20s} {tb:8.2f} {gbs} {xb}")
        if gb:
            cliffs.append((gb / tb, layer, shape, kb))
    cliffs.sort(reverse=True)
    print("\n# the largest cliffs (forward + backward: the any-shape kernels against the matrix-core ones AT the same shape):")
    for fac, layer, shape, kb in cliffs[:6]:
        print(f"#   {fac:6.1f} x  {layer} {shape}  ({kb})")
    slow = sorted(((tb, layer, shape) for layer, shape, dim, kf, tf, kb, tb, gf, gb in rows if "generic" in kb), reverse=True)
    print("# shapes WITHOUT a matrix-core gradient kernel, slowest first (ns per row, forward + backward):")
    for tb, layer, shape in slow[:8]:
        print(f"#   {tb:9.1f}  {layer} {shape}")


if __name__ == "__main__":
    main()
