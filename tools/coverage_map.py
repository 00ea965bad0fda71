"""Shape -> kernel -> speed map of the three coupling layers (verdict r4 item 6): which kernel family a layer call lands
on (torch_mnf_amd.last_kernel()) and what a row costs there, forward (no_grad) and forward + backward, over the shapes a
user of the reference might pick.  The specialised kernels are narrow templates; this table is where their edges are.

usage: python3 tools/coverage_map.py [rows] > profiles/r6/coverage_map.txt"""
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


def probe_once(layer, dim, call):
    """(forward kernel, ns/row, gradient kernel, fwd+bwd ns/row) of one layer in its current force_generic state."""
    layer.to(DEV)
    x = torch.randn(ROWS, dim, device=DEV)
    with torch.no_grad():
        tf = timed(lambda: call(layer, x))
    kf = amd.last_kernel()
    xg = x.clone().requires_grad_(True)

    def train():
        xg.grad = None
        y, ld = call(layer, xg)
        (y.sum() + ld.sum()).backward()

    tb = timed(train)
    kb = amd.last_kernel()
    for p in layer.parameters():
        p.grad = None
    return kf, tf, kb, tb


def probe(layer, dim, call):
    """Default path: (forward kernel, ns/row, gradient kernel, fwd+bwd ns/row); then the same two times with the layer
    forced onto the run-time-shaped matrix-core kernels (force_generic = 2; None where the default path is already one
    of them) and onto the VALU any-shape kernels (force_generic = 1)."""
    fast = probe_once(layer, dim, call)
    rt = (None, None)
    if not (fast[0].endswith("_rt") and fast[2].endswith("_rt")):
        layer.force_generic = 2
        g = probe_once(layer, dim, call)
        rt = (g[1] if g[0].endswith("_rt") else None, g[3] if g[2].endswith("_rt") else None)
    layer.force_generic = 1
    g = probe_once(layer, dim, call)
    layer.force_generic = False
    return (*fast, *rt, g[1], g[3])


def main():
    torch.manual_seed(0)
    print(f"# rows per call: {ROWS}; ns per row = best of 3 launches (HIP events); 'fwd+bwd' = forward with the autograd link, "
          "sum() of both outputs, backward")
    print("# kernel = torch_mnf_amd.last_kernel() after the call; rt = the same call forced onto the run-time-shaped matrix-core "
          "kernels (force_generic = 2), rt/x = rt / default; valu = forced onto the VALU any-shape kernels (force_generic = 1)")
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
    print(f"{'layer':15s} {'shape':30s} {'forward kernel':28s} {'ns/row':>8s} {'rt':>8s} {'rt/x':>5s} {'valu':>8s}   "
          f"{'gradient kernel':20s} {'fwd+bwd':>8s} {'rt':>8s} {'rt/x':>5s} {'valu':>8s}")

    def num(v, w=8):
        return f"{v:{w}.2f}" if v else " " * (w - 1) + "-"

    worst_f, worst_b, generic = [], [], []
    for layer, shape, dim, kf, tf, kb, tb, rf, rb, gf, gb in rows:
        xf = rf / tf if rf else None
        xb = rb / tb if rb else None
        print(f"{layer:15s} {shape:30s} {kf:28s} {tf:8.2f} {num(rf)} {num(xf, 5)} {num(gf)}   {kb:20s} {tb:8.2f} {num(rb)} "
              f"{num(xb, 5)} {num(gb)}")
        if xf:
            worst_f.append((xf, layer, shape, kf))
        if xb:
            worst_b.append((xb, layer, shape, kb))
        if "generic" in kf or "generic" in kb:
            generic.append((tb, layer, shape, kf, kb))
    for name, lst in (("forward", worst_f), ("forward + backward", worst_b)):
        lst.sort(reverse=True)
        print(f"\n# {name}: run-time-shaped / specialised at the same shape, worst first:")
        for fac, layer, shape, k in lst[:6]:
            print(f"#   {fac:5.2f} x  {layer} {shape}  ({k})")
    generic.sort(reverse=True)
    print(f"\n# shapes whose default path still has a VALU any-shape (*_generic) kernel: {len(generic)}")
    for tb, layer, shape, kf, kb in generic[:12]:
        print(f"#   {tb:9.1f} ns/row fwd+bwd  {layer} {shape}  ({kf} / {kb})")


if __name__ == "__main__":
    main()
