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
    print("# kernel = torch_mnf_amd.last_kernel() after the call (the gradient pass's name is its last launch's family)")
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
        for h in (50, 30, 64):
            f = amd.RNVP(dim, h_sizes=(h,))
            rows.append(("RNVP", f"dim={dim} hidden=({h},)", dim, *probe(f, dim, lambda m, x: m.forward(x, seed=3))))
    # ns per row per dim, and the factor against the best matrix-core shape of the same layer type
    best = {}
    for layer, _, dim, kf, tf, kb, tb in rows:
        if "generic" not in kf:
            best[(layer, "f")] = min(best.get((layer, "f"), 1e30), tf / dim)
        if "generic" not in kb:
            best[(layer, "b")] = min(best.get((layer, "b"), 1e30), tb / dim)
    print(f"{'layer':16s} {'shape':32s} {'forward kernel':18s} {'ns/row':>9s} {'x best/dim':>10s}   {'fwd+bwd kernel':20s} "
          f"{'ns/row':>9s} {'x best/dim':>10s}")
    for layer, shape, dim, kf, tf, kb, tb in rows:
        print(f"{layer:16s} {shape:32s} {kf:18s} {tf:9.2f} {tf / dim / best[(layer, 'f')]:10.1f}   {kb:20s} {tb:9.2f} "
              f"{tb / dim / best[(layer, 'b')]:10.1f}")
    worst = sorted(((tb / dim / best[(layer, 'b')], layer, shape, kb) for layer, shape, dim, kf, tf, kb, tb in rows), reverse=True)
    print("\n# the worst cliffs (fwd+bwd ns per row per dim against the layer type's best matrix-core shape):")
    for fac, layer, shape, kb in worst[:6]:
        print(f"#   {fac:6.1f} x  {layer} {shape} -> {kb}")


if __name__ == "__main__":
    main()
