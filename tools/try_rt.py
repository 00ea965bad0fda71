"""Run-time-shaped kernels (csrc/mnf_rt.h) against the CPU oracle on a grid of shapes, then their time per row.
usage: python3 tools/try_rt.py [ahf|nsf|rnvp|all] [rows-for-timing]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import recipes
import torch_mnf_amd as amd
from oracle import flow_oracle as O

which = sys.argv[1] if len(sys.argv) > 1 else "all"
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
DEV = "cuda"
bad = 0


def err(a, b):
    a, b = a.detach().cpu().double(), b.double()
    fin = torch.isfinite(b)
    if not bool((torch.isfinite(a) == fin).all()):
        return float("inf")
    return float((a[fin] - b[fin]).abs().max() / b[fin].abs().max().clamp_min(1e-30))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def check(name, e, tol=1e-5):
    global bad
    flag = "" if e <= tol else "   <-- FAIL"
    if flag:
        bad += 1
    print(f"  {name}: {e:.2e}{flag}")


if which in ("ahf", "all"):
    print("== AffineHalfFlow rt vs oracle")
    for dim, hs, kw in [(64, (24, 24), {}), (64, (24, 24, 24), {}), (64, (64, 64, 64), {}), (2, (24, 24), {}), (50, (17, 30), {}),
                        (512, (24, 24, 24), {}), (512, (64, 64, 64), {}), (128, (100,), {}), (256, (200, 130, 40, 7), {}),
                        (64, (24, 24), {"scale": False}), (64, (24, 24), {"shift": False}), (1024, (32, 32), {}), (6, (5, 9), {})]:
        sd = recipes.affine_half_params(11 + dim, dim, h_sizes=hs, s_last_gain=3.0, **kw)
        for rows in (1, 37, 3000):
            x = recipes.gaussian(5 + dim + rows, rows, dim)
            for parity in (False, True):
                f = amd.AffineHalfFlow(dim, parity=parity, h_sizes=hs, **kw)
                f.load_state_dict(sd)
                f.to(DEV)
                f.force_generic = 2
                for inverse in (False, True):
                    ry, rld = O.affine_half(x, sd, parity, inverse, **kw)
                    with torch.no_grad():
                        y, ld = f.forward(x.to(DEV), inverse=inverse)
                    k = amd.last_kernel()
                    e1, e2 = err(y, ry), err(ld, rld)
                    if max(e1, e2) > 1e-5 or k != "ahf_rt":
                        check(f"d={dim} h={hs} {kw} rows={rows} par={parity} inv={inverse} kernel={k}", max(e1, e2))
    # range: big inputs, big weights
    dim, hs = 64, (24, 24)
    sd = recipes.affine_half_params(3, dim, h_sizes=hs)
    x = recipes.gaussian(4, 500, dim)
    xb = x.clone()
    xb[::7] *= 3.0e5
    sdb = {k: (v * 1e-5 if k.endswith("0.weight") else v) for k, v in sd.items()}
    for tag, xx, ss in (("big rows", xb, sdb), ("big weights", x, {k: (v * 3000 if k.endswith("2.weight") else (v / 3000 if k.endswith("4.weight") else v)) for k, v in sd.items()})):
        f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs)
        f.load_state_dict(ss)
        f.to(DEV)
        f.force_generic = 2
        ry, rld = O.affine_half(xx, ss, False, False)
        with torch.no_grad():
            y, ld = f.forward(xx.to(DEV))
        check(f"range guard, {tag}: y", err(y, ry), 2e-5)
        check(f"range guard, {tag}: ld", err(ld, rld), 2e-5)
    print("== AffineHalfFlow time per row")
    for dim, hs in [(64, (24, 24)), (64, (24, 24, 24)), (64, (64, 64, 64)), (512, (24, 24, 24)), (512, (64, 64, 64)), (256, (32, 32, 32)), (2, (24, 24)), (128, (16, 16, 16))]:
        f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs).to(DEV)
        x = torch.randn(ROWS, dim, device=DEV)
        line = f"  d={dim} h={hs}:"
        for force in (0, 2, 1):
            f.force_generic = force
            with torch.no_grad():
                t = timed(lambda: f.forward(x))
            line += f"  {amd.last_kernel()} {t * 1e6 / ROWS:.3f} ns/row"
        print(line)

if which in ("nsf", "all"):
    print("== NSF_CL rt vs oracle")
    for dim, K, n_h in [(32, 8, 8), (64, 8, 16), (128, 8, 8), (128, 5, 32), (2, 5, 8), (6, 3, 5), (50, 10, 12), (16, 16, 64), (200, 4, 16), (48, 10, 32), (128, 10, 32)]:
        sd = recipes.nsf_cl_params(21 + dim + K, dim, K, n_h)
        for rows in (1, 37, 1500):
            x = recipes.gaussian(5 + dim + rows, rows, dim, scale=1.4)
            f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
            f.load_state_dict(sd)
            f.to(DEV)
            f.force_generic = 2
            for inverse in (False, True):
                ry, rld = O.nsf_cl(x, sd, K, 3.0, inverse=inverse)
                with torch.no_grad():
                    y, ld = (f.inverse if inverse else f.forward)(x.to(DEV))
                k = amd.last_kernel()
                e1, e2 = err(y, ry), err(ld, rld)
                if max(e1, e2) > 1e-5 or k != "nsf_rt":
                    check(f"d={dim} K={K} n_h={n_h} rows={rows} inv={inverse} kernel={k} y={e1:.2e} ld", e2)
    print("== NSF_CL time per row")
    for dim, K, n_h in [(32, 8, 8), (64, 8, 8), (64, 8, 16), (128, 8, 8), (128, 10, 32), (64, 10, 16), (48, 5, 32), (2, 8, 8)]:
        f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h).to(DEV)
        x = torch.randn(ROWS, dim, device=DEV) * 1.4
        line = f"  d={dim} K={K} n_h={n_h}:"
        for force in (0, 2, 1):
            f.force_generic = force
            with torch.no_grad():
                t = timed(lambda: f.forward(x), reps=3)
            line += f"  {amd.last_kernel()} {t * 1e6 / ROWS:.3f} ns/row"
        print(line)

if which in ("rnvp", "all"):
    print("== RNVP rt vs oracle")
    for dim, hs in [(800, (100,)), (50, (100,)), (128, (30,)), (784, (50, 40)), (50, (17,)), (37, (200,)), (1024, (64, 64)), (130, (130,))]:
        sd = recipes.mlp_params(np.random.default_rng(31 + dim), "net", (dim, *hs))
        rng = np.random.default_rng(32 + dim)
        for name in ("t", "s"):
            k = 1.0 / np.sqrt(hs[-1])
            sd[f"{name}.weight"] = torch.from_numpy(rng.uniform(-k, k, size=(dim, hs[-1])).astype(np.float32))
            sd[f"{name}.bias"] = torch.from_numpy(rng.uniform(-k, k, size=(dim,)).astype(np.float32))
        for rows in (1, 37, 700):
            z = recipes.gaussian(5 + dim + rows, rows, dim)
            mask = recipes.bernoulli_mask(97, rows, dim)
            f = amd.RNVP(dim, h_sizes=hs)
            f.load_state_dict(sd)
            f.to(DEV)
            f.force_generic = 2
            ry, rld = O.rnvp(z, sd, mask)
            with torch.no_grad():
                y, ld = f.forward(z.to(DEV), mask=mask.to(DEV))
            k = amd.last_kernel()
            e1, e2 = err(y, ry), err(ld, rld)
            if max(e1, e2) > 1e-5 or k != "rnvp_rt":
                check(f"d={dim} h={hs} rows={rows} kernel={k} x={e1:.2e} ld", e2)
            m_seed = f.mask_for(77, rows)
            with torch.no_grad():
                y, ld = f.forward(z.to(DEV), seed=77)
            ry, rld = O.rnvp(z, sd, m_seed.cpu())
            e1, e2 = err(y, ry), err(ld, rld)
            if max(e1, e2) > 1e-5 or amd.last_kernel() != "rnvp_rt":
                check(f"d={dim} h={hs} rows={rows} seeded kernel={amd.last_kernel()} x={e1:.2e} ld", e2)
    print("== RNVP time per row")
    for dim, hs in [(800, (100,)), (800, (50,)), (50, (100,)), (128, (100,)), (2048, (100,)), (784, (50,)), (50, (50,))]:
        f = amd.RNVP(dim, h_sizes=hs).to(DEV)
        x = torch.randn(ROWS, dim, device=DEV)
        line = f"  d={dim} h={hs}:"
        for force in (0, 2, 1):
            f.force_generic = force
            with torch.no_grad():
                t = timed(lambda: f.forward(x, seed=5), reps=3)
            line += f"  {amd.last_kernel()} {t * 1e6 / ROWS:.3f} ns/row"
        print(line)

print("FAILURES:", bad)
sys.exit(1 if bad else 0)
