"""Run-time-shaped gradient kernels against the float64 oracle (autograd through oracle/flow_oracle.py), then their time.
usage: python3 tools/try_rt_bwd.py [ahf|nsf|rnvp|all] [rows-for-timing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, recipes
import torch_mnf_amd as amd
from oracle import flow_oracle as O

which = sys.argv[1] if len(sys.argv) > 1 else "all"
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
DEV, bad = "cuda", 0


def nerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def oracle_grads(fn, x, sd, w_y, w_l, dt):
    xx = x.detach().to(dt).requires_grad_(True)
    p = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    y, ld = fn(xx, p)
    ((y * w_y.to(dt)).sum() + (ld * w_l.to(dt)).sum()).backward()
    return {"x": xx.grad, **{k: v.grad for k, v in p.items()}}


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best


if which in ("ahf", "all"):
    print("== AffineHalfFlow gradients, rt kernel vs float64 oracle (budget 1e-5 + 2 x dist(fp32 oracle, fp64 oracle))")
    for dim, hs, kw in [(64, (24, 24), {}), (64, (64, 64, 64), {}), (10, (16, 40), {}), (2, (24, 24), {}), (512, (24, 24, 24), {}),
                        (64, (24,), {}), (64, (20, 30, 40, 50), {}), (64, (24, 24), {"scale": False}), (64, (24, 24), {"shift": False}),
                        (256, (64, 64, 64), {})]:
        for rows in (300, 2100):
            for inverse in (False, True):
                for parity in (False, True):
                    sd = recipes.affine_half_params(31 + dim, dim, h_sizes=hs, s_last_gain=2.0, **kw)
                    x_cpu = recipes.gaussian(32 + dim, rows, dim)
                    w_y, w_l = recipes.gaussian(33, rows, dim), recipes.gaussian(34, rows, 1)[:, 0]
                    fn = lambda x, p: O.affine_half(x, p, parity, inverse, **kw)
                    g32, g64 = oracle_grads(fn, x_cpu, sd, w_y, w_l, torch.float32), oracle_grads(fn, x_cpu, sd, w_y, w_l, torch.float64)
                    f = amd.AffineHalfFlow(dim, parity, h_sizes=hs, **kw)
                    f.load_state_dict(sd); f.to(DEV); f.force_generic = 2
                    x = x_cpu.to(DEV).requires_grad_(True)
                    yg, ldg = f.forward(x, inverse=inverse)
                    ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
                    k = amd.last_kernel()
                    got = {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}
                    worst, wkey = 0.0, ""
                    for key in got:
                        budget = 1e-5 + 2 * nerr(g32[key], g64[key])
                        e = nerr(got[key], g64[key]) / budget
                        if e > worst: worst, wkey = e, key
                    flag = "" if worst <= 1.0 and k == "ahf_bwd_rt" else "   <-- FAIL"
                    if flag: bad += 1
                    if flag or (rows == 2100 and not inverse and not parity):
                        print(f"  d={dim} h={hs} {kw} rows={rows} inv={inverse} par={parity}: kernel={k} worst {worst:.2f} of budget ({wkey}){flag}")
    print("== AffineHalfFlow fwd+bwd time per row")
    for dim, hs in [(64, (24, 24)), (64, (24, 24, 24)), (64, (64, 64, 64)), (512, (24, 24, 24)), (256, (32, 32, 32)), (128, (64, 64, 64))]:
        f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs).to(DEV)
        x = torch.randn(ROWS, dim, device=DEV).requires_grad_(True)
        line = f"  d={dim} h={hs}:"
        for force in (0, 2, 1):
            f.force_generic = force
            kb = [None]
            def both():
                f.zero_grad()
                y, ld = f.forward(x)
                (y.sum() + ld.sum()).backward()
                kb[0] = amd.last_kernel()
            t = timed(both)
            line += f"  {kb[0]} {t * 1e6 / ROWS:.3f} ns/row"
        print(line)

if which in ("nsf", "all"):
    print("== NSF_CL gradients, rt kernel vs float64 oracle")
    for dim, K, n_h in [(32, 8, 8), (64, 8, 16), (128, 8, 8), (48, 5, 32), (2, 5, 8), (6, 3, 5), (50, 10, 12), (16, 16, 16), (64, 10, 32), (24, 13, 20)]:
        for rows in (300, 2100):
            for inverse in (False, True):
                sd = recipes.nsf_cl_params(51 + dim + K, dim, K, n_h)
                x_cpu = recipes.gaussian(52 + dim, rows, dim, scale=1.4)
                w_y, w_l = recipes.gaussian(53, rows, dim), recipes.gaussian(54, rows, 1)[:, 0]
                fn = lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse)
                g32, g64 = oracle_grads(fn, x_cpu, sd, w_y, w_l, torch.float32), oracle_grads(fn, x_cpu, sd, w_y, w_l, torch.float64)
                f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
                f.load_state_dict(sd); f.to(DEV); f.force_generic = 2
                x = x_cpu.to(DEV).requires_grad_(True)
                yg, ldg = (f.inverse if inverse else f.forward)(x)
                ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
                k = amd.last_kernel()
                got = {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}
                worst, wkey = 0.0, ""
                for key in got:
                    budget = 1e-5 + 2 * nerr(g32[key], g64[key])
                    e = nerr(got[key], g64[key]) / budget
                    if e > worst: worst, wkey = e, key
                flag = "" if worst <= 1.0 and k == "nsf_bwd_rt" else "   <-- FAIL"
                if flag: bad += 1
                if flag or rows == 2100:
                    print(f"  d={dim} K={K} n_h={n_h} rows={rows} inv={inverse}: kernel={k} worst {worst:.2f} of budget ({wkey}){flag}")
    print("== NSF_CL fwd+bwd time per row")
    for dim, K, n_h in [(32, 8, 8), (64, 8, 16), (128, 8, 8), (64, 8, 32), (128, 10, 32), (48, 10, 16)]:
        f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h).to(DEV)
        x = (torch.randn(ROWS, dim, device=DEV) * 1.4).requires_grad_(True)
        line = f"  d={dim} K={K} n_h={n_h}:"
        for force in (0, 2, 1):
            f.force_generic = force
            kb = [None]
            def both():
                f.zero_grad()
                y, ld = f.forward(x)
                (y.sum() + ld.sum()).backward()
                kb[0] = amd.last_kernel()
            t = timed(both, reps=2)
            line += f"  {kb[0]} {t * 1e6 / ROWS:.3f} ns/row"
        print(line)


def rnvp_sd(seed, dim, hs):
    rng = np.random.default_rng(seed)
    sd = recipes.mlp_params(rng, "net", (dim, *hs), gain=1.5)
    k = 1.5 / np.sqrt(hs[-1])
    for name in ("t", "s"):
        sd[f"{name}.weight"] = torch.from_numpy(rng.uniform(-k, k, size=(dim, hs[-1])).astype(np.float32))
        sd[f"{name}.bias"] = torch.from_numpy(rng.uniform(-k, k, size=(dim,)).astype(np.float32))
    return sd


if which in ("rnvp", "all"):
    print("== RNVP gradients, rt kernel vs float64 oracle")
    for dim, hs in [(800, (100,)), (50, (100,)), (128, (30,)), (784, (50, 40)), (50, (17,)), (37, (120,)), (256, (64, 64)), (64, (7, 9, 11))]:
        for rows in (300, 2100):
            for seeded in (False, True):
                sd = rnvp_sd(41 + dim, dim, hs)
                z_cpu = recipes.gaussian(42 + dim, rows, dim)
                w_y, w_l = recipes.gaussian(43, rows, dim), recipes.gaussian(44, rows, 1)[:, 0]
                f = amd.RNVP(dim, h_sizes=hs)
                f.load_state_dict(sd); f.to(DEV); f.force_generic = 2
                mask = f.mask_for(77, rows).cpu() if seeded else recipes.bernoulli_mask(97, rows, dim)
                fn = lambda x, p: O.rnvp(x, p, mask.to(x.dtype))
                g32, g64 = oracle_grads(fn, z_cpu, sd, w_y, w_l, torch.float32), oracle_grads(fn, z_cpu, sd, w_y, w_l, torch.float64)
                z = z_cpu.to(DEV).requires_grad_(True)
                xg, ldg = f.forward(z, seed=77) if seeded else f.forward(z, mask=mask.to(DEV))
                ((xg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
                k = amd.last_kernel()
                got = {"x": z.grad, **{n: q.grad for n, q in f.named_parameters()}}
                worst, wkey = 0.0, ""
                for key in got:
                    budget = 1e-5 + 2 * nerr(g32[key], g64[key])
                    e = nerr(got[key], g64[key]) / budget
                    if e > worst: worst, wkey = e, key
                flag = "" if worst <= 1.0 and k == "rnvp_bwd_rt" else "   <-- FAIL"
                if flag: bad += 1
                if flag or (rows == 2100 and not seeded):
                    print(f"  d={dim} h={hs} rows={rows} seeded={seeded}: kernel={k} worst {worst:.2f} of budget ({wkey}){flag}")
    print("== RNVP fwd+bwd time per row")
    for dim, hs in [(800, (100,)), (800, (50,)), (50, (100,)), (128, (100,)), (2048, (100,)), (784, (50,))]:
        f = amd.RNVP(dim, h_sizes=hs).to(DEV)
        x = torch.randn(ROWS, dim, device=DEV).requires_grad_(True)
        line = f"  d={dim} h={hs}:"
        for force in (0, 2, 1):
            f.force_generic = force
            kb = [None]
            def both():
                f.zero_grad()
                y, ld = f.forward(x, seed=3)
                (y.sum() + ld.sum()).backward()
                kb[0] = amd.last_kernel()
            t = timed(both, reps=2)
            line += f"  {kb[0]} {t * 1e6 / ROWS:.3f} ns/row"
        print(line)

print("FAILURES:", bad)
sys.exit(1 if bad else 0)
