"""Randomised shape fuzz of the run-time-shaped matrix-core kernels (csrc/mnf_rt.h): AffineHalfFlow (any even d <= 600, 1-4
hidden layers of widths 4..256, NICE / no-shift variants, both parities and directions), NSF_CL (any even d <= 200,
K 2..16, n_h 4..64) and RNVP (d 4..1200, 1-3 conditioner layers of widths 4..256, explicit and in-kernel masks), random
row counts -- forward results and, where the gradient kernels take the shape (AffineHalfFlow widths <= 64, RNVP <= 128),
the gradients -- with the layer forced onto them (force_generic = 2) against the same call on the VALU any-shape kernels
(force_generic = 1).  Two fp32 evaluations are compared, so a disagreement beyond the tolerance goes to a float64
referee (autograd through the oracle's formulas in float64 on the CPU): neither result may be more than twice as far from
it as the other, or 5e-5.  A disagreement that one or two rows carry alone -- a pre-activation within fp32 rounding of a
LeakyReLU kink, where two correct evaluations take different derivatives; about one draw in a hundred -- is confirmed
by repeating the case without those rows' cotangents.  (Split arithmetic carries ~22 bits per product against fp32's 24,
so of two correct kernels the run-time-shaped one is the one on the other side of the kink three times out of four.)  Not a pytest (minutes of GPU time; the file name keeps it out of the collection)
but test infrastructure -- it lives here because it uses oracle/ as the referee, which only tests may.  Exits non-zero on
a mismatch.

usage: python3 tests/fuzz_shapes_rt.py [cases] [seed]      (FUZZ_ONLY=i: case i of that run alone, for a closer look)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import recipes  # noqa: E402
import torch  # noqa: E402

import torch_mnf_amd as amd  # noqa: E402
from oracle import flow_oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)
only = int(os.environ.get("FUZZ_ONLY", "-1"))
DEV, TOL, LD_TOL, GRAD_TOL = "cuda", 1e-5, 5e-5, 1e-4


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.abs().max()), 1e-30))


def rnvp_sd(seed, dim, hs):
    r = np.random.default_rng(seed)
    sd = recipes.mlp_params(r, "net", (dim, *hs), gain=1.5)
    k = 1.5 / np.sqrt(hs[-1])
    for name in ("t", "s"):
        sd[f"{name}.weight"] = torch.from_numpy(r.uniform(-k, k, size=(dim, hs[-1])).astype(np.float32))
        sd[f"{name}.bias"] = torch.from_numpy(r.uniform(-k, k, size=(dim,)).astype(np.float32))
    return sd


def run(layer, call, x, grads, code):
    """(outputs, gradients or None, kernel names) of one call with the layer forced onto kernel family `code`."""
    layer.force_generic = code
    for p in layer.parameters():
        p.grad = None
    if not grads:
        with torch.no_grad():
            y, ld = call(layer, x)
        return (y, ld), None, (amd.last_kernel(), "")
    xg = x.clone().requires_grad_(True)
    y, ld = call(layer, xg)
    kf = amd.last_kernel()
    (y * w_y).sum().add((ld * w_l).sum()).backward()
    return (y.detach(), ld.detach()), [xg.grad] + [p.grad.clone() for p in layer.parameters()], (kf, amd.last_kernel())


bad = on_rt = with_grads = 0
for case in range(n_cases):
    r = rng.random()
    grads = bool(rng.random() < 0.5)
    if r < 0.45:
        dim = int(rng.integers(1, 301)) * 2
        n_hidden = int(rng.integers(1, 5))
        top = 64 if grads or rng.random() < 0.6 else 256
        hs = tuple(int(v) for v in rng.integers(4, top + 1, size=n_hidden))
        kw = {}
        v = rng.random()
        if v < 0.15:
            kw["scale"] = False
        elif v < 0.3:
            kw["shift"] = False
        parity, inverse, rows = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(1, 6000))
        sd = recipes.affine_half_params(int(rng.integers(1 << 30)), dim, h_sizes=hs, s_last_gain=1.5, **kw)
        layer = amd.AffineHalfFlow(dim, parity, h_sizes=hs, **kw)
        call = (lambda m, x: m.inverse(x)) if inverse else (lambda m, x: m.forward(x))
        ref = lambda x, p: O.affine_half(x, p, parity, inverse, **kw)  # noqa: E731
        want = ("ahf_rt", "ahf_bwd_rt")
        desc = f"ahf d={dim} h={hs} {kw} parity={parity} inv={inverse} rows={rows} grads={grads}"
        scale = float(rng.choice([0.1, 1.0, 3.0]))
    elif r < 0.7:
        dim, K, n_h = int(rng.integers(1, 101)) * 2, int(rng.integers(2, 17)), int(rng.integers(4, 65))
        inverse, rows = bool(rng.integers(0, 2)), int(rng.integers(1, 3000))
        sd = recipes.nsf_cl_params(int(rng.integers(1 << 30)), dim, K, n_h)
        layer = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
        call = (lambda m, x: m.inverse(x)) if inverse else (lambda m, x: m.forward(x))
        ref = lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse)  # noqa: E731
        want = ("nsf_rt", "nsf_bwd_rt")
        desc = f"nsf d={dim} K={K} n_h={n_h} inv={inverse} rows={rows} grads={grads}"
        scale = 1.5
    else:
        dim = int(rng.integers(4, 1201))
        top = 128 if grads or rng.random() < 0.6 else 256
        hs = tuple(int(v) for v in rng.integers(4, top + 1, size=int(rng.integers(1, 4))))
        rows, seed = int(rng.integers(1, 3000)), int(rng.integers(1 << 40))
        sd = rnvp_sd(int(rng.integers(1 << 30)), dim, hs)
        layer = amd.RNVP(dim, h_sizes=hs)
        if rng.random() < 0.5:
            mask = recipes.bernoulli_mask(int(rng.integers(1 << 20)), rows, dim)
            call = lambda m, z: m.forward(z, mask=mask.to(DEV))  # noqa: E731
        else:
            mask = None
            call = lambda m, z: m.forward(z, seed=seed)  # noqa: E731
        want = ("rnvp_rt", "rnvp_bwd_rt")
        desc = f"rnvp d={dim} h={hs} rows={rows} mask={'explicit' if mask is not None else 'seeded'} grads={grads}"
        scale = 1.0
    desc = f"[{case}] {desc}"
    if only >= 0 and case != only:
        continue
    torch.manual_seed(seed0 * 100003 + case)  # (per case: FUZZ_ONLY replays one with the same inputs)
    layer.load_state_dict(sd)
    layer.to(DEV)
    x = torch.randn(rows, dim, device=DEV) * scale
    w_y, w_l = torch.randn(rows, dim, device=DEV), torch.randn(rows, device=DEV)
    def compare(verbose):
        """rt vs VALU with the current cotangents; a disagreement goes to the float64 referee.  -> (ok, suspicious rows)"""
        global on_rt, with_grads
        out_r, g_r, k_r = run(layer, call, x, grads, 2)
        out_v, g_v, k_v = run(layer, call, x, grads, 1)
        if verbose:
            on_rt += int(k_r[0] == want[0])
            with_grads += int(grads and k_r[1] == want[1])
            if "generic" in k_r[0] or (grads and k_r[1] not in (want[1],) and "generic" not in k_r[1]):
                print(f"   note: forced call ran {k_r}: {desc}")
        errs = [rel(out_r[0], out_v[0]) / TOL, rel(out_r[1], out_v[1]) / LD_TOL]
        if grads:
            errs += [rel(a, b) / GRAD_TOL for a, b in zip(g_r, g_v)]
        if max(errs) <= 1.0 and all(bool(torch.isfinite(t).all()) == bool(torch.isfinite(u).all())
                                    for t, u in zip(out_r, out_v)):
            return True, []
        ref_fn = ref
        if isinstance(layer, amd.RNVP):
            m64 = (layer.mask_for(seed, rows) if mask is None else mask).cpu().double()
            ref_fn = lambda x, p: O.rnvp(x, p, m64)  # noqa: E731
        sd64 = {k: v.double().requires_grad_(grads) for k, v in sd.items()}
        x64 = x.double().cpu().requires_grad_(grads)
        y64, ld64 = ref_fn(x64, sd64)
        refs = [y64.detach(), ld64.detach()]
        if grads:
            ((y64 * w_y.double().cpu()).sum() + (ld64 * w_l.double().cpu()).sum()).backward()
            refs += [x64.grad] + [sd64[n].grad for n, _ in layer.named_parameters()]
        got_r = list(out_r) + (g_r or [])
        got_v = list(out_v) + (g_v or [])
        if not all(bool(torch.isfinite(t).all()) for t in refs):
            print("   (the draw overflows in float64 too: skipped)", desc)
            return True, []
        er = max(rel(a, b) for a, b in zip(got_r, refs))
        ev = max(rel(a, b) for a, b in zip(got_v, refs))
        fine = er <= max(2 * ev, 5e-5) and ev <= max(2 * er, 5e-5)  # (either kernel may be the one that is off)
        if fine and not verbose:
            return True, []
        print(f"   disagreement ({max(errs):.1f} x tolerance); vs float64: run-time-shaped {er:.1e}, VALU {ev:.1e}: {desc}")
        names = ["y", "log_det", "grad x"] + [n for n, _ in layer.named_parameters()]
        for n, a, b, c in zip(names, got_r, got_v, refs):
            if max(rel(a, c), rel(b, c)) > 1e-5:
                print(f"      {n}: run-time-shaped {rel(a, c):.1e}, VALU {rel(b, c):.1e}, max |float64| {float(c.abs().max()):.2e}, "
                      f"input scale {scale}")
        sus = []
        if grads:  # rows whose own gradient is off in either kernel (a row's grad x depends on that row alone)
            gmax = float(refs[2].abs().max())
            for got in (got_r[2], got_v[2]):
                off = ((got.double().cpu() - refs[2]).abs().amax(dim=1) > 2e-5 * gmax).nonzero().flatten().tolist()
                sus = sorted(set(sus) | set(off))
        return fine, sus

    ok, sus = compare(True)
    if not ok and 0 < len(sus) <= 2:
        # One or two rows carry the whole disagreement: a pre-activation within fp32 rounding of a LeakyReLU kink (or an
        # element of a spline knot), where two correct fp32 evaluations take different derivatives.  Without those rows'
        # cotangents everything must agree.
        w_y[sus] = 0
        w_l[sus] = 0
        ok, _ = compare(False)
        print(f"      rows {sus} carry it; without their cotangents: {'agreement -- a kink, not a kernel' if ok else 'STILL off'}")
    if not ok:
        bad += 1
        print("MISMATCH:", desc)
print(f"{n_cases} cases, {on_rt} forward on the run-time-shaped kernels, {with_grads} with gradients on them, {bad} mismatches")
sys.exit(1 if bad else 0)
