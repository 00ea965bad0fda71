"""Gradients of the HIP Flow modules against torch.autograd through the CPU
oracle, and the reference's own training contracts (tests/test_flows.py:14-31,53-55,76-86)."""
import numpy as np
import pytest
import torch

import recipes
from helpers import assert_close, normwise_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
# The gradient bar: a kernel's gradient must be within GBASE (the forward bar, 1e-5 normwise) PLUS twice the fp32
# oracle's own distance from the float64 oracle of the float64 oracle's gradient -- parameter gradients are sums over
# rows, and the reference's own fp32 autograd is that far from the exact sum; a flat 2e-5 / 5e-5 / 1e-4 would not
# tell kernel error from reference rounding.  Kernel-against-kernel comparisons (no float64 run of a kernel exists)
# keep a flat bar.
GBASE = 1e-5
# ... except where a test FORCES a kernel onto a case outside its operating range to reach a code path: the split
# (f16 hi + lo) gradient kernel on batches of a few rows (production sends batches below 49,152 rows to the fp32-MFMA
# kernel: in a sum over 17 rows the format's 2^-22 per product does not average out -- measured up to 2.0e-5 from
# float64), weights of 1e3 against inputs of 1e-3 (1.8e-5 on the fp32-MFMA kernel), and the row-per-lane spline
# gradient kernel on a two-unit conditioner next to a knot (3.3e-5 against an fp32-oracle-vs-fp64 distance of 1.0e-5;
# its reciprocals, exponentials and logarithms are the hardware's 1-ulp instructions).  Those call sites say so.
GBASE_STRESS = 2e-5
GTOL = 2e-5  # kernel vs kernel only: two fp32 evaluations, each carrying its own rounding of the row sums
from helpers import GRAD_LOG  # noqa: E402  one record per OracleGrads.check call (tests/test_zz_audit.py enforces the 80 % rule)


@pytest.fixture(scope="module")
def amd():
    import torch_mnf_amd

    return torch_mnf_amd


@pytest.fixture(scope="module")
def O():
    from oracle import flow_oracle

    return flow_oracle


def leaf(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


class OracleGrads:
    """Gradients of ``loss_fn(x, params, dtype) -> scalar`` by torch.autograd through the CPU oracle, evaluated in
    float32 (what the reference computes) AND in float64 (the exact answer to fp32 accuracy).  ``check`` holds a
    kernel's gradient to ``GBASE + 2 * dist(fp32 oracle, fp64 oracle)`` of the FLOAT64 gradient."""

    def __init__(self, loss_fn, x, sd):
        self.g, self.out = {}, {}
        for dt in (torch.float32, torch.float64):
            xx = x.detach().to(dt).requires_grad_(True)
            p = {k: (v.detach().to(dt).clone().requires_grad_(True) if torch.is_floating_point(v) else v)
                 for k, v in sd.items()}
            loss = loss_fn(xx, p, dt)
            loss.backward()
            self.g[dt] = {"x": xx.grad, **{k: v.grad for k, v in p.items() if torch.is_floating_point(v)}}
            self.loss = loss.detach()
        self.keys = [k for k in self.g[torch.float32] if k != "x"]

    def ref(self, key, dt=torch.float32):
        g = self.g[dt][key]
        return None if g is None else g

    def check(self, got, key, what="", base=GBASE):
        r32, r64 = self.g[torch.float32][key], self.g[torch.float64][key]
        if r64 is None or float(r64.abs().max()) == 0.0:  # a parameter the loss does not depend on
            assert got is None or float(got.abs().max()) == 0.0, f"{what}: grad {key} should be zero"
            return 0.0
        r32n = r32.numpy() if r32 is not None else np.zeros_like(r64.numpy(), dtype=np.float32)
        widening = 2.0 * normwise_err(r32n, r64.numpy())
        err = normwise_err(got.detach().double().cpu().numpy(), r64.numpy())
        GRAD_LOG.append({"what": f"{what} grad {key}", "err": err, "budget": base + widening, "widening": widening,
                         "stress": base != GBASE})
        assert err <= base + widening, (f"{what}: grad {key} is {err:.3e} from the float64 oracle; budget "
                                        f"{base + widening:.3e} = {base:.0e} + 2 x (fp32 oracle vs fp64 oracle = "
                                        f"{widening / 2:.3e})")
        return err

    def check_all(self, got: dict, what="", base=GBASE):
        return max(self.check(got[k], k, what, base) for k in ["x", *self.keys])


def check_vs_float64(got, r32, r64, what, base=GBASE):
    """The OracleGrads.check rule for tests that build their float32 / float64 oracle runs themselves."""
    widening = 2.0 * normwise_err(r32.detach().numpy(), r64.detach().numpy())
    err = normwise_err(got.detach().double().cpu().numpy(), r64.detach().numpy())
    GRAD_LOG.append({"what": what, "err": err, "budget": base + widening, "widening": widening, "stress": base != GBASE})
    assert err <= base + widening, (f"{what}: {err:.3e} from the float64 oracle; budget {base + widening:.3e} = "
                                    f"{base:.0e} + 2 x (fp32 oracle vs fp64 oracle = {widening / 2:.3e})")
    return err


def cot_loss(fn, w_y, w_l):
    """loss_fn for OracleGrads: sum(y * w_y) + sum(log_det * w_l) of ``fn(x, params) -> (y, log_det)``."""
    def loss(x, p, dt):
        y, ld = fn(x, p)
        return (y * w_y.to(dt)).sum() + (ld * w_l.to(dt)).sum()
    return loss


@pytest.mark.parametrize("dim,kw", [(64, {}), (10, dict(h_sizes=(16, 40))), (2, {}), (10, dict(scale=False)),
                                    (10, dict(shift=False)), (64, dict(scale=False)), (256, dict(shift=False))])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_half_gradients(amd, O, dim, kw, inverse):
    sd = recipes.affine_half_params(31 + dim, dim, s_last_gain=2.0, **kw)
    rows = 300
    x_cpu = recipes.gaussian(32 + dim, rows, dim).requires_grad_(True)
    w_y = recipes.gaussian(33, rows, dim)     # random cotangents
    w_l = recipes.gaussian(34, rows, 1)[:, 0]
    flags = {k: v for k, v in kw.items() if k in ("scale", "shift")}
    ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, True, inverse, **flags), w_y, w_l), x_cpu, sd)
    y, _ = O.affine_half(x_cpu.detach(), sd, True, inverse, **flags)

    f = amd.AffineHalfFlow(dim, True, **kw)
    f.load_state_dict(sd)
    f.to(DEV)
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = f.forward(x, inverse=inverse)
    assert yg.requires_grad and ldg.requires_grad
    ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    assert_close(yg, y.detach(), 1e-5, "y")
    ref.check_all({"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}, f"ahf d={dim} {kw} inv={inverse}")


def ahf_grads(amd, sd, dim, h_sizes, parity, inverse, x_cpu, w_y, w_l, mode):
    """Gradients of one AffineHalfFlow on the GPU through the kernel `mode` names: "split" (the default: f16 split
    MFMAs), "fp32" (fp32 MFMAs, forward too) or "generic"."""
    f = amd.AffineHalfFlow(dim, parity, h_sizes=h_sizes)
    f.load_state_dict(sd)
    f.to(DEV)
    assert f._bwd_index(torch.device(DEV, 0)) is not None
    f.force_generic = mode == "generic"
    f.force_fp32_mfma = mode == "fp32"
    assert (f._bwd_split_ok() and bool(f._bwd_split_index(torch.device(DEV, 0)))) == (mode == "split")
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = f.forward(x, inverse=inverse)
    floor, amd._dispatch.BWD_SPLIT_MIN_ROWS = amd._dispatch.BWD_SPLIT_MIN_ROWS, 0  # (small batches default to the fp32 kernel)
    try:
        ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    finally:
        amd._dispatch.BWD_SPLIT_MIN_ROWS = floor
    return {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}


@pytest.mark.parametrize("dim,hid", [(64, 24), (32, 24), (64, 16), (32, 16)])
@pytest.mark.parametrize("parity", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_half_mfma_gradient_kernel(amd, O, dim, hid, parity, inverse):
    """The split-MFMA and the fp32-MFMA gradient kernels (every shape they exist for, ragged row count, both
    parities) against autograd through the oracle and against the generic gradient kernel."""
    h_sizes = (hid, hid, hid)
    sd = recipes.affine_half_params(61 + dim + hid, dim, h_sizes=h_sizes, s_last_gain=2.0)
    rows = 1000 + 7
    x_cpu = recipes.gaussian(62 + dim, rows, dim).requires_grad_(True)
    w_y = recipes.gaussian(63, rows, dim)
    w_l = recipes.gaussian(64, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, parity, inverse), w_y, w_l), x_cpu, sd)
    grads = {mode: ahf_grads(amd, sd, dim, h_sizes, parity, inverse, x_cpu, w_y, w_l, mode)
             for mode in ("split", "fp32", "generic")}
    for mode in ("split", "fp32", "generic"):
        ref.check_all(grads[mode], f"{mode} d={dim} hid={hid}")
    for mode in ("split", "fp32"):
        for k in grads[mode]:
            assert_close(grads[mode][k], grads["generic"][k], GTOL, f"{mode} vs generic {k}")


@pytest.mark.parametrize("parity", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_half_fp32_mfma_gradient_kernel_d128(amd, O, parity, inverse):
    """d = 128: the fp32-MFMA gradient kernel with two waves per workgroup (its operand images take 86 KB of LDS
    there) against autograd through the oracle and against the generic gradient kernel; no split kernel at this
    width."""
    dim, h_sizes, rows = 128, (24, 24, 24), 1000 + 7
    sd = recipes.affine_half_params(261, dim, h_sizes=h_sizes, s_last_gain=2.0)
    x_cpu = recipes.gaussian(262, rows, dim).requires_grad_(True)
    w_y = recipes.gaussian(263, rows, dim)
    w_l = recipes.gaussian(264, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, parity, inverse), w_y, w_l), x_cpu, sd)
    grads = {mode: ahf_grads(amd, sd, dim, h_sizes, parity, inverse, x_cpu, w_y, w_l, mode)
             for mode in ("fp32", "generic")}
    ref.check_all(grads["fp32"], f"fp32 d={dim} hid={h_sizes}")
    for k in grads["fp32"]:
        assert_close(grads["fp32"][k], grads["generic"][k], GTOL, f"fp32 vs generic {k}")


@pytest.mark.parametrize("dim,hid", [(2, 24), (6, 24), (30, 16), (40, 24), (100, 24), (64, (20, 7, 24)), (2, (5, 16, 9)),
                                     (128, (16, 16, 16)), (32, (1, 1, 1)), (64, 32), (32, (32, 17, 25)), (2, 32),
                                     (256, 24), (200, (24, 9, 16))])
@pytest.mark.parametrize("parity", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_half_fp32_mfma_gradient_kernel_padded_halves(amd, O, dim, hid, parity, inverse):
    """Coupling halves narrower than the kernel's tile (d = 2: the reference's half-moons model; 6, 30 -> 16 columns,
    40 -> 32, 100 -> 64) and hidden layers narrower than its 16 / 24 / 32 unit slots (any three widths <= 32; <= 24 at d > 64): zero
    operands in the padded columns and units, rows read and written under a column mask.  Against autograd through
    the oracle and against the generic kernel."""
    h_sizes, rows = (hid, hid, hid) if isinstance(hid, int) else hid, 1000 + 7
    sd = recipes.affine_half_params(281 + dim + sum(h_sizes), dim, h_sizes=h_sizes, s_last_gain=2.0)
    x_cpu = recipes.gaussian(282 + dim, rows, dim).requires_grad_(True)
    w_y = recipes.gaussian(283, rows, dim)
    w_l = recipes.gaussian(284, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, parity, inverse), w_y, w_l), x_cpu, sd)
    grads = {mode: ahf_grads(amd, sd, dim, h_sizes, parity, inverse, x_cpu, w_y, w_l, mode)
             for mode in ("fp32", "generic")}
    ref.check_all(grads["fp32"], f"fp32 d={dim} hid={h_sizes}")
    for k in grads["fp32"]:
        assert_close(grads["fp32"][k], grads["generic"][k], GTOL, f"fp32 vs generic {k}")


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("magnitude", [1e-7, 3e-3, 40.0])
def test_split_gradient_kernel_does_not_depend_on_the_gradient_scale(amd, O, inverse, magnitude):
    """Cotangents of a mean over 2^20 rows are ~1e-6, far below f16's normal range: the split kernel normalises them
    by a power of two taken from a sample of the rows (exact) and scales the results back."""
    dim, h_sizes, rows = 64, (24, 24, 24), 4096 + 300   # (rows beyond the 512-row sample too)
    sd = recipes.affine_half_params(171, dim, h_sizes=h_sizes, s_last_gain=2.0)
    x_cpu = recipes.gaussian(172, rows, dim).requires_grad_(True)
    w_y = recipes.gaussian(173, rows, dim) * magnitude
    w_l = recipes.gaussian(174, rows, 1)[:, 0] * magnitude
    w_y[4200:] *= 37.0  # larger than anything in the sample
    ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, True, inverse), w_y, w_l), x_cpu, sd)
    g = ahf_grads(amd, sd, dim, h_sizes, True, inverse, x_cpu, w_y, w_l, "split")
    ref.check_all(g, f"split, cotangents x {magnitude}")


@pytest.mark.parametrize("seed", range(8))
def test_split_gradient_kernel_seeded_fuzz(amd, O, seed):
    """Random draws over what the split gradient kernel is templated or branches on: shape, row count (ragged last
    tile, fewer tiles than waves), parity, direction, input / weight / cotangent magnitudes, absent cotangents."""
    rng = torch.Generator().manual_seed(9000 + seed)
    pick = lambda xs: xs[int(torch.randint(len(xs), (1,), generator=rng))]
    dim, hid = pick([(64, 24), (32, 24), (64, 16), (32, 16)])
    rows = pick([1, 15, 16, 17, 63, 200, 777, 2048 + 3])
    parity, inverse = pick([False, True]), pick([False, True])
    x_scale, w_scale, g_scale = pick([1e-3, 0.3, 1.0, 3.0]), pick([0.5, 1.0, 2.0]), pick([1e-8, 1e-4, 1.0, 1e3])
    which = pick(["both", "y_only", "ld_only"])
    h_sizes = (hid, hid, hid)
    sd = {k: v * (w_scale if k.endswith("weight") else 1.0)
          for k, v in recipes.affine_half_params(9100 + seed, dim, h_sizes=h_sizes, s_last_gain=1.5).items()}
    x_cpu = (recipes.gaussian(9200 + seed, rows, dim) * x_scale).requires_grad_(True)
    w_y = recipes.gaussian(9300 + seed, rows, dim) * g_scale
    w_l = recipes.gaussian(9400 + seed, rows, 1)[:, 0] * g_scale
    def fuzz_loss(x, p, dt):
        y, ld = O.affine_half(x, p, parity, inverse)
        return (y * w_y.to(dt)).sum() * (which != "ld_only") + (ld * w_l.to(dt)).sum() * (which != "y_only")

    ref = OracleGrads(fuzz_loss, x_cpu, sd)
    f = amd.AffineHalfFlow(dim, parity, h_sizes=h_sizes)
    f.load_state_dict(sd)
    f.to(DEV)
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = f.forward(x, inverse=inverse)
    floor, amd._dispatch.BWD_SPLIT_MIN_ROWS = amd._dispatch.BWD_SPLIT_MIN_ROWS, 0
    try:
        terms = ([(yg * w_y.to(DEV)).sum()] if which != "ld_only" else []) + \
                ([(ldg * w_l.to(DEV)).sum()] if which != "y_only" else [])
        sum(terms).backward()
    finally:
        amd._dispatch.BWD_SPLIT_MIN_ROWS = floor
    what = f"d={dim} hid={hid} rows={rows} parity={parity} inverse={inverse} x*{x_scale} w*{w_scale} g*{g_scale} {which}"
    # (rows < 49,152 never reach the split kernel in production: the few-row draws are here for the ragged tiles)
    ref.check_all({"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}, what,
                  base=GBASE if rows >= 200 else GBASE_STRESS)


@pytest.mark.parametrize("case", ["big_rows", "big_gradients", "big_weights"])
def test_split_gradient_kernel_range_guard(amd, O, case):
    """16-row tiles with an operand outside the split range (inputs, activations or deltas) are handed to the fp32
    kernel (fix-up pass over the listed tiles); a layer whose weights exceed the weight limit runs there entirely."""
    dim, h_sizes, rows = 64, (24, 24, 24), 16 * 40 + 5
    sd = recipes.affine_half_params(181, dim, h_sizes=h_sizes, s_last_gain=1.0)
    x = recipes.gaussian(182, rows, dim)
    w_y = recipes.gaussian(183, rows, dim)
    w_l = recipes.gaussian(184, rows, 1)[:, 0]
    if case == "big_rows":
        x = x.clone()
        x[37] *= 3e4
        x[300:320] *= 1e5
        sd = {k: (v * 1e-5 if k.endswith("_net.0.weight") else v) for k, v in sd.items()}
    elif case == "big_gradients":
        w_y = w_y.clone()
        w_y[5] *= 1e6      # in the sample: sets the scale; the other rows become small, not wrong
        w_y[4000 % rows] *= 1e5
    elif case == "big_weights":
        sd = {k: (v * 1e3 if k == "s_net.0.weight" else v) for k, v in sd.items()}
        x = x * 1e-3
    x_cpu = x.requires_grad_(True)
    ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, False, False), w_y, w_l), x_cpu, sd)
    g = ahf_grads(amd, sd, dim, h_sizes, False, False, x_cpu, w_y, w_l, "split")
    ref.check_all(g, f"range guard {case}", base=GBASE_STRESS if case == "big_weights" else GBASE)


@pytest.mark.parametrize("dim", [64, 2, 10, 256])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_run_is_one_autograd_node(amd, O, inverse, dim):
    """With gradients wanted, a run of equal AffineHalfFlow layers is one autograd node (stack kernel forward;
    backward per layer on the saved intermediates: MFMA gradient kernel at d = 64, the generic one for the narrow
    halves of d = 2 / 10 and for d = 256): same gradients as layer-by-layer autograd and as autograd through the
    oracle, including a loss term on an intermediate tensor."""
    n, rows = 4, 777
    sds = [recipes.affine_half_params(71 + i, dim, s_last_gain=1.5) for i in range(n)]
    x_cpu = recipes.gaussian(72, rows, dim).requires_grad_(True)
    w_z, w_mid, w_l = recipes.gaussian(73, rows, dim), recipes.gaussian(74, rows, dim), recipes.gaussian(75, rows, 1)[:, 0]
    ps = [leaf(sd) for sd in sds]
    z, ld, mids = x_cpu, 0, []
    for i in (reversed(range(n)) if inverse else range(n)):
        z, l1 = O.affine_half(z, ps[i], bool(i % 2), inverse)
        ld = ld + l1
        mids.append(z)
    ((z * w_z).sum() + (mids[1] * w_mid).sum() + (ld * w_l).sum()).backward()

    results = {}
    for fused in (True, False):
        flows = []
        for i, sd in enumerate(sds):
            f = amd.AffineHalfFlow(dim, bool(i % 2))
            f.load_state_dict(sd)
            flows.append(f)
        model = amd.NormalizingFlow(flows).to(DEV)
        model.fuse_affine_runs = fused
        x = x_cpu.detach().to(DEV).requires_grad_(True)
        zs, ldg = model.inverse(x) if inverse else model.forward(x)
        assert len(zs) == n + 1 and ldg.requires_grad
        if fused:
            assert type(zs[-1].grad_fn).__name__.startswith("_AffineRunFn")
        ((zs[-1] * w_z.to(DEV)).sum() + (zs[2] * w_mid.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
        results[fused] = {"x": x.grad, **{k: q.grad for k, q in model.named_parameters()}}
        assert_close(zs[-1], z.detach(), 1e-5, "z")
    assert_close(results[True]["x"], x_cpu.grad, GTOL, "grad_x vs oracle")
    for i in range(n):
        for k, v in ps[i].items():
            assert_close(results[True][f"flows.{i}.{k}"], v.grad, GTOL, f"grad flows.{i}.{k} vs oracle")
    for k in results[True]:
        assert_close(results[True][k], results[False][k], GTOL, f"fused vs layer-by-layer {k}")


@pytest.mark.parametrize("cfg", [(32, 8, 8), (6, 5, 8), (2, 8, 16)])
@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_gradients(amd, O, cfg, inverse):
    dim, K, n_h = cfg
    sd = recipes.nsf_cl_params(71 + dim, dim, K, n_h)
    rows = 200
    x_cpu = recipes.gaussian(72 + dim, rows, dim, scale=1.3)
    x_cpu[0, :] = 5.0          # a row in the identity tails: gradient 1, no parameter gradient
    x_cpu.requires_grad_(True)
    w_y = recipes.gaussian(73, rows, dim)
    w_l = recipes.gaussian(74, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)

    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f.to(DEV)
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = (f.inverse if inverse else f.forward)(x)
    assert yg.requires_grad and ldg.requires_grad
    ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    ref.check_all({"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}, f"nsf_cl {cfg} inv={inverse}")


def nsf_grads(amd, sd, K, n_h, inverse, x_cpu, w_y, w_l, generic):
    f = amd.NSF_CL(32, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f.to(DEV)
    f.force_generic = generic
    lib = amd._lib.load()
    assert lib.mnf_nsf_cl_bwd_tile_supported(32, K, 3, f._hid) == 1
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = (f.inverse if inverse else f.forward)(x)
    ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    return {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}


@pytest.fixture(params=["tile"])
def nsf_rows_kernel(request):
    """The NSF_CL gradient kernel with matrix-core sums: the tile kernel (the conditioner as split MFMAs in both
    directions, 16 rows per wave).  (Rounds 3 and 4 had two lane-per-element kernels here; their measurements are in
    profiles/r4 and profiles/r5/README.md.)"""
    return request.param


@pytest.mark.parametrize("K,n_h", [(8, 8), (5, 8), (8, 6), (5, 3)])
@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_row_gradient_kernel(amd, O, K, n_h, inverse, nsf_rows_kernel):
    """The tile NSF_CL gradient kernel at d = 32 against autograd through the oracle and against the generic gradient
    kernel: ragged row count, rows in the identity tails, elements exactly on the tail bound, hidden widths below the
    kernel's 8 units."""
    rows = 1003
    sd = recipes.nsf_cl_params(371 + K + n_h, 32, K, n_h)
    x_cpu = recipes.gaussian(372 + K, rows, 32, scale=1.3)
    x_cpu[0, :] = 5.0
    x_cpu[1, ::2] = 3.0
    x_cpu[2, 1::2] = -3.0
    x_cpu.requires_grad_(True)
    w_y = recipes.gaussian(373, rows, 32)
    w_l = recipes.gaussian(374, rows, 1)[:, 0]
    oracle = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)
    got = nsf_grads(amd, sd, K, n_h, inverse, x_cpu, w_y, w_l, generic=False)
    ref = nsf_grads(amd, sd, K, n_h, inverse, x_cpu, w_y, w_l, generic=True)
    oracle.check_all(got, f"nsf rows kernel ({nsf_rows_kernel}) K={K} n_h={n_h} inv={inverse}")
    oracle.check_all(ref, f"nsf generic kernel K={K} n_h={n_h} inv={inverse}")
    for k in got:  # kernel vs kernel (flat bar: two fp32 evaluations with different summation orders)
        assert_close(got[k], ref[k], 5e-5, f"rows vs generic {k}")


@pytest.mark.parametrize("seed", range(8))
def test_nsf_cl_row_gradient_kernel_seeded_fuzz(amd, O, seed, nsf_rows_kernel):
    """Random row counts (1 ... 300: partial tiles, fewer tiles than waves), K, hidden widths, directions and input
    scales against autograd through the oracle."""
    rng = np.random.default_rng(7000 + seed)
    K, n_h = int(rng.choice([5, 8])), int(rng.integers(1, 9))
    rows, inverse, scale = int(rng.integers(1, 301)), bool(rng.integers(0, 2)), float(rng.choice([0.3, 1.0, 2.5]))
    sd = recipes.nsf_cl_params(7100 + seed, 32, K, n_h)
    x_cpu = recipes.gaussian(7200 + seed, rows, 32, scale=scale).requires_grad_(True)
    w_y = recipes.gaussian(7300 + seed, rows, 32)
    w_l = recipes.gaussian(7400 + seed, rows, 1)[:, 0]
    oracle = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)
    got = nsf_grads(amd, sd, K, n_h, inverse, x_cpu, w_y, w_l, generic=False)
    oracle.check_all(got, f"nsf rows fuzz seed {seed}: K={K} n_h={n_h} rows={rows} inverse={inverse} scale={scale}",
                     base=GBASE if n_h >= 4 else GBASE_STRESS)


@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_row_gradient_kernel_many_rows(amd, inverse, nsf_rows_kernel):
    """Enough rows for several trips of every wave of the persistent grid (and the one-trip-ahead row prefetch).  The
    batch is 70 copies of a 1,003-row batch: every row's gradient must equal the single batch's (row arithmetic
    does not depend on where the row sits) and the parameter gradients must be 70 times the single batch's.
    (A large random batch cannot be compared kernel against kernel: among ~10^6 hidden units some pre-activation
    lands within rounding of the LeakyReLU kink and the two kernels legitimately take different one-sided
    derivatives -- tests/tool_nsf_grad_outliers.py shows such rows.)"""
    rows, copies, K, n_h = 1003, 70, 8, 8
    sd = recipes.nsf_cl_params(391, 32, K, n_h)
    x_cpu = recipes.gaussian(392, rows, 32, scale=1.3)
    w_y = recipes.gaussian(383, rows, 32)
    w_l = recipes.gaussian(384, rows, 1)[:, 0]
    one = nsf_grads(amd, sd, K, n_h, inverse, x_cpu, w_y, w_l, generic=False)
    many = nsf_grads(amd, sd, K, n_h, inverse, x_cpu.repeat(copies, 1), w_y.repeat(copies, 1), w_l.repeat(copies),
                     generic=False)
    assert_close(many["x"], one["x"].repeat(copies, 1), 1e-6, "grad_x of the copies")
    for k in one:
        if k != "x":
            assert_close(many[k], copies * one[k], GTOL, f"{k}: {copies} copies")


def nsf_grads_dim(amd, sd, dim, K, n_h, inverse, x_cpu, w_y, w_l, generic=False):
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f.to(DEV)
    f.force_generic = generic
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = (f.inverse if inverse else f.forward)(x)
    ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    return {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}


@pytest.mark.parametrize("dim,K,n_h", [(16, 8, 8), (24, 5, 4), (8, 8, 8), (32, 8, 7), (64, 8, 8), (64, 5, 16), (48, 8, 8),
                                       (32, 8, 16), (32, 5, 12), (16, 8, 16), (64, 8, 16), (32, 10, 8), (32, 10, 16), (16, 10, 6)])
@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_tile_gradient_kernel_dims(amd, O, dim, K, n_h, inverse):
    """The tile gradient kernel away from d = 32, n_h = 8: halves narrower than its 16- or 32-element tile (whole float4
    groups of a lane are dead), d = 64 (two float4 groups per lane, eight slots, one wave per SIMD with the sums in
    accumulator registers), hidden widths up to 16 (no spare column for the bias sums: one-hot bias tiles) -- against
    autograd through the oracle and against the generic kernel."""
    lib = amd._lib.load()
    assert lib.mnf_nsf_cl_bwd_tile_supported(dim, K, 3, amd._lib.int_array((n_h,) * 3)) == 1
    rows = 777
    sd = recipes.nsf_cl_params(4100 + dim + K, dim, K, n_h)
    x_cpu = recipes.gaussian(4200 + dim, rows, dim, scale=1.3)
    x_cpu[0, :] = 5.0
    x_cpu[1, ::2] = 3.0
    x_cpu.requires_grad_(True)
    w_y = recipes.gaussian(4300, rows, dim)
    w_l = recipes.gaussian(4400, rows, 1)[:, 0]
    oracle = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)
    # the forward kernel on the same shape (a half narrower than its tile: dead float4 groups), against the oracle
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f.to(DEV)
    with torch.no_grad():
        y_k, ld_k = (f.inverse if inverse else f.forward)(x_cpu.detach().to(DEV))
        assert "generic" not in amd.last_kernel(), amd.last_kernel()
        y_o, ld_o = O.nsf_cl(x_cpu.detach(), sd, K, 3.0, inverse)
    assert_close(y_k, y_o, 1e-5, f"nsf forward kernel d={dim} y")
    assert_close(ld_k, ld_o, 2e-5, f"nsf forward kernel d={dim} log_det")
    got = nsf_grads_dim(amd, sd, dim, K, n_h, inverse, x_cpu, w_y, w_l)
    assert amd.last_kernel() == "nsf_bwd_tile", amd.last_kernel()
    ref = nsf_grads_dim(amd, sd, dim, K, n_h, inverse, x_cpu, w_y, w_l, generic=True)
    oracle.check_all(got, f"nsf tile kernel d={dim} K={K} n_h={n_h} inv={inverse}")
    for k in got:
        assert_close(got[k], ref[k], 5e-5, f"tile vs generic {k}")


@pytest.mark.parametrize("dim,K,n_h", [(2, 8, 16), (6, 5, 8), (10, 8, 8), (30, 8, 16), (4, 8, 4), (2, 10, 8)])
@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_padded_twin_matches_the_layer(amd, O, dim, K, n_h, inverse, monkeypatch):
    """Halves that are not whole float4 groups -- dim = 2 is the reference's own NSF_CL shape (tests/test_flows.py:89-99) --
    run the matrix-core kernels on a padded twin (NSF_CL._run_padded): padded columns sit beyond the tail bound, where the
    spline is the identity with log-derivative 0, and meet zero weights in the conditioners.  Outputs, log_det and every
    gradient against the oracle and against the any-shape kernels on the unpadded layer; then a parameter update must
    reach the twin."""
    import torch_mnf_amd.flows as fl

    monkeypatch.setattr(fl._dispatch, "NSF_PAD_MIN_ROWS", 0)
    rows = 531
    sd = recipes.nsf_cl_params(4700 + dim + K, dim, K, n_h)
    x_cpu = recipes.gaussian(4800 + dim, rows, dim, scale=1.3)
    x_cpu[0, :] = 5.0      # outside the tail bound: identity
    x_cpu[1, ::2] = -3.0   # on the bound
    x_cpu.requires_grad_(True)
    w_y = recipes.gaussian(4900, rows, dim)
    w_l = recipes.gaussian(4950, rows, 1)[:, 0]
    oracle = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f.to(DEV)
    assert f._pad_half() == (dim // 2 + 3) // 4 * 4
    with torch.no_grad():
        y_k, ld_k = (f.inverse if inverse else f.forward)(x_cpu.detach().to(DEV))
        assert amd.last_kernel().startswith("nsf_mfma"), amd.last_kernel()
        y_o, ld_o = O.nsf_cl(x_cpu.detach(), sd, K, 3.0, inverse)
    assert y_k.shape == (rows, dim)
    assert_close(y_k, y_o, 1e-5, f"nsf padded twin d={dim} y")
    assert_close(ld_k, ld_o, 2e-5, f"nsf padded twin d={dim} log_det")
    got = nsf_grads_dim(amd, sd, dim, K, n_h, inverse, x_cpu, w_y, w_l)
    assert amd.last_kernel() == "nsf_bwd_tile", amd.last_kernel()
    ref = nsf_grads_dim(amd, sd, dim, K, n_h, inverse, x_cpu, w_y, w_l, generic=True)
    assert set(got) == set(ref) and all(got[k].shape == ref[k].shape for k in got)
    oracle.check_all(got, f"nsf padded twin d={dim} K={K} n_h={n_h} inv={inverse}")
    for k in got:
        assert_close(got[k], ref[k], 5e-5, f"padded twin vs generic {k}")
    # an in-place parameter update (an optimizer step) must reach the twin's operand images
    with torch.no_grad():
        for q in f.parameters():
            q.mul_(0.5)
        sd2 = {k: v.detach().cpu().clone() for k, v in f.state_dict().items()}
        y2, ld2 = (f.inverse if inverse else f.forward)(x_cpu.detach().to(DEV))
        y2_o, ld2_o = O.nsf_cl(x_cpu.detach(), sd2, K, 3.0, inverse)
    assert_close(y2, y2_o, 1e-5, "after the update: y")
    assert_close(ld2, ld2_o, 2e-5, "after the update: log_det")


@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_tile_gradient_kernel_cold_tiles(amd, O, inverse):
    """Rows whose conditioner operands leave the split range (|x| >= 2^13: identity tails for the spline, but the other
    half's net sees them) send their 16-row tile to the fix-up pass: the tile kernel must add nothing for those tiles
    and the generic kernel everything."""
    rows, K, n_h = 1003, 8, 8
    sd = recipes.nsf_cl_params(4500, 32, K, n_h)
    x_cpu = recipes.gaussian(4501, rows, 32, scale=1.3)
    x_cpu[40, :16] = 2.0e4    # lower half: the first net's input (forward) / the second step's (inverse)
    x_cpu[333, 16:] = -1.5e4  # upper half
    x_cpu[1002, 3] = 9.0e3    # the ragged last tile
    x_cpu.requires_grad_(True)
    w_y = recipes.gaussian(4502, rows, 32)
    w_l = recipes.gaussian(4503, rows, 1)[:, 0]
    oracle = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)
    got = nsf_grads(amd, sd, K, n_h, inverse, x_cpu, w_y, w_l, generic=False)
    oracle.check_all(got, f"nsf tile kernel, cold tiles, inv={inverse}", base=GBASE_STRESS)


def test_nsf_cl_tile_gradient_kernel_poisoned_launch(amd, O):
    """A cotangent far above the scale the sampled rows set (row 900 of 1,003: the sample is the first 512) overflows
    f16 as a gradient operand: the launch is poisoned, the reduction adds nothing and the fix-up pass does every row."""
    rows, K, n_h = 1003, 8, 8
    sd = recipes.nsf_cl_params(4600, 32, K, n_h)
    x_cpu = recipes.gaussian(4601, rows, 32, scale=1.3).requires_grad_(True)
    w_y = recipes.gaussian(4602, rows, 32)
    w_l = recipes.gaussian(4603, rows, 1)[:, 0]
    w_l[900] = 3.0e7
    oracle = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, True), w_y, w_l), x_cpu, sd)
    got = nsf_grads(amd, sd, K, n_h, True, x_cpu, w_y, w_l, generic=False)
    oracle.check_all(got, "nsf tile kernel, poisoned launch", base=GBASE_STRESS)


def test_nsf_cl_tile_gradient_kernel_repeats_bit_for_bit(amd):
    """No float atomics: two runs of the same gradient pass give identical bits (fixed-order two-stage reduction)."""
    rows, K, n_h = 40000, 8, 8
    sd = recipes.nsf_cl_params(4700, 32, K, n_h)
    x_cpu = recipes.gaussian(4701, rows, 32, scale=1.3)
    w_y = recipes.gaussian(4702, rows, 32)
    w_l = recipes.gaussian(4703, rows, 1)[:, 0]
    a = nsf_grads(amd, sd, K, n_h, True, x_cpu, w_y, w_l, generic=False)
    b = nsf_grads(amd, sd, K, n_h, True, x_cpu, w_y, w_l, generic=False)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("dim", [50, 800])
def test_rnvp_gradients(amd, O, dim):
    sd = recipes.rnvp_params(81 + dim, dim, 50)
    rows = 70
    z_cpu = recipes.gaussian(82 + dim, rows, dim).requires_grad_(True)
    mask = recipes.bernoulli_mask(83, rows, dim)
    w_x = recipes.gaussian(84, rows, dim)
    w_l = recipes.gaussian(85, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(lambda z, p: O.rnvp(z, p, mask.to(z.dtype)), w_x, w_l), z_cpu, sd)

    f = amd.RNVP(dim, h_sizes=(50,))
    f.load_state_dict(sd)
    f.to(DEV)
    z = z_cpu.detach().to(DEV).requires_grad_(True)
    xg, ldg = f.forward(z, mask=mask.to(DEV))
    ((xg * w_x.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    ref.check_all({"x": z.grad, **{n: q.grad for n, q in f.named_parameters()}}, f"rnvp d={dim}")
    # seeded call: backward regenerates the same mask
    z2 = z_cpu.detach().to(DEV).requires_grad_(True)
    f.zero_grad()
    x2, ld2 = f.forward(z2, seed=77)
    (x2.sum() + ld2.sum()).backward()
    m77 = f.mask_for(77, rows)
    z3 = z_cpu.detach().to(DEV).requires_grad_(True)
    g_seeded = {n: q.grad.clone() for n, q in f.named_parameters()}
    f.zero_grad()
    x3, ld3 = f.forward(z3, mask=m77)
    (x3.sum() + ld3.sum()).backward()
    assert_close(z2.grad, z3.grad, 1e-6, "seeded grad_z")
    for n, q in f.named_parameters():
        assert_close(g_seeded[n], q.grad, 2e-5, f"seeded grad {n}")


@pytest.mark.parametrize("kind", ["affine_half", "nsf_cl", "actnorm", "glow"])
@pytest.mark.parametrize("inverse", [False, True])
def test_log_det_equals_log_abs_det_of_the_jacobian(amd, kind, inverse):
    """log_det returned by a layer == log|det dy/dx|, with the Jacobian assembled row by row from
    the HIP backward kernels (d backward passes on tiny d) -- ties values and gradients together."""
    dim, rows = 6, 5
    if kind == "affine_half":
        f = amd.AffineHalfFlow(dim, True)
        f.load_state_dict(recipes.affine_half_params(101, dim, s_last_gain=3.0))
    elif kind == "nsf_cl":
        f = amd.NSF_CL(dim, K=5, B=3, n_h=8)
        f.load_state_dict(recipes.nsf_cl_params(102, dim, 5, 8))
    elif kind == "actnorm":
        f = amd.ActNormFlow(dim)
        f.load_state_dict(recipes.actnorm_params(103, dim))
        f.data_dep_init_done = True
    else:
        f = amd.Glow(dim)
        gp = recipes.glow_params(104, dim)
        f.P = gp["P"]
        f.load_state_dict({k: gp[k] for k in "LSU"})
    f.to(DEV)
    x = recipes.gaussian(105, rows, dim, scale=1.2).to(DEV).requires_grad_(True)
    y, ld = (f.inverse if inverse else f.forward)(x)
    J = torch.zeros(rows, dim, dim, device=DEV)
    for i in range(dim):
        (g,) = torch.autograd.grad(y[:, i].sum(), x, retain_graph=True)
        J[:, i, :] = g  # rows are independent, so the batch Jacobian is block diagonal
    sign, logabs = torch.linalg.slogdet(J.double())
    ld_rows = ld.detach().double().expand(rows) if ld.dim() <= 1 and ld.numel() == 1 else ld.detach().double()
    assert float((logabs - ld_rows).abs().max()) <= 2e-5 * max(1.0, float(logabs.abs().max()))


@pytest.mark.parametrize("dim,rows", [(32, 100003), (32, 5), (50, 30001), (6, 70001), (300, 777)])
@pytest.mark.parametrize("inverse", [False, True])
def test_actnorm_and_glow_gradients_many_rows(amd, O, dim, rows, inverse):
    """ActNorm (one-pass kernel: grad_x and both column sums) and Glow (x^T g; at d = 32 on the matrix cores) against
    autograd through the oracle, float64: ragged row counts, widths that do and do not divide the workgroup."""
    ap, gp = recipes.actnorm_params(503 + dim, dim), recipes.glow_params(504 + dim, dim)
    x_cpu = recipes.gaussian(505 + dim, rows, dim, scale=1.2)
    w_y = recipes.gaussian(506, rows, dim)
    an, gl = amd.ActNormFlow(dim), amd.Glow(dim)
    an.load_state_dict(ap)
    an.data_dep_init_done = True
    gl.P = gp["P"]
    gl.load_state_dict({k: gp[k] for k in "LSU"})
    an.to(DEV), gl.to(DEV)
    x = x_cpu.to(DEV).requires_grad_(True)
    h, ld1 = (an.inverse if inverse else an.forward)(x)
    y, ld2 = (gl.inverse if inverse else gl.forward)(h)
    ((y * w_y.to(DEV)).sum() + (ld1.sum() + ld2.sum()) * rows).backward()

    def loss_fn(xc, p, dt):
        hc, l1 = O.affine_const(xc, p["an.s"], p["an.t"], inverse)
        yc, l2 = O.glow(hc, gp["P"].to(dt), p["gl.L"], p["gl.S"], p["gl.U"], inverse)
        return (yc * w_y.to(dt)).sum() + (l1.sum() + l2.sum()) * rows

    ref = OracleGrads(loss_fn, x_cpu, {**{f"an.{k}": v for k, v in ap.items()}, **{f"gl.{k}": gp[k] for k in "LSU"}})
    ref.check_all({"x": x.grad, **{f"an.{k}": getattr(an, k).grad for k in ap}, **{f"gl.{k}": getattr(gl, k).grad for k in "LSU"}},
                  f"actnorm+glow d={dim} rows={rows} inv={inverse}")


def test_mnf_linear_kl_and_forward_are_differentiable(amd):
    """The MNF caller trains: gradients reach q0, the RNVP flows and the weights."""
    torch.manual_seed(1)
    layer = amd.MNFLinear(32, 10, h_sizes=(50,)).to(DEV)
    xin = torch.randn(64, 32, device=DEV)
    loss = layer.forward(xin).pow(2).mean() + 1e-3 * layer.kl_div()
    loss.backward()
    for name, prm in layer.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
    assert float(layer.flow_q.flows[0].s.weight.grad.abs().sum()) > 0
    assert float(layer.flow_r.flows[0].net[0].weight.grad.abs().sum()) > 0


def test_stack_gradients_through_normalizing_flow(amd, O):
    """NLL of a 3-layer stack + ActNorm + Glow: d loss / d parameters vs the oracle's autograd."""
    dim, rows = 8, 257
    x_cpu = recipes.gaussian(50, rows, dim)
    specs, mods = [], []
    for i in range(2):
        an = recipes.actnorm_params(60 + i, dim)
        gl = recipes.glow_params(62 + i, dim)
        ah = recipes.affine_half_params(64 + i, dim, s_last_gain=2.0)
        specs += [{"kind": "affine_const", "params": leaf(an)},
                  {"kind": "glow", "params": {**{k: v.clone().requires_grad_(True) for k, v in gl.items() if k != "P"},
                                              "P": gl["P"]}},
                  {"kind": "affine_half", "parity": bool(i % 2), "params": leaf(ah)}]
        m_an = amd.ActNormFlow(dim); m_an.load_state_dict(an); m_an.data_dep_init_done = True
        m_gl = amd.Glow(dim); m_gl.P = gl["P"]; m_gl.load_state_dict({k: gl[k] for k in "LSU"})
        m_ah = amd.AffineHalfFlow(dim, bool(i % 2)); m_ah.load_state_dict(ah)
        mods += [m_an, m_gl, m_ah]
    zs, ld = O.flow_stack(x_cpu, specs, inverse=True)
    loss_ref = -(ld + O.std_normal_log_prob(zs[-1])).sum()
    loss_ref.backward()
    # the same stack in float64: what the fp32 oracle's own gradients are worth
    specs64 = [{**sp, "params": {k: (v.detach().double().requires_grad_(True) if v.requires_grad else v.double())
                                 for k, v in sp["params"].items()}} for sp in specs]
    zs64, ld64 = O.flow_stack(x_cpu.double(), specs64, inverse=True)
    (-(ld64 + O.std_normal_log_prob(zs64[-1])).sum()).backward()

    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), mods).to(DEV)
    lp = model.log_prob(x_cpu.to(DEV))
    loss = -lp.sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-5 * abs(float(loss_ref.detach()))
    for i, (spec, mod) in enumerate(zip(specs, mods)):
        for name, prm in mod.named_parameters():
            check_vs_float64(prm.grad, spec["params"][name].grad, specs64[i]["params"][name].grad,
                             f"stack layer {i} grad {name}")


@pytest.mark.parametrize("flat_home", [False, True])
def test_affine_run_backward_on_the_split_gradient_kernel(amd, O, flat_home):
    """The 9-layer d = 64 stack as one autograd node, mean-log-prob loss (cotangents ~1/rows): the backward pass runs
    the split gradient kernel layer by layer with one gradient scale for the run, one batched operand repack and a
    fix-up list per layer.  Parameter gradients against the oracle's autograd and against the fp32 gradient kernel;
    with train.FlatParameters the kernels add into the flat gradient buffer in place."""
    dim, rows = 64, 3000 + 5
    sds = recipes.c2_stack_params(dim)
    x_cpu = recipes.gaussian(501, rows, dim)
    specs = [{"kind": "affine_half", "parity": bool(i % 2), "params": leaf(sd)} for i, sd in enumerate(sds)]
    zs, ld = O.flow_stack(x_cpu, specs, inverse=True)
    (-(ld + O.std_normal_log_prob(zs[-1])).mean()).backward()
    specs64 = [{**sp, "params": {k: v.detach().double().requires_grad_(True) for k, v in sp["params"].items()}}
               for sp in specs]
    zs64, ld64 = O.flow_stack(x_cpu.double(), specs64, inverse=True)
    (-(ld64 + O.std_normal_log_prob(zs64[-1])).mean()).backward()

    def gpu_grads(min_rows):
        flows = []
        for i, sd in enumerate(sds):
            f = amd.AffineHalfFlow(dim, parity=bool(i % 2)); f.load_state_dict(sd); flows.append(f)
        model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)
        flat = amd.FlatParameters(model) if flat_home else None
        floor, amd._dispatch.BWD_SPLIT_MIN_ROWS = amd._dispatch.BWD_SPLIT_MIN_ROWS, min_rows
        try:
            if flat is not None:
                flat.zero_grad()
            (-model.log_prob(x_cpu.to(DEV)).mean()).backward()
        finally:
            amd._dispatch.BWD_SPLIT_MIN_ROWS = floor
        return [{n: q.grad.clone() for n, q in f.named_parameters()} for f in flows]

    split, fp32 = gpu_grads(0), gpu_grads(1 << 30)
    for i, spec in enumerate(specs):
        for name, ref in spec["params"].items():
            r64 = specs64[i]["params"][name].grad
            check_vs_float64(split[i][name], ref.grad, r64, f"9-layer run, split kernel: layer {i} grad {name}")
            check_vs_float64(fp32[i][name], ref.grad, r64, f"9-layer run, fp32 kernel: layer {i} grad {name}")
            assert_close(split[i][name], fp32[i][name], 5e-5, f"layer {i} grad {name} vs the fp32 kernel")
    assert any(not torch.equal(split[i][n], fp32[i][n]) for i in range(len(sds)) for n in split[i])  # (it did switch)


def moons(n):
    sk = pytest.importorskip("sklearn.datasets")
    pts, _ = sk.make_moons(n, noise=0.05, random_state=0)
    return torch.as_tensor(pts).float()


def train(model, optim, samples, steps):
    for _ in range(steps):
        _, log_det = model.inverse(samples)
        base_log_prob = model.base_log_prob(samples)
        loss = -(log_det + base_log_prob).sum()
        model.zero_grad()
        loss.backward()
        optim.step()
    return float(loss)


@pytest.mark.parametrize("name,bound", [("rnvp", 236), ("glow", 308), ("glow_actnorm", 246), ("nsfcl", 207),
                                        ("nsfcl_actnorm", 184), ("nsfar", 318), ("nsfar_actnorm", 213)])
def test_reference_training_contracts(amd, name, bound):
    """The reference's e2e tests: Adam, 1 step then 70 steps on 128 half-moon points; the loss must
    fall and end below the reference's bound (tests/test_flows.py:41-50, :53-55, :76-99)."""
    torch.manual_seed(0)
    samples = moons(128).to(DEV)
    if name == "rnvp":
        flows = [amd.AffineHalfFlow(dim=2, parity=i % 2 == 0) for i in range(2)]
    elif name.startswith("nsfar"):  # tests/test_flows.py:102-112
        flows = [amd.NSF_AR(dim=2, K=8, B=3, n_h=16) for _ in range(2)]
        if name == "nsfar_actnorm":
            for idx in reversed(range(len(flows))):
                flows.insert(idx, amd.ActNormFlow(dim=2))
    elif name.startswith("nsfcl"):  # tests/test_flows.py:89-99
        flows = [amd.NSF_CL(dim=2, K=8, B=3, n_h=16) for _ in range(2)]
        if name == "nsfcl_actnorm":
            for idx in reversed(range(len(flows))):
                flows.insert(idx, amd.ActNormFlow(dim=2))
    else:
        flows = [amd.Glow(dim=2) for _ in range(2)]
        if name == "glow_actnorm":
            for idx in reversed(range(len(flows))):
                flows.insert(idx, amd.ActNormFlow(dim=2))
    base = torch.distributions.MultivariateNormal(torch.zeros(2, device=DEV), torch.eye(2, device=DEV))
    model = amd.NormalizingFlowModel(base, flows).to(DEV)
    adam = torch.optim.Adam(model.parameters())
    loss1 = train(model, adam, samples, 1)
    loss2 = train(model, adam, samples, 70)
    assert loss1 > loss2
    assert loss2 < bound, f"{loss2=:.4} > {bound=}"


@pytest.mark.parametrize("dim,hid", [(64, 32), (6, 32), (6, 24), (40, 16)])
def test_runs_without_a_stack_kernel_fall_back_layer_by_layer(amd, O, dim, hid):
    """Shapes whose run does not go out as one launch (hidden width 32 at d = 64: the stack kernel declines, it would
    be no faster) still evaluate and train: same log-prob as the oracle, same gradients with and without run
    fusion.  (6, 32), (6, 24) and (40, 16) have the ragged stack kernel: the fused path itself."""
    h_sizes = (hid,) * 3
    sds = [recipes.affine_half_params(610 + dim + i, dim, h_sizes=h_sizes) for i in range(3)]
    layers = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd} for i, sd in enumerate(sds)]
    x_cpu = recipes.gaussian(611, 300, dim)
    ref_mean, ref_lp = O.mean_log_prob(x_cpu, layers)
    grads = {}
    for fused in (True, False):
        flows = []
        for i, sd in enumerate(sds):
            f = amd.AffineHalfFlow(dim, bool(i % 2), h_sizes=h_sizes)
            f.load_state_dict(sd)
            flows.append(f)
        model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)
        model.fuse_affine_runs = fused
        with torch.no_grad():
            lp = model.log_prob(x_cpu.to(DEV))
        assert_close(lp, ref_lp, 1e-5, "log_prob")
        loss = -model.log_prob(x_cpu.to(DEV)).mean()
        loss.backward()
        grads[fused] = {k: p.grad for k, p in model.named_parameters()}
        assert abs(float(loss) + ref_mean) <= 1e-5 * abs(ref_mean)
    for k in grads[True]:
        assert_close(grads[True][k], grads[False][k], GTOL, k)


def test_zz_gradient_budget_audit():
    """Runs last in this file: the worst gradient comparison per layer type of the session -- error from the float64
    oracle, budget, and how much of the budget is the fp32 oracle's own distance from float64 (pytest -s shows it;
    DESIGN.md section 1 quotes these)."""
    groups: dict[str, dict] = {}
    for r in GRAD_LOG:
        key = r["what"].split(" ")[0]
        if key not in groups or r["err"] / r["budget"] > groups[key]["err"] / groups[key]["budget"]:
            groups[key] = r
    for key, r in sorted(groups.items()):
        print(f"{key:12s} worst: {r['what'][:90]:90s} err {r['err']:.2e} budget {r['budget']:.2e} "
              f"(fp32-vs-fp64 share {r['widening']:.2e})")
