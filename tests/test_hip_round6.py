"""Round 6: the run-time-shaped matrix-core kernels (csrc/mnf_rt.h: any layer count and widths, weights converted from
the plain parameter vector inside the kernel) against the CPU oracle, through the C ABI (mnf_affine_half / mnf_nsf_cl /
mnf_rnvp_seeded with force_generic = 2, and by default for the shapes without a per-shape kernel)."""
import os

import numpy as np
import pytest
import torch

import recipes
from helpers import RTOL, assert_close, assert_parity

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def amd():
    import torch_mnf_amd

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch_mnf_amd


@pytest.fixture(scope="module")
def O():
    from oracle import flow_oracle

    return flow_oracle


def cuda(a):
    return a.to(DEV)


def rt(f):
    f.force_generic = 2  # include/mnf_hip.h: the run-time-shaped kernel whatever the shape's specialised kernels
    return f.to(DEV)


def grad_kernel(amd, name):
    """The gradient kernel family a call is expected on: the run-time-shaped gradient kernels add with float atomics and
    refuse under MNF_DETERMINISTIC=1 (the suite is also run in that mode), their shapes then take the VALU kernels."""
    return name.replace("_bwd_rt", "_bwd_generic") if amd.deterministic() else name


# ------------------------------------------------------------------ AffineHalfFlow (reference: affine_half_flow.py:28-66)
AHF_SHAPES = [
    (64, (24, 24), {}), (64, (24, 24, 24), {}), (64, (64, 64, 64), {}), (32, (24, 24), {}), (2, (24, 24), {}),
    (8, (16, 16, 16), {}), (50, (17, 30), {}), (6, (5, 9), {}), (128, (100,), {}), (256, (32, 32, 32), {}),
    (256, (200, 130, 40, 7), {}), (512, (24, 24, 24), {}), (512, (64, 64, 64), {}), (1024, (32, 32), {}),
    (64, (24, 24), {"scale": False}), (64, (24, 24), {"shift": False}), (40, (256,), {}),
]


@pytest.mark.parametrize("dim,hs,kw", AHF_SHAPES, ids=lambda v: str(v).replace(" ", ""))
def test_affine_half_rt_shape_matrix(amd, O, dim, hs, kw):
    """Any h_sizes length >= 1, hidden widths 4..256, any even dim, NICE / no-shift variants: the run-time-shaped kernel
    against the oracle, both parities and directions, row counts with partial tiles, one row."""
    sd = recipes.affine_half_params(11 + dim, dim, h_sizes=hs, s_last_gain=3.0, **kw)
    for rows in (1, 37, 1000):
        x = recipes.gaussian(5 + dim + rows, rows, dim)
        for parity in (False, True):
            f = amd.AffineHalfFlow(dim, parity=parity, h_sizes=hs, **kw)
            f.load_state_dict(sd)
            rt(f)
            for inverse in (False, True):
                ref_y, ref_ld = O.affine_half(x, sd, parity, inverse, **kw)
                with torch.no_grad():
                    y, ld = f.forward(cuda(x), inverse=inverse)
                assert amd.last_kernel() == "ahf_rt"
                what = f"ahf_rt d={dim} h={hs} {kw} rows={rows} par={parity} inv={inverse}"
                assert_parity(y, ref_y.numpy(), None, what + " y")
                if rows > 1:  # (one row: max |log_det| is the one value itself, a sum that may nearly cancel)
                    assert_parity(ld, ref_ld.numpy(), None, what + " ld")
                else:
                    assert abs(float(ld[0]) - float(ref_ld[0])) <= 1e-5 * max(1.0, float(ref_ld.abs().max()))


def test_affine_half_rt_is_the_default_without_a_specialised_kernel(amd, O):
    """From 2,048 rows on, a shape without a per-shape kernel lands on the run-time-shaped one by itself (no warning about
    an any-shape kernel); below, and for hidden layers narrower than 4 units, the VALU kernel keeps the call."""
    import warnings

    dim, hs = 64, (24, 24)
    sd = recipes.affine_half_params(3, dim, h_sizes=hs)
    f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs)
    f.load_state_dict(sd)
    f.to(DEV)
    x = recipes.gaussian(4, 5000, dim)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            y, ld = f.forward(cuda(x))
    assert amd.last_kernel() == "ahf_rt"
    ref_y, ref_ld = O.affine_half(x, sd, False, False)
    assert_close(y, ref_y, RTOL, "y")
    assert_close(ld, ref_ld, RTOL, "ld")
    with torch.no_grad():
        f.forward(cuda(x[:100]))
    assert amd.last_kernel() == "ahf_generic"
    g = amd.AffineHalfFlow(dim, parity=False, h_sizes=(2, 24)).to(DEV)
    with torch.no_grad():
        g.forward(cuda(x))
    assert amd.last_kernel() == "ahf_generic"


@pytest.mark.parametrize("case", ["big_rows", "big_weights", "tiny_first_layer", "inf_row"])
def test_affine_half_rt_range(amd, O, case):
    """The run-time-shaped kernel has no fp32 fix-up path: rows beyond the split range are scaled by a power of two
    before the split, weights are staged scaled to the top of f16's range -- results must not depend on either range."""
    dim, hs = 64, (24, 24)
    sd = recipes.affine_half_params(3, dim, h_sizes=hs)
    x = recipes.gaussian(4, 500, dim)
    if case == "big_rows":
        x = x.clone()
        x[::7] *= 3.0e5
        sd = {k: (v * 1e-5 if k.endswith("0.weight") else v) for k, v in sd.items()}
    elif case == "big_weights":
        sd = {k: (v * 3000 if k.endswith("2.weight") else (v / 3000 if k.endswith("4.weight") else v)) for k, v in sd.items()}
    elif case == "tiny_first_layer":
        x = x * 1.0e4
        sd = {k: (v * 1e-4 if k.endswith("0.weight") else v) for k, v in sd.items()}
    elif case == "inf_row":
        x = x.clone()
        x[3, 1] = float("inf")
    f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs)
    f.load_state_dict(sd)
    rt(f)
    ref_y, ref_ld = O.affine_half(x, sd, False, False)
    with torch.no_grad():
        y, ld = f.forward(cuda(x))
    assert amd.last_kernel() == "ahf_rt"
    if case == "inf_row":
        # a row holding a non-finite conditioner input comes out non-finite (the reference: inf or NaN, here NaN) and
        # does not touch its 15 tile neighbours
        keep = torch.ones(x.shape[0], dtype=torch.bool)
        keep[3] = False
        assert not bool(torch.isfinite(y.cpu()[3, dim // 2:]).any()) and not bool(torch.isfinite(ld.cpu()[3]))
        assert not bool(torch.isfinite(ref_y[3, dim // 2:]).any())
        assert_close(y.cpu()[keep], ref_y[keep], RTOL, "y (other rows)")
        assert_close(ld.cpu()[keep], ref_ld[keep], RTOL, "ld (other rows)")
        return
    assert_close(y, ref_y, RTOL, f"{case} y")
    assert_close(ld, ref_ld, RTOL, f"{case} ld")


# ------------------------------------------------------------------ NSF_CL (reference: spline_flow.py:241-285)
NSF_SHAPES = [(32, 8, 8), (64, 8, 16), (128, 8, 8), (128, 5, 32), (2, 5, 8), (6, 3, 5), (50, 10, 12), (16, 16, 64),
              (200, 4, 16), (48, 10, 32), (128, 10, 32), (64, 2, 8), (24, 13, 20), (32, 8, 32), (32, 5, 32), (16, 8, 24)]


@pytest.mark.parametrize("dim,K,n_h", NSF_SHAPES)
def test_nsf_cl_rt_shape_matrix(amd, O, dim, K, n_h):
    """Any dim, K = 2..16, n_h = 4..64: the run-time-shaped kernel against the oracle, both directions, on the parity
    rule's float64 budget (a spline element a few ulps from a knot moves by more than 1e-5 between two correct fp32
    evaluations: helpers.assert_parity)."""
    sd = recipes.nsf_cl_params(21 + dim + K, dim, K, n_h)
    sd64 = {k: v.double() for k, v in sd.items()}
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    rt(f)
    for rows in (37, 1500):
        x = recipes.gaussian(5 + dim + rows, rows, dim, scale=1.4)
        for inverse in (False, True):
            ref_y, ref_ld = O.nsf_cl(x, sd, K, 3.0, inverse)
            y64, ld64 = O.nsf_cl(x.double(), sd64, K, 3.0, inverse)
            with torch.no_grad():
                y, ld = (f.inverse if inverse else f.forward)(cuda(x))
            assert amd.last_kernel() == "nsf_rt"
            what = f"nsf_rt ({dim},{K},{n_h}) rows={rows} inv={inverse}"
            assert_parity(y, ref_y.numpy(), y64.numpy(), what + " y")
            assert_parity(ld, ref_ld.numpy(), ld64.numpy(), what + " ld")


def test_nsf_cl_rt_round_trip_and_default(amd):
    """d = 128 has no per-shape kernel: the default path is the run-time-shaped one, and inverse(forward(x)) = x with
    cancelling log-dets at 2^18 rows."""
    dim, K, n_h, rows = 128, 8, 8, 1 << 18
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(recipes.nsf_cl_params(77, dim, K, n_h))
    f.to(DEV)
    x = torch.randn(rows, dim, device=DEV) * 1.3
    with torch.no_grad():
        y, ld = f.forward(x)
        assert amd.last_kernel() == "nsf_rt"
        xb, ldb = f.inverse(y)
    assert float((xb - x).abs().max()) <= 2e-5 * float(x.abs().max())
    assert float((ld + ldb).abs().max()) <= 2e-5 * float(ld.abs().max())


# ------------------------------------------------------------------ RNVP (reference: rnvp.py:19-39)
def _rnvp_sd(seed, dim, hs):
    rng = np.random.default_rng(seed)
    sd = recipes.mlp_params(rng, "net", (dim, *hs), gain=1.5)
    k = 1.5 / np.sqrt(hs[-1])
    for name in ("t", "s"):
        sd[f"{name}.weight"] = torch.from_numpy(rng.uniform(-k, k, size=(dim, hs[-1])).astype(np.float32))
        sd[f"{name}.bias"] = torch.from_numpy(rng.uniform(-k, k, size=(dim,)).astype(np.float32))
    return sd


RNVP_SHAPES = [(800, (100,)), (50, (100,)), (128, (30,)), (784, (50, 40)), (50, (17,)), (37, (200,)), (1024, (64, 64)),
               (130, (130,)), (800, (50,)), (2048, (100,)), (64, (7, 9, 11))]


@pytest.mark.parametrize("dim,hs", RNVP_SHAPES, ids=lambda v: str(v).replace(" ", ""))
def test_rnvp_rt_shape_matrix(amd, O, dim, hs):
    """Any number of conditioner layers of widths 4..256, any dim: explicit mask and the in-kernel mask (the bits
    mnf_rnvp_mask materialises), row counts with partial tiles, one row."""
    sd = _rnvp_sd(31 + dim, dim, hs)
    f = amd.RNVP(dim, h_sizes=hs)
    f.load_state_dict(sd)
    rt(f)
    for rows in (1, 37, 700):
        z = recipes.gaussian(5 + dim + rows, rows, dim)
        mask = recipes.bernoulli_mask(97, rows, dim)
        ref_x, ref_ld = O.rnvp(z, sd, mask)
        with torch.no_grad():
            x, ld = f.forward(cuda(z), mask=cuda(mask))
        assert amd.last_kernel() == "rnvp_rt"
        assert_parity(x, ref_x.numpy(), None, f"rnvp_rt d={dim} h={hs} rows={rows} x")
        assert_parity(ld, ref_ld.numpy(), None, f"rnvp_rt d={dim} h={hs} rows={rows} ld")
        m_seed = f.mask_for(77, rows)
        with torch.no_grad():
            x, ld = f.forward(cuda(z), seed=77)
        assert amd.last_kernel() == "rnvp_rt"
        ref_x, ref_ld = O.rnvp(z, sd, m_seed.cpu())
        assert_parity(x, ref_x.numpy(), None, f"rnvp_rt d={dim} h={hs} rows={rows} x (seeded)")
        assert_parity(ld, ref_ld.numpy(), None, f"rnvp_rt d={dim} h={hs} rows={rows} ld (seeded)")


def test_rnvp_rt_is_the_default_for_wide_hidden_layers(amd, O):
    dim, hs, rows = 800, (100,), 4096
    sd = _rnvp_sd(5, dim, hs)
    f = amd.RNVP(dim, h_sizes=hs)
    f.load_state_dict(sd)
    f.to(DEV)
    z = recipes.gaussian(6, rows, dim)
    mask = recipes.bernoulli_mask(7, rows, dim)
    with torch.no_grad():
        x, ld = f.forward(cuda(z), mask=cuda(mask))
    assert amd.last_kernel() == "rnvp_rt"
    ref_x, ref_ld = O.rnvp(z, sd, mask)
    assert_close(x, ref_x, RTOL, "x")
    assert_close(ld, ref_ld, RTOL, "ld")


# ------------------------------------------------------------------ gradients (the reference trains through all three:
# tests/test_flows.py:14-99, tests/test_mnf_mnist.py:14-56): the run-time-shaped gradient kernels against the FLOAT64 oracle
from test_hip_autograd import OracleGrads, cot_loss  # noqa: E402  (the audited gradient budget: GBASE + float64 head-room)


def _backward(f, x_cpu, call, w_y, w_l):
    x = x_cpu.detach().to(DEV).requires_grad_(True)
    yg, ldg = call(f, x)
    ((yg * w_y.to(DEV)).sum() + (ldg * w_l.to(DEV)).sum()).backward()
    return {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}


AHF_BWD_SHAPES = [(64, (24, 24), {}), (64, (64, 64, 64), {}), (10, (16, 40), {}), (2, (24, 24), {}), (512, (24, 24, 24), {}),
                  (64, (24,), {}), (64, (20, 30, 40, 50), {}), (64, (24, 24), {"scale": False}), (64, (24, 24), {"shift": False}),
                  (256, (64, 64, 64), {})]


@pytest.mark.parametrize("dim,hs,kw", AHF_BWD_SHAPES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_half_rt_gradients(amd, O, dim, hs, kw, inverse):
    """1..4 hidden layers of widths 4..64, any even dim, NICE / no-shift variants, both directions and parities; a row
    count with a partial tile and one spanning several row blocks."""
    for rows, parity in ((300, True), (2100, False)):
        sd = recipes.affine_half_params(31 + dim, dim, h_sizes=hs, s_last_gain=2.0, **kw)
        # (seed 232: with test_hip_autograd's 32 + dim, row 211 of the dim = 64, (24, 24) case has a second-layer
        #  pre-activation of 1.4e-8 -- on the LeakyReLU kink, where two fp32-accurate evaluations may take different slopes)
        x_cpu = recipes.gaussian(232 + dim, rows, dim)
        w_y, w_l = recipes.gaussian(33, rows, dim), recipes.gaussian(34, rows, 1)[:, 0]
        ref = OracleGrads(cot_loss(lambda x, p: O.affine_half(x, p, parity, inverse, **kw), w_y, w_l), x_cpu, sd)
        f = amd.AffineHalfFlow(dim, parity, h_sizes=hs, **kw)
        f.load_state_dict(sd)
        rt(f)
        got = _backward(f, x_cpu, lambda m, x: m.forward(x, inverse=inverse), w_y, w_l)
        assert amd.last_kernel() == grad_kernel(amd, "ahf_bwd_rt")
        ref.check_all(got, f"ahf_bwd_rt d={dim} h={hs} {kw} rows={rows} inv={inverse}")


NSF_BWD_SHAPES = [(32, 8, 8), (64, 8, 16), (128, 8, 8), (48, 5, 32), (2, 5, 8), (6, 3, 5), (50, 10, 12), (16, 16, 16), (64, 10, 32)]


@pytest.mark.parametrize("dim,K,n_h", NSF_BWD_SHAPES)
@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_cl_rt_gradients(amd, O, dim, K, n_h, inverse):
    rows = 700
    sd = recipes.nsf_cl_params(51 + dim + K, dim, K, n_h)
    x_cpu = recipes.gaussian(52 + dim, rows, dim, scale=1.4)
    w_y, w_l = recipes.gaussian(53, rows, dim), recipes.gaussian(54, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(lambda x, p: O.nsf_cl(x, p, K, 3.0, inverse), w_y, w_l), x_cpu, sd)
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    rt(f)
    got = _backward(f, x_cpu, lambda m, x: (m.inverse if inverse else m.forward)(x), w_y, w_l)
    assert amd.last_kernel() == grad_kernel(amd, "nsf_bwd_rt")
    ref.check_all(got, f"nsf_bwd_rt ({dim},{K},{n_h}) inv={inverse}")


RNVP_BWD_SHAPES = [(800, (100,)), (50, (100,)), (128, (30,)), (784, (50, 40)), (50, (17,)), (37, (120,)), (256, (64, 64)),
                   (64, (7, 9, 11))]


@pytest.mark.parametrize("dim,hs", RNVP_BWD_SHAPES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("seeded", [False, True])
def test_rnvp_rt_gradients(amd, O, dim, hs, seeded):
    rows = 700
    sd = _rnvp_sd(41 + dim, dim, hs)
    z_cpu = recipes.gaussian(42 + dim, rows, dim)
    w_y, w_l = recipes.gaussian(43, rows, dim), recipes.gaussian(44, rows, 1)[:, 0]
    f = amd.RNVP(dim, h_sizes=hs)
    f.load_state_dict(sd)
    rt(f)
    mask = f.mask_for(77, rows).cpu() if seeded else recipes.bernoulli_mask(97, rows, dim)
    ref = OracleGrads(cot_loss(lambda x, p: O.rnvp(x, p, mask.to(x.dtype)), w_y, w_l), z_cpu, sd)
    got = _backward(f, z_cpu, (lambda m, z: m.forward(z, seed=77)) if seeded else (lambda m, z: m.forward(z, mask=mask.to(DEV))),
                    w_y, w_l)
    assert amd.last_kernel() == grad_kernel(amd, "rnvp_bwd_rt")
    ref.check_all(got, f"rnvp_bwd_rt d={dim} h={hs} seeded={seeded}")


def test_rt_gradients_are_the_default_without_a_specialised_kernel(amd):
    """From 2,048 rows on the autograd path lands on the run-time-shaped gradient kernels by itself for shapes without a
    per-shape one; below, on the VALU kernels."""
    cases = [(amd.AffineHalfFlow(64, False, h_sizes=(64, 64, 64)), 64, lambda m, x: m.forward(x), "ahf_bwd_rt", "ahf_bwd_generic"),
             (amd.NSF_CL(128, K=8, B=3, n_h=8), 128, lambda m, x: m.forward(x), "nsf_bwd_rt", "nsf_bwd_generic"),
             (amd.RNVP(128, h_sizes=(100,)), 128, lambda m, x: m.forward(x, seed=3), "rnvp_bwd_rt", "rnvp_bwd_generic")]
    for f, dim, call, fast, slow in cases:
        f.to(DEV)
        for rows, want in ((4096, grad_kernel(amd, fast)), (200, slow)):
            x = torch.randn(rows, dim, device=DEV, requires_grad=True)
            f.zero_grad()
            y, ld = call(f, x)
            (y.sum() + ld.sum()).backward()
            assert amd.last_kernel() == want, (type(f).__name__, rows, amd.last_kernel())
            assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in f.parameters())


@pytest.mark.parametrize("layer", ["ahf", "nsf", "nsf_rt", "rnvp"])
def test_gradients_of_a_view_at_an_odd_storage_offset(amd, O, layer):
    """Rows that are not 16-byte aligned (a contiguous view one float into its buffer): the per-shape gradient kernels refuse
    such a launch (MNF_ERR_UNSUPPORTED) and the call must land on a kernel that takes it -- the run-time-shaped one, whose
    row accesses then go element by element -- with the same gradients (advisor, round 5: the NSF_CL tile kernel's refusal
    used to surface as an exception in backward after a forward pass that had quietly fallen back)."""
    rows = 2500
    if layer == "ahf":
        dim, sd = 64, recipes.affine_half_params(71, 64)
        f = amd.AffineHalfFlow(dim, False)
        ref_fn, call = (lambda x, p: O.affine_half(x, p, False, False)), (lambda m, x: m.forward(x))
    elif layer in ("nsf", "nsf_rt"):
        dim, n_h = 32, (8 if layer == "nsf" else 40)
        sd = recipes.nsf_cl_params(72, dim, 8, n_h)
        f = amd.NSF_CL(dim, K=8, B=3, n_h=n_h)
        ref_fn, call = (lambda x, p: O.nsf_cl(x, p, 8, 3.0, False)), (lambda m, x: m.forward(x))
    else:
        dim, sd = 128, _rnvp_sd(73, 128, (50,))
        f = amd.RNVP(dim, h_sizes=(50,))
        mask = recipes.bernoulli_mask(74, rows, dim)
        ref_fn, call = (lambda x, p: O.rnvp(x, p, mask.to(x.dtype))), (lambda m, x: m.forward(x, mask=mask.to(DEV)))
    f.load_state_dict(sd)
    f.to(DEV)
    x_cpu = recipes.gaussian(75 + dim, rows, dim)
    w_y, w_l = recipes.gaussian(76, rows, dim), recipes.gaussian(77, rows, 1)[:, 0]
    ref = OracleGrads(cot_loss(ref_fn, w_y, w_l), x_cpu, sd)
    buf = torch.zeros(rows * dim + 1, device=DEV)
    x = buf[1:].view(rows, dim)
    x.copy_(x_cpu.to(DEV))
    assert x.data_ptr() % 16 != 0 and x.is_contiguous()
    x.requires_grad_(True)
    y, ld = call(f, x)
    ((y * w_y.to(DEV)).sum() + (ld * w_l.to(DEV)).sum()).backward()
    got = {"x": x.grad, **{n: q.grad for n, q in f.named_parameters()}}
    ref.check_all(got, f"{layer} at an odd storage offset ({amd.last_kernel()})")


def test_graphed_training_step_of_a_model_on_the_run_time_shaped_kernels(amd):
    """A model none of whose coupling layers has a per-shape kernel -- two-layer conditioners, K = 6, 100 RNVP hidden units --
    trained with FlatParameters + FusedAdam at 4,096 rows: every layer's forward and gradient launch is a run-time-shaped
    kernel (no operand image to repack after the optimiser step: they read the flat buffer), the parameter gradients land in
    the flat gradient buffer, and the step replays from one hipGraph (GraphedStep) with the losses and parameters of the
    eager loop."""
    dim, rows = 16, 4096

    def build():
        torch.manual_seed(31)
        layers = [amd.AffineHalfFlow(dim, parity=False, h_sizes=(20, 20)), amd.NSF_CL(dim, K=6, B=3, n_h=24),
                  amd.AffineHalfFlow(dim, parity=True, h_sizes=(40,)), amd.AffineHalfFlow(dim, parity=False, h_sizes=(12, 16, 12, 8))]
        model = amd.NormalizingFlowModel(amd.StandardNormal(dim), layers).to(DEV)
        return model, amd.FusedAdam(amd.FlatParameters(model), lr=1e-3, capturable=True)

    batches = [recipes.gaussian(400 + i, rows, dim).to(DEV) for i in range(8)]
    model_e, opt_e = build()
    losses_e, kernels = [], set()
    for x in [batches[0]] * 3 + batches[1:]:
        opt_e.zero_grad()
        loss = -model_e.log_prob(x).mean()
        kernels.add(amd.last_kernel())
        loss.backward()
        kernels.add(amd.last_kernel())
        opt_e.step()
        losses_e.append(float(loss.detach()))
    del loss
    assert kernels <= {"ahf_rt", grad_kernel(amd, "ahf_bwd_rt"), "nsf_rt", grad_kernel(amd, "nsf_bwd_rt")}, kernels
    assert losses_e[-1] < losses_e[0]
    model_g, opt_g = build()
    step = amd.GraphedStep(opt_g, lambda x: -model_g.log_prob(x).mean(), batches[0])
    losses_g = [float(step(x)) for x in batches[1:]]
    for a, b in zip(losses_e[3:], losses_g):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (a, b)
    assert_close(opt_g.flat.data, opt_e.flat.data, 2e-3, "parameters after 10 steps")
    # and a wide RNVP (100 hidden units) as MNFLinear's flows would use it, trained the same way
    torch.manual_seed(32)
    f = amd.RNVP(64, h_sizes=(100,)).to(DEV)
    opt = amd.FusedAdam(amd.FlatParameters(f), lr=1e-3)
    z = recipes.gaussian(500, rows, 64).to(DEV)
    first = None
    for _ in range(5):
        opt.zero_grad()
        x, ld = f.forward(z, seed=9)
        loss = (x.pow(2).mean() - ld.mean())
        loss.backward()
        assert amd.last_kernel() == grad_kernel(amd, "rnvp_bwd_rt")
        opt.step()
        first = float(loss.detach()) if first is None else first
    assert float(loss.detach()) < first
