"""GPU parity tests: the HIP path (through the C ABI, via torch_mnf_amd) against
(1) the golden fixtures written by the real reference and (2) the CPU oracle on seeded inputs,
plus size-independent properties at BASELINE.json's full sizes.

Tolerance (BASELINE.json north_star "1e-5 rel fp32"): normwise
max|a-b| <= 1e-5 * max|b| per output tensor; scalar mean log-prob to 1e-5 relative.
"""
import os

import numpy as np
import pytest
import torch

import recipes
from helpers import RTOL, normwise_err, assert_close, assert_parity, c2_layers, c3_layers, g1_layers, t, unpack_mask

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def amd():
    import torch_mnf_amd

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    if not os.path.exists(torch_mnf_amd.library_path()):  # normally prebuilt by __graft_entry__.build()
        import __graft_entry__

        __graft_entry__.build()
    torch_mnf_amd._lib.load()
    return torch_mnf_amd


@pytest.fixture(scope="module")
def O():
    from oracle import flow_oracle

    return flow_oracle


def cuda(a):
    return (a if isinstance(a, torch.Tensor) else t(a)).to(DEV)


def ahf_module(amd, sd, dim, parity, kernel="split", **kw):
    f = amd.AffineHalfFlow(dim, parity, **kw)
    f.load_state_dict(sd)
    return select_kernel(f.to(DEV), kernel)


KERNELS = ["split", "fp32", "generic"]  # f16 hi+lo MFMA (default) / fp32 MFMA / shape-generic


def select_kernel(f, kernel):
    f.force_generic = kernel == "generic"
    f.force_fp32_mfma = kernel == "fp32"
    return f


# ------------------------------------------------------------------ AffineHalfFlow
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dim", [64, 256])
@pytest.mark.parametrize("parity", [False, True])
def test_g2_affine_half_golden(amd, golden, dim, parity, kernel):
    fx = golden("g2_affine_half_single")
    tag = f"d{dim}_p{int(parity)}"
    f = ahf_module(amd, recipes.affine_half_params(200 + dim + int(parity), dim), dim, parity, kernel)
    z = cuda(fx[f"{tag}.z"])
    x, ld = f.forward(z)
    assert_close(x, fx[f"{tag}.fwd"], RTOL, "fwd")
    assert_close(ld, fx[f"{tag}.ld_fwd"], RTOL, "ld_fwd")
    x, ld = f.inverse(z)
    assert_close(x, fx[f"{tag}.inv"], RTOL, "inv")
    assert_close(ld, fx[f"{tag}.ld_inv"], RTOL, "ld_inv")
    x2, ld2 = f.forward(z, inverse=True)  # the reference's extra keyword (:44)
    assert torch.equal(x, x2) and torch.equal(ld, ld2)


def test_mfma_kernel_is_selected(amd):
    """The d=64 / d=256 default configuration must run the specialised kernel (image exists)."""
    lib = amd._lib.load()
    hid = amd._lib.int_array([24, 24, 24])
    for dim in (32, 64, 128, 256):
        assert lib.mnf_affine_half_image_floats(dim, 3, hid, 1, 1) > 0
        f = amd.AffineHalfFlow(dim, False).to(DEV)
        assert f._split_image(torch.device(DEV, 0)) is not None  # the split kernel is the default
        f.force_fp32_mfma = True
        assert f._split_image(torch.device(DEV, 0)) is None
    # d = 2 (config 1): a one-column half padded to a 16-column tile, served by the stack kernel's ragged variant
    assert lib.mnf_affine_half_image_floats(2, 3, hid, 1, 1) > 0
    assert amd.AffineHalfFlow(2, False).to(DEV)._split_image(torch.device(DEV, 0)) is not None
    assert lib.mnf_affine_half_image_floats(258, 3, hid, 1, 1) == 0


@pytest.mark.parametrize("tag,kw", [("nice", dict(scale=False)), ("noshift", dict(shift=False)),
                                    ("h2", dict(h_sizes=(16, 40))), ("h1", dict(h_sizes=(7,)))])
def test_g2_affine_half_variants(amd, golden, tag, kw):
    fx = golden("g2_affine_half_single")
    f = ahf_module(amd, recipes.affine_half_params(290, 10, **kw), 10, True, **kw)
    z = cuda(fx[f"{tag}.z"])
    for name, fn in (("fwd", f.forward), ("inv", f.inverse)):
        x, ld = fn(z)
        assert_close(x, fx[f"{tag}.{name}"], RTOL, f"{tag}.{name}")
        assert_close(ld, fx[f"{tag}.ld_{name}"], RTOL, f"{tag}.ld_{name}")


@pytest.mark.parametrize("kernel", ["split", "fp32"])
@pytest.mark.parametrize("rows", [1, 15, 17, 1000, 4099])
@pytest.mark.parametrize("dim", [32, 64, 128])
def test_affine_half_ragged_rows_vs_oracle(amd, O, rows, dim, kernel):
    """Tiles are 16 rows: partial last tile, single row, and a grid-stride wrap."""
    sd = recipes.affine_half_params(77 + dim, dim)
    x = recipes.gaussian(rows + dim, rows, dim)
    for parity in (False, True):
        f = ahf_module(amd, sd, dim, parity, kernel)
        for inverse in (False, True):
            ref_y, ref_ld = O.affine_half(x, sd, parity, inverse)
            y, ld = f.forward(cuda(x), inverse=inverse)
            assert_close(y, ref_y, RTOL, "y")
            assert_close(ld, ref_ld, RTOL, "ld")


@pytest.mark.parametrize("rows", [1, 17, 777, 1000, 4099])
@pytest.mark.parametrize("dim,hid", [(2, 24), (4, 24), (6, 16), (10, 24), (24, 24), (30, 16), (40, 24), (50, 24), (62, 24),
                                     (100, 24), (58, 16), (130, 24), (200, 24), (250, 24), (6, 32), (20, 32), (50, 32), (100, 32)])
def test_affine_half_narrow_halves_vs_oracle(amd, O, rows, dim, hid):
    """A coupling half that does not fill its 16/32/64/128-column MFMA tile (any even d <= 256, e.g. config 1's
    d = 2): zero-padded operand image, masked row accesses (16-byte when d % 8 == 0, else element by element).
    One layer and a 3-layer run, both directions, against the oracle and the shape-generic kernel."""
    h_sizes = (hid,) * 3
    sds = [recipes.affine_half_params(700 + dim + i, dim, h_sizes=h_sizes, s_last_gain=2.0) for i in range(3)]
    x = recipes.gaussian(rows + dim, rows, dim)
    flows = [ahf_module(amd, sd, dim, bool(i % 2), h_sizes=h_sizes) for i, sd in enumerate(sds)]
    assert all(f._split_image(torch.device(DEV, 0)) is not None for f in flows)
    for inverse in (False, True):
        f = flows[1]
        ref_y, ref_ld = O.affine_half(x, sds[1], True, inverse)
        y, ld = f.forward(cuda(x), inverse=inverse)
        assert_close(y, ref_y, RTOL, "y")
        assert_close(ld, ref_ld, RTOL, "ld")
        f.force_generic = True
        y_g, ld_g = f.forward(cuda(x), inverse=inverse)
        f.force_generic = False
        assert_close(y, y_g, RTOL, "vs generic y")
        assert_close(ld, ld_g, RTOL, "vs generic ld")
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)
    layers = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd} for i, sd in enumerate(sds)]
    with torch.no_grad():
        zs, ld = model.inverse(cuda(x))
        assert zs[1].data_ptr() + zs[1].numel() * 4 == zs[2].data_ptr()  # one launch, one buffer
        cur, ld_ref = x, 0
        for i in (2, 1, 0):
            cur, l1 = O.affine_half(cur, sds[i], bool(i % 2), True)
            ld_ref = ld_ref + l1
            assert_close(zs[3 - i], cur, RTOL, f"z after layer {i}")
        assert_close(ld, ld_ref, RTOL, "run log_det")
        xs, ld_f = model.forward(zs[-1])
        assert_close(xs[-1], x, 1e-4, "round trip")
        lp, total = model.log_prob(cuda(x), return_sum=True)
        assert model._logprob_done
        ref_mean, ref_lp = O.mean_log_prob(x, layers)
        assert_close(lp, ref_lp, RTOL, "log_prob")
        assert abs(float(total) / rows - ref_mean) <= RTOL * abs(ref_mean)


def test_narrow_half_unaligned_view(amd, O):
    """Row bases that are not 16-byte aligned (a column-offset view made contiguous at an odd float offset)."""
    dim = 8
    sd = recipes.affine_half_params(801, dim)
    f = ahf_module(amd, sd, dim, False)
    x = recipes.gaussian(802, 100, dim)
    buf = torch.empty(100 * dim + 1, device=DEV)
    xv = buf[1:].view(100, dim)
    xv.copy_(cuda(x))
    assert xv.data_ptr() % 16 != 0
    y, ld = f.forward(xv)
    ref_y, ref_ld = O.affine_half(x, sd, False, False)
    assert_close(y, ref_y, RTOL, "y")
    assert_close(ld, ref_ld, RTOL, "ld")


@pytest.mark.parametrize("kernel", ["split", "fp32"])
@pytest.mark.parametrize("dim,hid", [(32, 16), (64, 16), (32, 32), (64, 32), (128, 32), (32, 24), (128, 24), (64, 64), (32, 64), (128, 64),
                                     (64, 48)])
def test_affine_half_mfma_shape_matrix(amd, O, dim, hid, kernel):
    """Every (dim, hidden width) pair with a specialised kernel: MFMA result vs the oracle, and the
    library must actually have picked that kernel (an operand image exists)."""
    h_sizes = (hid, hid, hid)
    lib = amd._lib.load()
    assert lib.mnf_affine_half_image_floats(dim, 3, amd._lib.int_array(h_sizes), 1, 1) > 0
    sd = recipes.affine_half_params(90 + dim + hid, dim, h_sizes=h_sizes, s_last_gain=3.0)
    x = recipes.gaussian(91 + dim, 531, dim)
    for parity in (False, True):
        f = ahf_module(amd, sd, dim, parity, kernel, h_sizes=h_sizes)
        assert f._packed(torch.device(DEV, 0))[1] is not None
        assert (f._split_image(torch.device(DEV, 0)) is not None) == (kernel == "split")
        for inverse in (False, True):
            ref_y, ref_ld = O.affine_half(x, sd, parity, inverse)
            y, ld = f.forward(cuda(x), inverse=inverse)
            assert_close(y, ref_y, RTOL, "y")
            assert_close(ld, ref_ld, RTOL, "ld")


@pytest.mark.parametrize("kernel", ["split", "fp32"])
@pytest.mark.parametrize("dim,hid", [(64, 30), (800, 30), (128, 50), (784, 50), (80, 30),
                                     (50, 50), (49, 30), (70, 50), (100, 50), (130, 30), (500, 50), (790, 50),
                                     (800, 20), (64, 7), (100, 41), (50, 1), (800, 64), (100, 57), (50, 64)])
def test_rnvp_mfma_shape_matrix(amd, O, dim, hid, kernel):
    """(dim, hidden width) pairs of the MFMA kernels (a hidden width other than 30 / 50 / 64 runs at the next one up with
    structural-zero units); dim % 16 != 0 (MNFLinear(50, 10)'s flow, MNFFeedForward's
    100 / 500-wide layers) runs the ragged variants: zero-padded operand image, masked row accesses (16-byte
    when dim % 4 == 0, else element by element), padded scale bias such that nothing reaches log_det."""
    sd = recipes.rnvp_params(95 + dim + hid, dim, hid)
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    select_kernel(f.to(DEV), kernel)
    assert f._packed(torch.device(DEV, 0))[1] is not None
    assert (f._split_image(torch.device(DEV, 0)) is not None) == (kernel == "split")
    for rows in (131, 1000):
        z = recipes.gaussian(96 + dim, rows, dim)
        mask = recipes.bernoulli_mask(97, rows, dim)
        ref_x, ref_ld = O.rnvp(z, sd, mask)
        x, ld = f.forward(cuda(z), mask=cuda(mask))
        assert_close(x, ref_x, RTOL, "x")
        assert_close(ld, ref_ld, RTOL, "ld")
        # the in-kernel mask: same bits as mnf_rnvp_mask materialises
        m_seed = f.mask_for(77, rows)
        x_s, ld_s = f.forward(cuda(z), seed=77)
        ref_x, ref_ld = O.rnvp(z, sd, m_seed.cpu())
        assert_close(x_s, ref_x, RTOL, "x (seeded)")
        assert_close(ld_s, ref_ld, RTOL, "ld (seeded)")


@pytest.mark.parametrize("dim", [800, 50, 100])
@pytest.mark.parametrize("case", ["big_inputs", "big_weights", "inf_input", "one_big_row", "tiny_inputs"])
def test_rnvp_split_range_guard(amd, O, case, dim):
    """RNVP on the split path: a 128-row group whose operands leave the f16 range is recomputed with fp32
    MFMAs inside the same launch; the other groups stay on the split path (dim 50 / 100: ragged variants)."""
    hid, rows = 50, 700
    sd = recipes.rnvp_params(140, dim, hid)
    z = recipes.gaussian(141, rows, dim)
    if case == "big_inputs":
        z = z * 1.0e5
        sd = {k: (v * 1e-5 if k == "net.0.weight" else v) for k, v in sd.items()}
    elif case == "big_weights":
        sd = {k: (v * 1e5 if k == "net.0.weight" else v) for k, v in sd.items()}
        z = z * 1e-5
    elif case == "inf_input":
        z = z.clone()
        z[300, 17] = float("inf")
    elif case == "one_big_row":
        z = z.clone()
        z[5] *= 1e5
        z[650] *= 1e5
        sd = {k: (v * 1e-3 if k == "net.0.weight" else v) for k, v in sd.items()}
    elif case == "tiny_inputs":
        z = z * 1e-6
    mask = recipes.bernoulli_mask(142, rows, dim)
    ref_x, ref_ld = O.rnvp(z, sd, mask)
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    f.to(DEV)
    x, ld = f.forward(cuda(z), mask=cuda(mask))
    ok = torch.isfinite(ref_x).all(1)
    assert torch.equal(torch.isfinite(x).all(1).cpu(), ok)
    assert_close(x[ok.to(DEV)], ref_x[ok], RTOL, f"{case} x")
    assert_close(ld[ok.to(DEV)], ref_ld[ok], RTOL, f"{case} ld")
    # the in-kernel mask takes other kernels (d = 50: the narrow latency kernel, whose cold tiles are redone per wave)
    m_seed = f.mask_for(77, rows).cpu()
    ref_xs, ref_lds = O.rnvp(z, sd, m_seed)
    x_s, ld_s = f.forward(cuda(z), seed=77)
    ok = torch.isfinite(ref_xs).all(1)
    assert torch.equal(torch.isfinite(x_s).all(1).cpu(), ok)
    assert_close(x_s[ok.to(DEV)], ref_xs[ok], RTOL, f"{case} x (seeded)")
    assert_close(ld_s[ok.to(DEV)], ref_lds[ok], RTOL, f"{case} ld (seeded)")


def _f64_affine_half(x, sd, parity, inverse):
    """float64 evaluation of the layer (affine_half_flow.py:44-66) as the accuracy yardstick."""
    import numpy as np

    x = np.asarray(x, np.float64)
    h = x.shape[1] // 2
    x0, x1 = (x[:, h:], x[:, :h]) if parity else (x[:, :h], x[:, h:])

    def mlp(prefix, v):
        n = len([k for k in sd if k.startswith(prefix) and k.endswith("weight")])
        for i, li in enumerate(sorted({int(k.split(".")[1]) for k in sd if k.startswith(prefix)})):
            v = v @ np.asarray(sd[f"{prefix}.{li}.weight"], np.float64).T + np.asarray(sd[f"{prefix}.{li}.bias"], np.float64)
            if i < n - 1:
                v = np.where(v > 0, v, 0.2 * v)
        return v

    sv, tv = mlp("s_net", x0), mlp("t_net", x0)
    y1 = (x1 - tv) * np.exp(-sv) if inverse else np.exp(sv) * x1 + tv
    y = np.concatenate([y1, x0] if parity else [x0, y1], axis=1)
    return y, (-sv.sum(1) if inverse else sv.sum(1))


@pytest.mark.parametrize("dim", [32, 64, 256])
def test_split_kernel_accuracy_against_float64(amd, O, dim):
    """The split (f16 hi + lo) kernel is as close to the float64 value of the layer as fp32 arithmetic is:
    its error is of the size of the reference's own fp32 rounding error, far inside the 1e-5 bar."""
    sd = recipes.affine_half_params(300 + dim, dim, s_last_gain=2.0)
    x = recipes.gaussian(301 + dim, 2048, dim)
    errs = {}
    for inverse in (False, True):
        y64, ld64 = _f64_affine_half(x, sd, True, inverse)
        ref_y, ref_ld = O.affine_half(x, sd, True, inverse)  # the reference's fp32 arithmetic
        for kernel in ("split", "fp32"):
            with torch.no_grad():
                y, ld = ahf_module(amd, sd, dim, True, kernel).forward(cuda(x), inverse=inverse)
            errs[kernel, inverse] = max(normwise_err(y.cpu().numpy(), y64), normwise_err(ld.cpu().numpy(), ld64))
        errs["reference", inverse] = max(normwise_err(ref_y.numpy(), y64), normwise_err(ref_ld.numpy(), ld64))
        assert errs["split", inverse] < 1e-6, errs
        assert errs["split", inverse] < 4 * errs["reference", inverse] + 2e-7, errs


@pytest.mark.parametrize("case", ["big_inputs", "big_weights", "inf_input", "tiny_inputs", "one_big_row"])
def test_split_kernel_range_guard(amd, O, case):
    """Operands outside the f16 range (|v| >= 3e4) send the tile to the fp32 path inside the same launch:
    results must not depend on the input range."""
    import numpy as np

    dim = 64
    sd = recipes.affine_half_params(310, dim)
    x = recipes.gaussian(311, 333, dim)
    if case == "big_inputs":
        x = x * 1.0e5
        sd = {k: (v * 1e-5 if k.endswith(".0.weight") else v) for k, v in sd.items()}  # keeps s moderate
    elif case == "big_weights":
        sd = {k: (v * 1e5 if k.endswith(".0.weight") else v) for k, v in sd.items()}
        x = x * 1e-5
    elif case == "inf_input":
        x = x.clone()
        x[7, 3] = float("inf")
        x[100, 40] = -float("inf")
    elif case == "tiny_inputs":
        x = x * 1e-6
    elif case == "one_big_row":
        x = x.clone()
        x[200] *= 5e4
        sd = {k: (v * 1e-3 if k.endswith(".0.weight") else v) for k, v in sd.items()}
    for parity in (False, True):
        for inverse in (False, True):
            ref_y, ref_ld = O.affine_half(x, sd, parity, inverse)
            y, ld = ahf_module(amd, sd, dim, parity, "split").forward(cuda(x), inverse=inverse)
            fin = torch.isfinite(ref_y)
            assert torch.equal(torch.isfinite(y).cpu(), fin)
            rows_ok = fin.all(1)
            assert_close(y[rows_ok.to(DEV)], ref_y[rows_ok], RTOL, f"{case} y")
            assert_close(ld[rows_ok.to(DEV)], ref_ld[rows_ok], RTOL, f"{case} ld")
    # the stack kernel shares the guard
    flows = [ahf_module(amd, sd, dim, bool(i % 2), "split") for i in range(3)]
    fused = amd.FusedAffineStack(flows).to(DEV)
    z, ld = fused.inverse(cuda(x))
    ref_z, ref_ld = x, 0
    for i in reversed(range(3)):
        ref_z, l1 = O.affine_half(ref_z, sd, bool(i % 2), True)
        ref_ld = ref_ld + l1
    rows_ok = torch.isfinite(ref_z).all(1)
    assert_close(z[rows_ok.to(DEV)], ref_z[rows_ok], 3 * RTOL, f"{case} stack z")
    assert_close(ld[rows_ok.to(DEV)], ref_ld[rows_ok], 3 * RTOL, f"{case} stack ld")


@pytest.mark.parametrize("dim", [128, 256])
def test_wide_stack_kernel_range_guard(amd, O, dim):
    """d = 128 / 256 runs (one tile per wave; at 256 s and t are consumed chunk by chunk): rows outside the
    f16 range take the fp32 path inside the same launch."""
    sd = recipes.affine_half_params(320 + dim, dim)
    sd = {k: (v * 1e-3 if k.endswith(".0.weight") else v) for k, v in sd.items()}
    x = recipes.gaussian(321, 300, dim).clone()
    x[17] *= 5e4
    x[200:216] *= 2e4
    flows = [ahf_module(amd, sd, dim, bool(i % 2), "split") for i in range(3)]
    model = amd.NormalizingFlow(flows).to(DEV)
    with torch.no_grad():
        zs, ld = model.inverse(cuda(x))
    assert zs[1].data_ptr() + zs[1].numel() * 4 == zs[2].data_ptr()  # one launch wrote all three
    ref, ref_ld = x, 0
    for i in reversed(range(3)):
        ref, l1 = O.affine_half(ref, sd, bool(i % 2), True)
        ref_ld = ref_ld + l1
    ok = torch.isfinite(ref).all(1)
    assert_close(zs[-1][ok.to(DEV)], ref[ok], 3 * RTOL, "z")
    assert_close(ld[ok.to(DEV)], ref_ld[ok], 3 * RTOL, "log_det")


def test_empty_batches(amd):
    f = ahf_module(amd, recipes.affine_half_params(1, 64), 64, False)
    y, ld = f.forward(torch.empty(0, 64, device=DEV))
    assert y.shape == (0, 64) and ld.shape == (0,)
    zs, ld = amd.NormalizingFlow([f, amd.NSF_CL(64).to(DEV), amd.ActNormFlow(64).to(DEV)]).forward(
        torch.empty(0, 64, device=DEV))
    assert zs[-1].shape == (0, 64) and ld.shape == (0,)


def test_affine_half_overflow_matches_reference_semantics(amd, O):
    """s is unbounded (no clamp): huge weights overflow to inf/NaN exactly like the reference."""
    sd = recipes.affine_half_params(5, 64, s_last_gain=4000.0)
    x = recipes.gaussian(6, 64, 64)
    ref_y, ref_ld = O.affine_half(x, sd, False, False)
    f = ahf_module(amd, sd, 64, False)
    y, _ = f.forward(cuda(x))
    assert torch.isinf(ref_y).any()
    assert torch.equal(torch.isfinite(y).cpu(), torch.isfinite(ref_y))


@pytest.mark.parametrize("seed", range(6))
def test_generic_kernels_random_shapes_vs_oracle(amd, O, seed):
    """Seeded sweep over shapes only the generic kernels serve: odd dims, 0-3 hidden layers of
    assorted widths, K from 2 to 12, row counts around the workgroup tile."""
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([2, 4, 6, 10, 18, 34, 50]))
    rows = int(rng.choice([1, 3, 63, 64, 65, 257]))
    h_sizes = tuple(int(v) for v in rng.integers(1, 40, size=int(rng.integers(0, 4))))
    sd = recipes.affine_half_params(2000 + seed, dim, h_sizes=h_sizes, s_last_gain=2.0)
    x = recipes.gaussian(3000 + seed, rows, dim)
    parity = bool(seed % 2)
    f = ahf_module(amd, sd, dim, parity, h_sizes=h_sizes)
    for inverse in (False, True):
        ref_y, ref_ld = O.affine_half(x, sd, parity, inverse)
        y, ld = f.forward(cuda(x), inverse=inverse)
        assert_close(y, ref_y, RTOL, f"ahf d={dim} h={h_sizes} rows={rows}")
        assert_close(ld, ref_ld, RTOL, "ahf ld")
    K = int(rng.choice([2, 3, 5, 8, 12]))
    n_h = int(rng.choice([3, 8, 13]))
    sdn = recipes.nsf_cl_params(4000 + seed, dim, K, n_h)
    g = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    g.load_state_dict(sdn)
    g.to(DEV)
    xs = recipes.gaussian(5000 + seed, rows, dim, scale=1.5)
    for inverse in (False, True):
        ref_y, ref_ld = O.nsf_cl(xs, sdn, K, 3.0, inverse)
        y, ld = (g.inverse if inverse else g.forward)(cuda(xs))
        assert_close(y, ref_y, 2e-5, f"nsf d={dim} K={K} n_h={n_h} rows={rows}")
        assert_close(ld, ref_ld, 2e-5 if abs(float(ref_ld.abs().max())) > 1e-3 else 1.0, "nsf ld")
    hq = int(rng.choice([7, 30, 50]))
    sdr = recipes.rnvp_params(6000 + seed, dim, hq)
    r = amd.RNVP(dim, h_sizes=(hq,))
    r.load_state_dict(sdr)
    r.to(DEV)
    mask = recipes.bernoulli_mask(7000 + seed, rows, dim)
    ref_x, ref_ld = O.rnvp(x, sdr, mask)
    xg, ldg = r.forward(cuda(x), mask=cuda(mask))
    assert_close(xg, ref_x, RTOL, f"rnvp d={dim} h={hq}")
    assert_close(ldg, ref_ld, RTOL, "rnvp ld")


# ------------------------------------------------------------------ stacks (C1, C2, C4)
def build_ahf_stack(amd, layers, dim):
    flows = []
    for spec in layers:
        flows.append(ahf_module(amd, spec["params"], dim, spec["parity"]))
    return amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)


@pytest.mark.parametrize("tag", ["init", "trained"])
def test_g1_c1_stack(amd, golden, tag):
    """Config 1: 9 x AffineHalfFlow d=2 on half-moons (h = 1: the stack kernel's ragged variant, one launch)."""
    fx = golden(f"g1_c1_stack_{tag}")
    model = build_ahf_stack(amd, g1_layers(fx), 2)
    x = cuda(fx["x"])
    zs, ld = model.inverse(x)
    assert zs[0] is x and len(zs) == 10
    assert_close(ld, fx["ld_inv"], RTOL, "ld_inv")
    for i in fx["keep"]:
        assert_close(zs[i], fx[f"zs{i}"], RTOL, f"zs{i}")
    assert_close(model.base_log_prob(x), fx["base_log_prob"], RTOL, "base_log_prob")
    xs, ld_f = model.forward(zs[-1])
    assert_close(ld_f, fx["ld_fwd"], RTOL, "ld_fwd")
    assert_close(xs[-1], fx["xs9"], RTOL, "xs9")
    lp, total = model.log_prob(x, return_sum=True)
    mean = float(total.item()) / x.shape[0]
    ref = float(fx["mean_log_prob"])
    assert abs(mean - ref) <= RTOL * abs(ref)


@pytest.mark.parametrize("dim", [64, 256])
def test_g3_c2_stack(amd, golden, dim):
    """Configs 2 / 4: the benchmark stack's weights on the fixture rows."""
    fx = golden("g3_c2_stack")
    model = build_ahf_stack(amd, c2_layers(dim), dim)
    x = cuda(fx[f"d{dim}.x"])
    zs, ld = model.inverse(x)
    assert_close(zs[-1], fx[f"d{dim}.z_last"], RTOL, "z_last")
    assert_close(zs[4], fx[f"d{dim}.z_mid"], RTOL, "z_mid")
    assert_close(ld, fx[f"d{dim}.ld_inv"], RTOL, "ld_inv")
    cur = x
    for i, f in enumerate(reversed(model.flows)):
        cur, l1 = f.inverse(cur)
        assert_close(l1, fx[f"d{dim}.ld_incr"][i], RTOL, f"ld_incr[{i}]")
    xs, ld_f = model.forward(x)
    assert_close(xs[-1], fx[f"d{dim}.x_fwd_last"], RTOL, "x_fwd_last")
    assert_close(ld_f, fx[f"d{dim}.ld_fwd"], RTOL, "ld_fwd")
    lp, total = model.log_prob(x, return_sum=True)
    assert_close(lp, fx[f"d{dim}.ld_inv"] + fx[f"d{dim}.base_log_prob"], RTOL, "log_prob")
    ref = float(fx[f"d{dim}.mean_log_prob"])
    assert abs(float(total.item()) / x.shape[0] - ref) <= RTOL * abs(ref)
    # against the reference's fp64 run: the HIP fp32 path is as close to it as the reference's own fp32
    assert_close(zs[-1], fx[f"d{dim}.z_last_f64"].astype(np.float32), RTOL, "vs fp64")


def test_c2_full_size_properties(amd, O):
    """BASELINE configs[1] at full size (2^20 x 64): properties that need no CPU reference, and a 4,096-row slice of the
    full-size launches' every intermediate, log_det and log-prob against the oracle."""
    dim, rows = 64, 1 << 20
    model = build_ahf_stack(amd, c2_layers(dim), dim)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(rows, dim, device=DEV, generator=g)
    zs, ld = model.inverse(x)
    xs, ld_f = model.forward(zs[-1])
    # forward(inverse(x)) == x ; the two log-dets cancel
    assert float((xs[-1] - x).abs().max()) <= 2e-4 * float(x.abs().max())
    assert float((ld + ld_f).abs().max()) <= 1e-4 * float(ld.abs().max())
    assert torch.isfinite(zs[-1]).all() and torch.isfinite(ld).all()
    # MFMA kernel == generic kernel on the full batch (first layer of the inverse pass)
    f = model.flows[-1]
    y_fast, ld_fast = f.inverse(x)
    f.force_generic = True
    y_gen, ld_gen = f.inverse(x)
    f.force_generic = False
    assert_close(y_fast, y_gen, RTOL, "mfma vs generic y")
    assert_close(ld_fast, ld_gen, RTOL, "mfma vs generic ld")
    # the FULL-SIZE launches against the oracle: every intermediate, log_det and log-prob of the 2^20-row passes,
    # sliced at 4,096 rows spread over the batch (every workgroup position is sampled) -- as C3, C4 and C5 do
    sel = torch.arange(0, rows, rows // 4096, device=DEV)[:4096]
    xo = x[sel].cpu()
    layers = c2_layers(dim)
    ref_zs, ref_ld = O.flow_stack(xo, layers, inverse=True)
    assert len(zs) == len(ref_zs)
    for k in range(len(zs)):
        assert_close(zs[k][sel], ref_zs[k], RTOL, f"c2 full-size inverse zs[{k}] slice")
    assert_close(ld[sel], ref_ld, RTOL, "c2 full-size inverse log_det slice")
    ref_xs, ref_ldf = O.flow_stack(ref_zs[-1], layers, inverse=False)
    xs_o, ld_o = model.forward(zs[-1][sel].contiguous())  # (the forward pass above started from the GPU's own z)
    assert_close(xs_o[-1], ref_xs[-1], 2 * RTOL, "c2 forward of the oracle's z")
    ref_mean, ref_lp = O.mean_log_prob(xo, layers)
    lp_all, total_all = model.log_prob(x, return_sum=True)
    assert_close(lp_all[sel], ref_lp, RTOL, "c2 full-size log_prob slice")
    lp, total = model.log_prob(x[sel].contiguous(), return_sum=True)
    assert_close(lp, ref_lp, RTOL, "log_prob of the slice as its own launch")
    assert abs(float(total.item()) / 4096 - ref_mean) <= RTOL * abs(ref_mean)
    # sum of the per-row log-probs over the whole batch == the epilogue's fp64 sum
    assert abs(float(total_all.item()) - float(lp_all.double().sum())) <= 1e-9 * abs(float(total_all.item()))


@pytest.mark.parametrize("dim", [64, 32])
def test_fused_affine_stack_matches_layer_by_layer(amd, golden, O, dim):
    """Opt-in whole-stack fusion: one launch for all nine layers, same numbers as the layer-by-layer
    pass and as the reference (G3 fixture at d=64)."""
    layers = c2_layers(dim)
    plain = build_ahf_stack(amd, layers, dim)
    fused = amd.NormalizingFlowModel(amd.StandardNormal(dim), [amd.FusedAffineStack(list(plain.flows))]).to(DEV)
    x = cuda(golden("g3_c2_stack")["d64.x"]) if dim == 64 else cuda(recipes.gaussian(8, 300, dim))
    zs, ld = fused.inverse(x)
    assert len(zs) == 2  # input and output only
    zs_p, ld_p = plain.inverse(x)
    assert_close(zs[-1], zs_p[-1], 1e-6, "fused vs layer-by-layer z")
    assert_close(ld, ld_p, 1e-6, "fused vs layer-by-layer log_det")
    xs, ld_f = fused.forward(x)
    xs_p, ld_fp = plain.forward(x)
    assert_close(xs[-1], xs_p[-1], 1e-6, "forward")
    assert_close(ld_f, ld_fp, 1e-6, "forward log_det")
    if dim == 64:
        fx = golden("g3_c2_stack")
        assert_close(zs[-1], fx["d64.z_last"], RTOL, "vs reference z")
        assert_close(ld, fx["d64.ld_inv"], RTOL, "vs reference log_det")
    lp, total = fused.log_prob(x, return_sum=True)       # fused |z|^2 epilogue too
    lp_p, total_p = plain.log_prob(x, return_sum=True)
    assert_close(lp, lp_p, 1e-6, "log_prob")
    # ragged rows + big batch
    xr = cuda(recipes.gaussian(9, 1000 + 77, dim))
    assert_close(fused.inverse(xr)[0][-1], plain.inverse(xr)[0][-1], 1e-6, "ragged")
    assert list(fused.flows[0].state_dict())[0] == "layers.0.s_net.0.weight"


@pytest.mark.parametrize("kernel", ["split", "fp32"])
@pytest.mark.parametrize("dim", [64, 32, 128, 256])
def test_run_fusion_keeps_every_intermediate(amd, dim, kernel):
    """NormalizingFlow sends a run of equal AffineHalfFlow layers out as ONE launch that still writes every
    intermediate: same list of tensors, same log_det as launching the layers one by one."""
    layers = c2_layers(dim)
    model = build_ahf_stack(amd, layers, dim)
    for f in model.flows:
        select_kernel(f, kernel)
    assert model.fuse_affine_runs
    x = cuda(recipes.gaussian(21, 1000 + 13, dim))
    with torch.no_grad():
        for direction in ("inverse", "forward"):
            model.fuse_affine_runs = True
            zs_f, ld_f = getattr(model, direction)(x)
            model.fuse_affine_runs = False
            zs_u, ld_u = getattr(model, direction)(x)
            assert len(zs_f) == len(zs_u) == len(layers) + 1
            if kernel == "split" or dim <= 64:  # one launch: one buffer holds the run's outputs
                assert zs_f[1].data_ptr() + zs_f[1].numel() * 4 == zs_f[2].data_ptr()
            for i, (a, b) in enumerate(zip(zs_f, zs_u)):
                assert_close(a, b, 1e-6, f"{direction} tensor {i}")
            assert_close(ld_f, ld_u, 1e-6, f"{direction} log_det")
        model.fuse_affine_runs = True
        lp_f, tot_f = model.log_prob(x, return_sum=True)
        # the split stack kernel also does the standard-normal epilogue (log p and its fp64 sum) in the same launch
        assert model._logprob_done == (kernel == "split")
        lp_only = model.log_prob(x)
        model.fuse_affine_runs = False
        lp_u, tot_u = model.log_prob(x, return_sum=True)
        assert not model._logprob_done
    assert_close(lp_f, lp_u, 1e-6, "log_prob")
    assert torch.equal(lp_only, lp_f)
    assert abs(float(tot_f) - float(tot_u)) <= 1e-6 * abs(float(tot_u))
    assert abs(float(tot_f) - float(lp_f.double().sum())) <= 1e-9 * abs(float(tot_f))


def test_run_fusion_in_mixed_stacks(amd):
    """Runs are found between other layers, split where the shape changes, and skipped for layers that want
    gradients; a single AffineHalfFlow is not a run."""
    dim = 64
    mk = lambda i, **kw: ahf_module(amd, recipes.affine_half_params(400 + i, dim, **kw), dim, bool(i % 2), **kw)
    gl = amd.Glow(dim)  # fixed parameters: Glow's own init draws from torch's global generator
    gp = recipes.glow_params(411, dim)
    gl.P = gp["P"]
    gl.load_state_dict({k: gp[k] for k in "LSU"})
    flows = [mk(0), mk(1), mk(2), amd.ActNormFlow(dim).to(DEV), mk(3), mk(4, h_sizes=(16, 16, 16)),
             mk(5, h_sizes=(16, 16, 16)), gl.to(DEV), mk(6)]
    flows[3].load_state_dict(recipes.actnorm_params(410, dim))
    flows[3].data_dep_init_done = True
    model = amd.NormalizingFlow(flows).to(DEV)
    runs = model._affine_runs()
    assert sorted((s, len(r.layers)) for s, r in runs.items()) == [(0, 3), (5, 2)]
    x = cuda(recipes.gaussian(22, 777, dim))
    with torch.no_grad():
        for direction in ("forward", "inverse"):
            model.fuse_affine_runs = True
            zs_f, ld_f = getattr(model, direction)(x)
            model.fuse_affine_runs = False
            zs_u, ld_u = getattr(model, direction)(x)
            assert len(zs_f) == len(zs_u) == len(flows) + 1
            for i, (a, b) in enumerate(zip(zs_f, zs_u)):
                assert_close(a, b, 2e-6, f"{direction} tensor {i}")
            assert_close(ld_f, ld_u, 2e-6, f"{direction} log_det")
    # with gradients requested the layers run one by one through their autograd functions
    model.fuse_affine_runs = True
    zs, ld = model.inverse(x)
    assert ld.requires_grad and zs[-1].requires_grad


@pytest.mark.parametrize("case", ["big_inputs", "big_weights", "one_big_row", "tiny_inputs"])
def test_nsf_split_range_guard(amd, O, case):
    """NSF_CL on the split path: a half step whose conditioner operands leave the f16 range is redone with
    fp32 MFMAs in the same launch (inputs inside the tails still go through the spline)."""
    dim, K, n_h, rows = 32, 8, 8, 555
    sd = recipes.nsf_cl_params(150, dim, K, n_h)
    x = recipes.gaussian(151, rows, dim)
    if case == "big_inputs":      # far outside the tail bound: identity there, but the conditioner sees 1e5
        x = x * 1.0e5
        sd = {k: (v * 1e-5 if k.endswith(".0.weight") else v) for k, v in sd.items()}
    elif case == "big_weights":
        sd = {k: (v * 1e5 if k.endswith(".0.weight") else v) for k, v in sd.items()}
        x = x * 1e-5
    elif case == "one_big_row":
        x = x.clone()
        x[77] *= 3e4
        x[400] *= 3e4
        sd = {k: (v * 1e-3 if k.endswith(".0.weight") else v) for k, v in sd.items()}
    elif case == "tiny_inputs":
        x = x * 1e-6
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f.to(DEV)
    with torch.no_grad():
        for inverse in (False, True):
            ref_y, ref_ld = O.nsf_cl(x, sd, K, 3.0, inverse)
            y, ld = (f.inverse if inverse else f.forward)(cuda(x))
            select_kernel(f, "fp32")
            y32, ld32 = (f.inverse if inverse else f.forward)(cuda(x))
            select_kernel(f, "split")
            ok = torch.isfinite(ref_y).all(1) & torch.isfinite(ref_ld)
            assert torch.equal((torch.isfinite(y).all(1) & torch.isfinite(ld)).cpu(), ok)
            # ill-conditioned on purpose (1e5-scale products feeding a spline): the yardstick is the fp32 MFMA
            # kernel on the same inputs, which the guarded split path must reproduce, and the oracle loosely
            tol32 = 1e-5 if case == "tiny_inputs" else 2e-6  # tiny: not guarded, 2^-36 absolute operand error
            assert_close(y[ok.to(DEV)], y32[ok.to(DEV)], tol32, f"{case} y vs fp32 kernel")
            assert_close(ld[ok.to(DEV)], ld32[ok.to(DEV)], tol32, f"{case} ld vs fp32 kernel")
            assert_close(y[ok.to(DEV)], ref_y[ok], 2e-3 if case == "big_weights" else 2e-5, f"{case} y")
            assert_close(ld[ok.to(DEV)], ref_ld[ok], 2e-3 if case == "big_weights" else 2e-5, f"{case} ld")


def test_spline_block_run_keeps_every_intermediate(amd, golden):
    """[ActNorm, Glow, NSF_CL] blocks go out as one launch each that still writes both intermediates."""
    fx = golden("g6_c3_stack")
    model = build_c3(amd, fx)
    runs = model._affine_runs()
    assert sorted(runs) == [0, 3, 6] and all(type(r).__name__ == "_SplineBlockRun" for r in runs.values())
    x = cuda(fx["x"])
    with torch.no_grad():
        for direction in ("inverse", "forward"):
            model.fuse_affine_runs = True
            zs_f, ld_f = getattr(model, direction)(x)
            model.fuse_affine_runs = False
            zs_u, ld_u = getattr(model, direction)(x)
            assert len(zs_f) == len(zs_u) == 10
            for i, (a, b) in enumerate(zip(zs_f, zs_u)):
                assert_close(a, b, 3e-6, f"{direction} tensor {i}")
            assert_close(ld_f, ld_u, 3e-6, f"{direction} log_det")
        # density pass: the last block's launch also does the standard-normal log-prob epilogue and its fp64 sum
        model.fuse_affine_runs = True
        lp_f, tot_f = model.log_prob(x, return_sum=True)
        assert model._logprob_done
        model.fuse_affine_runs = False
        lp_u, tot_u = model.log_prob(x, return_sum=True)
        assert not model._logprob_done
    assert_close(lp_f, lp_u, 3e-6, "log_prob")
    assert abs(float(tot_f) - float(tot_u)) <= 3e-6 * abs(float(tot_u))
    assert abs(float(tot_f) - float(lp_f.double().sum())) <= 1e-9 * abs(float(tot_f))
    model.fuse_affine_runs = True


def test_log_det_accumulates_in_layer_order(amd):
    """NormalizingFlow's fused `log_det += ld` equals summing the per-layer log-dets."""
    dim = 64
    model = build_ahf_stack(amd, c2_layers(dim, 4), dim)
    x = cuda(recipes.gaussian(11, 333, dim))
    zs, ld = model.inverse(x)
    acc = torch.zeros(333, device=DEV)
    cur = x
    for f in reversed(model.flows):
        cur, l1 = f.inverse(cur)
        acc += l1
    assert torch.equal(cur, zs[-1])
    assert_close(ld, acc, 1e-6, "accumulated log_det")


# ------------------------------------------------------------------------- splines
@pytest.mark.parametrize("K", [5, 8])
def test_g4_rqs_direct(amd, golden, K):
    fx = golden("g4_rqs_direct")
    v, W, H, D = (cuda(fx[f"K{K}.{n}"]) for n in "vWHD")
    for inv, name in ((False, "fwd"), (True, "inv")):
        out, lad = amd.rqs(v, W, H, D, inverse=inv, tail_bound=3.0)
        # extreme (W, H, D): the budget includes the reference's own fp32-vs-fp64 distance
        # (a stress fixture by construction -- knots a few ulps from the samples, |W|, |H|, |D| up to 8: the reference's
        #  own fp32 run is up to 1.5e-4 from its float64 run here, so the 5e-5 cap on the head-room does not apply)
        assert_parity(out, fx[f"K{K}.out_{name}"], fx[f"K{K}.out64_{name}"], f"g4 K{K} out_{name}", max_widening=None)
        assert_parity(lad, fx[f"K{K}.lad_{name}"], fx[f"K{K}.lad64_{name}"], f"g4 K{K} lad_{name}", max_widening=None)
    out, lad = amd.rqs(v, W, H, D, inverse=False, tail_bound=3.0)
    out, lad = out.cpu(), lad.cpu()
    assert out[5] == 3.5 and lad[5] == 0 and out[6] == -7.0 and torch.isnan(out[7]) and lad[7] == 0


def test_rqs_all_outside_is_identity_and_too_many_bins_raises(amd):
    v = torch.tensor([4.0, -5.0, float("nan")], device=DEV)
    W = torch.zeros(3, 5, device=DEV)
    out, lad = amd.rqs(v, W, W.clone(), torch.zeros(3, 4, device=DEV), tail_bound=3.0)
    assert torch.equal(out[:2], v[:2]) and torch.isnan(out[2]) and (lad == 0).all()
    Wbig = torch.zeros(2, 1001, device=DEV)
    with pytest.raises(ValueError):
        amd.rqs(torch.zeros(2, device=DEV), Wbig, Wbig.clone(), torch.zeros(2, 1000, device=DEV), tail_bound=3.0)


@pytest.mark.parametrize("cfg", [(32, 8, 8, 1.0), (32, 8, 16, 1.0), (2, 8, 16, 1.0), (6, 5, 8, 1.0),
                                 (32, 8, 8, 2.0), (2, 8, 16, 2.0)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_g5_nsf_cl_layer(amd, golden, cfg, kernel):
    dim, K, n_h, gain = cfg
    fx = golden("g5_nsf_cl_layer")
    tag = f"d{dim}_K{K}_h{n_h}" + ("" if gain == 1.0 else "_stress")
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(recipes.nsf_cl_params(500 + dim + n_h, dim, K, n_h, gain=gain))
    f = select_kernel(f.to(DEV), kernel)
    if dim == 32 and kernel != "generic":
        assert (f._split_image(torch.device(DEV, 0)) is not None) == (kernel == "split")
    z = cuda(fx[f"{tag}.z"])
    for name, fn in (("fwd", f.forward), ("inv", f.inverse)):
        x, ld = fn(z)
        if gain == 1.0:  # nn.Linear-scale weights: the plain 1e-5 rule
            assert_close(x, fx[f"{tag}.{name}"], RTOL, f"{tag}.{name}")
            assert_close(ld, fx[f"{tag}.ld_{name}"], RTOL, f"{tag}.ld_{name}")
        else:  # stress weights: budget widened by the reference's own fp32-vs-fp64 distance
            assert_parity(x, fx[f"{tag}.{name}"], fx[f"{tag}.{name}64"], f"g5 {tag}.{name}", max_widening=None)
            assert_parity(ld, fx[f"{tag}.ld_{name}"], fx[f"{tag}.ld_{name}64"], f"g5 {tag}.ld_{name}", max_widening=None)
    assert ld.device.type == "cuda"  # the reference allocates log_det on the CPU (:250); fixed here


def build_c3(amd, fx=None):
    flows = []
    for i in range(3):
        an = amd.ActNormFlow(32)
        if fx is not None:
            an.load_state_dict({"s": t(fx[f"actnorm{i}.s"]), "t": t(fx[f"actnorm{i}.t"])})
            an.data_dep_init_done = True
        gl = amd.Glow(32)
        gp = recipes.glow_params(600 + i, 32)
        gl.P = gp["P"]
        gl.load_state_dict({"L": gp["L"], "S": gp["S"], "U": gp["U"]})
        sp = amd.NSF_CL(32, K=8, B=3, n_h=8)
        sp.load_state_dict(recipes.nsf_cl_params(610 + i, 32, 8, 8))
        flows += [an, gl, sp]
    return amd.NormalizingFlowModel(amd.StandardNormal(32), flows).to(DEV)


def test_g6_c3_stack(amd, golden):
    """Config 3: 3 x [ActNorm, Glow, NSF_CL] d=32 K=8."""
    fx = golden("g6_c3_stack")
    x = cuda(fx["x"])
    model = build_c3(amd, fx)
    zs, ld = model.inverse(x)
    # nine modules deep: the plain 1e-5 rule, plus the reference's own fp32-vs-fp64 distance
    # (1e-6 .. 2.5e-6 here) as head-room
    assert_parity(zs[-1], fx["z_last"], fx["z_last64"], "g6 z_last")
    assert_close(zs[5], fx["z_mid"], RTOL, "z_mid")
    assert_parity(ld, fx["ld_inv"], fx["ld_inv64"], "g6 ld_inv")
    xs, ld_f = model.forward(x)
    assert_parity(xs[-1], fx["x_fwd_last"], fx["x_fwd_last64"], "g6 x_fwd_last")
    assert_parity(ld_f, fx["ld_fwd"], fx["ld_fwd64"], "g6 ld_fwd")


def test_fused_spline_block_matches_the_three_layers(amd, golden):
    """Opt-in [ActNorm, Glow, NSF_CL] fusion: same numbers as the three modules (and as the
    reference's C3 stack), in one launch per block and direction."""
    fx = golden("g6_c3_stack")
    x = cuda(fx["x"])
    plain = build_c3(amd, fx)
    fused = amd.NormalizingFlowModel(amd.StandardNormal(32), [
        amd.FusedSplineBlock(plain.flows[3 * i], plain.flows[3 * i + 1], plain.flows[3 * i + 2]) for i in range(3)
    ]).to(DEV)
    zs, ld = fused.inverse(x)
    assert len(zs) == 4  # one tensor per block: the intermediates inside a block are gone
    assert_parity(zs[-1], fx["z_last"], fx["z_last64"], "fused z_last")
    assert_parity(ld, fx["ld_inv"], fx["ld_inv64"], "fused ld_inv")
    xs, ld_f = fused.forward(x)
    assert_parity(xs[-1], fx["x_fwd_last"], fx["x_fwd_last64"], "fused x_fwd_last")
    assert_parity(ld_f, fx["ld_fwd"], fx["ld_fwd64"], "fused ld_fwd")
    zs_p, ld_p = plain.inverse(x)
    assert_close(zs[-1], zs_p[-1], 2e-6, "fused vs unfused")
    assert_close(ld, ld_p, 2e-6, "fused vs unfused log_det")
    # a parameter update invalidates the folded affine map
    with torch.no_grad():
        plain.flows[0].t.add_(0.25)
    zs2, _ = fused.inverse(x)
    zs2_p, _ = plain.inverse(x)
    assert_close(zs2[-1], zs2_p[-1], 2e-6, "after update")
    assert not torch.equal(zs2[-1], zs[-1])
    # state_dict nests the three modules
    assert list(fused.flows[0].state_dict())[:5] == ["actnorm.s", "actnorm.t", "glow.L", "glow.S", "glow.U"]


def test_g6_actnorm_data_dependent_init(amd, golden):
    """First inverse call initialises each ActNorm from the batch it sees (affine_constant_flow.py:42-50)."""
    fx = golden("g6_c3_stack")
    torch.manual_seed(6)
    model = build_c3(amd, None)
    x = cuda(fx["x"])
    zs, ld = model.inverse(x)
    for i in range(3):
        an = model.flows[3 * i]
        assert an.data_dep_init_done is True
        assert_close(an.s, fx[f"actnorm{i}.s"], 2e-5, f"actnorm{i}.s")
        assert_close(an.t, fx[f"actnorm{i}.t"], 2e-5, f"actnorm{i}.t")
    assert_close(zs[-1], fx["z_last_first_call"], 2e-5, "first call z")
    assert_close(ld, fx["ld_first_call"], 2e-5, "first call ld")


def test_c3_full_size_round_trip(amd, O):
    """Config 3 at 2^20 rows: inverse(forward(x)) == x and log-dets cancel (reference: ~7e-6 abs); and -- as C2, C4
    and C5 have it -- every tensor of both passes, log_det and log_prob on a 4,096-row slice spread over the batch
    against the oracle (a self-consistent wrong spline at scale would pass the round trip)."""
    rows = 1 << 20
    model = build_c3(amd, None)
    for i in range(3):
        model.flows[3 * i].load_state_dict(recipes.actnorm_params(630 + i, 32))
        model.flows[3 * i].data_dep_init_done = True
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(rows, 32, device=DEV, generator=g)
    xs, ld_f = model.forward(x)
    zs, ld_i = model.inverse(xs[-1])
    assert float((zs[-1] - x).abs().max()) <= 5e-4
    assert float((ld_f + ld_i).abs().max()) <= 5e-4 * max(1.0, float(ld_f.abs().max()))
    outside = (x.abs() > 3).float().mean().item()
    assert 0.001 < outside < 0.005  # ~0.27 % of N(0,1) falls in the identity tails
    # the oracle slice: rows chosen across the batch (so every workgroup position is sampled), both directions
    layers = c3_layers(None)
    sel = torch.arange(0, rows, rows // 4096, device=DEV)[:4096]
    xo = x[sel].cpu()
    zs_d, ld_d = model.inverse(x)          # the density direction bench.py --workload c3 times, full batch
    lp_d, total = model.log_prob(x, return_sum=True)
    worst = 0.0
    for name, got_list, got_ld, inverse in (("inverse", zs_d, ld_d, True), ("forward", xs, ld_f, False)):
        ref_list, ref_ld = O.flow_stack(xo, layers, inverse=inverse)
        # float64 run of the same oracle: the head-room the spline legitimately needs near a knot (helpers.assert_parity)
        ref_list64, ref_ld64 = O.flow_stack(xo.double(), _layers_f64(layers), inverse=inverse)
        assert len(got_list) == len(ref_list) == 10
        for k in range(1, 10):
            worst = max(worst, assert_parity(got_list[k][sel], ref_list[k], ref_list64[k].float().numpy(),
                                             f"c3 {name} tensor {k}", max_widening=5e-5))
        worst = max(worst, assert_parity(got_ld[sel], ref_ld, ref_ld64.float().numpy(), f"c3 {name} log_det",
                                         max_widening=5e-5))
    ref_mean, ref_lp = O.mean_log_prob(xo, layers)
    assert_close(lp_d[sel], ref_lp, RTOL, "c3 log_prob slice")
    lp_sel, total_sel = model.log_prob(x[sel].contiguous(), return_sum=True)
    assert abs(float(total_sel.item()) / 4096 - ref_mean) <= RTOL * abs(ref_mean)
    assert abs(float(total.item()) - float(lp_d.double().sum())) <= 1e-9 * abs(float(total.item()))
    print(f"c3 full size: worst normwise error of the oracle slice {worst:.2e}")


def _layers_f64(layers):
    """The same layer specs with float64 parameters (the oracle computes in the dtype it is given)."""
    out = []
    for spec in layers:
        s2 = dict(spec)
        s2["params"] = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v)
                        for k, v in spec["params"].items()}
        out.append(s2)
    return out


@pytest.mark.parametrize("dim", [4, 16, 32, 64, 128])
def test_glow_and_actnorm_layers_vs_oracle(amd, O, dim):
    """Glow (MFMA kernel for d in {16,32,64,128}, generic otherwise) and AffineConstantFlow."""
    gp = recipes.glow_params(40 + dim, dim)
    gl = amd.Glow(dim)
    gl.P = gp["P"]
    gl.load_state_dict({k: gp[k] for k in "LSU"})
    gl.to(DEV)
    x = recipes.gaussian(41 + dim, 1037, dim)
    for inverse in (False, True):
        ref_y, ref_ld = O.glow(x, gp["P"], gp["L"], gp["S"], gp["U"], inverse)
        y, ld = (gl.inverse if inverse else gl.forward)(cuda(x))
        assert_close(y, ref_y, 2e-5 if inverse else RTOL, "glow y")  # inverse: dense torch.inverse on both sides
        assert abs(float(ld) - float(ref_ld)) <= 1e-5 * max(1.0, abs(float(ref_ld))) and ld.dim() == 0
        gl.force_generic = True
        gl._w_img = {}
        y_gen, _ = (gl.inverse if inverse else gl.forward)(cuda(x))
        gl.force_generic = False
        gl._w_img = {}
        assert_close(y, y_gen, RTOL, "mfma vs generic")
    ap = recipes.actnorm_params(42 + dim, dim)
    an = amd.AffineConstantFlow(dim)
    an.load_state_dict(ap)
    an.to(DEV)
    for inverse in (False, True):
        ref_y, ref_ld = O.affine_const(x, ap["s"], ap["t"], inverse)
        y, ld = (an.inverse if inverse else an.forward)(cuda(x))
        assert_close(y, ref_y, RTOL, "affine_const y")
        assert_close(ld, ref_ld, RTOL, "affine_const ld")
        assert tuple(ld.shape) == (1,)


# --------------------------------------------------------------------------- RNVP
@pytest.mark.parametrize("dim", [50, 800, 784])
@pytest.mark.parametrize("kernel", KERNELS)
def test_g7_rnvp(amd, golden, dim, kernel):
    fx = golden("g7_rnvp")
    f = amd.RNVP(dim, h_sizes=(50,))
    f.load_state_dict(recipes.rnvp_params(700 + dim, dim, 50))
    f = select_kernel(f.to(DEV), kernel)
    if dim >= 64 and kernel != "generic":
        assert (f._split_image(torch.device(DEV, 0)) is not None) == (kernel == "split")
    z = cuda(fx[f"d{dim}.z"])
    mask = unpack_mask(fx[f"d{dim}.mask_bits"], dim).to(DEV)
    x, ld = f.forward(z, mask=mask)
    assert_close(x, fx[f"d{dim}.x"], RTOL, "x")
    assert_close(ld, fx[f"d{dim}.ld"], RTOL, "ld")
    assert not hasattr(f, "inverse")
    # without a mask argument a Bernoulli(0.5) mask is drawn per element per call
    x1, _ = f.forward(z)
    x2, _ = f.forward(z)
    assert not torch.equal(x1, x2)
    # ... from the library's counter-based generator: reproducible under torch.manual_seed, and a
    # seeded call equals an explicit-mask call with the mask mask_for() reports
    torch.manual_seed(5)
    xa, lda = f.forward(z)
    torch.manual_seed(5)
    xb, ldb = f.forward(z)
    assert torch.equal(xa, xb) and torch.equal(lda, ldb)
    xs, lds = f.forward(z, seed=1234)
    m = f.mask_for(1234, z.shape[0])
    xm, ldm = f.forward(z, mask=m)
    # (the in-kernel-mask path evaluates the same formula in its binary-mask form,
    #  x = (1 - gate) t + (m ? z : gate z), so the two agree to rounding, not bit for bit)
    assert_close(xs, xm, 1e-6, "seeded vs explicit mask x")
    assert_close(lds, ldm, 1e-6, "seeded vs explicit mask log_det")
    assert set(m.unique().tolist()) <= {0.0, 1.0}


def test_rnvp_in_kernel_mask_statistics(amd):
    """The generated mask is Bernoulli(0.5) per element: mean, per-column and per-row balance, and
    no correlation between neighbouring rows / columns or between seeds."""
    f = amd.RNVP(800, h_sizes=(50,)).to(DEV)
    m = f.mask_for(99, 4096)
    assert abs(float(m.mean()) - 0.5) < 2e-3
    assert float((m.mean(0) - 0.5).abs().max()) < 0.05 and float((m.mean(1) - 0.5).abs().max()) < 0.09
    c = lambda a, b: float(((a - 0.5) * (b - 0.5)).mean() * 4)  # noqa: E731
    assert abs(c(m[1:], m[:-1])) < 5e-3 and abs(c(m[:, 1:], m[:, :-1])) < 5e-3
    assert abs(c(m, f.mask_for(100, 4096))) < 5e-3


def test_g8_mnf_linear_sample_z(amd, golden):
    """MNFLinear(800, 50).sample_z(64) with the reference's captured noise and masks (row a14)."""
    fx = golden("g8_sample_z")
    layer = amd.MNFLinear(800, 50)
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(800 + i, 800, 50))
    layer.load_state_dict({"q0_mean": t(fx["q0_mean"]), "q0_log_var": t(fx["q0_log_var"])}, strict=False)
    layer.to(DEV)
    masks = [unpack_mask(fx[f"mask{i}_bits"], 800).to(DEV) for i in range(2)]
    z, ld = layer.sample_z(64, eps=cuda(fx["eps"]), masks=masks)
    assert_close(z, fx["z"], RTOL, "z")
    assert_close(ld, fx["log_det"], RTOL, "log_det")
    # the prologue fused into the first flow's kernel (the default) against mnf_sample_z0 + the flows
    with torch.no_grad():
        assert layer._fused_prologue_ok(layer.flow_q.flows[0], cuda(fx["eps"]))
        layer.fuse_prologue = False
        z_u, ld_u = layer.sample_z(64, eps=cuda(fx["eps"]), masks=masks)
        torch.manual_seed(7)
        zs_u, lds_u = layer.sample_z(500, eps=cuda(recipes.gaussian(88, 500, 800)))
        layer.fuse_prologue = True
        z_f, ld_f = layer.sample_z(64, eps=cuda(fx["eps"]), masks=masks)
        torch.manual_seed(7)
        zs_f, lds_f = layer.sample_z(500, eps=cuda(recipes.gaussian(88, 500, 800)))
    assert_close(z_f, fx["z"], RTOL, "fused z vs reference")
    assert_close(z_f, z_u, 1e-6, "fused vs unfused z")
    assert_close(ld_f, ld_u, 1e-6, "fused vs unfused log_det")
    assert_close(zs_f, zs_u, 1e-6, "fused vs unfused z (in-kernel masks)")
    assert_close(lds_f, lds_u, 1e-6, "fused vs unfused log_det (in-kernel masks)")
    # default path: noise and masks drawn on the device; shapes and finiteness
    z2, ld2 = layer.sample_z(32)
    assert z2.shape == (32, 800) and ld2.shape == (32,) and torch.isfinite(z2).all()
    out = layer.forward(torch.randn(16, 800, device=DEV))
    assert out.shape == (16, 50) and torch.isfinite(out).all()
    assert torch.isfinite(layer.kl_div())
    ref_keys = ["W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var", "r0_c", "r0_b1", "r0_b2"]
    assert list(layer.state_dict())[:9] == ref_keys


def test_mnf_linear_50_10_sample_z_vs_oracle(amd, O):
    """Config 5's second workload: MNFLinear(50, 10) -- its flow is 50 wide, i.e. not a multiple of the kernels'
    16-dim groups: the ragged RNVP variants, with the sample_z prologue fused into the first one."""
    n_in, rows = 50, 777
    layer = amd.MNFLinear(n_in, 10)
    sds = [recipes.rnvp_params(850 + i, n_in, 50) for i in range(2)]
    for f, sd in zip(layer.flow_q.flows, sds):
        f.load_state_dict(sd)
    q0_mean, q0_log_var = recipes.gaussian(851, 1, n_in)[0], recipes.gaussian(852, 1, n_in)[0] * 0.3 - 2.0
    layer.load_state_dict({"q0_mean": q0_mean, "q0_log_var": q0_log_var}, strict=False)
    layer.to(DEV)
    eps = recipes.gaussian(853, rows, n_in)
    masks = [recipes.bernoulli_mask(854 + i, rows, n_in) for i in range(2)]
    z = q0_mean + q0_log_var.exp().sqrt() * eps   # mnf_linear.py:59-62
    ld_ref = 0
    for sd, m in zip(sds, masks):
        z, l1 = O.rnvp(z, sd, m)
        ld_ref = ld_ref + l1
    with torch.no_grad():
        assert layer._fused_prologue_ok(layer.flow_q.flows[0], cuda(eps))
        z_f, ld_f = layer.sample_z(rows, eps=cuda(eps), masks=[cuda(m) for m in masks])
        layer.fuse_prologue = False
        z_u, ld_u = layer.sample_z(rows, eps=cuda(eps), masks=[cuda(m) for m in masks])
    assert_close(z_f, z, RTOL, "z")
    assert_close(ld_f, ld_ref, RTOL, "log_det")
    assert_close(z_u, z_f, 1e-6, "fused vs unfused z")
    assert_close(ld_u, ld_f, 1e-6, "fused vs unfused log_det")


# ------------------------------------------------------- contracts of the boundary
def test_g9_log_det_shapes(amd, golden):
    fx = golden("g9_logdet_shapes")
    x = cuda(recipes.gaussian(900, 8, 4))
    mods = {"affine_half": amd.AffineHalfFlow(4, False), "nsf_cl": amd.NSF_CL(4, K=5),
            "actnorm": amd.ActNormFlow(4), "affine_const": amd.AffineConstantFlow(4),
            "glow": amd.Glow(4), "rnvp": amd.RNVP(4)}
    for name, m in mods.items():
        m.to(DEV)
        assert tuple(m.forward(x)[1].shape) == tuple(fx[f"{name}.fwd"]), name
        if name != "rnvp":
            assert tuple(m.inverse(x)[1].shape) == tuple(fx[f"{name}.inv"]), name
    stack = amd.NormalizingFlow([mods["actnorm"], mods["glow"], mods["nsf_cl"]])
    zs, ld = stack.forward(x)
    assert tuple(ld.shape) == tuple(fx["stack.ld_shape"]) and zs[0] is x
    assert len(zs) == int(fx["stack.n_intermediates"])


def test_state_dict_keys_match_reference(amd):
    assert list(amd.AffineHalfFlow(64, False).state_dict()) == list(recipes.affine_half_params(0, 64))
    assert list(amd.NSF_CL(32, K=8).state_dict()) == list(recipes.nsf_cl_params(0, 32, 8, 8))
    assert list(amd.RNVP(800, (50,)).state_dict()) == list(recipes.rnvp_params(0, 800, 50))
    assert list(amd.Glow(4).state_dict()) == ["L", "S", "U"]
    assert list(amd.ActNormFlow(4).state_dict()) == ["s", "t"]


def test_launches_are_graph_capturable(amd):
    """The library never allocates or synchronises, so a whole log_prob pass (9 coupling kernels +
    epilogue) can be captured into a HIP graph and replayed on new data."""
    dim, rows = 64, 4096
    model = build_ahf_stack(amd, c2_layers(dim), dim)
    replay = model.graphed_log_prob(cuda(recipes.gaussian(21, rows, dim)))
    x_new = cuda(recipes.gaussian(22, rows, dim))
    lp_g, total_g = replay(x_new)
    torch.cuda.synchronize()
    with torch.no_grad():
        lp_eager, total_eager = model.log_prob(x_new, return_sum=True)
    assert torch.equal(lp_g, lp_eager)
    assert abs(float(total_g) - float(total_eager)) <= 1e-9 * abs(float(total_eager))
    # small batches are launch-bound: the replay must not be slower than the eager pass
    import time
    def timed(fn, n=50):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    with torch.no_grad():
        t_eager = timed(lambda: model.log_prob(x_new, return_sum=True))
    t_graph = timed(lambda: replay(x_new))
    print(f"4096-row pass: eager {t_eager * 1e6:.0f} us, graph replay {t_graph * 1e6:.0f} us")
    assert t_graph < 3.0 * t_eager  # informational (measured 61 vs 284 us); loose so a busy host cannot flake it


def test_cpu_input_is_an_error_not_a_fallback(amd):
    f = amd.AffineHalfFlow(64, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f.forward(torch.zeros(4, 64))
    with pytest.raises(ValueError):
        f.to(DEV).forward(torch.zeros(4, 32, device=DEV))


def test_foreign_duck_typed_flow_in_stack(amd):
    """NormalizingFlow calls unknown flows by name and adds their (possibly scalar) log_det."""

    class Shift(torch.nn.Module):
        def forward(self, z):
            return z + 1.0, torch.tensor(0.5, device=z.device)

        def inverse(self, x):
            return x - 1.0, torch.tensor(-0.5, device=x.device)

    f = ahf_module(amd, recipes.affine_half_params(3, 64), 64, False)
    stack = amd.NormalizingFlow([Shift(), f])
    x = cuda(recipes.gaussian(4, 20, 64))
    xs, ld = stack.forward(x)
    y, l1 = f.forward(x + 1.0)
    assert torch.equal(xs[-1], y)
    assert_close(ld, l1 + 0.5, 1e-6)
    zs, ldi = stack.inverse(xs[-1])
    assert float((zs[-1] - x).abs().max()) < 1e-4


def test_pipelined_all_reduce_on_rccl(amd):
    """The N > 1 step of bench.py on RCCL itself (a one-rank group on this one-GPU box): the 16-byte all-reduce is
    left in flight (async_op) while the next pass is enqueued, and collected afterwards."""
    import socket

    import torch.distributed as dist
    from torch_mnf_amd.dist import reduce_sum_count

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(DEV, 0))
    except Exception as err:  # the box cannot bring RCCL up at all: nothing of ours to test here
        pytest.skip(f"RCCL process group unavailable: {err}")
    try:
        dim, rows = 64, 5000
        model = build_ahf_stack(amd, c2_layers(dim), dim)
        x = cuda(recipes.gaussian(31, rows, dim))
        with torch.no_grad():
            lp, total = model.log_prob(x, return_sum=True)
            first = reduce_sum_count(total, rows, async_op=True, force_collective=True)
            lp2, total2 = model.log_prob(2 * x, return_sum=True)  # enqueued while the reduction is in flight
            second = reduce_sum_count(total2, rows, async_op=True, force_collective=True)
            m1, m2 = float(first.result()), float(second.result())
            s_sync, c_sync = reduce_sum_count(total, rows, force_collective=True)
        assert abs(m1 - float(lp.double().mean())) <= 1e-9 * abs(m1)
        assert abs(m2 - float(lp2.double().mean())) <= 1e-9 * abs(m2)
        assert float(c_sync) == rows and abs(float(s_sync) / rows - m1) <= 1e-12 * abs(m1)
    finally:
        dist.destroy_process_group()


def test_every_even_width_up_to_256_matches_the_generic_kernel(amd):
    """Sweep of all 128 even dims: the MFMA path (full tiles, or the ragged variant for a narrow half) against the
    shape-generic kernel, one layer, both directions, a row count that leaves a partial tile."""
    worst = 0.0
    for dim in range(2, 257, 2):
        sd = recipes.affine_half_params(900 + dim, dim, s_last_gain=1.5)
        f = ahf_module(amd, sd, dim, bool((dim // 2) % 2))
        assert f._split_image(torch.device(DEV, 0)) is not None, dim
        x = cuda(recipes.gaussian(dim, 37, dim))
        for inverse in (False, True):
            y, ld = f.forward(x, inverse=inverse)
            f.force_generic = True
            y_g, ld_g = f.forward(x, inverse=inverse)
            f.force_generic = False
            err = max(float((y - y_g).abs().max() / y_g.abs().max()), float((ld - ld_g).abs().max() / ld_g.abs().max()))
            worst = max(worst, err)
            assert err <= RTOL, (dim, inverse, err)
    assert worst > 0.0  # two different kernels, not the same one twice


def test_rnvp_every_width_from_49_to_130_matches_the_generic_kernel(amd):
    """Sweep over 82 consecutive widths (every residue mod 16, both access modes of the ragged variants) with the
    in-kernel mask: MFMA path against the shape-generic kernel."""
    for dim in range(49, 131):
        f = amd.RNVP(dim, h_sizes=(50,))
        f.load_state_dict(recipes.rnvp_params(1200 + dim, dim, 50))
        f.to(DEV)
        assert f._split_image(torch.device(DEV, 0)) is not None, dim
        z = cuda(recipes.gaussian(dim, 141, dim))
        x, ld = f.forward(z, seed=dim)
        f.force_generic = True
        x_g, ld_g = f.forward(z, seed=dim)
        f.force_generic = False
        assert_close(x, x_g, RTOL, f"x d={dim}")
        assert_close(ld, ld_g, RTOL, f"ld d={dim}")


@pytest.mark.parametrize("kw", [dict(scale=False), dict(shift=False)], ids=["nice", "noshift"])
@pytest.mark.parametrize("dim", [2, 10, 64, 256])
def test_nice_and_no_shift_variants_on_the_mfma_path(amd, O, dim, kw):
    """scale=False (NICE, readme.md) / shift=False: the absent net is the zero function in the reference
    (affine_half_flow.py:38); here its operands are structural zeros of the same MFMA kernels -- s = 0 (log_det
    exactly 0) or t = 0.  Single layers and a 3-layer run, both directions, vs the oracle."""
    sds = [recipes.affine_half_params(1300 + dim + i, dim, s_last_gain=2.0, **kw) for i in range(3)]
    flows = [ahf_module(amd, sd, dim, bool(i % 2), **kw) for i, sd in enumerate(sds)]
    assert all(f._split_image(torch.device(DEV, 0)) is not None for f in flows)
    x = recipes.gaussian(1301 + dim, 333, dim)
    for inverse in (False, True):
        ref_y, ref_ld = O.affine_half(x, sds[0], False, inverse, **kw)
        y, ld = flows[0].forward(cuda(x), inverse=inverse)
        assert_close(y, ref_y, RTOL, "y")
        if kw.get("scale", True):
            assert_close(ld, ref_ld, RTOL, "ld")
        else:
            assert float(ld.abs().max()) == 0.0 and float(torch.as_tensor(ref_ld).abs().max()) == 0.0
    model = amd.NormalizingFlow(flows).to(DEV)
    with torch.no_grad():
        zs, ld = model.inverse(cuda(x))
        assert zs[1].data_ptr() + zs[1].numel() * 4 == zs[2].data_ptr()  # one launch
        cur, ld_ref = x, 0
        for i in (2, 1, 0):
            cur, l1 = O.affine_half(cur, sds[i], bool(i % 2), True, **kw)
            ld_ref = ld_ref + l1
        assert_close(zs[-1], cur, RTOL, "run z")
        if kw.get("scale", True):
            assert_close(ld, ld_ref, RTOL, "run log_det")
        else:
            assert float(ld.abs().max()) == 0.0


@pytest.mark.parametrize("kernel", ["split", "fp32"])
@pytest.mark.parametrize("dim", [64, 6, 256])
@pytest.mark.parametrize("h_sizes", [(20, 20, 20), (8, 24, 17), (30, 12, 32), (3, 1, 2)])
def test_affine_half_any_three_hidden_widths_up_to_32(amd, O, dim, h_sizes, kernel):
    """Three hidden layers of any widths <= 32 run on the MFMA kernels at the next of 16 / 24 / 32 (structural-zero
    units): single layer and 3-layer run vs the oracle and the generic kernel."""
    if dim == 256 and not 16 < max(h_sizes) <= 24:
        pytest.skip("d = 256 only has the kernels that run the hidden layers at 24 units")
    sds = [recipes.affine_half_params(1400 + dim + i, dim, h_sizes=h_sizes, s_last_gain=2.0) for i in range(3)]
    flows = [ahf_module(amd, sd, dim, bool(i % 2), kernel, h_sizes=h_sizes) for i, sd in enumerate(sds)]
    # (a hidden layer of fewer than 4 units sends the layer to the fp32 MFMA kernels by itself: flows._MIN_SPLIT_HIDDEN)
    assert all((f._split_image(torch.device(DEV, 0)) is not None) == (kernel == "split" and min(h_sizes) >= 4)
               for f in flows)
    assert all(f._packed(torch.device(DEV, 0))[1] is not None for f in flows)
    x = recipes.gaussian(1401 + dim, 333, dim)
    for inverse in (False, True):
        ref_y, ref_ld = O.affine_half(x, sds[1], True, inverse)
        y, ld = flows[1].forward(cuda(x), inverse=inverse)
        assert_close(y, ref_y, RTOL, "y")
        assert_close(ld, ref_ld, RTOL, "ld")
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)
    layers = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd} for i, sd in enumerate(sds)]
    with torch.no_grad():
        lp, total = model.log_prob(cuda(x), return_sum=True)
    ref_mean, ref_lp = O.mean_log_prob(x, layers)
    assert_close(lp, ref_lp, RTOL, "log_prob")
    assert abs(float(total) / 333 - ref_mean) <= RTOL * abs(ref_mean)


def test_g10_padded_shapes_vs_reference(amd, golden):
    """Reference-generated fixture for the shapes the kernels run padded: AffineHalfFlow with narrow halves, odd hidden
    widths or an absent net, RNVP with dim % 16 != 0 / other hidden widths -- every one on an MFMA kernel."""
    fx = golden("g10_padded_shapes")
    for k, (tag, (dim, kw)) in enumerate(recipes.G10_AHF.items()):
        f = ahf_module(amd, recipes.affine_half_params(1000 + 10 * k, dim, s_last_gain=2.0, **kw), dim, bool(k % 2), **kw)
        assert f._split_image(torch.device(DEV, 0)) is not None, tag
        z = cuda(fx[f"ahf.{tag}.z"])
        for name, fn in (("fwd", f.forward), ("inv", f.inverse)):
            y, ld = fn(z)
            assert_close(y, fx[f"ahf.{tag}.{name}"], RTOL, f"{tag}.{name}")
            if kw.get("scale", True):
                assert_close(ld, fx[f"ahf.{tag}.ld_{name}"], RTOL, f"{tag}.ld_{name}")
            else:
                assert float(ld.abs().max()) == 0.0
    for k, (tag, (dim, hid)) in enumerate(recipes.G10_RNVP.items()):
        f = amd.RNVP(dim, h_sizes=(hid,))
        f.load_state_dict(recipes.rnvp_params(1100 + 10 * k, dim, hid))
        f.to(DEV)
        assert f._split_image(torch.device(DEV, 0)) is not None, tag
        x, ld = f.forward(cuda(fx[f"rnvp.{tag}.z"]), mask=unpack_mask(fx[f"rnvp.{tag}.mask_bits"], dim).to(DEV))
        assert_close(x, fx[f"rnvp.{tag}.x"], RTOL, f"{tag}.x")
        assert_close(ld, fx[f"rnvp.{tag}.ld"], RTOL, f"{tag}.ld")


@pytest.mark.parametrize("kernel", ["split", "fp32"])
@pytest.mark.parametrize("dim,K,n_h", [(32, 8, 8), (32, 8, 16), (32, 5, 8), (32, 5, 16), (64, 8, 8), (64, 5, 8), (64, 8, 16),
                                       (32, 8, 12), (64, 5, 3), (32, 5, 1), (32, 10, 8), (32, 10, 16), (16, 10, 8)])
def test_nsf_cl_mfma_shape_matrix(amd, O, dim, K, n_h, kernel):
    """Every (dim, K, n_h) triple with an MFMA spline kernel -- incl. the reference's default K = 5 and the n_h = 16 of
    its tests at both dims; an n_h other than 8 / 16 runs at the next one up (n_h > 16: the run-time-shaped kernel,
    tests/test_hip_round6.py): result vs the oracle, both directions, a row count with a partial tile."""
    sd = recipes.nsf_cl_params(1500 + dim + K + n_h, dim, K, n_h)
    f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    f = select_kernel(f.to(DEV), kernel)
    assert f._packed(torch.device(DEV, 0))[1] is not None
    assert (f._split_image(torch.device(DEV, 0)) is not None) == (kernel == "split" and n_h >= 4)
    x = recipes.gaussian(1501 + dim, 531, dim, scale=1.5)
    sd64 = {k: v.double() for k, v in sd.items()}
    for inverse in (False, True):
        ref_y, ref_ld = O.nsf_cl(x, sd, K, 3.0, inverse)
        # the parity rule with its float64 budget spelled out: 1e-5 plus twice the distance of the fp32 oracle from
        # its own float64 evaluation on these inputs (a spline element a few ulps from a knot moves by more than 1e-5
        # between two correct fp32 evaluations)
        y64, ld64 = O.nsf_cl(x.double(), sd64, K, 3.0, inverse)
        y, ld = (f.inverse if inverse else f.forward)(cuda(x))
        # (tame nn.Linear-scale weights: the head-room is capped at 5e-5, helpers.MAX_WIDENING)
        assert_parity(y, ref_y.numpy(), y64.numpy(), f"nsf shape ({dim},{K},{n_h}) {kernel} inv={inverse} y")
        assert_parity(ld, ref_ld.numpy(), ld64.numpy(), f"nsf shape ({dim},{K},{n_h}) {kernel} inv={inverse} ld")


@pytest.mark.parametrize("dim,K,n_h", [(64, 8, 8), (64, 5, 8), (64, 8, 16), (32, 5, 8), (32, 5, 16)])
def test_spline_block_run_at_other_shapes(amd, dim, K, n_h):
    """[ActNorm, Glow, NSF_CL] blocks as one launch each for the other MFMA spline shapes (d = 64; K = 5): same
    tensors, log_det and log-prob as running the nine layers one by one."""
    torch.manual_seed(5)
    flows = []
    for i in range(3):
        a = amd.ActNormFlow(dim)
        a.load_state_dict({"s": 0.1 * torch.randn(1, dim), "t": 0.1 * torch.randn(1, dim)}, strict=False)
        a.data_dep_init_done = True
        nsf = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
        nsf.load_state_dict(recipes.nsf_cl_params(1600 + dim + i, dim, K, n_h))
        flows += [a, amd.Glow(dim), nsf]
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)
    x = cuda(recipes.gaussian(1601 + dim, 700, dim))
    with torch.no_grad():
        for direction in ("inverse", "forward"):
            model.fuse_affine_runs = True
            zs_f, ld_f = getattr(model, direction)(x)
            assert zs_f[1].data_ptr() + zs_f[1].numel() * 4 == zs_f[2].data_ptr()  # one launch, one buffer
            model.fuse_affine_runs = False
            zs_u, ld_u = getattr(model, direction)(x)
            for i, (a, b) in enumerate(zip(zs_f, zs_u)):
                assert_close(a, b, 5e-6, f"{direction} tensor {i}")
            assert_close(ld_f, ld_u, 5e-6, f"{direction} log_det")
        model.fuse_affine_runs = True
        lp_f = model.log_prob(x)
        assert model._logprob_done
        model.fuse_affine_runs = False
        assert_close(lp_f, model.log_prob(x), 5e-6, "log_prob")


def test_zz_parity_budget_audit():
    """Runs last in this file: prints, for every assert_parity call of the session that used float64 head-room, the
    error, the budget and the share of the budget that was head-room (pytest -s / -rP shows it), and checks that no
    NON-stress comparison needed more than half of its budget."""
    from helpers import PARITY_LOG, parity_report

    print("\n" + parity_report())
    tame = [r for r in PARITY_LOG if 0 < r["widening"] <= 5e-5 and not r["what"].startswith(("g4 ", "g5 "))]
    if tame:
        worst = max(tame, key=lambda r: r["used"])
        print(f"worst non-stress use of a widened budget: {worst['what']}: {100 * worst['used']:.0f} % "
              f"(err {worst['err']:.2e}, budget {worst['budget']:.2e})")
