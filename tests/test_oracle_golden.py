"""Pin the CPU oracle to the reference: every fixture under tests/golden/ was written by
the real reference (tests/golden/gen_golden.py); the oracle must reproduce it.

Where the oracle issues the reference's own ATen op sequence the match is required
bit-for-bit; the spline path (evaluated in place instead of on a gathered subset) and
the closed-form base log-prob are held to 1e-6 normwise.
"""
import numpy as np
import pytest
import torch

import recipes
from helpers import assert_parity, assert_close, c2_layers, c3_layers, g1_layers, t, unpack_mask
from oracle import flow_oracle as O

torch.set_num_threads(4)


def same(a, b, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == np.asarray(b).shape, what
    assert np.array_equal(a, b, equal_nan=True), f"{what}: not bit-identical, max diff {np.nanmax(np.abs(a - b))}"


@pytest.mark.parametrize("tag", ["init", "trained"])
def test_g1_c1_stack(golden, tag):
    fx = golden(f"g1_c1_stack_{tag}")
    layers = g1_layers(fx)
    x = t(fx["x"])
    zs, ld = O.flow_stack(x, layers, inverse=True)
    assert zs[0] is x and len(zs) == 10
    same(ld, fx["ld_inv"], "ld_inv")
    for i in fx["keep"]:
        same(zs[i], fx[f"zs{i}"], f"zs{i}")
    xs, ld_f = O.flow_stack(zs[-1], layers, inverse=False)
    same(ld_f, fx["ld_fwd"], "ld_fwd")
    for i in fx["keep"]:
        same(xs[i], fx[f"xs{i}"], f"xs{i}")
    assert_close(O.std_normal_log_prob(zs[-1]), fx["base_log_prob"], 1e-6, "base_log_prob")
    mean_lp, _ = O.mean_log_prob(x, layers)
    assert abs(mean_lp - float(fx["mean_log_prob"])) <= 1e-6 * abs(float(fx["mean_log_prob"]))
    # round trip sanity (~7e-7 abs in the reference)
    assert (xs[-1] - x).abs().max() < 5e-5


def test_g2_affine_half_single(golden):
    fx = golden("g2_affine_half_single")
    for dim in (64, 256):
        for parity in (False, True):
            sd = recipes.affine_half_params(200 + dim + int(parity), dim)
            tag = f"d{dim}_p{int(parity)}"
            z = t(fx[f"{tag}.z"])
            x, ld = O.affine_half(z, sd, parity, inverse=False)
            same(x, fx[f"{tag}.fwd"], tag + ".fwd")
            same(ld, fx[f"{tag}.ld_fwd"], tag + ".ld_fwd")
            x, ld = O.affine_half(z, sd, parity, inverse=True)
            same(x, fx[f"{tag}.inv"], tag + ".inv")
            same(ld, fx[f"{tag}.ld_inv"], tag + ".ld_inv")
            assert np.abs(fx[f"{tag}.ld_fwd"]).max() > 1.0  # fixture exercises exp() away from 1
    for tag, kw in (("nice", dict(scale=False)), ("noshift", dict(shift=False)),
                    ("h2", dict(h_sizes=(16, 40))), ("h1", dict(h_sizes=(7,)))):
        sd = recipes.affine_half_params(290, 10, **kw)
        flags = {k: v for k, v in kw.items() if k in ("scale", "shift")}
        z = t(fx[f"{tag}.z"])
        for inv, name in ((False, "fwd"), (True, "inv")):
            x, ld = O.affine_half(z, sd, True, inverse=inv, **flags)
            same(x, fx[f"{tag}.{name}"], f"{tag}.{name}")
            same(ld, fx[f"{tag}.ld_{name}"], f"{tag}.ld_{name}")


@pytest.mark.parametrize("dim", [64, 256])
def test_g3_c2_stack(golden, dim):
    fx = golden("g3_c2_stack")
    layers = c2_layers(dim)
    x = t(fx[f"d{dim}.x"])
    zs, ld = O.flow_stack(x, layers, inverse=True)
    same(zs[-1], fx[f"d{dim}.z_last"], "z_last")
    same(zs[4], fx[f"d{dim}.z_mid"], "z_mid")
    same(ld, fx[f"d{dim}.ld_inv"], "ld_inv")
    cur = x
    for i, spec in enumerate(reversed(layers)):
        cur, l1 = O.apply_layer(spec, cur, inverse=True)
        same(l1, fx[f"d{dim}.ld_incr"][i], f"ld_incr[{i}]")
    xs, ld_f = O.flow_stack(x, layers, inverse=False)
    same(xs[-1], fx[f"d{dim}.x_fwd_last"], "x_fwd_last")
    same(ld_f, fx[f"d{dim}.ld_fwd"], "ld_fwd")
    assert_close(O.std_normal_log_prob(zs[-1]), fx[f"d{dim}.base_log_prob"], 1e-6)
    mean_lp, _ = O.mean_log_prob(x, layers)
    ref = float(fx[f"d{dim}.mean_log_prob"])
    assert abs(mean_lp - ref) <= 1e-6 * abs(ref)
    # fp64 run of the oracle against the reference's own fp64 run: the error budget
    layers64 = [{**s, "params": {k: v.double() for k, v in s["params"].items()}} for s in layers]
    zs64, ld64 = O.flow_stack(x.double(), layers64, inverse=True)
    assert_close(zs64[-1], fx[f"d{dim}.z_last_f64"], 1e-12)
    # the reference accumulates log_det in an fp32 buffer even for fp64 inputs (core.py:29)
    assert_close(ld64, fx[f"d{dim}.ld_inv_f64"], 5e-7)
    # the fp32 reference sits within 1e-5 normwise of its own fp64 run (tolerance is meaningful)
    assert_close(fx[f"d{dim}.z_last"], fx[f"d{dim}.z_last_f64"].astype(np.float32), 1e-5)


@pytest.mark.parametrize("K", [5, 8])
def test_g4_rqs_direct(golden, K):
    fx = golden("g4_rqs_direct")
    v, W, H, D = (t(fx[f"K{K}.{n}"]) for n in "vWHD")
    for inv, name in ((False, "fwd"), (True, "inv")):
        out, lad = O.unconstrained_rqs(v, W, H, D, inverse=inv, tail_bound=3.0)
        assert_close(out, fx[f"K{K}.out_{name}"], 1e-6, f"out_{name}")
        assert_close(lad, fx[f"K{K}.lad_{name}"], 1e-6, f"lad_{name}")
    # identity tails, NaN passes through, +-T are inside
    out, lad = O.unconstrained_rqs(v, W, H, D, inverse=False, tail_bound=3.0)
    assert out[5] == 3.5 and lad[5] == 0 and out[6] == -7.0 and torch.isnan(out[7]) and lad[7] == 0
    assert abs(float(out[1]) - 3.0) < 1e-5 and abs(float(out[2]) + 3.0) < 1e-5


def test_rqs_all_outside_returns_identity():
    v = torch.tensor([4.0, -5.0, float("nan")])
    W = torch.zeros(3, 5)
    out, lad = O.unconstrained_rqs(v, W, W.clone(), torch.zeros(3, 4), inverse=False, tail_bound=3.0)
    assert torch.equal(out[:2], v[:2]) and torch.isnan(out[2]) and (lad == 0).all()


def test_rqs_too_many_bins_raises():
    v = torch.zeros(2)
    W = torch.zeros(2, 1001)
    with pytest.raises(ValueError):
        O.unconstrained_rqs(v, W, W.clone(), torch.zeros(2, 1000), inverse=False, tail_bound=3.0)


@pytest.mark.parametrize("cfg", [(32, 8, 8, 1.0), (32, 8, 16, 1.0), (2, 8, 16, 1.0), (6, 5, 8, 1.0),
                                 (32, 8, 8, 2.0), (2, 8, 16, 2.0)])
def test_g5_nsf_cl_layer(golden, cfg):
    dim, K, n_h, gain = cfg
    fx = golden("g5_nsf_cl_layer")
    tag = f"d{dim}_K{K}_h{n_h}" + ("" if gain == 1.0 else "_stress")
    sd = recipes.nsf_cl_params(500 + dim + n_h, dim, K, n_h, gain=gain)
    z = t(fx[f"{tag}.z"])
    for inv, name in ((False, "fwd"), (True, "inv")):
        x, ld = O.nsf_cl(z, sd, K, 3.0, inverse=inv)
        assert_close(x, fx[f"{tag}.{name}"], 1e-6, f"{tag}.{name}")
        assert_close(ld, fx[f"{tag}.ld_{name}"], 2e-6, f"{tag}.ld_{name}")


def test_g6_c3_stack(golden):
    fx = golden("g6_c3_stack")
    x = t(fx["x"])
    # ActNorm's data-dependent init, replayed layer by layer as the first inverse call does
    layers = c3_layers(fx)
    cur = x
    for i in reversed(range(3)):
        nsf, glow, _ = layers[3 * i + 2], layers[3 * i + 1], layers[3 * i]
        cur, _ = O.apply_layer(nsf, cur, True)
        cur, _ = O.apply_layer(glow, cur, True)
        s, tt = O.actnorm_init(cur)
        assert_close(s, fx[f"actnorm{i}.s"], 1e-5, f"actnorm{i}.s")
        assert_close(tt, fx[f"actnorm{i}.t"], 1e-5, f"actnorm{i}.t")
        cur, _ = O.affine_const(cur, t(fx[f"actnorm{i}.s"]), t(fx[f"actnorm{i}.t"]), True)
    assert_close(cur, fx["z_last_first_call"], 2e-6)
    zs, ld = O.flow_stack(x, layers, inverse=True)
    assert_close(zs[-1], fx["z_last"], 2e-6, "z_last")
    assert_close(zs[5], fx["z_mid"], 2e-6, "z_mid")
    assert_close(ld, fx["ld_inv"], 2e-6, "ld_inv")
    xs, ld_f = O.flow_stack(x, layers, inverse=False)
    assert_close(xs[-1], fx["x_fwd_last"], 2e-6, "x_fwd_last")
    assert_close(ld_f, fx["ld_fwd"], 2e-6, "ld_fwd")


@pytest.mark.parametrize("dim", [50, 800, 784])
def test_g7_rnvp(golden, dim):
    fx = golden("g7_rnvp")
    sd = recipes.rnvp_params(700 + dim, dim, 50)
    z = t(fx[f"d{dim}.z"])
    mask = unpack_mask(fx[f"d{dim}.mask_bits"], dim)
    x, ld = O.rnvp(z, sd, mask)
    # K=800 GEMMs: MKL's summation order moves with the thread count, so not bitwise
    assert_close(x, fx[f"d{dim}.x"], 1e-6, "x")
    assert_close(ld, fx[f"d{dim}.ld"], 1e-6, "ld")


def test_g8_sample_z(golden):
    fx = golden("g8_sample_z")
    layers = [{"kind": "rnvp", "params": recipes.rnvp_params(800 + i, 800, 50),
               "mask": unpack_mask(fx[f"mask{i}_bits"], 800)} for i in range(2)]
    z, ld = O.sample_z(t(fx["q0_mean"]), t(fx["q0_log_var"]), t(fx["eps"]), layers)
    assert_close(z, fx["z"], 1e-6, "z")
    assert_close(ld, fx["log_det"], 1e-6, "log_det")


G11_CASES = {"l800": (800, 50, 11), "l50": (50, 10, 12)}


def g11_oracle(fx, tag):
    """The oracle's MNFLinear.forward on fixture G11's captured noise: sample_z, then the layer."""
    n_in, n_out, seed = G11_CASES[tag]
    layers = [{"kind": "rnvp", "params": recipes.rnvp_params(1100 + seed + i, n_in, 50),
               "mask": unpack_mask(fx[f"{tag}.mask{i}_bits"], n_in)} for i in range(2)]
    z, _ = O.sample_z(t(fx[f"{tag}.q0_mean"]), t(fx[f"{tag}.q0_log_var"]), t(fx[f"{tag}.eps_z"]), layers)
    y = O.mnf_linear_forward(t(fx[f"{tag}.x"]), z, t(fx[f"{tag}.W_mean"]), t(fx[f"{tag}.W_log_var"]),
                             t(fx[f"{tag}.b_mean"]), t(fx[f"{tag}.b_log_var"]), t(fx[f"{tag}.eps_out"]))
    return z, y


@pytest.mark.parametrize("tag", sorted(G11_CASES))
def test_g11_mnf_linear_forward(golden, tag):
    """MNFLinear.forward of the reference with every random draw captured (mnf_linear.py:46-56)."""
    fx = golden("g11_mnf_linear_forward")
    _, y = g11_oracle(fx, tag)
    assert_close(y, fx[f"{tag}.y"], 1e-6, "y")


def test_g16_mnf_linear_wide_forward_and_gradients(golden):
    """MNFLinear(784, 256): the reference's forward with every draw captured, and its own autograd gradients of
    sum(y w), against autograd through the oracle (mnf_linear.py:46-64; models/mnf_feed_forward.py:27-31)."""
    fx = golden("g16_mnf_linear_wide")
    n_in, n_out = 784, 256
    rows = fx["y"].shape[0]
    p = {"W_mean": 0.1 * recipes.gaussian(1600, n_out, n_in), "W_log_var": -9 + 0.1 * recipes.gaussian(1601, n_out, n_in),
         "b_mean": 0.3 * recipes.gaussian(1602, 1, n_out)[0], "b_log_var": -9 + 0.1 * recipes.gaussian(1603, 1, n_out)[0],
         "q0_mean": 1 + 0.1 * recipes.gaussian(1604, 1, n_in)[0], "q0_log_var": -9 + 0.1 * recipes.gaussian(1605, 1, n_in)[0]}
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    flow_p = [{k: v.clone().requires_grad_(True) for k, v in recipes.rnvp_params(1610 + i, n_in, 50).items()} for i in range(2)]
    layers = [{"kind": "rnvp", "params": flow_p[i], "mask": unpack_mask(fx[f"mask{i}_bits"], n_in)} for i in range(2)]
    x = recipes.gaussian(1620, rows, n_in, scale=1.5).abs().requires_grad_(True)
    w = recipes.gaussian(1621, rows, n_out) / (rows * n_out)
    z, _ = O.sample_z(p["q0_mean"], p["q0_log_var"], t(fx["eps_z"]), layers)
    y = O.mnf_linear_forward(x, z, p["W_mean"], p["W_log_var"], p["b_mean"], p["b_log_var"], t(fx["eps_out"]))
    assert_close(y, fx["y"], 2e-6, "y")
    (y * w).sum().backward()
    assert_close(x.grad, fx["grad.x"], 5e-6, "grad x")
    for k, v in p.items():
        assert_close(v.grad, fx[f"grad.{k}"], 5e-6, f"grad {k}")
    for i in range(2):
        for k, v in flow_p[i].items():
            assert_close(v.grad, fx[f"grad.flow_q.{i}.{k}"], 5e-6, f"grad flow_q.{i}.{k}")


G12_CASES = {"d2_k8": (2, 8, 16, 1.0), "d6_k5": (6, 5, 8, 1.0), "d16_k8": (16, 8, 8, 1.5)}  # dim, K, n_h, gain


@pytest.mark.parametrize("tag", sorted(G12_CASES))
def test_g12_nsf_ar(golden, tag):
    """NSF_AR forward / inverse of the reference (spline_flow.py:182-235); budget as for the other spline fixtures."""
    fx = golden("g12_nsf_ar")
    dim, K, n_h, gain = G12_CASES[tag]
    sd = recipes.nsf_ar_params(1200 + dim + K, dim, K, n_h, gain=gain)
    x = t(fx[f"{tag}.x"])
    for direction, inverse in (("fwd", False), ("inv", True)):
        y, ld = O.nsf_ar(x, sd, K, 3.0, inverse)
        assert_parity(y, fx[f"{tag}.{direction}"], fx[f"{tag}.{direction}64"], f"{direction} y", rtol=2e-6)
        assert_close(ld, fx[f"{tag}.ld_{direction}"], 2e-5, f"{direction} log_det")


def test_g9_logdet_shapes(golden):
    fx = golden("g9_logdet_shapes")
    x = recipes.gaussian(900, 8, 4)
    specs = {
        "affine_half": {"kind": "affine_half", "parity": False, "params": recipes.affine_half_params(1, 4)},
        "nsf_cl": {"kind": "nsf_cl", "K": 5, "B": 3.0, "params": recipes.nsf_cl_params(2, 4, 5, 8)},
        "actnorm": {"kind": "affine_const", "params": recipes.actnorm_params(3, 4)},
        "affine_const": {"kind": "affine_const", "params": recipes.actnorm_params(4, 4)},
        "glow": {"kind": "glow", "params": recipes.glow_params(5, 4)},
    }
    for name, spec in specs.items():
        for inv, d in ((False, "fwd"), (True, "inv")):
            _, ld = O.apply_layer(spec, x, inv)
            assert tuple(ld.shape) == tuple(fx[f"{name}.{d}"]), name
    assert int(fx["rnvp.has_inverse"]) == 0
    with pytest.raises(AttributeError):
        O.apply_layer({"kind": "rnvp", "params": {}, "mask": None}, x, True)
    zs, ld = O.flow_stack(x, [specs["actnorm"], specs["glow"], specs["nsf_cl"]], False)
    assert tuple(ld.shape) == tuple(fx["stack.ld_shape"]) and zs[0] is x
    assert len(zs) == int(fx["stack.n_intermediates"])


def test_g10_padded_shapes(golden):
    """The shapes the HIP kernels run padded (narrow halves, odd hidden widths, an absent net; RNVP widths that are
    not multiples of 16 / hidden widths other than 30, 50): the oracle against the real reference."""
    fx = golden("g10_padded_shapes")
    for k, (tag, (dim, kw)) in enumerate(recipes.G10_AHF.items()):
        sd = recipes.affine_half_params(1000 + 10 * k, dim, s_last_gain=2.0, **kw)
        z = t(fx[f"ahf.{tag}.z"])
        flags = {key: kw[key] for key in ("scale", "shift") if key in kw}
        for name, inverse in (("fwd", False), ("inv", True)):
            y, ld = O.affine_half(z, sd, bool(k % 2), inverse, **flags)
            assert_close(y, fx[f"ahf.{tag}.{name}"], 1e-6, f"{tag}.{name}")
            assert_close(ld, fx[f"ahf.{tag}.ld_{name}"], 1e-6 if flags.get("scale", True) else 1.0, f"{tag}.ld_{name}")
    for k, (tag, (dim, hid)) in enumerate(recipes.G10_RNVP.items()):
        sd = recipes.rnvp_params(1100 + 10 * k, dim, hid)
        x, ld = O.rnvp(t(fx[f"rnvp.{tag}.z"]), sd, unpack_mask(fx[f"rnvp.{tag}.mask_bits"], dim))
        assert_close(x, fx[f"rnvp.{tag}.x"], 1e-6, f"{tag}.x")
        assert_close(ld, fx[f"rnvp.{tag}.ld"], 1e-6, f"{tag}.ld")


# ------------------------------------------------------------------ G13: MNFConv2d
G13_CASES = {"c1": (1, 20, 5, 21), "c2": (20, 50, 5, 22)}
G13_KEYS = ("W_mean", "W_log_var", "b_log_var", "q0_mean", "q0_log_var", "r0_c", "r0_b1", "r0_b2")


def g13_specs(tag, which, masks):
    n_in, n_out, k, seed = G13_CASES[tag]
    return [{"kind": "rnvp", "params": recipes.rnvp_params(1300 + seed + 10 * (which == "r") + i, n_out, 50),
             "mask": torch.from_numpy(masks[i])} for i in range(2)]


def g13_oracle(fx, tag):
    """The oracle's MNFConv2d.forward and kl_div on fixture G13's captured draws."""
    p = {k: torch.from_numpy(fx[f"{tag}.{k}"]) for k in G13_KEYS}
    z, _ = O.mnf_conv2d_sample_z(p["q0_mean"], p["q0_log_var"], torch.from_numpy(fx[f"{tag}.fwd.eps_z"]),
                                           g13_specs(tag, "q", fx[f"{tag}.fwd.masks"]))
    y = O.mnf_conv2d_forward(torch.from_numpy(fx[f"{tag}.x"]), z, p["W_mean"], p["W_log_var"], p["b_log_var"],
                                       torch.from_numpy(fx[f"{tag}.fwd.eps_out"]))
    zk, ldq = O.mnf_conv2d_sample_z(p["q0_mean"], p["q0_log_var"], torch.from_numpy(fx[f"{tag}.kl.eps_z"]),
                                              g13_specs(tag, "q", fx[f"{tag}.kl.masks_q"]))
    kl = O.mnf_conv2d_kl(p, zk, ldq, torch.from_numpy(fx[f"{tag}.kl.eps_w"]), torch.from_numpy(fx[f"{tag}.kl.eps_b"]),
                                   g13_specs(tag, "r", fx[f"{tag}.kl.masks_r"]))
    return y, kl


@pytest.mark.parametrize("tag", sorted(G13_CASES))
def test_g13_mnf_conv2d(golden, tag):
    """Oracle vs the reference's MNFConv2d.forward / kl_div (layers/mnf_conv.py:67-133), every draw replayed."""
    fx = golden("g13_mnf_conv2d")
    y, kl = g13_oracle(fx, tag)
    assert_parity(y, fx[f"{tag}.y"], what="MNFConv2d.forward")
    assert abs(float(kl) - float(fx[f"{tag}.kl"])) <= 1e-5 * abs(float(fx[f"{tag}.kl"])), (float(kl), float(fx[f"{tag}.kl"]))


# ------------------------------------------------------------------ G14: MNFLinear.kl_div
G14_CASES = {"l800": (800, 50, 31), "l50": (50, 10, 32)}
G14_KEYS = ("W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var", "r0_c", "r0_b1", "r0_b2")


def g14_specs(tag, which, masks, dtype=torch.float32):
    n_in, n_out, seed = G14_CASES[tag]
    return [{"kind": "rnvp", "mask": torch.from_numpy(masks[i]).to(dtype),
             "params": {k: v.to(dtype) for k, v in
                        recipes.rnvp_params(1400 + seed + 10 * (which == "r") + i, n_in, 50).items()}}
            for i in range(2)]


def g14_oracle(fx, tag, dtype=torch.float32):
    """The oracle's MNFLinear.kl_div on fixture G14's captured draws: (kl, parameter dict, z)."""
    p = {k: torch.from_numpy(fx[f"{tag}.{k}"]).to(dtype) for k in G14_KEYS}
    z, ldq = O.sample_z(p["q0_mean"], p["q0_log_var"], torch.from_numpy(fx[f"{tag}.eps_z"]).to(dtype),
                            g14_specs(tag, "q", fx[f"{tag}.masks_q"], dtype))
    kl = O.mnf_linear_kl(p, z, ldq, torch.from_numpy(fx[f"{tag}.eps_w"]).to(dtype),
                         g14_specs(tag, "r", fx[f"{tag}.masks_r"], dtype))
    return kl, p, z


@pytest.mark.parametrize("tag", sorted(G14_CASES))
def test_g14_mnf_linear_kl(golden, tag):
    """Oracle vs the reference's MNFLinear.kl_div (layers/mnf_linear.py:66-90), every draw replayed."""
    fx = golden("g14_mnf_linear_kl")
    kl, _, _ = g14_oracle(fx, tag)
    assert abs(float(kl) - float(fx[f"{tag}.kl"])) <= 1e-5 * abs(float(fx[f"{tag}.kl"])), (float(kl), float(fx[f"{tag}.kl"]))


# ------------------------------------------------------------------ G15: MAF / IAF
G15_CASES = {"d2": (2, (24, 24, 24), 300), "d5": (5, (24, 24, 24), 257), "d7_h16": (7, (16,), 129),
             "d12": (12, (24, 24, 24), 64)}


def g15_params(tag, parity, dtype=torch.float32):
    dim, h_sizes, _ = G15_CASES[tag]
    return {k: v.to(dtype) for k, v in recipes.maf_params(1500 + dim + int(parity), dim, h_sizes, gain=1.5,
                                                          last_gain=0.7).items()}


@pytest.mark.parametrize("tag", sorted(G15_CASES))
def test_g15_made_masks(golden, tag):
    """The oracle's MADE connectivity masks are the reference's (layers/made.py:58-94), bit for bit."""
    fx = golden("g15_maf_iaf")
    dim, h_sizes, _ = G15_CASES[tag]
    masks = O.made_masks(dim, h_sizes, 2 * dim)
    assert len(masks) == len(h_sizes) + 1
    for i, m in enumerate(masks):
        assert np.array_equal(m.numpy().astype(np.uint8), fx[f"{tag}.mask{i}"]), (tag, i)


@pytest.mark.parametrize("parity", [False, True])
@pytest.mark.parametrize("tag", sorted(G15_CASES))
def test_g15_maf_iaf(golden, tag, parity):
    """Oracle vs the reference's MAF.forward (sequential) and MAF.inverse (one pass) (flows/maf.py:39-62); IAF is the
    same layer with the directions swapped (:65-72)."""
    fx = golden("g15_maf_iaf")
    dim, h_sizes, _ = G15_CASES[tag]
    x = torch.from_numpy(fx[f"{tag}.x"])
    masks = O.made_masks(dim, h_sizes, 2 * dim)
    key = f"{tag}.p{int(parity)}"
    for name, inverse in (("fwd", False), ("inv", True)):
        y, ld = O.maf(x, g15_params(tag, parity), masks, parity, inverse)
        y64, ld64 = O.maf(x.double(), g15_params(tag, parity, torch.float64), masks, parity, inverse)
        assert_parity(y, fx[f"{key}.{name}"], fx[f"{key}.{name}64"], what=f"MAF {key} {name}")
        assert_parity(ld, fx[f"{key}.ld_{name}"], fx.get(f"{key}.ld_{name}64") if hasattr(fx, "get") else None,
                      what=f"MAF {key} ld_{name}")
        assert_parity(y64, fx[f"{key}.{name}64"], what=f"MAF {key} {name} (float64)", rtol=1e-12)
        spec = {"kind": "iaf", "params": g15_params(tag, parity), "masks": masks, "parity": parity}
        y_iaf, ld_iaf = O.apply_layer(spec, x, not inverse)
        assert torch.equal(y_iaf, y) and torch.equal(ld_iaf, ld)
