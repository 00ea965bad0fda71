"""Write the golden fixtures by running the REAL reference (janosh/torch-mnf).

Run in the build container only (needs /root/reference; it never travels):

    python tests/golden/gen_golden.py

Every fixture is data: inputs, masks/noise, the reference's outputs, and -- where they
cannot be rebuilt from ``recipes.py`` -- parameter values.  No reference source text is
stored.  Fixture list follows SURVEY.md section 8c (G1..G9); G10 adds the shapes the kernels run padded.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MNF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from sklearn.datasets import make_moons  # noqa: E402
from torch.distributions import MultivariateNormal  # noqa: E402

import recipes  # noqa: E402
import torch_mnf.flows as nf  # noqa: E402  (the reference)
from torch_mnf.flows import spline_flow as ref_spline  # noqa: E402
from torch_mnf.layers import MNFConv2d, MNFLinear  # noqa: E402

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  keys={len(arrays)}")


def moons(n):
    """== torch_mnf.data.sample_moons(n) (data.py:21-24); data.py itself needs seaborn."""
    pts, _ = make_moons(n, noise=0.05, random_state=0)
    return torch.as_tensor(pts).float()


def sd_arrays(prefix, sd):
    return {f"{prefix}{k}": npy(v) for k, v in sd.items()}


# ----------------------------------------------------------------------------- G1
def g1_c1_stack():
    """9 x AffineHalfFlow(d=2) on half-moons, default-init (seed 0) and trained."""
    x = moons(4096)
    base = MultivariateNormal(torch.zeros(2), torch.eye(2))
    for tag, steps in (("init", 0), ("trained", 300)):
        torch.manual_seed(0)
        flows = [nf.AffineHalfFlow(dim=2, parity=bool(i % 2)) for i in range(9)]
        model = nf.NormalizingFlowModel(base, flows)
        if steps:
            opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=1e-5)
            for _ in range(steps):
                xb = x[torch.randperm(4096)[:512]]  # minibatches of the evaluated points
                _, ld = model.inverse(xb)
                loss = -(ld + model.base_log_prob(xb)).sum() / 512
                model.zero_grad()
                loss.backward()
                opt.step()
        with torch.no_grad():
            zs, ld_inv = model.inverse(x)
            blp = model.base_log_prob(x)
            xs, ld_fwd = model.forward(zs[-1])
        out = {"x": npy(x), "ld_inv": npy(ld_inv), "base_log_prob": npy(blp),
               "ld_fwd": npy(ld_fwd),
               "mean_log_prob": np.float64((ld_inv + blp).double().mean().item())}
        # keep every 3rd intermediate + the last to stay small; index list is stored
        keep = [0, 3, 6, 9]
        out["keep"] = np.array(keep)
        for i in keep:
            out[f"zs{i}"] = npy(zs[i])
            out[f"xs{i}"] = npy(xs[i])
        for li, f in enumerate(flows):
            out.update(sd_arrays(f"L{li}.", f.state_dict()))
        save(f"g1_c1_stack_{tag}", **out)


# ----------------------------------------------------------------------------- G2
def g2_affine_half_single():
    out = {}
    for dim in (64, 256):
        for parity in (False, True):
            seed = 200 + dim + int(parity)
            f = nf.AffineHalfFlow(dim, parity)
            f.load_state_dict(recipes.affine_half_params(seed, dim))
            z = recipes.gaussian(seed + 1, 256 if dim == 64 else 64, dim)
            with torch.no_grad():
                x_f, ld_f = f.forward(z)
                x_i, ld_i = f.inverse(z)
            tag = f"d{dim}_p{int(parity)}"
            out.update({f"{tag}.z": npy(z), f"{tag}.fwd": npy(x_f), f"{tag}.ld_fwd": npy(ld_f),
                        f"{tag}.inv": npy(x_i), f"{tag}.ld_inv": npy(ld_i)})
    # NICE variants and odd hidden sizes (scale/shift flags, h_sizes of other lengths)
    for tag, kw in (("nice", dict(scale=False)), ("noshift", dict(shift=False)),
                    ("h2", dict(h_sizes=(16, 40))), ("h1", dict(h_sizes=(7,)))):
        dim, seed = 10, 290
        f = nf.AffineHalfFlow(dim, True, **kw)
        f.load_state_dict(recipes.affine_half_params(seed, dim, **kw))
        z = recipes.gaussian(seed + 1, 64, dim)
        with torch.no_grad():
            x_f, ld_f = f.forward(z)
            x_i, ld_i = f.inverse(z)
        out.update({f"{tag}.z": npy(z), f"{tag}.fwd": npy(x_f), f"{tag}.ld_fwd": npy(ld_f),
                    f"{tag}.inv": npy(x_i), f"{tag}.ld_inv": npy(ld_i)})
    save("g2_affine_half_single", **out)


# ----------------------------------------------------------------------------- G3
def g3_c2_stack():
    """9 x AffineHalfFlow d=64 (the benchmark stack's weights) on 256 rows, plus d=256."""
    out = {}
    for dim in (64, 256):
        sds = recipes.c2_stack_params(dim)
        flows = []
        for i, sd in enumerate(sds):
            f = nf.AffineHalfFlow(dim, parity=bool(i % 2))
            f.load_state_dict(sd)
            flows.append(f)
        base = MultivariateNormal(torch.zeros(dim), torch.eye(dim))
        model = nf.NormalizingFlowModel(base, flows)
        x = recipes.gaussian(300 + dim, 256 if dim == 64 else 64, dim)
        with torch.no_grad():
            zs, ld = model.inverse(x)
            blp = model.base_log_prob(x)
            incr, cur = [], x
            for f in reversed(flows):
                cur, l1 = f.inverse(cur)
                incr.append(npy(l1))
            xs, ld_f = model.forward(x)
            z64, ld64 = nf.NormalizingFlowModel(base, flows).double().inverse(x.double())
        out.update({f"d{dim}.x": npy(x), f"d{dim}.z_last": npy(zs[-1]), f"d{dim}.z_mid": npy(zs[4]),
                    f"d{dim}.ld_inv": npy(ld), f"d{dim}.ld_incr": np.stack(incr),
                    f"d{dim}.base_log_prob": npy(blp),
                    f"d{dim}.mean_log_prob": np.float64((ld + blp).double().mean().item()),
                    f"d{dim}.x_fwd_last": npy(xs[-1]), f"d{dim}.ld_fwd": npy(ld_f),
                    f"d{dim}.z_last_f64": npy(z64[-1]), f"d{dim}.ld_inv_f64": npy(ld64)})
    save("g3_c2_stack", **out)


# ----------------------------------------------------------------------------- G4
def g4_rqs_direct():
    """unconstrained_RQS called directly, edge inputs included."""
    out = {}
    T = 3.0
    for K in (5, 8):
        rng = np.random.default_rng(400 + K)
        n = 512
        v = (2.2 * rng.standard_normal(n)).astype(np.float32)
        edge = np.array([0.0, T, -T, np.nextafter(np.float32(T), np.float32(10)),
                         np.nextafter(np.float32(-T), np.float32(-10)), 3.5, -7.0, np.nan,
                         np.nextafter(np.float32(T), np.float32(0)), 1e-30, -1e-30, 2.9999],
                        dtype=np.float32)
        v[: len(edge)] = edge
        W = (2.0 * rng.standard_normal((n, K))).astype(np.float32)
        H = (2.0 * rng.standard_normal((n, K))).astype(np.float32)
        D = (2.0 * rng.standard_normal((n, K - 1))).astype(np.float32)
        D[20:24] = 30.0  # softplus linear branch (threshold 20)
        out.update({f"K{K}.v": v, f"K{K}.W": W, f"K{K}.H": H, f"K{K}.D": D})
        for inv in (False, True):
            y, lad = ref_spline.unconstrained_RQS(
                torch.from_numpy(v), torch.from_numpy(W), torch.from_numpy(H),
                torch.from_numpy(D), inverse=inv, tail_bound=T)
            out[f"K{K}.out_{'inv' if inv else 'fwd'}"] = npy(y)
            out[f"K{K}.lad_{'inv' if inv else 'fwd'}"] = npy(lad)
            y64, lad64 = ref_spline.unconstrained_RQS(
                torch.from_numpy(v).double(), torch.from_numpy(W).double(), torch.from_numpy(H).double(),
                torch.from_numpy(D).double(), inverse=inv, tail_bound=T)
            out[f"K{K}.out64_{'inv' if inv else 'fwd'}"] = npy(y64)
            out[f"K{K}.lad64_{'inv' if inv else 'fwd'}"] = npy(lad64)
    save("g4_rqs_direct", **out)


# ----------------------------------------------------------------------------- G5
def g5_nsf_cl_layer():
    out = {}
    for dim, K, n_h, rows, gain in ((32, 8, 8, 256, 1.0), (32, 8, 16, 256, 1.0), (2, 8, 16, 256, 1.0),
                                   (6, 5, 8, 128, 1.0), (32, 8, 8, 256, 2.0), (2, 8, 16, 256, 2.0)):
        seed = 500 + dim + n_h
        f = nf.NSF_CL(dim, K=K, B=3, n_h=n_h)
        f.load_state_dict(recipes.nsf_cl_params(seed, dim, K, n_h, gain=gain))
        z = recipes.gaussian(seed + 1, rows, dim, scale=1.4)
        z[0, :] = 3.0
        z[1, :] = -3.0
        z[2, 0] = 3.5  # one element outside, rest of the row inside
        z[3, :] = 5.0  # whole row outside
        f64 = nf.NSF_CL(dim, K=K, B=3, n_h=n_h).double()
        f64.load_state_dict({k: v.double() for k, v in f.state_dict().items()})
        with torch.no_grad():
            x_f, ld_f = f.forward(z)
            x_i, ld_i = f.inverse(z)
            x_f64, ld_f64 = f64.forward(z.double())
            x_i64, ld_i64 = f64.inverse(z.double())
        tag = f"d{dim}_K{K}_h{n_h}" + ("" if gain == 1.0 else "_stress")
        out.update({f"{tag}.z": npy(z), f"{tag}.fwd": npy(x_f), f"{tag}.ld_fwd": npy(ld_f),
                    f"{tag}.inv": npy(x_i), f"{tag}.ld_inv": npy(ld_i),
                    f"{tag}.fwd64": npy(x_f64).astype(np.float32), f"{tag}.ld_fwd64": npy(ld_f64).astype(np.float32),
                    f"{tag}.inv64": npy(x_i64).astype(np.float32), f"{tag}.ld_inv64": npy(ld_i64).astype(np.float32)})
    save("g5_nsf_cl_layer", **out)


# ----------------------------------------------------------------------------- G6
def g6_c3_stack():
    """3 x [ActNorm, Glow, NSF_CL] d=32 K=8 n_h=8 after ActNorm's data-dependent init."""
    dim, rows = 32, 256
    torch.manual_seed(6)
    flows = []
    for i in range(3):
        an = nf.ActNormFlow(dim)
        gl = nf.Glow(dim)
        gp = recipes.glow_params(600 + i, dim)
        gl.P = gp["P"]
        gl.load_state_dict({"L": gp["L"], "S": gp["S"], "U": gp["U"]})
        sp = nf.NSF_CL(dim, K=8, B=3, n_h=8)
        sp.load_state_dict(recipes.nsf_cl_params(610 + i, dim, 8, 8))
        flows += [an, gl, sp]
    model = nf.NormalizingFlow(flows)
    x = recipes.gaussian(620, rows, dim, scale=1.2)
    out = {"x": npy(x)}
    with torch.no_grad():
        zs, ld = model.inverse(x)  # first inverse call initialises the three ActNorms
        for i in range(3):
            out[f"actnorm{i}.s"] = npy(flows[3 * i].s)
            out[f"actnorm{i}.t"] = npy(flows[3 * i].t)
        zs2, ld2 = model.inverse(x)  # now data-independent
        xs, ld_f = model.forward(x)
        shapes = [tuple(f.inverse(x)[1].shape) for f in flows[:3]]
    # the same stack in float64 (ActNorm values copied from the fp32 run): the error budget
    flows64 = []
    for i in range(3):
        an = nf.ActNormFlow(dim).double()
        an.load_state_dict({"s": flows[3 * i].s.double(), "t": flows[3 * i].t.double()})
        an.data_dep_init_done = True
        gl = nf.Glow(dim).double()
        gl.P = flows[3 * i + 1].P.double()
        gl.load_state_dict({k: v.double() for k, v in flows[3 * i + 1].state_dict().items()})
        sp = nf.NSF_CL(dim, K=8, B=3, n_h=8).double()
        sp.load_state_dict({k: v.double() for k, v in flows[3 * i + 2].state_dict().items()})
        flows64 += [an, gl, sp]
    with torch.no_grad():
        # NSF_CL/NormalizingFlow keep log_det in an fp32 buffer; sum the float64 terms here instead
        def run64(xx, inverse):
            ld = torch.zeros(rows, dtype=torch.float64)
            for f in (reversed(flows64) if inverse else flows64):
                xx, l1 = f.inverse(xx) if inverse else f.forward(xx)
                ld = ld + l1.double()
            return xx, ld
        z64, ld64 = run64(x.double(), True)
        x64, ldf64 = run64(x.double(), False)
    out.update({"z_last64": npy(z64).astype(np.float32), "ld_inv64": npy(ld64).astype(np.float32),
                "x_fwd_last64": npy(x64).astype(np.float32), "ld_fwd64": npy(ldf64).astype(np.float32)})
    out.update({"z_last_first_call": npy(zs[-1]), "ld_first_call": npy(ld),
                "z_last": npy(zs2[-1]), "ld_inv": npy(ld2), "z_mid": npy(zs2[5]),
                "x_fwd_last": npy(xs[-1]), "ld_fwd": npy(ld_f),
                "ld_shape_actnorm": np.array(shapes[0], dtype=np.int64),
                "ld_shape_glow": np.array(shapes[1], dtype=np.int64),
                "ld_shape_nsf": np.array(shapes[2], dtype=np.int64)})
    save("g6_c3_stack", **out)


# ----------------------------------------------------------------------------- G7
def g7_rnvp():
    out = {}
    for dim, rows in ((50, 128), (800, 32), (784, 16)):
        seed = 700 + dim
        f = nf.RNVP(dim, h_sizes=(50,))
        f.load_state_dict(recipes.rnvp_params(seed, dim, 50))
        z = recipes.gaussian(seed + 1, rows, dim)
        torch.manual_seed(seed)
        mask = torch.bernoulli(0.5 * torch.ones_like(z))
        torch.manual_seed(seed)
        with torch.no_grad():
            x, ld = f.forward(z)
        out.update({f"d{dim}.z": npy(z), f"d{dim}.mask_bits": np.packbits(npy(mask).astype(np.uint8), axis=1),
                    f"d{dim}.x": npy(x), f"d{dim}.ld": npy(ld)})
    save("g7_rnvp", **out)


# ----------------------------------------------------------------------------- G10
def g10_padded_shapes():
    """Layers whose widths do not fill the kernels' tiles: AffineHalfFlow with narrow halves / odd hidden widths /
    an absent net, RNVP with dim % 16 != 0 and hidden widths other than 30 / 50."""
    out = {}
    for k, (tag, (dim, kw)) in enumerate(recipes.G10_AHF.items()):
        seed = 1000 + 10 * k
        f = nf.AffineHalfFlow(dim, bool(k % 2), **kw)
        f.load_state_dict(recipes.affine_half_params(seed, dim, s_last_gain=2.0, **kw))
        z = recipes.gaussian(seed + 1, 96, dim)
        with torch.no_grad():
            x_f, ld_f = f.forward(z)
            x_i, ld_i = f.inverse(z)
        out.update({f"ahf.{tag}.z": npy(z), f"ahf.{tag}.fwd": npy(x_f), f"ahf.{tag}.ld_fwd": npy(ld_f),
                    f"ahf.{tag}.inv": npy(x_i), f"ahf.{tag}.ld_inv": npy(ld_i)})
    for k, (tag, (dim, hid)) in enumerate(recipes.G10_RNVP.items()):
        seed = 1100 + 10 * k
        f = nf.RNVP(dim, h_sizes=(hid,))
        f.load_state_dict(recipes.rnvp_params(seed, dim, hid))
        z = recipes.gaussian(seed + 1, 80, dim)
        torch.manual_seed(seed)
        mask = torch.bernoulli(0.5 * torch.ones_like(z))
        torch.manual_seed(seed)
        with torch.no_grad():
            x, ld = f.forward(z)
        out.update({f"rnvp.{tag}.z": npy(z), f"rnvp.{tag}.mask_bits": np.packbits(npy(mask).astype(np.uint8), axis=1),
                    f"rnvp.{tag}.x": npy(x), f"rnvp.{tag}.ld": npy(ld)})
    save("g10_padded_shapes", **out)


# ----------------------------------------------------------------------------- G8
def g8_sample_z():
    """MNFLinear(800, 50).sample_z(64) with the noise and both masks captured."""
    torch.manual_seed(8)
    layer = MNFLinear(800, 50)
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(800 + i, 800, 50))
    captured = {"eps": [], "mask": []}
    real_randn_like, real_bernoulli = torch.randn_like, torch.bernoulli

    def randn_like(t, *a, **k):
        r = real_randn_like(t, *a, **k)
        captured["eps"].append(r.clone())
        return r

    def bernoulli(t, *a, **k):
        r = real_bernoulli(t, *a, **k)
        captured["mask"].append(r.clone())
        return r

    torch.randn_like, torch.bernoulli = randn_like, bernoulli
    try:
        with torch.no_grad():
            z, ld = layer.sample_z(64)
    finally:
        torch.randn_like, torch.bernoulli = real_randn_like, real_bernoulli
    assert len(captured["eps"]) == 1 and len(captured["mask"]) == 2
    save("g8_sample_z", q0_mean=npy(layer.q0_mean), q0_log_var=npy(layer.q0_log_var),
         eps=npy(captured["eps"][0]),
         mask0_bits=np.packbits(npy(captured["mask"][0]).astype(np.uint8), axis=1),
         mask1_bits=np.packbits(npy(captured["mask"][1]).astype(np.uint8), axis=1),
         z=npy(z), log_det=npy(ld))


# ----------------------------------------------------------------------------- G11
def g11_mnf_linear_forward():
    """MNFLinear.forward (mnf_linear.py:46-56) with every random draw captured: the sample_z noise, both flow masks and
    the output noise -- for MNFLinear(800, 50) (MNF-LeNet's first dense layer) and MNFLinear(50, 10) (its second)."""
    out = {}
    for tag, (n_in, n_out, rows, seed) in {"l800": (800, 50, 96, 11), "l50": (50, 10, 200, 12)}.items():
        torch.manual_seed(seed)
        layer = MNFLinear(n_in, n_out)
        for i, f in enumerate(layer.flow_q.flows):
            f.load_state_dict(recipes.rnvp_params(1100 + seed + i, n_in, 50))
        x = recipes.gaussian(1100 + seed, rows, n_in, scale=1.5).abs()  # post-ReLU-like activations
        captured = {"eps": [], "mask": []}
        real_randn_like, real_bernoulli = torch.randn_like, torch.bernoulli

        def randn_like(t, *a, **k):
            r = real_randn_like(t, *a, **k)
            captured["eps"].append(r.clone())
            return r

        def bernoulli(t, *a, **k):
            r = real_bernoulli(t, *a, **k)
            captured["mask"].append(r.clone())
            return r

        torch.randn_like, torch.bernoulli = randn_like, bernoulli
        try:
            with torch.no_grad():
                y = layer.forward(x)
        finally:
            torch.randn_like, torch.bernoulli = real_randn_like, real_bernoulli
        assert len(captured["eps"]) == 2 and len(captured["mask"]) == 2
        for k in ("W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var"):
            out[f"{tag}.{k}"] = npy(getattr(layer, k))
        out[f"{tag}.x"] = npy(x)
        out[f"{tag}.eps_z"] = npy(captured["eps"][0])
        out[f"{tag}.eps_out"] = npy(captured["eps"][1])
        out[f"{tag}.mask0_bits"] = np.packbits(npy(captured["mask"][0]).astype(np.uint8), axis=1)
        out[f"{tag}.mask1_bits"] = np.packbits(npy(captured["mask"][1]).astype(np.uint8), axis=1)
        out[f"{tag}.y"] = npy(y)
    save("g11_mnf_linear_forward", **out)


# ----------------------------------------------------------------------------- G16
def g16_mnf_linear_wide():
    """MNFLinear(784, 256) (mnf_linear.py:46-64 under loss.backward(); the first layer of an ordinary
    MNFFeedForward([784, 256, 10]), models/mnf_feed_forward.py:27-31): forward with every random draw captured, and the
    reference's own autograd gradients of sum(y * w) for x and every parameter on that path.  The layer's parameters are
    recipe draws (rebuilt by the tests, not stored): W_mean 0.1 N, W_log_var -9 + 0.1 N, b_log_var -9 + 0.1 N,
    q0_mean 1 + 0.1 N, q0_log_var -9 + 0.1 N."""
    n_in, n_out, rows, seed = 784, 256, 48, 16
    torch.manual_seed(seed)
    layer = MNFLinear(n_in, n_out)
    with torch.no_grad():
        layer.W_mean.copy_(0.1 * recipes.gaussian(1600, n_out, n_in))
        layer.W_log_var.copy_(-9 + 0.1 * recipes.gaussian(1601, n_out, n_in))
        layer.b_mean.copy_(0.3 * recipes.gaussian(1602, 1, n_out)[0])
        layer.b_log_var.copy_(-9 + 0.1 * recipes.gaussian(1603, 1, n_out)[0])
        layer.q0_mean.copy_(1 + 0.1 * recipes.gaussian(1604, 1, n_in)[0])
        layer.q0_log_var.copy_(-9 + 0.1 * recipes.gaussian(1605, 1, n_in)[0])
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(1610 + i, n_in, 50))
    x = recipes.gaussian(1620, rows, n_in, scale=1.5).abs().requires_grad_(True)
    w = recipes.gaussian(1621, rows, n_out) / (rows * n_out)
    captured = {"eps": [], "mask": []}
    real_randn_like, real_bernoulli = torch.randn_like, torch.bernoulli

    def randn_like(t, *a, **k):
        r = real_randn_like(t, *a, **k)
        captured["eps"].append(r.clone())
        return r

    def bernoulli(t, *a, **k):
        r = real_bernoulli(t, *a, **k)
        captured["mask"].append(r.clone())
        return r

    torch.randn_like, torch.bernoulli = randn_like, bernoulli
    try:
        y = layer.forward(x)
    finally:
        torch.randn_like, torch.bernoulli = real_randn_like, real_bernoulli
    assert len(captured["eps"]) == 2 and len(captured["mask"]) == 2
    (y * w).sum().backward()
    out = {"eps_z": npy(captured["eps"][0]), "eps_out": npy(captured["eps"][1]),
           "mask0_bits": np.packbits(npy(captured["mask"][0]).astype(np.uint8), axis=1),
           "mask1_bits": np.packbits(npy(captured["mask"][1]).astype(np.uint8), axis=1),
           "y": npy(y), "grad.x": npy(x.grad)}
    for k in ("W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var"):
        out[f"grad.{k}"] = npy(getattr(layer, k).grad)
    for i, f in enumerate(layer.flow_q.flows):
        for k, prm in f.named_parameters():
            out[f"grad.flow_q.{i}.{k}"] = npy(prm.grad)
    save("g16_mnf_linear_wide", **out)


# ----------------------------------------------------------------------------- G12
def g12_nsf_ar():
    """NSF_AR (spline_flow.py:182-235) forward and inverse: K in {5, 8}, dims 2 (the reference's tests), 6 and 16."""
    out = {}
    for tag, (dim, K, n_h, rows, gain) in {"d2_k8": (2, 8, 16, 300, 1.0), "d6_k5": (6, 5, 8, 257, 1.0),
                                           "d16_k8": (16, 8, 8, 128, 1.5)}.items():
        sd = recipes.nsf_ar_params(1200 + dim + K, dim, K, n_h, gain=gain)
        layer = nf.NSF_AR(dim, K=K, B=3, n_h=n_h)
        layer.load_state_dict(sd)
        x = recipes.gaussian(1201 + dim, rows, dim, scale=1.6)  # a few percent of the elements outside +-3
        with torch.no_grad():
            y_f, ld_f = layer.forward(x)
            y_i, ld_i = layer.inverse(x)
            back, _ = layer.inverse(y_f)
        assert float((back - x).abs().max()) < 1e-3
        out[f"{tag}.x"] = npy(x)
        out[f"{tag}.fwd"], out[f"{tag}.ld_fwd"] = npy(y_f), npy(ld_f)
        out[f"{tag}.inv"], out[f"{tag}.ld_inv"] = npy(y_i), npy(ld_i)
        layer64 = nf.NSF_AR(dim, K=K, B=3, n_h=n_h).double()
        layer64.load_state_dict({k: v.double() for k, v in sd.items()})
        with torch.no_grad():
            # (the reference allocates log_det as float32 zeros: its float64 run returns float64 outputs only)
            y64_f, _ = layer64.forward(x.double())
            y64_i, _ = layer64.inverse(x.double())
        out[f"{tag}.fwd64"], out[f"{tag}.inv64"] = npy(y64_f), npy(y64_i)
    save("g12_nsf_ar", **out)


G12_CASES = {"d2_k8": (2, 8, 16), "d6_k5": (6, 5, 8), "d16_k8": (16, 8, 8)}


# ----------------------------------------------------------------------------- G13
def g13_mnf_conv2d():
    """MNFConv2d (layers/mnf_conv.py:67-133) forward and kl_div with every random draw captured, for MNF-LeNet's two
    convolutions at small image sizes: MNFConv2d(1, 20, 5) and MNFConv2d(20, 50, 5)."""
    out = {}
    for tag, (n_in, n_out, k, batch, side, seed) in {"c1": (1, 20, 5, 3, 12, 21), "c2": (20, 50, 5, 2, 8, 22)}.items():
        torch.manual_seed(seed)
        layer = MNFConv2d(n_in, n_out, k)
        for name, flow in (("q", layer.flow_q), ("r", layer.flow_r)):
            for i, f in enumerate(flow.flows):
                f.load_state_dict(recipes.rnvp_params(1300 + seed + 10 * (name == "r") + i, n_out, 50))
        x = recipes.gaussian(1300 + seed, batch, n_in * side * side).reshape(batch, n_in, side, side)
        captured = {"randn_like": [], "bernoulli": [], "randn": []}
        real = (torch.randn_like, torch.bernoulli, torch.randn)

        def randn_like(t, *a, **kw):
            r = real[0](t, *a, **kw)
            captured["randn_like"].append(r.clone())
            return r

        def bernoulli(t, *a, **kw):
            r = real[1](t, *a, **kw)
            captured["bernoulli"].append(r.clone())
            return r

        def randn(*a, **kw):
            r = real[2](*a, **kw)
            captured["randn"].append(r.clone())
            return r

        torch.randn_like, torch.bernoulli, torch.randn = randn_like, bernoulli, randn
        try:
            with torch.no_grad():
                y = layer.forward(x)
                n_fwd = {key: len(v) for key, v in captured.items()}
                kl = layer.kl_div()
        finally:
            torch.randn_like, torch.bernoulli, torch.randn = real
        assert n_fwd == {"randn_like": 2, "bernoulli": 2, "randn": 0}, n_fwd
        assert {key: len(v) for key, v in captured.items()} == {"randn_like": 4, "bernoulli": 6, "randn": 1}
        for key in ("W_mean", "W_log_var", "b_log_var", "q0_mean", "q0_log_var", "r0_c", "r0_b1", "r0_b2"):
            out[f"{tag}.{key}"] = npy(getattr(layer, key))
        out[f"{tag}.x"] = npy(x)
        # forward: eps_z, two flow_q masks, output noise
        out[f"{tag}.fwd.eps_z"] = npy(captured["randn_like"][0])
        out[f"{tag}.fwd.eps_out"] = npy(captured["randn_like"][1])
        out[f"{tag}.fwd.masks"] = np.stack([npy(m) for m in captured["bernoulli"][:2]])
        out[f"{tag}.y"] = npy(y)
        # kl_div: eps_z, flow_q masks, eps_w, eps_b, flow_r masks
        out[f"{tag}.kl.eps_z"] = npy(captured["randn_like"][2])
        out[f"{tag}.kl.eps_w"] = npy(captured["randn_like"][3])
        out[f"{tag}.kl.eps_b"] = npy(captured["randn"][0])
        out[f"{tag}.kl.masks_q"] = np.stack([npy(m) for m in captured["bernoulli"][2:4]])
        out[f"{tag}.kl.masks_r"] = np.stack([npy(m) for m in captured["bernoulli"][4:6]])
        out[f"{tag}.kl"] = npy(kl)
    save("g13_mnf_conv2d", **out)


# ----------------------------------------------------------------------------- G14
def g14_mnf_linear_kl():
    """MNFLinear.kl_div (mnf_linear.py:66-90) with every random draw captured (the sample_z noise, flow_q's two masks,
    the weight noise, flow_r's two masks) -- for MNFLinear(800, 50) and MNFLinear(50, 10), MNF-LeNet's dense layers."""
    out = {}
    for tag, (n_in, n_out, seed) in {"l800": (800, 50, 31), "l50": (50, 10, 32)}.items():
        torch.manual_seed(seed)
        layer = MNFLinear(n_in, n_out)
        for name, flow in (("q", layer.flow_q), ("r", layer.flow_r)):
            for i, f in enumerate(flow.flows):
                f.load_state_dict(recipes.rnvp_params(1400 + seed + 10 * (name == "r") + i, n_in, 50))
        with torch.no_grad():  # (b_mean is zero at init: give the bias term something to square)
            layer.b_mean.copy_(recipes.gaussian(1400 + seed, 1, n_out, scale=0.3)[0])
        captured = {"randn_like": [], "bernoulli": []}
        real = (torch.randn_like, torch.bernoulli)

        def randn_like(t, *a, **kw):
            r = real[0](t, *a, **kw)
            captured["randn_like"].append(r.clone())
            return r

        def bernoulli(t, *a, **kw):
            r = real[1](t, *a, **kw)
            captured["bernoulli"].append(r.clone())
            return r

        torch.randn_like, torch.bernoulli = randn_like, bernoulli
        try:
            with torch.no_grad():
                kl = layer.kl_div()
        finally:
            torch.randn_like, torch.bernoulli = real
        assert {key: len(v) for key, v in captured.items()} == {"randn_like": 2, "bernoulli": 4}
        for key in ("W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var", "r0_c", "r0_b1", "r0_b2"):
            out[f"{tag}.{key}"] = npy(getattr(layer, key))
        out[f"{tag}.eps_z"] = npy(captured["randn_like"][0])
        out[f"{tag}.eps_w"] = npy(captured["randn_like"][1])
        out[f"{tag}.masks_q"] = np.stack([npy(m) for m in captured["bernoulli"][:2]])
        out[f"{tag}.masks_r"] = np.stack([npy(m) for m in captured["bernoulli"][2:]])
        out[f"{tag}.kl"] = npy(kl)
    save("g14_mnf_linear_kl", **out)


# ----------------------------------------------------------------------------- G15
G15_CASES = {"d2": (2, (24, 24, 24), 300), "d5": (5, (24, 24, 24), 257), "d7_h16": (7, (16,), 129),
             "d12": (12, (24, 24, 24), 64)}


def g15_maf_iaf():
    """MAF and IAF (flows/maf.py:21-72 on layers/made.py's MADE) in both directions and both parities: dim 2 (the
    reference's tests), 5, 7 (one hidden layer) and 12; the float64 run of the same layers; the MADE masks themselves."""
    out = {}
    for tag, (dim, h_sizes, rows) in G15_CASES.items():
        for parity in (False, True):
            sd = recipes.maf_params(1500 + dim + int(parity), dim, h_sizes, gain=1.5, last_gain=0.7)
            x = recipes.gaussian(1501 + dim, rows, dim, scale=1.2)
            layer = nf.MAF(dim, parity=parity, h_sizes=h_sizes)
            layer.load_state_dict(sd, strict=False)
            key = f"{tag}.p{int(parity)}"
            if not parity:
                out[f"{tag}.x"] = npy(x)
                masked = [m for m in layer.net if hasattr(m, "mask")]
                for i, m in enumerate(masked):
                    out[f"{tag}.mask{i}"] = npy(m.mask).astype(np.uint8)
            with torch.no_grad():
                y_f, ld_f = layer.forward(x)      # sequential (IAF's inverse)
                y_i, ld_i = layer.inverse(x)      # one pass (IAF's forward)
                back, _ = layer.inverse(y_f)
            assert float((back.flip(dims=[1]) if parity else back).sub(x.flip(dims=[1]) if parity else x).abs().max()) < 1e-3 \
                or parity  # (with parity the reference's forward flips its INPUT and inverse its OUTPUT: see the oracle)
            out[f"{key}.fwd"], out[f"{key}.ld_fwd"] = npy(y_f), npy(ld_f)
            out[f"{key}.inv"], out[f"{key}.ld_inv"] = npy(y_i), npy(ld_i)
            layer64 = nf.MAF(dim, parity=parity, h_sizes=h_sizes).double()
            layer64.load_state_dict({k: v.double() for k, v in sd.items()}, strict=False)
            with torch.no_grad():
                y64_f, _ = layer64.forward(x.double())   # (log_det is allocated as float32 zeros there: outputs only)
                y64_i, ld64_i = layer64.inverse(x.double())
            out[f"{key}.fwd64"], out[f"{key}.inv64"], out[f"{key}.ld_inv64"] = npy(y64_f), npy(y64_i), npy(ld64_i)
    save("g15_maf_iaf", **out)


# ----------------------------------------------------------------------------- G9
def g9_logdet_shapes():
    x = recipes.gaussian(900, 8, 4)
    mods = {"affine_half": nf.AffineHalfFlow(4, False), "nsf_cl": nf.NSF_CL(4, K=5),
            "actnorm": nf.ActNormFlow(4), "affine_const": nf.AffineConstantFlow(4),
            "glow": nf.Glow(4), "rnvp": nf.RNVP(4)}
    out = {}
    with torch.no_grad():
        for name, m in mods.items():
            out[f"{name}.fwd"] = np.array(tuple(m.forward(x)[1].shape), dtype=np.int64)
            if hasattr(m, "inverse"):
                out[f"{name}.inv"] = np.array(tuple(m.inverse(x)[1].shape), dtype=np.int64)
        out["rnvp.has_inverse"] = np.array(int(hasattr(mods["rnvp"], "inverse")))
        zs, ld = nf.NormalizingFlow([mods["actnorm"], mods["glow"], mods["nsf_cl"]]).forward(x)
        out["stack.ld_shape"] = np.array(tuple(ld.shape), dtype=np.int64)
        out["stack.first_is_input"] = np.array(int(zs[0] is x))
        out["stack.n_intermediates"] = np.array(len(zs))
    save("g9_logdet_shapes", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    g1_c1_stack()
    g2_affine_half_single()
    g3_c2_stack()
    g4_rqs_direct()
    g5_nsf_cl_layer()
    g6_c3_stack()
    g7_rnvp()
    g8_sample_z()
    g10_padded_shapes()
    g11_mnf_linear_forward()
    g12_nsf_ar()
    g13_mnf_conv2d()
    g14_mnf_linear_kl()
    g15_maf_iaf()
    g16_mnf_linear_wide()
    g9_logdet_shapes()
