"""CPU oracle for the coupling-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over stock PyTorch *CPU* ops, the arithmetic
of the reference's (janosh/torch-mnf, pure Python on ATen) coupling layers and the
log-det accumulation loop.  It is the checker the HIP kernels are compared with.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``torch_mnf_amd/`` imports it; the product path
raises when the HIP library is missing instead of falling back to this file.

Parity pin: the reference holds no golden vectors for forward/inverse values
(its tests only bound training losses), so the pin is the set
of fixtures under ``tests/golden/`` that ``tests/golden/gen_golden.py`` wrote by
importing the real reference in the build container.  ``tests/test_oracle_golden.py``
checks every function below against those fixtures (bit-exact where the op sequence
is the reference's own, 1e-6 otherwise).

All functions take parameters as a mapping with the reference's ``state_dict`` key
names (``s_net.0.weight`` ...), and work in whatever dtype the tensors carry (fp32
for parity, fp64 for error budgets).

Reference citations are ``file:line`` below ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Mapping, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Mapping[str, Tensor]

LEAKY_SLOPE = 0.2  # torch_mnf/models/mlp.py:7
MIN_BIN_WIDTH = 1e-3  # torch_mnf/flows/spline_flow.py:17
MIN_BIN_HEIGHT = 1e-3  # torch_mnf/flows/spline_flow.py:18
MIN_DERIVATIVE = 1e-3  # torch_mnf/flows/spline_flow.py:19
# boundary-knot constant, evaluated in float64 as numpy does at spline_flow.py:47
EDGE_DERIVATIVE_CONST = math.log(math.exp(1.0 - MIN_DERIVATIVE) - 1.0)


# --------------------------------------------------------------------------- MLP
def linear_indices(params: Params, prefix: str) -> list[int]:
    """Indices ``i`` of ``{prefix}.{i}.weight`` in order (0, 2, 4, ... for the MLP)."""
    idx = []
    for key in params:
        if key.startswith(prefix + ".") and key.endswith(".weight"):
            idx.append(int(key[len(prefix) + 1 : -len(".weight")]))
    return sorted(idx)


def mlp(x: Tensor, params: Params, prefix: str) -> Tensor:
    """Linear / LeakyReLU(0.2) chain without a trailing activation.

    torch_mnf/models/mlp.py:4-12: nn.Sequential of Linear, LeakyReLU pairs with the
    last LeakyReLU dropped; with two sizes it is one bare Linear.
    """
    layer_ids = linear_indices(params, prefix)
    for n, i in enumerate(layer_ids):
        x = F.linear(x, params[f"{prefix}.{i}.weight"], params[f"{prefix}.{i}.bias"])
        if n + 1 < len(layer_ids):
            x = F.leaky_relu(x, LEAKY_SLOPE)
    return x


# ---------------------------------------------------------------- AffineHalfFlow
def affine_half(
    x: Tensor,
    params: Params,
    parity: bool,
    inverse: bool,
    scale: bool = True,
    shift: bool = True,
) -> tuple[Tensor, Tensor]:
    """RealNVP/NICE half coupling.  torch_mnf/flows/affine_half_flow.py:44-66.

    The conditioner half passes through; the other half becomes ``exp(s)*v + t``
    (forward) or ``(v - t) / exp(s)`` (inverse, log-det ``-sum s``).  ``parity``
    selects the upper half as conditioner (:47-48) and the halves keep their places
    in the output (:58-60).
    """
    h = x.shape[1] // 2
    lo, hi = x[:, :h], x[:, h:]
    cond, act = (hi, lo) if parity else (lo, hi)
    zeros = cond.new_zeros(cond.shape[0], h)  # :38, scale/shift switched off
    s = mlp(cond, params, "s_net") if scale else zeros
    t = mlp(cond, params, "t_net") if shift else zeros
    if inverse:
        new = (act - t) / s.exp()  # :53
        s = -s  # :55
    else:
        new = s.exp() * act + t  # :57
    out = torch.cat([new, cond] if parity else [cond, new], dim=1)
    return out, s.sum(1)  # :61


# ------------------------------------------------------ rational-quadratic spline
def _knots(unnorm: Tensor, lo: float, hi: float, min_size: float) -> tuple[Tensor, Tensor]:
    """Bin sizes and K+1 knot positions on [lo, hi].  spline_flow.py:95-102 / :106-113."""
    k = unnorm.shape[-1]
    frac = F.softmax(unnorm, dim=-1)
    frac = min_size + (1 - min_size * k) * frac
    knots = torch.cumsum(frac, dim=-1)
    knots = F.pad(knots, pad=(1, 0), mode="constant", value=0.0)
    knots = (hi - lo) * knots + lo
    knots[..., 0] = lo
    knots[..., -1] = hi
    return knots[..., 1:] - knots[..., :-1], knots


def _bin_index(knots: Tensor, v: Tensor, eps: float = 1e-6) -> Tensor:
    """count(v >= knot) - 1 with the last knot nudged up by eps.  spline_flow.py:22-24."""
    knots = knots.clone()
    knots[..., -1] += eps
    return (v[..., None] >= knots).sum(dim=-1) - 1


def rqs(
    v: Tensor,
    un_w: Tensor,
    un_h: Tensor,
    un_d: Tensor,
    inverse: bool,
    left: float,
    right: float,
    bottom: float,
    top: float,
) -> tuple[Tensor, Tensor]:
    """Monotone rational-quadratic spline on a box.  spline_flow.py:71-179.

    ``un_d`` has K+1 entries (already padded).  Every ``v`` must lie inside the box.
    Returns (outputs, log|dy/dx|) -- the inverse branch returns the negated
    log-derivative evaluated at the root (:158-159).
    """
    k = un_w.shape[-1]
    if MIN_BIN_WIDTH * k > 1.0:  # :90-91
        raise ValueError("Minimal bin width too large for the number of bins")
    if MIN_BIN_HEIGHT * k > 1.0:  # :92-93
        raise ValueError("Minimal bin height too large for the number of bins")

    widths, xk = _knots(un_w, left, right, MIN_BIN_WIDTH)
    deriv = MIN_DERIVATIVE + F.softplus(un_d)  # :104
    heights, yk = _knots(un_h, bottom, top, MIN_BIN_HEIGHT)

    idx = _bin_index(yk if inverse else xk, v)[..., None]  # :115-118

    def pick(t: Tensor) -> Tensor:
        return t.gather(-1, idx)[..., 0]

    x_k, w_k = pick(xk), pick(widths)
    y_k, h_k = pick(yk), pick(heights)
    delta = pick(heights / widths)  # :124-125
    d_k, d_k1 = pick(deriv), pick(deriv[..., 1:])  # :127-129

    if inverse:
        dy = v - y_k
        curv = d_k + d_k1 - 2 * delta
        a = dy * curv + h_k * (delta - d_k)  # :134-136
        b = h_k * d_k - dy * curv  # :137-139
        c = -delta * dy  # :140
        disc = b.pow(2) - 4 * a * c  # :142
        assert (disc >= 0).all()  # :143
        root = (2 * c) / (-b - torch.sqrt(disc))  # :145
        out = root * w_k + x_k  # :146
        tomt = root * (1 - root)
        denom = delta + curv * tomt  # :149-152
        dnum = delta.pow(2) * (
            d_k1 * root.pow(2) + 2 * delta * tomt + d_k * (1 - root).pow(2)
        )  # :153-157
        return out, -(torch.log(dnum) - 2 * torch.log(denom))

    theta = (v - x_k) / w_k  # :161
    tomt = theta * (1 - theta)
    numer = h_k * (delta * theta.pow(2) + d_k * tomt)  # :164-166
    denom = delta + (d_k + d_k1 - 2 * delta) * tomt  # :167-170
    out = y_k + numer / denom  # :171
    dnum = delta.pow(2) * (
        d_k1 * theta.pow(2) + 2 * delta * tomt + d_k * (1 - theta).pow(2)
    )  # :173-177
    return out, torch.log(dnum) - 2 * torch.log(denom)  # :178


def unconstrained_rqs(
    v: Tensor, un_w: Tensor, un_h: Tensor, un_d: Tensor, inverse: bool, tail_bound: float
) -> tuple[Tensor, Tensor]:
    """Spline on [-T, T] with identity tails.  spline_flow.py:29-68.

    Elements outside the interval -- NaN included, both comparisons are false --
    pass through with log-derivative 0 (:51-52).  The derivative vector is padded to
    K+1 with the constant that makes the boundary slope 1 (:46-49).

    Divergence, documented: the reference gathers the inside elements into a 1-D
    tensor first and raises from ``torch.min`` of an empty tensor when *no* element
    is inside.  Here every element is evaluated in place
    (outside ones on a harmless stand-in value) and an all-outside batch returns the
    identity.  Per-element results are unaffected: the spline acts element by element.
    """
    inside = (v >= -tail_bound) & (v <= tail_bound)  # :40
    un_d = F.pad(un_d, pad=(1, 1))
    un_d[..., 0] = EDGE_DERIVATIVE_CONST
    un_d[..., -1] = EDGE_DERIVATIVE_CONST
    safe = torch.where(inside, v, torch.zeros_like(v))
    out, lad = rqs(
        safe, un_w, un_h, un_d, inverse, -tail_bound, tail_bound, -tail_bound, tail_bound
    )
    return torch.where(inside, out, v), torch.where(inside, lad, torch.zeros_like(lad))


def _nsf_half_step(
    cond: Tensor, act: Tensor, params: Params, net: str, K: int, T: float, inverse: bool
) -> tuple[Tensor, Tensor]:
    """One of the two dependent half-updates of NSF_CL.  spline_flow.py:252-258."""
    h = act.shape[1]
    raw = mlp(cond, params, net).reshape(-1, h, 3 * K - 1)
    W, H, D = torch.split(raw, K, dim=2)
    W, H = torch.softmax(W, dim=2), torch.softmax(H, dim=2)
    W, H = 2 * T * W, 2 * T * H  # first normalisation; rqs() normalises again
    D = F.softplus(D)
    new, lad = unconstrained_rqs(act, W, H, D, inverse=inverse, tail_bound=T)
    return new, torch.sum(lad, dim=1)


def nsf_cl(x: Tensor, params: Params, K: int, T: float, inverse: bool) -> tuple[Tensor, Tensor]:
    """Neural-spline coupling layer.  spline_flow.py:249-285.

    forward: f1(lower) moves upper, then f2(upper') moves lower; inverse undoes them
    in the opposite order.  Output is cat([lower, upper]) (:266).
    """
    h = x.shape[1] // 2
    lower, upper = x[:, :h], x[:, h:]
    log_det = torch.zeros(x.shape[0], dtype=x.dtype)
    if not inverse:
        upper, ld = _nsf_half_step(lower, upper, params, "f1", K, T, False)
        log_det += ld
        lower, ld = _nsf_half_step(upper, lower, params, "f2", K, T, False)
        log_det += ld
    else:
        lower, ld = _nsf_half_step(upper, lower, params, "f2", K, T, True)
        log_det += ld
        upper, ld = _nsf_half_step(lower, upper, params, "f1", K, T, True)
        log_det += ld
    return torch.cat([lower, upper], dim=1), log_det


def nsf_ar(x: Tensor, params: Params, K: int, T: float, inverse: bool) -> tuple[Tensor, Tensor]:
    """Neural-spline autoregressive layer.  spline_flow.py:201-235.

    ``params``: ``init_param`` (3K-1,) and ``layers.{i-1}.{0,2,4,6}.{weight,bias}`` for the conditioner of element i.
    forward (:201-218) conditions element i on the first i OUTPUT elements and runs the spline inverted;
    inverse (:220-235) conditions on the INPUT and runs it forward.
    """
    dim = x.shape[1]
    out = torch.zeros_like(x)
    log_det = torch.zeros(x.shape[0], dtype=x.dtype)
    for i in range(dim):
        if i == 0:
            raw = params["init_param"].expand(x.shape[0], 3 * K - 1)
        else:
            raw = mlp((x if inverse else out)[:, :i], params, f"layers.{i - 1}")
        W, H, D = torch.split(raw, K, dim=1)
        W, H = torch.softmax(W, dim=1), torch.softmax(H, dim=1)
        W, H = 2 * T * W, 2 * T * H
        D = F.softplus(D)
        col, ld = unconstrained_rqs(x[:, i], W, H, D, inverse=not inverse, tail_bound=T)
        out = torch.cat([out[:, :i], col[:, None], out[:, i + 1:]], dim=1)  # (out-of-place: autograd friendly)
        log_det = log_det + ld
    return out, log_det


# ------------------------------------------------------------- masked/gated RNVP
def rnvp(z: Tensor, params: Params, mask: Tensor) -> tuple[Tensor, Tensor]:
    """Forward-only gated coupling with an explicit Bernoulli mask.  rnvp.py:25-39.

    The reference draws ``mask`` itself (:28); the oracle takes it as an input so a
    GPU run can be compared on the same mask.  ``(1-gate)*shift`` reaches every
    position, masked-in ones too (:37) -- reproduced as written.
    """
    z_gated, z_kept = (1 - mask) * z, mask * z  # :30
    y = mlp(z_kept, params, "net")
    shift = F.linear(y, params["t.weight"], params["t.bias"])
    scale = F.linear(y, params["s.weight"], params["s.bias"])
    gate = torch.sigmoid(scale)  # :35
    log_det = ((1 - mask) * gate.log()).sum(1)  # :36
    x = (z_gated * gate + (1 - gate) * shift) + z_kept  # :37
    return x, log_det


# ------------------------------------------------------ data-independent layers
def affine_const(x: Tensor, s: Tensor, t: Tensor, inverse: bool) -> tuple[Tensor, Tensor]:
    """Per-dimension affine.  affine_constant_flow.py:18-26.  log_det has shape (1,)."""
    if inverse:
        return (x - t) * torch.exp(-s), torch.sum(-s, dim=1)
    return x * torch.exp(s) + t, torch.sum(s, dim=1)


def actnorm_init(x: Tensor) -> tuple[Tensor, Tensor]:
    """Data-dependent (s, t) set on the first inverse call.  affine_constant_flow.py:45-48."""
    s = x.std(dim=0, keepdim=True).log()
    t = (x * s.exp()).mean(dim=0, keepdim=True)
    return s, t


def glow_weight(P: Tensor, L: Tensor, S: Tensor, U: Tensor) -> Tensor:
    """W = P (tril(L,-1)+I) (triu(U,1)+diag(S)).  glow.py:20-24."""
    n = L.shape[0]
    lower = torch.tril(L, diagonal=-1) + torch.eye(n, dtype=L.dtype)
    upper = torch.triu(U, diagonal=1)
    return P @ lower @ (upper + S.diag())


def glow(x: Tensor, P: Tensor, L: Tensor, S: Tensor, U: Tensor, inverse: bool) -> tuple[Tensor, Tensor]:
    """Invertible linear map; 0-dim log_det.  glow.py:26-37 (dense inverse at :34)."""
    W = glow_weight(P, L, S, U)
    ld = S.abs().log().sum()
    if inverse:
        return x @ torch.inverse(W), -ld
    return x @ W, ld


# ------------------------------------------------------------------ the stack
def made_masks(n_in: int, hidden_sizes: Sequence[int], n_out: int, natural_ordering: bool = True, seed: int = 0) -> list:
    """The binary (n_in_l, n_out_l) connectivity masks of a MADE network.  torch_mnf/layers/made.py:58-94 (num_masks = 1):
    input degrees are the natural order (or a permutation from numpy.random.RandomState(seed)); hidden unit degrees are
    drawn with ``rng.randint(min of the previous layer's degrees, n_in - 1, size)``; a hidden connection needs
    degree_prev <= degree_next, an output connection degree_hidden < degree_output; the output mask is tiled when
    n_out is a multiple of n_in."""
    import numpy as np

    rng = np.random.RandomState(seed)
    deg = {-1: np.arange(n_in) if natural_ordering else rng.permutation(n_in)}
    for layer, size in enumerate(hidden_sizes):
        deg[layer] = rng.randint(deg[layer - 1].min(), n_in - 1, size=size)
    n_layers = len(hidden_sizes)
    masks = [deg[layer - 1][:, None] <= deg[layer][None, :] for layer in range(n_layers)]
    masks.append(deg[n_layers - 1][:, None] < deg[-1][None, :])
    if n_out > n_in:
        masks[-1] = np.concatenate([masks[-1]] * (n_out // n_in), axis=1)
    return [torch.from_numpy(m) for m in masks]


def made(x: Tensor, params: Params, masks: Sequence[Tensor], prefix: str = "net.") -> Tensor:
    """MADE forward: MaskedLinear layers ``x @ (W.T * mask) + b`` with ReLU between them, none after the last.
    torch_mnf/layers/made.py:24-25, 46-49.  ``params``: state_dict entries ``{prefix}{2l}.weight / .bias``."""
    for layer, mask in enumerate(masks):
        W, b = params[f"{prefix}{2 * layer}.weight"], params[f"{prefix}{2 * layer}.bias"]
        x = x @ (W.T * mask.to(x.dtype)) + b
        if layer + 1 < len(masks):
            x = torch.relu(x)
    return x


def maf(x: Tensor, params: Params, masks: Sequence[Tensor], parity: bool, inverse: bool) -> tuple[Tensor, Tensor]:
    """MAF.  torch_mnf/flows/maf.py:39-62.  ``inverse`` (:54-62, one pass): ``z = x * exp(s) + t`` with
    ``s, t = net(x).split(dim)``, flipped along the features when ``parity``, ``log_det = sum(s)``.  ``forward``
    (:39-52, sequential): the input is flipped first when ``parity``; starting from zeros, element i becomes
    ``(z_i - t_i) * exp(-s_i)`` with s, t from the net on the elements decoded so far; ``log_det = -sum(s_i)``."""
    dim = x.shape[1]
    if inverse:
        s, t = made(x, params, masks).split(dim, dim=1)
        z = x * s.exp() + t
        return (z.flip(dims=[1]) if parity else z), s.sum(1)
    z = x.flip(dims=[1]) if parity else x
    out = torch.zeros_like(z)
    log_det = torch.zeros(z.shape[0], dtype=z.dtype)
    for i in range(dim):
        s, t = made(out, params, masks).split(dim, dim=1)
        out = torch.cat([out[:, :i], ((z[:, i] - t[:, i]) * torch.exp(-s[:, i]))[:, None], out[:, i + 1:]], dim=1)
        log_det = log_det - s[:, i]
    return out, log_det


def apply_layer(spec: dict, x: Tensor, inverse: bool) -> tuple[Tensor, Tensor]:
    """Dispatch on ``spec['kind']`` -- the oracle's stand-in for duck-typed flow modules."""
    kind, p = spec["kind"], spec["params"]
    if kind == "affine_half":
        return affine_half(
            x, p, spec["parity"], inverse, spec.get("scale", True), spec.get("shift", True)
        )
    if kind == "nsf_cl":
        return nsf_cl(x, p, spec["K"], spec["B"], inverse)
    if kind == "nsf_ar":
        return nsf_ar(x, p, spec["K"], spec["B"], inverse)
    if kind == "affine_const":
        return affine_const(x, p["s"], p["t"], inverse)
    if kind == "glow":
        return glow(x, p["P"], p["L"], p["S"], p["U"], inverse)
    if kind == "rnvp":
        if inverse:
            raise AttributeError("RNVP has no inverse (rnvp.py)")
        return rnvp(x, p, spec["mask"])
    if kind in ("maf", "iaf"):  # IAF swaps the two directions (maf.py:65-72)
        return maf(x, p, spec["masks"], spec["parity"], inverse if kind == "maf" else not inverse)
    raise ValueError(f"unknown layer kind {kind!r}")


def flow_stack(x: Tensor, layers: Sequence[dict], inverse: bool) -> tuple[list[Tensor], Tensor]:
    """Log-det accumulation loop.  torch_mnf/flows/core.py:17-35.

    Layers run in order (forward) or reversed (inverse); ``log_det`` starts as zeros
    (B,) and receives ``+= ld`` per layer, where ``ld`` may be (B,), (1,) or ().
    Every intermediate is kept; element 0 is the caller's tensor itself.
    """
    log_det = torch.zeros(x.shape[0], dtype=x.dtype)
    seen = [x]
    for spec in reversed(layers) if inverse else layers:
        x, ld = apply_layer(spec, x, inverse)
        log_det += ld
        seen.append(x)
    return seen, log_det


def std_normal_log_prob(z: Tensor) -> Tensor:
    """log N(z; 0, I_d) = -|z|^2/2 - d/2 log(2 pi): the base distribution the
    notebooks and tests pair with the flows (tests/test_flows.py:38,
    examples/half_moons.ipynb:80 use MultivariateNormal(zeros, eye))."""
    d = z.shape[1]
    return -0.5 * (z * z).sum(1) - 0.5 * d * math.log(2 * math.pi)


def mean_log_prob(x: Tensor, layers: Sequence[dict]) -> tuple[float, Tensor]:
    """The metric's unit of work: one inverse pass + base log-prob.

    Returns (mean over rows accumulated in float64, per-row log-prob)."""
    zs, log_det = flow_stack(x, layers, inverse=True)
    lp = log_det + std_normal_log_prob(zs[-1])
    return float(lp.double().mean()), lp


def sample_z(q0_mean: Tensor, q0_log_var: Tensor, eps: Tensor, layers: Sequence[dict]) -> tuple[Tensor, Tensor]:
    """MNFLinear.sample_z with injected noise.  torch_mnf/layers/mnf_linear.py:58-64."""
    q0_std = q0_log_var.exp().sqrt().repeat(eps.shape[0], 1)
    z = q0_mean + q0_std * eps
    zs, log_det = flow_stack(z, layers, inverse=False)
    return zs[-1], log_det.squeeze()


def mnf_linear_forward(x: Tensor, z: Tensor, W_mean: Tensor, W_log_var: Tensor, b_mean: Tensor, b_log_var: Tensor,
                       eps: Tensor) -> Tensor:
    """MNFLinear.forward behind sample_z, with the output noise injected.  torch_mnf/layers/mnf_linear.py:46-56:
    ``mean = x * z @ W_mean.T + b_mean``; ``var = x**2 @ exp(W_log_var).T + exp(b_log_var)``;
    ``mean + var.sqrt() * epsilon``."""
    mean = x * z @ W_mean.T + b_mean
    var = x**2 @ W_log_var.exp().T + b_log_var.exp()
    return mean + var.sqrt() * eps


def mnf_linear_kl(p: dict, z: Tensor, log_det_q: Tensor, eps_w: Tensor, r_layers: Sequence[dict]) -> Tensor:
    """MNFLinear.kl_div behind sample_z with the weight noise injected and flow_r given as oracle layer specs.
    torch_mnf/layers/mnf_linear.py:66-90 (tanh auxiliary activation, eqs. (9), (10))."""
    W_mean = z * p["W_mean"]
    W_var = p["W_log_var"].exp()
    weight = W_mean + W_var.sqrt() * eps_w
    kl_W = 0.5 * torch.sum(-W_var.log() + W_var + W_mean**2 - 1)
    kl_b = 0.5 * torch.sum(-p["b_log_var"] + p["b_log_var"].exp() + p["b_mean"] ** 2 - 1)
    log_q = -log_det_q - 0.5 * p["q0_log_var"].sum()
    act = torch.tanh(p["r0_c"] @ weight.T)
    mean_r = torch.outer(p["r0_b1"], act).mean(1)
    log_var_r = torch.outer(p["r0_b2"], act).mean(1)
    zs, log_det_r = flow_stack(z, r_layers, inverse=False)
    log_r = log_det_r.squeeze() + 0.5 * torch.sum(-log_var_r.exp() * (zs[-1] - mean_r) ** 2 + log_var_r)
    return kl_W + kl_b + log_q - log_r


def mnf_conv2d_sample_z(q0_mean: Tensor, q0_log_var: Tensor, eps_z: Tensor, layers: Sequence[dict]) -> tuple[Tensor, Tensor]:
    """MNFConv2d.sample_z with injected noise: one n_out-vector through flow_q.  layers/mnf_conv.py:90-98."""
    z0 = q0_mean + q0_log_var.exp().sqrt() * eps_z
    zs, log_det = flow_stack(z0[None, :], layers, inverse=False)
    return zs[-1], log_det.squeeze()


def mnf_conv2d_forward(x: Tensor, z: Tensor, W_mean: Tensor, W_log_var: Tensor, b_log_var: Tensor, eps: Tensor) -> Tensor:
    """MNFConv2d.forward behind sample_z with the output noise injected.  layers/mnf_conv.py:67-88 (b_mean is the
    zero tensor of :45)."""
    mean = F.conv2d(x, weight=W_mean * z.view(-1, 1, 1, 1), bias=torch.zeros(W_mean.shape[0], dtype=x.dtype))
    var = F.conv2d(x**2, weight=W_log_var.exp(), bias=b_log_var.exp())
    return mean + var.sqrt() * eps


def mnf_conv2d_kl(p: dict, z: Tensor, log_det_q: Tensor, eps_w: Tensor, eps_b: Tensor, r_layers: Sequence[dict]) -> Tensor:
    """MNFConv2d.kl_div behind sample_z with eps_w / eps_b injected and flow_r given as oracle layer specs.
    layers/mnf_conv.py:100-133 (linear auxiliary activation: no tanh, :119-123)."""
    W_var, b_var = p["W_log_var"].exp(), p["b_log_var"].exp()
    W_mean = p["W_mean"] * z.view(-1, 1, 1, 1)
    b_mean = torch.zeros_like(p["b_log_var"]) * z
    kl_W = 0.5 * torch.sum(-W_var.log() + W_var + W_mean**2 - 1)
    kl_b = 0.5 * torch.sum(-b_var.log() + b_var + b_mean**2 - 1)
    log_q = -log_det_q - 0.5 * p["q0_log_var"].sum()
    n = p["r0_c"].numel()
    act = W_mean.reshape(-1, n) @ p["r0_c"] + (W_var.sqrt().reshape(-1, n) @ p["r0_c"]) * eps_w  # eqs. (11), (12)
    act = act + torch.sum(b_mean * p["r0_c"]) + torch.sum(b_var * p["r0_c"] ** 2).sqrt() * eps_b
    mean_r = torch.outer(p["r0_b1"], act).mean(1)
    log_var_r = torch.outer(p["r0_b2"], act).mean(1)
    zs, log_det_r = flow_stack(z, r_layers, inverse=False)
    log_r = log_det_r.squeeze() + 0.5 * torch.sum(-log_var_r.exp() * (zs[-1] - mean_r) ** 2 + log_var_r)
    return kl_W + kl_b + log_q - log_r
