"""Deterministic parameter / input recipes shared by the benchmark, the fixture generator and the tests.

The big weight tensors are not stored in the fixtures: both sides rebuild them from
``numpy.random.default_rng(seed)`` with the functions below, so a fixture only
carries inputs, masks and the reference's outputs.  Nothing here imports the
reference or the oracle.
"""
from __future__ import annotations

import numpy as np
import torch


def _linear(rng: np.random.Generator, n_out: int, n_in: int, gain: float = 1.0):
    """nn.Linear-like init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias, times gain."""
    bound = gain / np.sqrt(n_in)
    w = rng.uniform(-bound, bound, size=(n_out, n_in)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(n_out,)).astype(np.float32)
    return torch.from_numpy(w), torch.from_numpy(b)


def mlp_params(rng, prefix: str, sizes, last_gain: float = 1.0, gain: float = 1.0) -> dict:
    """state_dict entries ``{prefix}.{2i}.weight/bias`` for MLP(*sizes)."""
    out = {}
    n = len(sizes) - 1
    for i, (a, b) in enumerate(zip(sizes, sizes[1:])):
        g = gain * (last_gain if i == n - 1 else 1.0)
        w, bias = _linear(rng, b, a, g)
        out[f"{prefix}.{2 * i}.weight"] = w
        out[f"{prefix}.{2 * i}.bias"] = bias
    return out


def affine_half_params(seed: int, dim: int, h_sizes=(24, 24, 24), s_last_gain: float = 4.0,
                       scale: bool = True, shift: bool = True) -> dict:
    """AffineHalfFlow state_dict.  Only the last Linear of s_net is amplified so |s|
    reaches ~2 without overflowing through a deep stack."""
    rng = np.random.default_rng(seed)
    h = dim // 2
    sd = {}
    if scale:
        sd.update(mlp_params(rng, "s_net", (h, *h_sizes, h), last_gain=s_last_gain))
    if shift:
        sd.update(mlp_params(rng, "t_net", (h, *h_sizes, h)))
    return sd


def nsf_cl_params(seed: int, dim: int, K: int, n_h: int, gain: float = 1.0) -> dict:
    """NSF_CL state_dict: f1, f2 = MLP(h, n_h, n_h, n_h, (3K-1)*h).

    gain 1 is nn.Linear's own init range; gain 2 is the stress variant: there the
    reference's fp32 output already sits 1e-5..4e-5 (normwise) from its own fp64 run."""
    rng = np.random.default_rng(seed)
    h = dim // 2
    sizes = (h, n_h, n_h, n_h, (3 * K - 1) * dim // 2)
    sd = mlp_params(rng, "f1", sizes, gain=gain)
    sd.update(mlp_params(rng, "f2", sizes, gain=gain))
    return sd


def nsf_ar_params(seed: int, dim: int, K: int, n_h: int, gain: float = 1.0) -> dict:
    """NSF_AR state_dict: init_param ~ U(-1/2, 1/2) (3K-1,), layers.{i-1} = MLP(i, n_h, n_h, n_h, 3K-1)."""
    rng = np.random.default_rng(seed)
    sd = {"init_param": torch.from_numpy(rng.uniform(-0.5, 0.5, size=(3 * K - 1,)).astype(np.float32))}
    for i in range(1, dim):
        sd.update(mlp_params(rng, f"layers.{i - 1}", (i, n_h, n_h, n_h, 3 * K - 1), gain=gain))
    return sd


def rnvp_params(seed: int, dim: int, h: int, gain: float = 1.5) -> dict:
    """flows.RNVP state_dict: net = Linear(dim, h); t, s = Linear(h, dim)."""
    rng = np.random.default_rng(seed)
    sd = mlp_params(rng, "net", (dim, h), gain=gain)
    for name in ("t", "s"):
        w, b = _linear(rng, dim, h, gain)
        sd[f"{name}.weight"] = w
        sd[f"{name}.bias"] = b
    return sd


def maf_params(seed: int, dim: int, h_sizes=(24, 24, 24), gain: float = 1.0, last_gain: float = 1.0) -> dict:
    """flows.MAF / IAF state_dict WITHOUT the mask buffers: net = MADE(dim, h_sizes, 2 dim) -> net.{2l}.weight / .bias
    (the masks are the deterministic construction of layers/made.py and stay what the constructor built)."""
    return mlp_params(np.random.default_rng(seed), "net", (dim, *h_sizes, 2 * dim), last_gain=last_gain, gain=gain)


def actnorm_params(seed: int, dim: int) -> dict:
    rng = np.random.default_rng(seed)
    s = (0.3 * rng.standard_normal((1, dim))).astype(np.float32)
    t = (0.5 * rng.standard_normal((1, dim))).astype(np.float32)
    return {"s": torch.from_numpy(s), "t": torch.from_numpy(t)}


def glow_params(seed: int, dim: int) -> dict:
    """P (permutation), L, S, U of a random orthogonal matrix's LU factorisation, with
    S pushed away from 0 so the matrix stays well conditioned."""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
    P, L, U = torch.linalg.lu(torch.from_numpy(q.astype(np.float32)))
    S = U.diag().clone()
    S = torch.where(S.abs() < 0.2, torch.sign(S) * 0.2 + (S == 0) * 0.2, S)
    return {"P": P.contiguous(), "L": L.contiguous(), "S": S.contiguous(),
            "U": torch.triu(U, diagonal=1).contiguous()}


def gaussian(seed: int, rows: int, dim: int, scale: float = 1.0) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy((scale * rng.standard_normal((rows, dim))).astype(np.float32))


def bernoulli_mask(seed: int, rows: int, dim: int) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.random((rows, dim)) < 0.5).astype(np.float32))


def c2_stack_params(dim: int = 64, n_layers: int = 9, base_seed: int = 1000) -> list[dict]:
    """The benchmark stack's weights (C2 / C4): layer i uses seed base_seed + i, parity
    = bool(i % 2) as in examples/half_moons.ipynb:90.  s_last_gain 2 keeps a 9-layer
    pass finite while exercising exp() away from 1."""
    return [affine_half_params(base_seed + i, dim, s_last_gain=2.0) for i in range(n_layers)]


# Fixture G10 (gen_golden.g10_padded_shapes): k-th entry uses seed 1000 + 10 k (AffineHalfFlow, parity = k odd)
# or 1100 + 10 k (RNVP)
G10_AHF = {  # tag -> (dim, constructor keywords): shapes the MFMA kernels run padded (narrow halves, hidden widths)
    "d6_h8_24_17": (6, dict(h_sizes=(8, 24, 17))),
    "d64_h20": (64, dict(h_sizes=(20, 20, 20))),
    "d50": (50, dict()),
    "d2": (2, dict()),
    "d64_nice": (64, dict(scale=False)),
    "d130_noshift": (130, dict(shift=False)),
    "d100_h32": (100, dict(h_sizes=(32, 32, 32))),
}
G10_RNVP = {"d100_h41": (100, 41), "d70_h50": (70, 50), "d49_h7": (49, 7)}
