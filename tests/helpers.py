"""Shared test helpers: tolerance rule, fixture -> oracle layer specs."""
from __future__ import annotations

import numpy as np
import torch

import recipes

# SURVEY.md 8c / BASELINE.json north_star: "1e-5 rel fp32", read normwise:
# max|a - b| <= RTOL * max|b| per output tensor.
RTOL = 1e-5


def normwise_err(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    same_nan = np.isnan(a) == np.isnan(b)
    assert same_nan.all(), "NaN pattern differs"
    m = ~np.isnan(b)
    if not m.any():
        return 0.0
    scale = max(np.abs(b[m]).max(), 1e-30)
    return float(np.abs(a[m] - b[m]).max() / scale)


def assert_close(a, b, rtol: float = RTOL, what: str = ""):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    err = normwise_err(a, b)
    assert err <= rtol, f"{what}: normwise error {err:.3e} > {rtol:.1e}"
    return err


# every assert_parity call leaves a record here (what, err, budget, widening): the audit trail of how much of each
# budget was used and how much of it was float64 head-room (tests/test_hip_parity.py::test_parity_budget_audit prints it)
PARITY_LOG: list[dict] = []
# widening a NON-stress fixture may claim: beyond this the float64 head-room is hiding something (the G4 edge cases
# and the x2-weight stress fixtures pass max_widening=None and say why)
MAX_WIDENING = 5e-5


def assert_parity(got, ref32, ref64=None, what: str = "", rtol: float = RTOL, max_widening: float | None = MAX_WIDENING):
    """The parity rule.  max|got - ref32| <= rtol * max|ref32| (BASELINE.json: 1e-5 rel fp32,
    normwise).  Where the fixture also carries the reference's own float64 run, the budget is
    widened by twice the reference's fp32-vs-fp64 distance on the same inputs: on
    ill-conditioned inputs (spline knots a few ulps from the sample, stress-gain weights) two
    correct fp32 evaluations differ by about that much, and 1e-5 alone is below the
    reference's own rounding noise (measured: up to 1.5e-4 on the G4 edge cases).

    The widening is capped: above ``max_widening`` (5e-5) the call fails unless the caller passes
    ``max_widening=None`` -- the stress fixtures, which are built to sit on knots.  Every call is recorded in
    ``PARITY_LOG`` with the share of the budget that was float64 head-room, and returns the error."""
    if isinstance(got, torch.Tensor):
        got = got.detach().cpu().numpy()
    widening = 0.0
    if ref64 is not None:
        widening = 2.0 * normwise_err(np.asarray(ref32), np.asarray(ref64))
    budget = rtol + widening
    err = normwise_err(got, np.asarray(ref32))
    PARITY_LOG.append({"what": what, "err": err, "budget": budget, "widening": widening,
                       "widening_share": widening / budget, "used": err / budget, "stress": max_widening is None})
    if max_widening is not None:
        assert widening <= max_widening, (f"{what}: the float64 head-room {widening:.3e} exceeds {max_widening:.1e} "
                                          f"on a fixture that is not a stress case (err {err:.3e})")
    assert err <= budget, (f"{what}: normwise error {err:.3e} > budget {budget:.3e} "
                           f"(= {rtol:.1e} + float64 head-room {widening:.3e})")
    return err


# one record per gradient / optimiser comparison against a float64 oracle (test_hip_autograd.OracleGrads.check,
# check_vs_float64, budgeted()): what, err, budget, widening -- tests/test_zz_audit.py enforces the 80 % rule on these too
GRAD_LOG: list[dict] = []


def budgeted(err: float, budget: float, what: str, stress: bool = False) -> float:
    """A comparison with a fixed, stated budget (no float64 head-room to compute): recorded for the audit like the
    others.  ``stress``: exempt from the audit's 80 % rule (say why at the call)."""
    GRAD_LOG.append({"what": what, "err": err, "budget": budget, "widening": 0.0, "stress": stress})
    assert err <= budget, f"{what}: {err:.3e} > budget {budget:.3e}"
    return err


def parity_report() -> str:
    """One line per recorded assert_parity call that used float64 head-room, worst first."""
    rows = sorted((r for r in PARITY_LOG if r["widening"] > 0), key=lambda r: -r["widening"])
    return "\n".join(f"{r['what']:48s} err {r['err']:.2e}  budget {r['budget']:.2e}  of which fp64 head-room "
                     f"{r['widening']:.2e} ({100 * r['widening_share']:.0f} %)  used {100 * r['used']:.0f} %"
                     for r in rows)


def t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a))


def unpack_mask(bits, dim: int) -> torch.Tensor:
    return t(np.unpackbits(bits, axis=1)[:, :dim].astype(np.float32))


def g1_layers(fx) -> list[dict]:
    layers = []
    for i in range(9):
        pre = f"L{i}."
        sd = {k[len(pre):]: t(v) for k, v in fx.items() if k.startswith(pre)}
        layers.append({"kind": "affine_half", "parity": bool(i % 2), "params": sd})
    return layers


def c2_layers(dim: int = 64, n_layers: int = 9) -> list[dict]:
    return [{"kind": "affine_half", "parity": bool(i % 2), "params": sd}
            for i, sd in enumerate(recipes.c2_stack_params(dim, n_layers))]


def c3_layers(fx=None, dim: int = 32, K: int = 8, n_h: int = 8) -> list[dict]:
    """3 x [ActNorm, Glow, NSF_CL]; ActNorm (s, t) from the G6 fixture when given."""
    layers = []
    for i in range(3):
        if fx is not None:
            an = {"s": t(fx[f"actnorm{i}.s"]), "t": t(fx[f"actnorm{i}.t"])}
        else:
            an = recipes.actnorm_params(630 + i, dim)
        layers.append({"kind": "affine_const", "params": an})
        layers.append({"kind": "glow", "params": recipes.glow_params(600 + i, dim)})
        layers.append({"kind": "nsf_cl", "K": K, "B": 3.0,
                       "params": recipes.nsf_cl_params(610 + i, dim, K, n_h)})
    return layers
