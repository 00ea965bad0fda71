"""GPU tests of Glow's parameter preparation in one launch each way (``mnf_glow_weight`` / ``_bwd``): W, its inverse by
triangular substitution and log_det against the oracle's composition (flows/glow.py:20-37) in float64, their gradients
against autograd through it, and the layer end to end against the stock-op composition it replaces."""
import numpy as np
import pytest
import torch

import recipes
from helpers import assert_close, normwise_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def amd():
    import torch_mnf_amd

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    torch_mnf_amd._lib.load()
    return torch_mnf_amd


@pytest.fixture(scope="module")
def O():
    from oracle import flow_oracle

    return flow_oracle


def plu(seed, d):
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(d, d, generator=g))
    P, L, U = torch.linalg.lu(q)
    # (full matrices, as the reference keeps them: entries outside the triangles must not matter)
    return P, L + 0.3 * torch.triu(torch.randn(d, d, generator=g), 0), U.diag().clone(), \
        torch.triu(U, 1) + 0.3 * torch.tril(torch.randn(d, d, generator=g), 0)


@pytest.mark.parametrize("d", [1, 2, 5, 32, 47, 64])
@pytest.mark.parametrize("inverse", [False, True])
def test_glow_weight_and_gradients_vs_float64_oracle(amd, O, d, inverse):
    from torch_mnf_amd.flows import _GlowWeightFn

    P, L, S, U = plu(9000 + d, d)
    g = torch.Generator().manual_seed(9100 + d)
    G, gl = torch.randn(d, d, generator=g), torch.randn((), generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        Ld, Sd, Ud = (t.to(dt).clone().requires_grad_(True) for t in (L, S, U))
        W = O.glow_weight(P.to(dt), Ld, Sd, Ud)
        M = torch.inverse(W) if inverse else W
        ld = Sd.abs().log().sum() * (-1 if inverse else 1)
        ((M * G.to(dt)).sum() + ld * gl.to(dt)).backward()
        ref[dt] = (M.detach(), ld.detach(), Ld.grad, Sd.grad, Ud.grad)
    Lg, Sg, Ug = (t.to(DEV).requires_grad_(True) for t in (L, S, U))
    M, ld = _GlowWeightFn.apply(Lg, Sg, Ug, P.to(DEV).contiguous(), inverse, None)
    ((M * G.to(DEV)).sum() + ld * gl.to(DEV)).backward()
    m64, ld64, gL64, gS64, gU64 = ref[torch.float64]
    m32, ld32, gL32, gS32, gU32 = ref[torch.float32]
    for name, got, r64, r32 in (("matrix", M, m64, m32), ("dL", Lg.grad, gL64, gL32), ("dS", Sg.grad, gS64, gS32),
                                ("dU", Ug.grad, gU64, gU32)):
        widen = 2 * normwise_err(r32.double().numpy(), r64.numpy())
        err = normwise_err(got.detach().cpu().double().numpy(), r64.numpy())
        assert err <= 1e-5 + widen, f"d={d} inverse={inverse} {name}: {err:.2e} > 1e-5 + {widen:.2e}"
    assert abs(float(ld.detach()) - float(ld64)) <= 1e-5 * max(1.0, abs(float(ld64)))
    # only the triangles the layer uses receive gradient (glow.py:21-23)
    assert float(torch.triu(Lg.grad, 0).abs().max()) == 0.0 and float(torch.tril(Ug.grad, 0).abs().max()) == 0.0


@pytest.mark.parametrize("d,rows", [(2, 128), (32, 3000)])
@pytest.mark.parametrize("inverse", [False, True])
def test_glow_layer_gradients_match_the_stock_composition(amd, d, rows, inverse, monkeypatch):
    """The layer end to end (x @ W or x @ W^-1, log_det) with the fused preparation against the stock-op composition
    it replaces (the path dims above 64 still take); a FlatParameters home receives the same gradients in place."""
    import torch_mnf_amd.flows as fl

    torch.manual_seed(3)
    x = recipes.gaussian(9200 + d, rows, d).to(DEV)
    w = recipes.gaussian(9201 + d, rows, d).to(DEV)

    def run(fused, homed=False):
        monkeypatch.setattr(fl, "_GLOW_WEIGHT_MAX_DIM", 64 if fused else 0)
        torch.manual_seed(11)
        layer = amd.Glow(d).to(DEV)
        flat = amd.FlatParameters(layer) if homed else None
        xx = x.clone().requires_grad_(True)
        y, ld = layer.inverse(xx) if inverse else layer.forward(xx)
        ((y * w).sum() + 3.0 * ld.sum()).backward()
        return y.detach(), ld.detach(), {"x": xx.grad, **{n: p.grad.clone() for n, p in layer.named_parameters()}}, flat

    y1, ld1, g1, _ = run(True)
    y0, ld0, g0, _ = run(False)
    assert_close(y1, y0, 2e-5, "y")
    assert abs(float(ld1) - float(ld0)) <= 1e-5 * max(1.0, abs(float(ld0)))
    for k in g0:
        assert_close(g1[k], g0[k], 5e-5, f"gradient {k}")
    _, _, g2, flat = run(True, homed=True)
    for k in g0:
        assert_close(g2[k], g1[k], 2e-6, f"flat-home gradient {k}")
    assert all(p.grad is v for p, v in zip(flat.params, flat._grad_views))
