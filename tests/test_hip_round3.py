"""Round-3 GPU tests: the stale-operand-image detector on real layers, FlatParameters / FusedAdam under the reference's
``model.zero_grad()`` loop, the MNF training path on the HIP library (RNVP gradient kernels on the matrix cores,
``MNFLinear.forward`` as an autograd function)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipes
from helpers import RTOL, assert_close, normwise_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


# ------------------------------------------------------------------------------------------------ stale-image detector
def test_stale_operand_images_are_detected_under_the_switch(amd, monkeypatch):
    """``p.data.mul_(2)`` bumps no version counter: the packed operand images go stale silently.  Under
    MNF_CHECK_PARAMS the next call raises and names ``invalidate()``; after ``invalidate()`` the layer computes with
    the new weights."""
    monkeypatch.setattr(amd.flows, "_CHECK_PARAMS_EVERY", 1)
    x = recipes.gaussian(5, 300, 64).to(DEV)
    f = amd.AffineHalfFlow(64, True)
    f.load_state_dict(recipes.affine_half_params(6, 64, s_last_gain=1.5))
    f.to(DEV)
    with torch.no_grad():
        y0, _ = f.forward(x)
        y0b, _ = f.forward(x)                       # cache hit, parameters unchanged: passes the check
        assert torch.equal(y0, y0b)
        f.t_net[6].bias.data.mul_(2).add_(1.0)       # a write the caches cannot see
        with pytest.raises(RuntimeError, match="invalidate"):
            f.forward(x)
        f.invalidate()
        y1, _ = f.forward(x)
        assert not torch.equal(y0, y1)
        # the sanctioned way to write in place bumps the version: no invalidate needed
        f.t_net[6].bias.mul_(0.5)
        y2, _ = f.forward(x)
        assert not torch.equal(y1, y2)
    # a run of layers (one stack launch) and an RNVP layer have their own caches
    layers = [amd.AffineHalfFlow(64, bool(i % 2)) for i in range(3)]
    for i, l in enumerate(layers):
        l.load_state_dict(recipes.affine_half_params(10 + i, 64))
    model = amd.NormalizingFlow(layers).to(DEV)
    with torch.no_grad():
        model.forward(x)
        layers[1].s_net[0].weight.data.mul_(1.5)
        with pytest.raises(RuntimeError, match="invalidate"):
            model.forward(x)
        model.invalidate()
        model.forward(x)
    r = amd.RNVP(800, h_sizes=(50,)).to(DEV)
    z = torch.randn(64, 800, device=DEV)
    with torch.no_grad():
        r.forward(z, seed=3)
        r.s.weight.data.mul_(2)
        with pytest.raises(RuntimeError, match="invalidate"):
            r.forward(z, seed=3)
        r.invalidate()
        r.forward(z, seed=3)


def test_without_the_switch_a_data_write_is_not_checked(amd):
    assert amd.flows._CHECK_PARAMS_EVERY == int(os.environ.get("MNF_CHECK_PARAMS", "0") or 0)


# ------------------------------------------------------------------------------------------------ FlatParameters
def test_reference_loop_with_flat_parameters_and_fused_adam(amd):
    """The reference's loop -- ``model.zero_grad(); loss.backward(); opt.step()`` (tests/test_flows.py:27) -- with
    FlatParameters + FusedAdam on a mixed model (an AffineHalfFlow run: in-place gradient sums; ActNorm / Glow /
    NSF_CL: ordinary autograd accumulation): same parameters as torch.optim.Adam on a copy, step by step.  ADVICE
    round 2: model.zero_grad() used to detach every gradient view and silently stop the non-run layers' training."""
    def build():
        torch.manual_seed(3)
        flows = [amd.ActNormFlow(8), amd.Glow(8), amd.NSF_CL(8, K=5, B=3, n_h=8)]
        flows += [amd.AffineHalfFlow(8, bool(i % 2)) for i in range(3)]
        flows[0].data_dep_init_done = True
        return amd.NormalizingFlowModel(amd.StandardNormal(8), flows).to(DEV)

    a, b = build(), build()
    b.load_state_dict(a.state_dict())
    flat = amd.FlatParameters(a)
    opt_a = amd.FusedAdam(flat, lr=1e-2)
    opt_b = torch.optim.Adam(b.parameters(), lr=1e-2)
    x = recipes.gaussian(77, 512, 8).to(DEV)
    for step in range(6):
        a.zero_grad()                      # the reference's call, not opt.zero_grad()
        la = -a.log_prob(x).mean()
        la.backward()
        opt_a.step()
        b.zero_grad()
        lb = -b.log_prob(x).mean()
        lb.backward()
        opt_b.step()
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb)), (step, float(la), float(lb))
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb
        assert_close(pa, pb, 2e-4, f"parameter {na} after 6 steps")
    # every gradient is still a view of the one buffer
    assert all(p.grad is flat._grad_views[i] for i, p in enumerate(flat.params))


def test_torch_optimizer_on_a_flat_homed_model_still_trains_the_run_layers(amd):
    """A torch optimizer's zero_grad(set_to_none=True) detaches p.grad from the flat gradient buffer; the affine run
    must then hand its gradients to autograd instead of adding them into the (now invisible) buffer."""
    torch.manual_seed(5)
    layers = [amd.AffineHalfFlow(64, bool(i % 2)) for i in range(3)]
    model = amd.NormalizingFlowModel(amd.StandardNormal(64), layers).to(DEV)
    amd.FlatParameters(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = recipes.gaussian(9, 400, 64).to(DEV)
    before = [p.detach().clone() for p in model.parameters()]
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        (-model.log_prob(x).mean()).backward()
        assert all(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in model.parameters())
        opt.step()
    assert all(not torch.equal(p.detach(), q) for p, q in zip(model.parameters(), before))


def test_fused_stack_falls_back_without_double_counting(amd, O):
    """FusedAffineStack over more than 32 layers goes out in chunks; if a LATER chunk has no stack launch after an
    earlier one has already added its log-dets, the rest runs layer by layer from where the chunks stopped (ADVICE
    round 2: it used to restart from x and count the first chunks twice)."""
    dim, n = 8, 40
    sds = [recipes.affine_half_params(900 + i, dim, s_last_gain=0.5) for i in range(n)]
    layers = []
    for i, sd in enumerate(sds):
        f = amd.AffineHalfFlow(dim, bool(i % 2))
        f.load_state_dict(sd)
        layers.append(f)
    stack = amd.FusedAffineStack(layers).to(DEV)
    x = recipes.gaussian(901, 333, dim)
    specs = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd} for i, sd in enumerate(sds)]
    ref_list, ref_ld = O.flow_stack(x, specs, inverse=False)
    with torch.no_grad():
        y, ld = stack.forward(x.to(DEV))
        assert_close(y, ref_list[-1], RTOL, "y")
        assert_close(ld, ref_ld, RTOL, "ld")
        # make the SECOND chunk's launch fail: its run reports "unsupported" at launch time
        runs = stack._run_helpers
        assert len(runs) == 2
        real_launch = runs[1].launch
        runs[1].launch = lambda *a, **k: None
        try:
            y2, ld2 = stack.forward(x.to(DEV))
        finally:
            runs[1].launch = real_launch
        assert_close(y2, ref_list[-1], RTOL, "y after a failed second chunk")
        assert_close(ld2, ref_ld, RTOL, "ld after a failed second chunk")


def test_gradient_scale_sample_spans_the_batch(amd):
    """mnf_affine_half_grad_scale samples up to 512 rows spread evenly over the batch: a batch whose first rows
    carry zero cotangents (sorted / masked batches) still gets the scale of the rows that matter."""
    rows, dim = 8192, 64
    g = torch.zeros(rows, dim, device=DEV)
    g[rows // 2:] = 3e-6 * torch.randn(rows // 2, dim, device=DEV)
    scale = amd.flows._grad_scale(g, None, rows, dim, torch.device(DEV))
    top = float(g[::rows // 512].abs().max())
    assert 1.0 <= float(scale) * top < 2.0
