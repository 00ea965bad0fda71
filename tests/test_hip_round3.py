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
    # (the optimiser-equivalence budget of tests/test_hip_round2.py: 1 % of the distance six steps can move a parameter)
    from helpers import budgeted
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb
        scale = float(pb.detach().abs().max())
        budgeted(normwise_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()), 0.01 * 6 * 1e-2 / max(scale, 1e-30),
                 f"FusedAdam vs torch.optim.Adam (mixed model), 6 steps: {na}")
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


# ------------------------------------------------------------------------------------------------ RNVP gradients (MFMA)
def _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l):
    """(fp32, fp64) gradient dicts of sum(x * w_x) + sum(log_det * w_l) through the oracle."""
    out = []
    for dt in (torch.float32, torch.float64):
        zz = z.detach().to(dt).requires_grad_(True)
        p = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        x, ld = O.rnvp(zz, p, mask.to(dt))
        loss = 0
        if w_x is not None:
            loss = loss + (x * w_x.to(dt)).sum()
        if w_l is not None:
            loss = loss + (ld * w_l.to(dt)).sum()
        loss.backward()
        out.append({"z": zz.grad, **{k: v.grad for k, v in p.items()}})
    return out


def _check_grads(got, g32, g64, what, base=1e-5):
    worst = 0.0
    for k, r64 in g64.items():
        if r64 is None or float(r64.abs().max()) == 0.0:  # the loss does not depend on this parameter
            assert got[k] is None or float(got[k].abs().max()) == 0.0, f"{what}: grad {k} should be zero"
            continue
        widening = 2.0 * normwise_err(g32[k].numpy(), r64.numpy())
        err = normwise_err(got[k].detach().double().cpu().numpy(), r64.numpy())
        assert err <= base + widening, (f"{what}: grad {k} is {err:.3e} from the float64 oracle; budget "
                                        f"{base + widening:.3e} (fp32 oracle vs fp64 {widening / 2:.2e})")
        worst = max(worst, err)
    return worst


@pytest.fixture(autouse=True)
def _mfma_gradient_path_at_every_size(monkeypatch):
    """The library sends few-row / narrow RNVP layers to the generic gradient kernel (it is faster there); these tests
    exercise the matrix-core kernels at every size."""
    import torch_mnf_amd.flows as fl

    monkeypatch.setattr(fl._dispatch, "RNVP_BWD_MFMA_MIN_ROWS", 0)
    monkeypatch.setattr(fl._dispatch, "RNVP_BWD_MFMA_MIN_DIM", 0)
    monkeypatch.setattr(fl._dispatch, "RNVP_BWD_FEW_GRID_OFF", True)  # (batches of <= 512 rows otherwise take mnf_rnvp_bwd_few)


def _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask=None, seed=None, generic=False):
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    f.to(DEV)
    f.force_generic = generic
    zz = z.detach().to(DEV).requires_grad_(True)
    x, ld = f.forward(zz, mask=None if mask is None else mask.to(DEV), seed=seed)
    loss = 0
    if w_x is not None:
        loss = loss + (x * w_x.to(DEV)).sum()
    if w_l is not None:
        loss = loss + (ld * w_l.to(DEV)).sum()
    loss.backward()
    return f, {"z": zz.grad, **{n: q.grad for n, q in f.named_parameters()}}


@pytest.mark.parametrize("dim,hid,rows", [(800, 50, 70), (800, 50, 1000), (50, 50, 129), (64, 30, 33), (784, 50, 300),
                                          (100, 17, 257), (96, 50, 1), (128, 8, 4099), (800, 64, 300), (100, 57, 131), (96, 64, 1)])
@pytest.mark.parametrize("masked", ["explicit", "seeded"])
def test_rnvp_mfma_gradient_kernels(amd, O, dim, hid, rows, masked):
    """mnf_rnvp_bwd_mfma (row-parallel launch A + dims-slab launch B) against autograd through the oracle in float64
    and against the generic gradient kernel: MNF-LeNet's widths (800, 50), ragged widths (784, 100, 50), hidden widths
    below the kernels' 30 / 50 units, a single row, ragged row counts, explicit float masks and the in-kernel mask."""
    sd = recipes.rnvp_params(3100 + dim + hid, dim, hid)
    z = recipes.gaussian(3200 + dim, rows, dim)
    w_x = recipes.gaussian(3300 + dim, rows, dim)
    w_l = recipes.gaussian(3400 + dim, rows, 1)[:, 0]
    if masked == "explicit":
        mask, seed = recipes.bernoulli_mask(3500 + dim, rows, dim), None
    else:
        seed = 4242 + dim
        probe = amd.RNVP(dim, h_sizes=(hid,))
        mask = probe.mask_for(seed, rows).cpu()
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    f, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask if masked == "explicit" else None, seed)
    assert f._bwd_index(torch.device(DEV, 0)), "this shape should have the matrix-core gradient kernels"
    worst = _check_grads(got, g32, g64, f"rnvp mfma d={dim} h={hid} rows={rows} {masked}")
    _, gen = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask if masked == "explicit" else None, seed, generic=True)
    for k in got:
        assert_close(got[k], gen[k], 2e-5, f"mfma vs generic {k}")
    print(f"rnvp mfma gradients d={dim} h={hid} rows={rows} {masked}: worst {worst:.2e} from float64")


@pytest.mark.parametrize("which", ["x_only", "ld_only"])
def test_rnvp_mfma_gradient_kernels_absent_cotangents(amd, O, which):
    dim, hid, rows = 800, 50, 200
    sd = recipes.rnvp_params(3601, dim, hid)
    z = recipes.gaussian(3602, rows, dim)
    mask = recipes.bernoulli_mask(3603, rows, dim)
    w_x = recipes.gaussian(3604, rows, dim) if which == "x_only" else None
    w_l = recipes.gaussian(3605, rows, 1)[:, 0] if which == "ld_only" else None
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    _, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask)
    _check_grads(got, g32, g64, f"rnvp mfma {which}")


@pytest.mark.parametrize("magnitude", [1e-7, 1.0, 300.0])
def test_rnvp_mfma_gradient_kernels_do_not_depend_on_the_gradient_scale(amd, O, magnitude):
    """Cotangents of a mean over 256,000 rows are ~4e-6 (far below f16's normal range): the kernels normalise them by a
    power of two from a sample of the rows and scale the results back."""
    dim, hid, rows = 800, 50, 600
    sd = recipes.rnvp_params(3701, dim, hid)
    z = recipes.gaussian(3702, rows, dim)
    mask = recipes.bernoulli_mask(3703, rows, dim)
    w_x = recipes.gaussian(3704, rows, dim) * magnitude
    w_l = recipes.gaussian(3705, rows, 1)[:, 0] * magnitude
    w_x[550:] *= 20.0   # outside the 512-row sample
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    _, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask)
    _check_grads(got, g32, g64, f"rnvp mfma, cotangents x {magnitude}")


@pytest.mark.parametrize("case", ["big_rows", "big_weights", "one_huge_gradient"])
def test_rnvp_mfma_gradient_kernels_range_guard(amd, O, case):
    """128-row groups with an operand outside the split range are flagged by launch A, skipped by launch B and redone
    by the generic fp32 kernel on exactly those groups; weights beyond the limit send every group there."""
    dim, hid, rows = 800, 50, 128 * 5 + 37
    sd = recipes.rnvp_params(3801, dim, hid)
    z = recipes.gaussian(3802, rows, dim)
    mask = recipes.bernoulli_mask(3803, rows, dim)
    w_x = recipes.gaussian(3804, rows, dim)
    w_l = recipes.gaussian(3805, rows, 1)[:, 0]
    if case == "big_rows":
        z = z.clone()
        z[130] *= 3e4          # one row of group 1
        z[400:410] *= 1e5      # group 3
        sd = {k: (v * 1e-5 if k == "net.0.weight" else v) for k, v in sd.items()}
    elif case == "big_weights":
        sd = {k: (v * 1e4 if k == "net.0.weight" else v) for k, v in sd.items()}
        z = z * 1e-4
    else:
        w_x = w_x.clone()
        w_x[300, 5] = 1e7      # in the sample: it sets the scale, every other row becomes small (not wrong)
        w_x[640] *= 3e4        # last group, beyond the sample: 3e4 x the scale -> out of range there
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    _, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask)
    _check_grads(got, g32, g64, f"rnvp mfma range guard {case}", base=2e-5)


def test_rnvp_mfma_gradient_kernels_many_rows(amd):
    """Enough rows for several row parts per XCD and several trips per wave (65,536 x 800: 70 copies of a 937-row
    batch, so that every copy's grad_z must equal the single batch's and the parameter gradients 70 x it)."""
    dim, hid, rows, copies = 800, 50, 937, 70
    sd = recipes.rnvp_params(3901, dim, hid)
    z = recipes.gaussian(3902, rows, dim)
    mask = recipes.bernoulli_mask(3903, rows, dim)
    w_x = recipes.gaussian(3904, rows, dim)
    w_l = recipes.gaussian(3905, rows, 1)[:, 0]
    _, one = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask)
    _, many = _rnvp_gpu_grads(amd, sd, dim, hid, z.repeat(copies, 1), w_x.repeat(copies, 1), w_l.repeat(copies),
                              mask.repeat(copies, 1))
    assert_close(many["z"], one["z"].repeat(copies, 1), 2e-6, "grad_z of the copies")
    for k in one:
        if k != "z":
            assert_close(many[k], copies * one[k], 1e-5, f"{k}: {copies} copies")


# ------------------------------------------------------------------------------------------------ MNFLinear.forward gradients
def _mnf_linear_oracle_grads(O, p, x, z, eps, w_out):
    """(fp32, fp64) gradients of sum(out * w_out) through the oracle's MNFLinear.forward (mnf_linear.py:46-56)."""
    out = []
    for dt in (torch.float32, torch.float64):
        xx = x.detach().to(dt).requires_grad_(True)
        zz = z.detach().to(dt).requires_grad_(True)
        q = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in p.items()}
        o = O.mnf_linear_forward(xx, zz, q["W_mean"], q["W_log_var"], q["b_mean"], q["b_log_var"], eps.to(dt))
        (o * w_out.to(dt)).sum().backward()
        out.append({"x": xx.grad, "z": zz.grad, **{k: v.grad for k, v in q.items()}})
        val = o.detach()
    return out[0], out[1], val


def _mnf_linear_params(seed, n_in, n_out):
    g = torch.Generator().manual_seed(seed)
    return {"W_mean": 0.1 * torch.randn(n_out, n_in, generator=g), "W_log_var": -9 + 0.5 * torch.randn(n_out, n_in, generator=g),
            "b_mean": 0.1 * torch.randn(n_out, generator=g), "b_log_var": -9 + 0.5 * torch.randn(n_out, generator=g)}


class _FixedZ:
    """Stand-in for sample_z: MNFLinear.forward draws z itself; the gradient tests need a known z with a grad slot."""

    def __init__(self, z):
        self.z = z

    def __call__(self, batch_size=1, eps=None, masks=None):
        return self.z, torch.zeros(self.z.shape[0], device=self.z.device)


@pytest.mark.parametrize("n_in,n_out,rows", [(800, 50, 300), (50, 10, 129), (800, 50, 1), (784, 50, 70), (100, 64, 257),
                                             (96, 17, 4099), (33, 3, 40)])
def test_mnf_linear_forward_gradients(amd, O, n_in, n_out, rows):
    """MNFLinear.forward with gradients = mnf_mnf_linear_fwd_train + mnf_mnf_linear_bwd: grad x, grad z, dW_mean,
    dW_log_var, db_mean, db_log_var against autograd through the oracle evaluated in float64 (MNF-LeNet's two dense
    shapes, ragged widths, a single row, ragged row counts)."""
    p = _mnf_linear_params(5000 + n_in + n_out, n_in, n_out)
    x = recipes.gaussian(5100 + n_in, rows, n_in).abs()          # (activations after a ReLU in the reference's models)
    z = 1.0 + 0.3 * recipes.gaussian(5200 + n_in, rows, n_in)
    eps = recipes.gaussian(5300 + n_out, rows, n_out)
    w_out = recipes.gaussian(5400 + n_out, rows, n_out)
    g32, g64, ref_out = _mnf_linear_oracle_grads(O, p, x, z, eps, w_out)
    layer = amd.MNFLinear(n_in, n_out)
    layer.load_state_dict(p, strict=False)
    layer.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    zg = z.to(DEV).requires_grad_(True)
    layer.sample_z = _FixedZ(zg)
    out = layer.forward(xg, eps=eps.to(DEV))
    assert out.requires_grad and type(out.grad_fn).__name__.startswith("_MnfLinearFn")
    assert_close(out, ref_out.float(), RTOL, "out")
    (out * w_out.to(DEV)).sum().backward()
    got = {"x": xg.grad, "z": zg.grad, **{k: getattr(layer, k).grad for k in p}}
    worst = _check_grads(got, g32, g64, f"MNFLinear({n_in},{n_out}) rows={rows}")
    print(f"MNFLinear({n_in},{n_out}) rows={rows}: worst gradient {worst:.2e} from float64")


def test_mnf_linear_forward_gradients_with_in_kernel_noise_and_tiny_cotangents(amd, O):
    """The default call: noise generated in the kernel from a seed (the backward pass regenerates it), cotangents of a
    mean over the batch (~1e-6: normalised by the gradient scale)."""
    n_in, n_out, rows = 800, 50, 2000
    p = _mnf_linear_params(5501, n_in, n_out)
    x = recipes.gaussian(5502, rows, n_in).abs()
    z = 1.0 + 0.3 * recipes.gaussian(5503, rows, n_in)
    layer = amd.MNFLinear(n_in, n_out)
    layer.load_state_dict(p, strict=False)
    layer.to(DEV)
    xg, zg = x.to(DEV).requires_grad_(True), z.to(DEV).requires_grad_(True)
    layer.sample_z = _FixedZ(zg)
    torch.manual_seed(77)
    out = layer.forward(xg)
    torch.manual_seed(77)
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
    eps = layer.noise_for(seed, rows).cpu()
    w_out = recipes.gaussian(5504, rows, n_out) / (rows * n_out)
    g32, g64, ref_out = _mnf_linear_oracle_grads(O, p, x, z, eps, w_out)
    assert_close(out, ref_out.float(), RTOL, "out (in-kernel noise)")
    (out * w_out.to(DEV)).sum().backward()
    got = {"x": xg.grad, "z": zg.grad, **{k: getattr(layer, k).grad for k in p}}
    _check_grads(got, g32, g64, "MNFLinear, in-kernel noise, mean-loss cotangents")


def test_mnf_linear_forward_gradients_range_guard(amd, O):
    """Rows whose x z or x^2 leave the split range are flagged by the FORWARD launch; the backward pass skips their
    128-row groups on the matrix cores and redoes them in fp32."""
    n_in, n_out, rows = 800, 50, 128 * 3 + 50
    p = _mnf_linear_params(5601, n_in, n_out)
    x = recipes.gaussian(5602, rows, n_in).abs()
    x[140] *= 400.0          # x^2 ~ 1e5 in group 1
    x[300:303] *= 1e3        # group 2
    z = 1.0 + 0.3 * recipes.gaussian(5603, rows, n_in)
    eps = recipes.gaussian(5604, rows, n_out)
    w_out = recipes.gaussian(5605, rows, n_out)
    g32, g64, ref_out = _mnf_linear_oracle_grads(O, p, x, z, eps, w_out)
    layer = amd.MNFLinear(n_in, n_out)
    layer.load_state_dict(p, strict=False)
    layer.to(DEV)
    xg, zg = x.to(DEV).requires_grad_(True), z.to(DEV).requires_grad_(True)
    layer.sample_z = _FixedZ(zg)
    out = layer.forward(xg, eps=eps.to(DEV))
    assert_close(out, ref_out.float(), RTOL, "out")
    (out * w_out.to(DEV)).sum().backward()
    got = {"x": xg.grad, "z": zg.grad, **{k: getattr(layer, k).grad for k in p}}
    _check_grads(got, g32, g64, "MNFLinear range guard", base=2e-5)


def test_mnf_linear_training_path_issues_hip_launches_only(amd):
    """MNFLinear.forward + kl_div with gradients: no stock-PyTorch matrix product (aten::mm / addmm / bmm / linear) in
    the forward + backward trace of the layer's forward -- the products are libmnf_hip.so launches (the
    training path used to fall back to `(x * z) @ W_mean.T`)."""
    from torch.profiler import ProfilerActivity, profile

    torch.manual_seed(3)
    layer = amd.MNFLinear(800, 50).to(DEV)
    x = torch.rand(128, 800, device=DEV)
    layer.forward(x).sum().backward()  # warm-up: index tables, images
    layer.zero_grad()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = layer.forward(x)
        out.pow(2).mean().backward()
    names = {e.name for e in prof.events()}
    gemms = sorted(n for n in names if n in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear", "aten::matmul"))
    assert not gemms, f"stock-PyTorch matrix products on the MNFLinear.forward training path: {gemms}"
    assert any("_MnfLinearFn" in n for n in names) and any("_RnvpFn" in n for n in names)
    for name, prm in layer.named_parameters():
        if name.startswith(("r0_", "flow_r")):
            continue  # (kl_div's side of the layer: not touched by forward)
        assert prm.grad is not None and torch.isfinite(prm.grad).all() and float(prm.grad.abs().sum()) > 0, name


def test_mnf_linear_rejects_what_the_kernels_cannot_take(amd):
    layer = amd.MNFLinear(32, 10).to(DEV)
    with pytest.raises(ValueError):
        layer.forward(torch.randn(4, 31, device=DEV))          # wrong width (used to read out of bounds)
    with pytest.raises(RuntimeError):
        layer.forward(torch.randn(4, 32))                      # CPU input: no fallback
    wide = amd.MNFLinear(32, 100).to(DEV)
    with torch.no_grad():
        assert wide.forward(torch.randn(4, 32, device=DEV)).shape == (4, 100)  # (round 4: 64-output slabs, no width limit)
    with pytest.raises(TypeError):
        layer.forward(torch.randn(4, 32, device=DEV, dtype=torch.float64))      # float64: documented in INTEGRATION.md


# ------------------------------------------------------------------------------------------------ bench lines
def _run_bench(*args):
    import json

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env={**os.environ, "MNF_BENCH_SECONDARY_STEPS": "3"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_c5t_training_step_line(amd):
    """--workload c5t: one Adam step of MNFLinear(800, 50) at 256,000 rows per step; the line carries launch B-ts's
    roofline, the oracle's training step as cpu_baseline and the gradients' parity against the float64 oracle at
    1e-5 + twice the fp32 oracle's own distance from it."""
    line = _run_bench("--workload", "c5t", "--steps", "3", "--warmup", "1", "--prime-ms", "5")
    r = line["roofline"]
    assert line["unit"] == "rows/s" and r["bound"] == "hbm" and "rnvp_bwd_ts" in r["kernel"]
    assert r["launches_timed"] == 3 * 2 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert line["loss_last_step"] < line["loss_first_step"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["parity"]["worst_parameter_gradient_normwise_err"] <= line["parity"]["tolerance"]
    assert line["parity"]["loss_gpu_vs_cpu_rel_err"] <= 1e-6


def test_bench_lenet_training_step_line(amd):
    """--workload lenet: the MNF-LeNet training step at batch 128 replayed from one hipGraph; the loss falls, a replay
    is faster than the eager step it records, and -- the round-2 verdict's bar -- takes less than 3 ms."""
    line = _run_bench("--workload", "lenet", "--steps", "30", "--warmup", "5", "--prime-ms", "5")
    assert line["unit"] == "images/s" and line["config"]["batch"] == 128
    assert line["loss_last_step"] < line["loss_first_step"]
    assert line["ms_per_step"] < line["eager_ms_per_step"]
    assert line["ms_per_step"] < 3.0, line["ms_per_step"]


def test_bench_default_line_carries_the_other_configurations(amd):
    """The default (driver-run) invocation attaches 20-step runs of c3, c4, c5, c5b, c6, c2t, c3t, c5t and the MNF-LeNet step as `secondary`
    (only c2 used to be driver-observable), the per-step median / min and a one-thread CPU
    figure."""
    line = _run_bench("--steps", "5", "--warmup", "2", "--prime-ms", "5", "--no-cpu-baseline")
    assert line["config"]["workload"].startswith("9xAffineHalfFlow d=64")
    sec = line["secondary"]
    assert set(sec) == {"c3", "c4", "c5", "c5b", "c6", "c1", "c2_fp32", "c2t", "c3t", "c5t", "lenet"}
    assert "ahf_rt" in sec["c6"]["kernel"]  # (round 6: a shape without per-shape kernel, on the run-time-shaped one)
    for w, d in sec.items():
        assert "error" not in d, (w, d)
        assert d["ms_per_step"] > 0 and (w == "c1" or d["bound"] in ("hbm", "valu", "mfma")), (w, d)
        assert w in ("lenet", "c1") or d["avg_kernel_us"] > 0, (w, d)  # (lenet, c1: launch-bound, no dominant kernel)
    assert sec["c1"]["latency_us"]["graphed_pass"] > 0 and sec["c1"]["latency_us"]["eager_pass"] > 0
    assert sec["c2_fp32"]["arithmetic"] == "fp32 MFMA" and "mfma" in sec["c2_fp32"]["kernel"]
    assert line["min_ms"] <= line["median_ms"]


def test_bench_cpu_baseline_has_a_one_thread_figure(amd):
    line = _run_bench("--steps", "3", "--warmup", "1", "--prime-ms", "5", "--no-secondary")
    cb = line["cpu_baseline"]
    assert cb["one_thread"]["value"] > 0 and cb["value"] > 0
    # one thread and many threads measured on the SAME rows
    assert cb["one_thread"]["multi_thread_on_the_same_rows"] > 0 and cb["one_thread"]["rows"] >= 1 << 16
    assert line["parity"]["rel_err"] <= line["parity"]["tolerance"]


@pytest.mark.parametrize("dim,hid,rows,masked", [(784, 50, 4500, "explicit"), (800, 50, 4200, "explicit"),
                                                 (100, 17, 5000, "seeded"), (784, 30, 4100, "seeded")])
def test_rnvp_gradient_pass_with_the_forward_pass_y_streaming_kernels(amd, O, dim, hid, rows, masked, monkeypatch):
    """The streaming split forward kernel keeps y too (explicit masks, ragged widths, hidden widths below the tile):
    same gradients as with launch A's own first sweep, and the float64 oracle on a slice."""
    import torch_mnf_amd.flows as fl

    monkeypatch.setattr(fl._dispatch, "RNVP_BWD_MFMA_MIN_DIM", 0)
    sd = recipes.rnvp_params(8200 + dim, dim, hid)
    z = recipes.gaussian(8201 + dim, rows, dim)
    z[7] *= 3.0e4  # one group through the fp32 bodies
    w_x = recipes.gaussian(8202 + dim, rows, dim)
    w_l = recipes.gaussian(8203 + dim, rows, 1)[:, 0]
    probe = amd.RNVP(dim, h_sizes=(hid,))
    seed = 5 + dim
    mask = recipes.bernoulli_mask(8204 + dim, rows, dim) if masked == "explicit" else probe.mask_for(seed, rows).cpu()

    def run(keep):
        monkeypatch.setattr(fl._dispatch, "RNVP_KEEP_Y_MIN_ROWS", 4096 if keep else 1 << 60)
        f = amd.RNVP(dim, h_sizes=(hid,))
        f.load_state_dict(sd)
        f.to(DEV)
        zz = z.to(DEV).requires_grad_(True)
        x, ld = f.forward(zz, mask=mask.to(DEV)) if masked == "explicit" else f.forward(zz, seed=seed)
        kept = x.grad_fn.kept_y
        ((x * w_x.to(DEV)).sum() + (ld * w_l.to(DEV)).sum()).backward()
        return kept, {"z": zz.grad, **{n: p.grad for n, p in f.named_parameters()}}

    kept, g1 = run(True)
    none, g0 = run(False)
    assert kept is not None and none is None and torch.isnan(kept[7]).all() and not torch.isnan(kept[200:]).any()
    for k in g0:
        assert_close(g1[k], g0[k], 2e-6, f"gradient {k} with the kept y vs recomputed")
    sl = slice(1024, 1400)
    g32, g64 = _rnvp_oracle_grads(O, sd, z[sl], mask[sl], w_x[sl], w_l[sl])
    widen = 2.0 * normwise_err(g32["z"].numpy(), g64["z"].numpy())
    assert normwise_err(g1["z"][sl].cpu().double().numpy(), g64["z"].numpy()) <= 1e-5 + widen


@pytest.mark.parametrize("case", ["plain", "cold_rows", "ragged_tail"])
def test_rnvp_gradient_pass_with_the_forward_pass_y(amd, O, case, monkeypatch):
    """Large batches with the in-kernel mask: the register-resident forward kernel keeps y = net(mask * z) and launch A
    of the gradient pass skips its own first sweep over z (mnf_rnvp_seeded_train -> mnf_rnvp_bwd_mfma_phases(y)).  Same
    gradients as without (to rounding: the same y either way), and the float64 oracle on a slice.  cold_rows: rows
    whose operands leave the f16 range make the forward pass recompute their 64-row groups in fp32 and leave NaN rows in
    y, which must send those groups to the gradient pass's own fp32 fix-up; ragged_tail: the last, short group."""
    import torch_mnf_amd.flows as fl

    dim, hid = 800, 50
    rows = 8192 + (37 if case == "ragged_tail" else 0)
    sd = recipes.rnvp_params(8100, dim, hid)
    z = recipes.gaussian(8101, rows, dim)
    if case == "cold_rows":
        z[100] *= 3.0e4
        z[5000, 17] = 2.0e5
    w_x = recipes.gaussian(8102, rows, dim)
    w_l = recipes.gaussian(8103, rows, 1)[:, 0]
    seed = 99

    def run(keep):
        monkeypatch.setattr(fl._dispatch, "RNVP_KEEP_Y_MIN_ROWS", 4096 if keep else 1 << 60)
        f = amd.RNVP(dim, h_sizes=(hid,))
        f.load_state_dict(sd)
        f.to(DEV)
        zz = z.to(DEV).requires_grad_(True)
        x, ld = f.forward(zz, seed=seed)
        kept = x.grad_fn.kept_y
        ((x * w_x.to(DEV)).sum() + (ld * w_l.to(DEV)).sum()).backward()
        return kept, x.detach(), {"z": zz.grad, **{n: p.grad for n, p in f.named_parameters()}}

    kept, x1, g1 = run(True)
    none, x0, g0 = run(False)
    assert kept is not None and none is None
    assert torch.equal(x1, x0)
    if case == "cold_rows":
        assert torch.isnan(kept[100]).all() and torch.isnan(kept[5000]).all() and not torch.isnan(kept[300]).any()
    elif case == "ragged_tail":  # the short last group takes the fp32 body in the forward kernel too
        assert not torch.isnan(kept[:8192]).any() and torch.isnan(kept[8192:]).all()
    else:
        assert not torch.isnan(kept).any()
    for k in g0:
        assert_close(g1[k], g0[k], 2e-6, f"{case}: gradient {k} with the kept y vs recomputed")
    # the float64 oracle on a slice of ordinary rows (a row's grad_z does not depend on the other rows)
    sl = slice(1024, 1536)
    probe = amd.RNVP(dim, h_sizes=(hid,))
    mask = probe.mask_for(seed, rows).cpu()
    g32, g64 = _rnvp_oracle_grads(O, sd, z[sl], mask[sl], w_x[sl], w_l[sl])
    widen = 2.0 * normwise_err(g32["z"].numpy(), g64["z"].numpy())
    err = normwise_err(g1["z"][sl].cpu().double().numpy(), g64["z"].numpy())
    assert err <= 1e-5 + widen, (case, err, widen)
