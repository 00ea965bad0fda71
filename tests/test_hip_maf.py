"""GPU tests of MAF / IAF on the library (``mnf_maf`` / ``mnf_maf_bwd`` behind ``flows.MAF`` / ``flows.IAF``): both
directions and parities against the reference's own runs (fixture G15), gradients against autograd through the float64
oracle, and the reference's three training contracts for these layers (tests/test_flows.py:58-73)."""
import numpy as np
import pytest
import torch

import recipes
from helpers import RTOL, assert_close, assert_parity, normwise_err
from test_hip_autograd import moons, train
from test_oracle_golden import G15_CASES, g15_params

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


def build(amd, cls, tag, parity):
    dim, h_sizes, _ = G15_CASES[tag]
    layer = cls(dim, parity=parity, h_sizes=h_sizes)
    missing = layer.load_state_dict(g15_params(tag, parity), strict=False)
    assert all(k.endswith(".mask") for k in missing.missing_keys) and not missing.unexpected_keys
    return layer.to(DEV)


@pytest.mark.parametrize("parity", [False, True])
@pytest.mark.parametrize("tag", sorted(G15_CASES))
def test_g15_maf_iaf_vs_reference(amd, O, golden, tag, parity):
    """Fixture G15: the reference's MAF.forward (sequential) / MAF.inverse (one pass), flows/maf.py:39-62; IAF is the
    same layer with the directions swapped (:65-72).  The masks are the reference's (layers/made.py:58-94)."""
    fx = golden("g15_maf_iaf")
    x = torch.from_numpy(fx[f"{tag}.x"]).to(DEV)
    key = f"{tag}.p{int(parity)}"
    maf, iaf = build(amd, amd.MAF, tag, parity), build(amd, amd.IAF, tag, parity)
    for i, m in enumerate(maf._masked()):
        assert np.array_equal(m.mask.cpu().numpy().astype(np.uint8), fx[f"{tag}.mask{i}"])
    with torch.no_grad():
        y_f, ld_f = maf.forward(x)
        y_i, ld_i = maf.inverse(x)
        yi_f, ldi_f = iaf.forward(x)
        yi_i, ldi_i = iaf.inverse(x)
    assert_parity(y_f, fx[f"{key}.fwd"], fx[f"{key}.fwd64"], what=f"MAF.forward {key}")
    # (the reference's float64 forward allocates a float32 log-det: the float64 head-room comes from the oracle in
    # float64 on the same inputs; round 3: a flat 2e-5)
    sd64 = {k: v.detach().cpu().double() for k, v in maf.state_dict().items() if not k.endswith("mask")}
    masks64 = [m.mask.detach().cpu() for m in maf._masked()]
    _, ld_f64 = O.maf(x.cpu().double(), sd64, masks64, parity, False)
    assert_parity(ld_f, fx[f"{key}.ld_fwd"], ld_f64.numpy(), what=f"MAF.forward log_det {key}")
    assert_parity(y_i, fx[f"{key}.inv"], fx[f"{key}.inv64"], what=f"MAF.inverse {key}")
    assert_parity(ld_i, fx[f"{key}.ld_inv"], fx[f"{key}.ld_inv64"], what=f"MAF.inverse log_det {key}")
    assert torch.equal(yi_f, y_i) and torch.equal(ldi_f, ld_i) and torch.equal(yi_i, y_f) and torch.equal(ldi_i, ld_f)
    # log_det accumulation, as NormalizingFlow's loop uses it
    acc = torch.full((x.shape[0],), 0.25, device=DEV)
    with torch.no_grad():
        maf._run(x, True, acc)
    assert_close(acc, fx[f"{key}.ld_inv"] + 0.25, RTOL, "accumulated log_det")


def oracle_grads(O, tag, parity, inverse, x, w_y, w_l, dtype):
    dim, h_sizes, _ = G15_CASES[tag]
    p = {k: v.clone().requires_grad_(True) for k, v in g15_params(tag, parity, dtype).items()}
    xx = x.to(dtype).clone().requires_grad_(True)
    y, ld = O.maf(xx, p, O.made_masks(dim, h_sizes, 2 * dim), parity, inverse)
    ((y * w_y.to(dtype)).sum() + (ld * w_l.to(dtype)).sum()).backward()
    return {"x": xx.grad, **{k: v.grad for k, v in p.items()}}


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("parity", [False, True])
@pytest.mark.parametrize("tag", sorted(G15_CASES))
def test_maf_gradients_vs_float64_oracle(amd, O, golden, tag, parity, inverse):
    """sum(y w_y) + sum(log_det w_l) differentiated by ``mnf_maf_bwd`` (one pass: one back-propagation; sequential: one
    net evaluation at the output and dim back-propagations of a one-hot cotangent pair) within 1e-5 + 2 dist(fp32 oracle,
    fp64 oracle) of autograd through the float64 oracle."""
    fx = golden("g15_maf_iaf")
    dim, _, rows = G15_CASES[tag]
    x = torch.from_numpy(fx[f"{tag}.x"])
    w_y, w_l = recipes.gaussian(1600 + dim, rows, dim), recipes.gaussian(1601 + dim, rows, 1)[:, 0]
    g64 = oracle_grads(O, tag, parity, inverse, x, w_y, w_l, torch.float64)
    g32 = oracle_grads(O, tag, parity, inverse, x, w_y, w_l, torch.float32)
    layer = build(amd, amd.MAF, tag, parity)
    xx = x.to(DEV).requires_grad_(True)
    y, ld = layer.inverse(xx) if inverse else layer.forward(xx)
    ((y * w_y.to(DEV)).sum() + (ld * w_l.to(DEV)).sum()).backward()
    got = {"x": xx.grad, **{n: p.grad for n, p in layer.named_parameters()}}
    assert set(got) == set(g64)
    worst = 0.0
    for k, r64 in g64.items():
        if float(r64.abs().max()) == 0.0:
            assert float(got[k].abs().max()) == 0.0, k
            continue
        widen = 2 * normwise_err(g32[k].double().numpy(), r64.numpy())
        err = normwise_err(got[k].detach().cpu().double().numpy(), r64.numpy())
        worst = max(worst, err - widen)
        assert err <= 1e-5 + widen, f"MAF {tag} parity={parity} inverse={inverse} grad {k}: {err:.2e} > 1e-5 + {widen:.2e}"
    # masked-out weights receive exactly no gradient
    for m in layer._masked():
        assert float((m.weight.grad * (m.mask.T == 0)).abs().max()) == 0.0
    print(f"MAF {tag} parity={parity} inverse={inverse}: worst gradient error beyond the oracle's own fp32 distance {worst:.2e}")


def test_maf_round_trip_and_model_loop(amd):
    """inverse(forward(x)) = x through a stack of alternating parities, and NormalizingFlowModel.log_prob / sample run
    the layers (log_det accumulated across them)."""
    torch.manual_seed(3)
    flows = [amd.MAF(dim=6, parity=i % 2 == 0, h_sizes=(16, 16)) for i in range(3)]
    model = amd.NormalizingFlowModel(amd.StandardNormal(6), flows).to(DEV)
    x = recipes.gaussian(1700, 200, 6).to(DEV)
    with torch.no_grad():
        zs, ld_inv = model.inverse(x)
        xs, ld_fwd = model.forward(zs[-1])
        lp = model.log_prob(x)
    assert_close(xs[-1], x, 2e-5, "forward(inverse(x))")
    assert_close(ld_fwd, -ld_inv, 2e-5, "log_det of the round trip")
    assert torch.isfinite(lp).all()


@pytest.mark.parametrize("name,bound", [("maf", 250), ("maf_actnorm", 226), ("iaf", 300)])
def test_reference_training_contracts_maf_iaf(amd, name, bound):
    """The reference's e2e tests for these layers (tests/test_flows.py:58-73): Adam, 1 step then 70 steps on 128
    half-moon points; the loss must fall and end below the reference's bound."""
    torch.manual_seed(0)
    samples = moons(128).to(DEV)
    cls = amd.IAF if name == "iaf" else amd.MAF
    flows = [cls(dim=2, parity=i % 2 == 0) for i in range(2)]
    if name == "maf_actnorm":
        for idx in reversed(range(len(flows))):
            flows.insert(idx, amd.ActNormFlow(dim=2))
    base = torch.distributions.MultivariateNormal(torch.zeros(2, device=DEV), torch.eye(2, device=DEV))
    model = amd.NormalizingFlowModel(base, flows).to(DEV)
    adam = torch.optim.Adam(model.parameters())
    loss1 = train(model, adam, samples, 1)
    loss2 = train(model, adam, samples, 70)
    assert loss1 > loss2
    assert loss2 < bound, f"{loss2=:.4} > {bound=}"


def test_maf_in_a_flat_parameter_buffer_and_errors(amd):
    torch.manual_seed(5)
    plain, homed = amd.MAF(4, True, h_sizes=(8,)).to(DEV), amd.MAF(4, True, h_sizes=(8,)).to(DEV)
    homed.load_state_dict(plain.state_dict())
    flat = amd.FlatParameters(homed)
    x = recipes.gaussian(1800, 70, 4).to(DEV)
    for layer in (plain, homed):
        for direction in (layer.forward, layer.inverse):
            y, ld = direction(x)
            (y.pow(2).sum() + ld.sum()).backward()
    for (n0, p0), (n1, p1) in zip(plain.named_parameters(), homed.named_parameters()):
        assert normwise_err(p1.grad.cpu().numpy(), p0.grad.cpu().numpy()) <= 2e-6, n0
    assert all(p.grad is v for p, v in zip(flat.params, flat._grad_views))
    with pytest.raises(NotImplementedError, match="MADE"):
        amd.MAF(4, False, net=torch.nn.Linear(4, 8))
    with pytest.raises(ValueError, match="expected dim"):
        plain.forward(torch.zeros(3, 5, device=DEV))
    y, ld = plain.forward(torch.zeros(0, 4, device=DEV))
    assert y.shape == (0, 4) and ld.shape == (0,)


@pytest.mark.parametrize("dim,h_sizes,rows", [(64, (32, 32), 70), (40, (128,), 33), (3, (5,), 257), (96, (24, 24, 24), 65),
                                              (2, (200, 100), 40)])
@pytest.mark.parametrize("parity", [False, True])
def test_maf_wide_and_odd_shapes(amd, O, dim, h_sizes, rows, parity):
    """Nets wider than the reference's defaults: fewer rows per workgroup than lanes (the activation slots of 64 rows do
    not fit LDS), hidden layers wider than the input, a ragged last workgroup.  Both directions and their gradients
    against the oracle (float64 for the gradients)."""
    sd = recipes.maf_params(1900 + dim, dim, h_sizes, gain=1.2, last_gain=0.5)
    masks = O.made_masks(dim, h_sizes, 2 * dim)
    layer = amd.MAF(dim, parity=parity, h_sizes=h_sizes)
    layer.load_state_dict(sd, strict=False)
    layer.to(DEV)
    x = recipes.gaussian(1901 + dim, rows, dim)
    w_y, w_l = recipes.gaussian(1902 + dim, rows, dim), recipes.gaussian(1903 + dim, rows, 1)[:, 0]
    for inverse in (True, False):
        xx = x.to(DEV).requires_grad_(True)
        y, ld = layer.inverse(xx) if inverse else layer.forward(xx)
        y_ref, ld_ref = O.maf(x, sd, masks, parity, inverse)
        y64, ld64 = O.maf(x.double(), {k: v.double() for k, v in sd.items()}, masks, parity, inverse)
        assert_parity(y, y_ref.numpy(), y64.numpy(), what=f"MAF d={dim} h={h_sizes} inverse={inverse}")
        assert_parity(ld, ld_ref.numpy(), ld64.numpy(), what=f"MAF d={dim} h={h_sizes} inverse={inverse} log_det")
        layer.zero_grad()
        ((y * w_y.to(DEV)).sum() + (ld * w_l.to(DEV)).sum()).backward()
        got = {"x": xx.grad, **{n: p.grad.clone() for n, p in layer.named_parameters()}}
        ref = {}
        for dt in (torch.float32, torch.float64):
            p = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
            xr = x.to(dt).clone().requires_grad_(True)
            yy, ll = O.maf(xr, p, masks, parity, inverse)
            ((yy * w_y.to(dt)).sum() + (ll * w_l.to(dt)).sum()).backward()
            ref[dt] = {"x": xr.grad, **{k: v.grad for k, v in p.items()}}
        for k, r64 in ref[torch.float64].items():
            if float(r64.abs().max()) == 0.0:
                continue
            widen = 2 * normwise_err(ref[torch.float32][k].double().numpy(), r64.numpy())
            err = normwise_err(got[k].detach().cpu().double().numpy(), r64.numpy())
            assert err <= 1e-5 + 2 * widen, f"MAF d={dim} h={h_sizes} inverse={inverse} grad {k}: {err:.2e} vs {widen:.2e}"


def test_maf_with_a_permuted_made_values_match_and_sequential_gradients_are_refused(amd, O):
    """A MADE with natural_ordering=False (made.py's default): both directions' VALUES are the reference's; the one-pass
    direction trains; the element-by-element direction's gradient pass (which reuses one network evaluation for every
    step, valid only for masks that are autoregressive in index order) raises instead of returning wrong gradients
    (ADVICE round 3)."""
    from torch_mnf_amd.flows import MADE

    dim = 6
    torch.manual_seed(4)
    net = MADE(dim, (16, 16), 2 * dim, natural_ordering=False)
    flow = amd.MAF(dim, True, net=net).to(DEV)
    assert not flow._autoregressive_in_index_order() and amd.MAF(dim, True)._autoregressive_in_index_order()
    x = (0.5 * recipes.gaussian(31, 200, dim)).to(DEV)
    with torch.no_grad():
        z, ld_inv = flow.inverse(x)          # one pass
        xr, ld_fwd = flow.forward(z)         # element by element: values fine
    # the oracle on the same masked network (CPU)
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    masks = [m.mask.detach().cpu() for m in net if hasattr(m, "mask")]  # (in, out), as the oracle's made() takes them
    z_ref, ld_ref = O.maf(x.cpu(), sd, masks, True, inverse=True)
    assert_close(z, z_ref, 1e-5, "z (permuted MADE)")
    assert_close(ld_inv, ld_ref, 2e-5, "log_det (permuted MADE)")
    xr_ref, _ = O.maf(z_ref, sd, masks, True, inverse=False)
    assert_close(xr, xr_ref, 1e-4, "element-by-element values (permuted MADE)")
    zg = x.clone().requires_grad_(True)
    out, ld = flow.inverse(zg)               # one pass: gradients exist
    (out.sum() + ld.sum()).backward()
    assert zg.grad is not None and all(p.grad is not None for p in flow.parameters())
    with pytest.raises(NotImplementedError, match="natural_ordering"):
        flow.forward(x.clone().requires_grad_(True))
