"""GPU tests of the few-rows RNVP kernels (mnf_rnvp_few.hip: one or two rows, one hidden layer, one workgroup, no atomics) --
the path one row takes through flow_q / flow_r in the MNF layers' kl_div: forward against the oracle, gradients
against autograd through the float64 oracle, bit-for-bit repeatability, gradients ADDED to what grad_flat holds."""
import pytest
import torch

import recipes
from helpers import RTOL, assert_close
from test_hip_round3 import _check_grads, _rnvp_gpu_grads, _rnvp_oracle_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"
# (gradients at 3 rows: past the one-workgroup kernel -- the streaming kernels, same checks; hidden 70: wider than a
#  wave, ditto.  Both kernels also run as a grid, a workgroup per two rows: FWD_SHAPES / GRID_SHAPES add batch-sized cases.)
SHAPES = [(800, 50, 1), (800, 50, 2), (800, 50, 3), (50, 50, 1), (20, 50, 1), (20, 50, 2), (784, 30, 2), (33, 7, 1),
          (1100, 64, 2), (2, 16, 1), (96, 70, 1)]


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


FWD_SHAPES = SHAPES + [(800, 50, 128), (800, 50, 129), (50, 50, 300), (20, 50, 63), (784, 30, 512)]


@pytest.mark.parametrize("dim,hid,rows", FWD_SHAPES)
@pytest.mark.parametrize("masked", ["explicit", "seeded"])
def test_few_rows_forward_vs_oracle(amd, O, dim, hid, rows, masked):
    sd = recipes.rnvp_params(5100 + dim + hid, dim, hid)
    z = recipes.gaussian(5200 + dim, rows, dim)
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    f.to(DEV)
    if masked == "explicit":
        mask, kw = recipes.bernoulli_mask(5300 + dim, rows, dim), {}
        kw = {"mask": mask.to(DEV)}
    else:
        mask, kw = f.mask_for(77 + dim, rows).cpu(), {"seed": 77 + dim}
    x_ref, ld_ref = O.rnvp(z, sd, mask)
    with torch.no_grad():
        x, ld = f.forward(z.to(DEV), **kw)
        x2, ld2 = f.forward(z.to(DEV), **kw)
    assert_close(x, x_ref, RTOL, f"few-rows RNVP x d={dim} h={hid} rows={rows}")
    assert_close(ld, ld_ref, RTOL, f"few-rows RNVP log_det d={dim} h={hid} rows={rows}")
    assert torch.equal(x, x2) and torch.equal(ld, ld2)  # fixed-order sums
    # the streaming kernels agree (force_generic takes the generic kernel at every row count)
    f.force_generic = True
    with torch.no_grad():
        xg, ldg = f.forward(z.to(DEV), **kw)
    assert_close(x, xg, 2e-6, "few-rows vs generic x")
    assert_close(ld, ldg, 2e-6, "few-rows vs generic log_det")
    # log_det accumulation (NormalizingFlow's running sum)
    f.force_generic = False
    acc = torch.full((rows,), 0.5, device=DEV)
    with torch.no_grad():
        f._run(z.to(DEV), False, acc, **kw)
    assert_close(acc, ld_ref + 0.5, RTOL, "few-rows RNVP accumulated log_det")


@pytest.mark.parametrize("dim,hid,rows", SHAPES)
def test_few_rows_gradients_vs_float64_oracle(amd, O, dim, hid, rows):
    sd = recipes.rnvp_params(5400 + dim + hid, dim, hid)
    z = recipes.gaussian(5500 + dim, rows, dim)
    w_x = recipes.gaussian(5600 + dim, rows, dim)
    w_l = recipes.gaussian(5700 + dim, rows, 1)[:, 0]
    mask = recipes.bernoulli_mask(5800 + dim, rows, dim)
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    f, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask, None)
    worst = _check_grads(got, g32, g64, f"few-rows rnvp d={dim} h={hid} rows={rows}")
    f2, again = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask, None)
    for k in got:
        if rows <= 2 and hid <= 64:
            assert torch.equal(got[k], again[k]), f"{k}: the few-rows gradient kernel has no atomics, runs must repeat"
    print(f"few-rows rnvp gradients d={dim} h={hid} rows={rows}: worst {worst:.2e} from float64")


GRID_SHAPES = [(800, 50, 3), (800, 50, 128), (800, 50, 129), (50, 50, 128), (20, 50, 300), (784, 30, 512), (33, 7, 65)]


@pytest.mark.parametrize("dim,hid,rows", GRID_SHAPES)
@pytest.mark.parametrize("masked", ["explicit", "seeded"])
def test_few_rows_gradient_grid_vs_float64_oracle(amd, O, dim, hid, rows, masked):
    """Batch-sized gradient calls (3 .. 512 rows): ``mnf_rnvp_bwd_few`` -- a workgroup per two rows writing its own copy
    of the parameter gradients, one reduction launch.  Against autograd through the float64 oracle, against the
    streaming gradient kernels, bit-for-bit repeatable (no atomics), and ADDING to what the gradient buffer holds."""
    import torch_mnf_amd.flows as fl

    sd = recipes.rnvp_params(6100 + dim + hid, dim, hid)
    z = recipes.gaussian(6200 + dim, rows, dim)
    w_x = recipes.gaussian(6300 + dim, rows, dim)
    w_l = recipes.gaussian(6400 + dim, rows, 1)[:, 0]
    probe = amd.RNVP(dim, h_sizes=(hid,))
    seed = 31 + dim
    mask = recipes.bernoulli_mask(6500 + dim, rows, dim) if masked == "explicit" else probe.mask_for(seed, rows).cpu()
    kw = (mask, None) if masked == "explicit" else (None, seed)
    assert amd._lib.load().mnf_rnvp_bwd_few_workspace_floats(rows, dim, 1, probe._hid) > 0
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    f, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, *kw)
    worst = _check_grads(got, g32, g64, f"few-rows grid rnvp d={dim} h={hid} rows={rows} {masked}")
    _, again = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, *kw)
    for k in got:
        assert torch.equal(got[k], again[k]), k
    fl._dispatch.RNVP_BWD_FEW_GRID_OFF = True
    try:
        _, streaming = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, *kw)
    finally:
        fl._dispatch.RNVP_BWD_FEW_GRID_OFF = False
    for k in got:
        assert_close(got[k], streaming[k], 2e-5, f"grid vs streaming gradient kernels: {k}")
    # a second backward pass adds to the first (FlatParameters home: in place)
    homed = amd.RNVP(dim, h_sizes=(hid,))
    homed.load_state_dict(sd)
    homed.to(DEV)
    flat = amd.FlatParameters(homed)
    for n in (1, 2):
        x, ld = homed.forward(z.to(DEV), mask=None if kw[0] is None else kw[0].to(DEV), seed=kw[1])
        ((x * w_x.to(DEV)).sum() + (ld * w_l.to(DEV)).sum()).backward()
        if n == 1:
            once = flat.grad.clone()
    assert_close(flat.grad, 2 * once, 1e-6, "two passes into the flat gradient buffer")
    for name, p in homed.named_parameters():
        assert_close(p.grad / 2, got[name], 1e-6, f"flat-home gradient {name}")
    print(f"few-rows grid gradients d={dim} h={hid} rows={rows} {masked}: worst {worst:.2e} from float64")


@pytest.mark.parametrize("which", ["x_only", "ld_only"])
def test_few_rows_gradients_absent_cotangents(amd, O, which):
    dim, hid, rows = 800, 50, 1
    sd = recipes.rnvp_params(5900, dim, hid)
    z = recipes.gaussian(5901, rows, dim)
    w_x = recipes.gaussian(5902, rows, dim) if which == "x_only" else None
    w_l = recipes.gaussian(5903, rows, 1)[:, 0] if which == "ld_only" else None
    mask = recipes.bernoulli_mask(5904, rows, dim)
    g32, g64 = _rnvp_oracle_grads(O, sd, z, mask, w_x, w_l)
    _, got = _rnvp_gpu_grads(amd, sd, dim, hid, z, w_x, w_l, mask, None)
    _check_grads(got, g32, g64, f"few-rows rnvp {which}")


def test_few_rows_gradients_are_added_to_a_flat_gradient_buffer(amd, O):
    """Two backward passes through a layer living in a FlatParameters buffer: the second adds to the first."""
    dim, hid = 800, 50
    sd = recipes.rnvp_params(6000, dim, hid)
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    f.to(DEV)
    flat = amd.FlatParameters(f)
    z = recipes.gaussian(6001, 1, dim).to(DEV)
    mask = recipes.bernoulli_mask(6002, 1, dim).to(DEV)
    for n in (1, 2):
        x, ld = f.forward(z, mask=mask)
        (x.sum() + ld.sum()).backward()
        if n == 1:
            once = flat.grad.clone()
    assert torch.equal(flat.grad, 2 * once)
    g32, g64 = _rnvp_oracle_grads(O, sd, z.cpu(), mask.cpu(), torch.ones(1, dim), torch.ones(1))
    got = {n: p.grad / 2 for n, p in f.named_parameters()}
    got["z"] = None
    g64.pop("z"), g32.pop("z")
    _check_grads(got, g32, g64, "few-rows rnvp into a flat gradient buffer")
