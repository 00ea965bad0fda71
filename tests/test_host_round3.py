"""Host-side logic added in round 3 (no GPU): FlatParameters keeps its gradient views attached through the
reference's ``model.zero_grad()`` loop and a torch optimizer's ``zero_grad(set_to_none=True)``; the stale-parameter
detector's comparison."""
import torch
from torch import nn

import torch_mnf_amd as amd
from torch_mnf_amd import flows


def _flat_model():
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(4, 8), nn.Tanh(), nn.Linear(8, 2))
    return model, amd.FlatParameters(model)


def test_model_zero_grad_keeps_the_gradient_views_attached():
    """The reference's training loop (tests/test_flows.py:27, examples/half_moons.ipynb:188) calls
    ``model.zero_grad()``, whose default sets p.grad to None: the next backward would give every parameter a fresh
    .grad outside FlatParameters.grad and a fused optimizer would see zeros.  On a flat-homed model
    zero_grad is the buffer's memset and the views stay."""
    model, flat = _flat_model()
    x = torch.randn(16, 4)
    for step in range(3):
        model.zero_grad()
        assert float(flat.grad.abs().sum()) == 0.0
        model(x).pow(2).sum().backward()
        for i, p in enumerate(flat.params):
            assert p.grad is flat._grad_views[i]
            assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offset[id(p)]
        assert float(flat.grad.abs().sum()) > 0.0
        expect = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        assert torch.equal(expect, flat.grad)


def test_stray_gradients_are_folded_back_before_a_step():
    """``opt.zero_grad(set_to_none=True)`` of a torch optimizer (or ``p.grad = None``) detaches the views; autograd
    then allocates stray gradient tensors.  ``_reattach(fold=True)`` -- what FusedAdam.step runs first -- adds them
    into the buffer and re-attaches; ``zero_grad`` re-attaches and clears."""
    model, flat = _flat_model()
    x = torch.randn(16, 4)
    model(x).pow(2).sum().backward()
    want = flat.grad.clone()
    flat.zero_grad()
    for p in model.parameters():
        p.grad = None                      # what set_to_none does
    model(x).pow(2).sum().backward()       # stray .grad tensors
    assert float(flat.grad.abs().sum()) == 0.0
    assert all(p.grad is not flat._grad_views[i] for i, p in enumerate(flat.params))
    flat._reattach(fold=True)
    assert torch.allclose(flat.grad, want)
    assert all(p.grad is flat._grad_views[i] for i, p in enumerate(flat.params))
    for p in model.parameters():
        p.grad = None
    flat.zero_grad()
    assert all(p.grad is flat._grad_views[i] for i, p in enumerate(flat.params))


def test_flat_home_validation():
    """home_is_valid: parameters moved out of the buffer (``.double().float()`` re-homes them) or frozen afterwards
    are no longer a valid home for in-place gradient sums."""
    model, flat = _flat_model()
    params = list(model.parameters())
    assert flat.home_is_valid(params)
    params[1].requires_grad_(False)
    assert not flat.home_is_valid(params)
    params[1].requires_grad_(True)
    assert flat.home_is_valid(params)
    model.double().float()
    assert not flat.home_is_valid(list(model.parameters()))


def test_stale_parameter_detector(monkeypatch):
    """_check_params_fresh: equal parameters pass, a write through p.data (no version bump) raises, every N-th call."""
    p = nn.Parameter(torch.arange(6.0).reshape(2, 3))
    packed = p.detach().reshape(-1).clone()
    monkeypatch.setattr(flows, "_CHECK_PARAMS_EVERY", 1)
    flows._check_params_fresh([p], packed, "test")            # fresh: fine
    version = p._version
    p.data.mul_(2)
    assert p._version == version                               # the write the caches cannot see
    try:
        flows._check_params_fresh([p], packed, "test")
    except RuntimeError as err:
        assert "invalidate()" in str(err)
    else:
        raise AssertionError("a stale image went undetected")
    monkeypatch.setattr(flows, "_CHECK_PARAMS_EVERY", 0)
    flows._check_params_fresh([p], packed, "test")            # switched off: no check


def test_flat_homed_model_pickles_and_a_copy_clears_its_own_gradients():
    """FlatParameters used to install ``model.zero_grad`` as a local lambda closing over itself --
    unpicklable, and a ``copy.deepcopy`` of the model kept the ORIGINAL's function object (``copy.zero_grad()`` cleared
    the original's gradient buffer).  Now a bound module-level function that finds the flat buffer through the module."""
    import copy
    import pickle

    import torch_mnf_amd as amd

    model = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    flat = amd.FlatParameters(model)
    flat.grad.fill_(1.0)
    clone = copy.deepcopy(model)
    clone.zero_grad()
    assert float(flat.grad.abs().sum()) == flat.grad.numel(), "the copy cleared the original's gradient buffer"
    assert float(clone.__dict__["_mnf_flat"].grad.abs().sum()) == 0.0
    model.zero_grad()
    assert float(flat.grad.abs().sum()) == 0.0
    restored = pickle.loads(pickle.dumps(model))
    restored.zero_grad()
    # a middle parameter that left the buffer invalidates the fused layers' in-place gradient path
    params = list(model.parameters())
    assert flat.home_is_valid(params)
    params[1].data = params[1].data.clone()
    assert not flat.home_is_valid(params)
