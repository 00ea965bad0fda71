"""Round-4 GPU tests: the seeded sample_z prologue (noise generated inside the forward and the gradient launch), MNFLinear
wider than 64 outputs, and what else round 4 added behind the C ABI."""
import ctypes

import pytest
import torch

import recipes
from helpers import RTOL, assert_close, normwise_err

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


def _err(a, b) -> float:
    return normwise_err(a.detach().cpu().numpy(), b.detach().cpu().numpy())


def _noise(amd, seed, rows, cols):
    eps = torch.empty(rows, cols, device=DEV)
    amd._lib.check("mnf_sample_z0_noise", amd._lib.load().mnf_sample_z0_noise(
        ctypes.c_uint64(seed), eps.data_ptr(), rows, cols, None))
    torch.cuda.synchronize()
    return eps


# ------------------------------------------------------------------------------------------------ seeded prologue
@pytest.mark.parametrize("rows,dim", [(1, 7), (333, 50), (4096, 800), (1000, 27)])
def test_seeded_sample_z0_is_the_explicit_prologue_on_the_materialised_noise(amd, rows, dim):
    """mnf_sample_z0_seeded / _seeded_bwd == mnf_sample_z0 / _bwd fed the numbers mnf_sample_z0_noise(seed) writes
    (mnf_linear.py:59-62 with `epsilon` never stored): values bit for bit, the two parameter gradients to summation
    order."""
    lib = amd._lib.load()
    seed = 0x1234_5678_9ABC_DEF0 + rows
    mean = (0.1 * recipes.gaussian(61, 1, dim)[0]).to(DEV).contiguous()
    log_var = (-9 + 0.1 * recipes.gaussian(62, 1, dim)[0]).to(DEV).contiguous()
    eps = _noise(amd, seed, rows, dim)
    assert abs(float(eps.mean())) < 4.0 / (rows * dim) ** 0.5 + 1e-3 and (rows * dim < 1000 or abs(float(eps.std()) - 1) < 0.1)
    z_ref, z_got = torch.empty(rows, dim, device=DEV), torch.empty(rows, dim, device=DEV)
    amd._lib.check("z0", lib.mnf_sample_z0(mean.data_ptr(), log_var.data_ptr(), eps.data_ptr(), z_ref.data_ptr(), rows, dim, None))
    amd._lib.check("z0s", lib.mnf_sample_z0_seeded(mean.data_ptr(), log_var.data_ptr(), ctypes.c_uint64(seed),
                                                    z_got.data_ptr(), rows, dim, None))
    torch.cuda.synchronize()
    assert torch.equal(z_ref, z_got)
    g = (recipes.gaussian(63, rows, dim) / rows).to(DEV).contiguous()
    out_ref, out_got = torch.zeros(2 * dim, device=DEV), torch.zeros(2 * dim, device=DEV)
    amd._lib.check("bwd", lib.mnf_sample_z0_bwd(g.data_ptr(), eps.data_ptr(), log_var.data_ptr(), out_ref.data_ptr(),
                                                 out_ref.data_ptr() + 4 * dim, rows, dim, None))
    amd._lib.check("bwds", lib.mnf_sample_z0_seeded_bwd(g.data_ptr(), ctypes.c_uint64(seed), log_var.data_ptr(),
                                                         out_got.data_ptr(), out_got.data_ptr() + 4 * dim, rows, dim, None))
    torch.cuda.synchronize()
    # against float64 autograd of the reference's expression
    m64 = mean.double().cpu().requires_grad_(True)
    v64 = log_var.double().cpu().requires_grad_(True)
    z64 = m64 + v64.exp().sqrt() * eps.double().cpu()
    (z64 * g.double().cpu()).sum().backward()
    ref = torch.cat([m64.grad, v64.grad])
    assert _err(out_got[:dim], ref[:dim]) < 2e-6 and _err(out_got[dim:], ref[dim:]) < 2e-6
    assert _err(out_ref[:dim], ref[:dim]) < 2e-6 and _err(out_ref[dim:], ref[dim:]) < 2e-6


def test_mnf_linear_training_draws_no_noise_tensor_for_sample_z(amd):
    """MNFLinear.sample_z under autograd with nothing injected: z0 comes from the seeded prologue (no (batch, n_in)
    randn launch), and the q0 gradients are those of the explicit prologue on the same noise."""
    n_in, n_out, rows = 96, 10, 640
    layer = amd.MNFLinear(n_in, n_out).to(DEV)
    torch.manual_seed(123)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        z, ld = layer.sample_z(rows)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("sample_z0_seeded" in n for n in names), names
    assert not any("normal" in n.lower() for n in names), names
    w = (recipes.gaussian(64, rows, n_in) / rows).to(DEV)
    ((z * w).sum() + ld.mean()).backward()
    got = {k: getattr(layer, k).grad.clone() for k in ("q0_mean", "q0_log_var")}
    # replay: same seed (torch's CPU generator), noise materialised, masks reproduced through the same in-kernel seeds
    torch.manual_seed(123)
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
    eps = _noise(amd, seed, rows, n_in)
    layer.zero_grad()
    z2, ld2 = layer.sample_z(rows, eps=eps)
    assert torch.equal(z, z2) and torch.equal(ld, ld2)
    ((z2 * w).sum() + ld2.mean()).backward()
    for k, v in got.items():
        assert _err(v, getattr(layer, k).grad) < 1e-6, k
