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


# ------------------------------------------------------------------------------------------------ MNFLinear wider than 64 outputs
def _g16_layer(amd):
    n_in, n_out = 784, 256
    layer = amd.MNFLinear(n_in, n_out)
    with torch.no_grad():
        layer.W_mean.copy_(0.1 * recipes.gaussian(1600, n_out, n_in))
        layer.W_log_var.copy_(-9 + 0.1 * recipes.gaussian(1601, n_out, n_in))
        layer.b_mean.copy_(0.3 * recipes.gaussian(1602, 1, n_out)[0])
        layer.b_log_var.copy_(-9 + 0.1 * recipes.gaussian(1603, 1, n_out)[0])
        layer.q0_mean.copy_(1 + 0.1 * recipes.gaussian(1604, 1, n_in)[0])
        layer.q0_log_var.copy_(-9 + 0.1 * recipes.gaussian(1605, 1, n_in)[0])
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(1610 + i, n_in, 50))
    return layer.to(DEV), n_in, n_out


def test_g16_wide_mnf_linear_forward_and_gradients_vs_reference(amd, golden):
    """Fixture G16: the reference's MNFLinear(784, 256).forward with its draws captured and its own autograd gradients
    of sum(y w) (mnf_linear.py:46-64; the first layer of an ordinary MNFFeedForward([784, 256, 10])).  Here the layer
    runs the <= 64-output kernels once per 64-output slab, forward and backward."""
    from helpers import unpack_mask

    fx = golden("g16_mnf_linear_wide")
    layer, n_in, n_out = _g16_layer(amd)
    rows = fx["y"].shape[0]
    x = recipes.gaussian(1620, rows, n_in, scale=1.5).abs().to(DEV).requires_grad_(True)
    w = (recipes.gaussian(1621, rows, n_out) / (rows * n_out)).to(DEV)
    masks = [unpack_mask(fx[f"mask{i}_bits"], n_in).to(DEV) for i in range(2)]
    eps_z, eps_out = torch.from_numpy(fx["eps_z"]).to(DEV), torch.from_numpy(fx["eps_out"]).to(DEV)
    real = layer.sample_z
    layer.sample_z = lambda n: real(n, eps=eps_z, masks=masks)
    try:
        y = layer.forward(x, eps=eps_out)
    finally:
        layer.sample_z = real
    assert_close(y, fx["y"], RTOL, "y vs the reference")
    (y * w).sum().backward()
    worst = {}
    got = {"x": x.grad, **{k: getattr(layer, k).grad for k in ("W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean",
                                                                 "q0_log_var")}}
    for i, f in enumerate(layer.flow_q.flows):
        for k, prm in f.named_parameters():
            got[f"flow_q.{i}.{k}"] = prm.grad
    for k, g in got.items():
        assert g is not None, k
        worst[k] = normwise_err(g.detach().cpu().numpy(), fx[f"grad.{k}"])
    # the reference's gradients are fp32 autograd: two fp32 evaluations of these sums differ by ~1e-6 normwise
    assert max(worst.values()) < 2e-5, sorted(worst.items(), key=lambda kv: -kv[1])[:4]


def test_mnf_feed_forward_784_256_10_runs_and_trains(amd):
    """models/mnf_feed_forward.py:27-31 at an ordinary layout: forward, kl_div and a few Adam steps on the library."""
    torch.manual_seed(5)
    model = amd.MNFFeedForward([784, 256, 10]).to(DEV)
    x = torch.rand(96, 784, device=DEV)
    yb = torch.randint(0, 10, (96,), device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x), yb) + 1e-4 * model.kl_div()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    with torch.no_grad():
        assert model(x).shape == (96, 10)
    # in-kernel noise of the wide layer: forward(x) with a seed == forward(x, eps=noise_for(seed))
    wide = model[0]
    real = wide.sample_z
    z = real(96)[0].detach()
    wide.sample_z = lambda n: (z, None)
    try:
        with torch.no_grad():
            torch.manual_seed(77)
            a = wide.forward(x)
            torch.manual_seed(77)
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
            b = wide.forward(x, eps=wide.noise_for(seed, 96))
    finally:
        wide.sample_z = real
    assert torch.equal(a, b)


@pytest.mark.parametrize("graphed", [False, True])
def test_wide_mnf_linear_sees_fused_optimizer_updates(amd, graphed):
    """A layer wider than 64 outputs runs per 64-output slab, and the slabs' packed operands are keyed on the PARENT's
    train.FlatParameters generation: FusedAdam (eager, or replayed from a hipGraph) writes the flat buffer through raw
    pointers and bumps nothing else.  After training steps the layer's forward must be the forward of a fresh layer
    holding the same parameters (round-4 advisor finding: the slabs kept serving the operands of the first pack)."""
    torch.manual_seed(11)
    model = amd.MNFFeedForward([784, 256, 10]).to(DEV)
    x = torch.rand(64, 784, device=DEV)
    yb = torch.randint(0, 10, (64,), device=DEV)
    wide = model[0]
    eps = torch.randn(64, 256, device=DEV)
    z = torch.ones(64, 784, device=DEV)

    def probe(layer):
        real = layer.sample_z
        layer.sample_z = lambda n: (z, None)
        try:
            with torch.no_grad():
                return layer.forward(x, eps=eps).clone()
        finally:
            layer.sample_z = real

    before = probe(wide)  # (packs every slab's operands from the initial parameters)
    opt = amd.FusedAdam(amd.FlatParameters(model), lr=5e-2, capturable=graphed)
    loss_fn = lambda xb: torch.nn.functional.cross_entropy(model(xb), yb) + 1e-4 * model.kl_div()  # noqa: E731
    if graphed:
        step = amd.GraphedStep(opt, loss_fn, x)
        for _ in range(3):
            step(x)
    else:
        for _ in range(3):
            opt.zero_grad()
            loss_fn(x).backward()
            opt.step()
    after = probe(wide)
    fresh = amd.MNFLinear(784, 256).to(DEV)
    fresh.load_state_dict(wide.state_dict())
    want = probe(fresh)
    assert float((after - before).abs().max()) > 1e-3, "the parameters did not move"
    assert_close(after, want, 1e-5, "wide layer after fused optimizer steps vs a fresh layer with the same parameters")


@pytest.mark.parametrize("rows,n_layers,x_scale,s_gain", [(1007, 9, 1.0, 2.0), (4096, 9, 1.0, 2.0), (1007, 2, 2500.0, 0.002)])
def test_log_prob_training_node_matches_the_unfused_route(amd, rows, n_layers, x_scale, s_gain):
    """`-model.log_prob(x).mean()` under a standard-normal base with the whole model one run of AffineHalfFlow layers:
    ONE autograd node from x to log p (the stack kernel's log-prob epilogue forward; the last layer's cotangents formed
    inside mnf_affine_half_bwd_split_lp backward) against the unfused route (log_det + base.log_prob(z) with its own
    autograd link, validated against the float64 oracle by tests/test_hip_autograd.py).  The third case scales x
    beyond the split range (two layers, |x| up to ~1e4 > 8192): most 16-row tiles go to the fp32 fix-up pass, which
    then reads the grad_y rows the split kernel left in its scratch buffer."""
    dim = 64
    flows_mod = amd.flows
    floor, flows_mod._dispatch.BWD_SPLIT_MIN_ROWS = flows_mod._dispatch.BWD_SPLIT_MIN_ROWS, 0
    try:
        results = {}
        for fused in (True, False):
            layers = []
            for i in range(n_layers):
                f = amd.AffineHalfFlow(dim, parity=bool(i % 2))
                f.load_state_dict(recipes.affine_half_params(1000 + i, dim, s_last_gain=s_gain))
                layers.append(f)
            model = amd.NormalizingFlowModel(amd.StandardNormal(dim), layers).to(DEV)
            x = (recipes.gaussian(77, rows, dim) * x_scale).to(DEV).requires_grad_(True)
            w = recipes.gaussian(78, rows, 1)[:, 0].to(DEV)  # a row-dependent cotangent, not just 1 / rows
            env, flows_mod._dispatch.NO_FUSED_LOGPROB = flows_mod._dispatch.NO_FUSED_LOGPROB, not fused
            try:
                lp = model.log_prob(x)
                (-(lp * w).mean()).backward()
            finally:
                flows_mod._dispatch.NO_FUSED_LOGPROB = env
            results[fused] = (lp.detach(), x.grad, {n: p.grad for n, p in model.named_parameters()})
        assert bool(torch.isfinite(results[False][0]).all()) and bool(torch.isfinite(results[False][1]).all())
        assert float((results[True][0] - results[False][0]).abs().max()) <= 2e-5 * float(results[False][0].abs().max())
        assert normwise_err(results[True][1].cpu().numpy(), results[False][1].cpu().numpy()) <= 2e-6
        for n, g in results[False][2].items():
            assert normwise_err(results[True][2][n].cpu().numpy(), g.cpu().numpy()) <= 5e-6, n
    finally:
        flows_mod._dispatch.BWD_SPLIT_MIN_ROWS = floor


# ------------------------------------------------------------------------ Glow.inverse + ActNormFlow.inverse, fused
def _pair_case(seed, rows, dim=32):
    """(u, M, s, t, cotangent) in float64 on the host; M a well-conditioned dense matrix."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(rows, dim, generator=g, dtype=torch.float64) * 1.7 + 0.3
    M = torch.eye(dim, dtype=torch.float64) + 0.2 * torch.randn(dim, dim, generator=g, dtype=torch.float64)
    s = 0.5 * torch.randn(1, dim, generator=g, dtype=torch.float64)
    t = torch.randn(1, dim, generator=g, dtype=torch.float64)
    gz = torch.randn(rows, dim, generator=g, dtype=torch.float64) / max(rows, 1)
    return u, M, s, t, gz


@pytest.mark.parametrize("rows,dim", [(1, 32), (15, 32), (16, 32), (17, 32), (63, 32), (1000, 32), (4099, 32), (70001, 32),
                                      (1, 16), (37, 16), (5000, 16), (1, 64), (37, 64), (5000, 64)])
def test_glow_actnorm_inverse_pair_kernels_vs_float64_oracle(amd, O, rows, dim):
    """mnf_glow_actnorm_inv / _bwd against the oracle's two layers composed in float64 (oracle.glow's product with M
    given, then oracle.affine_const inverse) and torch.autograd through them.  fp32 MFMA products, fp32 sums over the
    rows by atomics: 2e-6 normwise on the rows, 2e-5 on the sums over up to 70,001 rows."""
    lib = amd._lib.load()
    u, M, s, t, gz = _pair_case(100 + rows + dim, rows, dim)
    u64, M64, s64, t64 = (a.clone().requires_grad_(True) for a in (u, M, s, t))
    z64, _ = O.affine_const(u64 @ M64, s64, t64, inverse=True)
    z64.backward(gz)
    ud, Md, gzd = u.float().to(DEV), M.float().to(DEV).contiguous(), gz.float().to(DEV)
    # (s and t handed over 4-byte aligned only, as views into a flat parameter buffer are)
    st = torch.cat((torch.zeros(1), s.float().reshape(-1), t.float().reshape(-1))).to(DEV)
    sd, td = st[1:1 + dim], st[1 + dim:1 + 2 * dim]
    z = torch.empty_like(ud)
    ld_in, ld = torch.full((1,), 0.75, device=DEV), torch.empty(1, device=DEV)
    amd._lib.check("mnf_glow_actnorm_inv", lib.mnf_glow_actnorm_inv(
        ud.data_ptr(), Md.data_ptr(), sd.data_ptr(), td.data_ptr(), z.data_ptr(), ld_in.data_ptr(), ld.data_ptr(), rows,
        dim, None))
    gu = torch.full_like(ud, float("nan"))
    gM, gs, gt = torch.zeros(dim, dim, device=DEV), torch.zeros(dim, device=DEV), torch.zeros(dim, device=DEV)
    g_ld = torch.full((1,), -0.3, device=DEV)  # cotangent of the pair's log|det J| = 0.75 - sum s
    amd._lib.check("mnf_glow_actnorm_inv_bwd", lib.mnf_glow_actnorm_inv_bwd(
        ud.data_ptr(), gzd.data_ptr(), Md.data_ptr(), sd.data_ptr(), td.data_ptr(), gu.data_ptr(), gM.data_ptr(),
        gs.data_ptr(), gt.data_ptr(), g_ld.data_ptr(), rows, dim, None))
    assert abs(float(ld) - (0.75 - float(s.sum()))) <= 1e-5
    gs = gs - 0.3  # (checked below against the gradient of z alone)
    torch.cuda.synchronize()
    assert _err(z.double(), z64) <= 2e-6
    assert _err(gu.double(), u64.grad) <= 2e-6
    assert _err(gM.double(), M64.grad) <= 2e-5
    assert _err(gs.double(), s64.grad.reshape(-1)) <= 2e-5
    assert _err(gt.double(), t64.grad.reshape(-1)) <= 2e-5
    # the sums are ADDED to, and either column sum may be left out
    amd._lib.check("mnf_glow_actnorm_inv_bwd", lib.mnf_glow_actnorm_inv_bwd(
        ud.data_ptr(), gzd.data_ptr(), Md.data_ptr(), sd.data_ptr(), td.data_ptr(), gu.data_ptr(), gM.data_ptr(),
        None, gt.data_ptr(), None, rows, dim, None))
    torch.cuda.synchronize()
    assert _err(gM.double(), 2 * M64.grad) <= 2e-5 and _err(gt.double(), 2 * t64.grad.reshape(-1)) <= 2e-5


def test_glow_actnorm_inverse_pair_rejects_what_it_has_no_kernel_for(amd):
    lib = amd._lib.load()
    a = torch.zeros(64, 64, device=DEV)
    assert lib.mnf_glow_actnorm_inv(a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), None, None, 64, 32, None) \
        == amd._lib.MNF_ERR_INVALID_ARG  # in place
    b = torch.zeros(64, 64, device=DEV)
    assert lib.mnf_glow_actnorm_inv(a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), b.data_ptr(), None, None, 64, 48, None) \
        == amd._lib.MNF_ERR_UNSUPPORTED
    assert lib.mnf_glow_actnorm_inv(a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), b.data_ptr(), None, None, 0, 32, None) \
        == amd._lib.MNF_OK
    assert lib.mnf_glow_actnorm_inv_bwd(a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), None,
                                        b.data_ptr(), None, None, None, 64, 32, None) == amd._lib.MNF_ERR_INVALID_ARG


@pytest.mark.parametrize("rows,dim", [(777, 32), (8192, 32), (777, 64), (777, 16)])
def test_spline_block_training_pass_with_the_fused_pair_matches_layer_by_layer(amd, rows, dim):
    """-mean log_prob of 3 x [ActNormFlow, Glow, NSF_CL] (d = 32: bench.py's c3t model; 16 and 64: the other dims the pair kernels are built for): the pass with Glow.inverse +
    ActNormFlow.inverse as one autograd node (the default under log_prob) against the same pass layer by layer
    (MNF_NO_PAIR_FUSION=1, whose gradients tests/test_hip_autograd.py checks against the float64 oracle)."""
    flows_mod = amd.flows
    results = {}
    for fused in (True, False):
        torch.manual_seed(11)
        layers = []
        for _ in range(3):
            layers += [amd.ActNormFlow(dim), amd.Glow(dim), amd.NSF_CL(dim, K=8, B=3, n_h=8)]
        model = amd.NormalizingFlowModel(amd.StandardNormal(dim), layers).to(DEV)
        x = recipes.gaussian(5, rows, dim).to(DEV).requires_grad_(True)
        with torch.no_grad():
            model.log_prob(x.detach())  # ActNorm's data-dependent initialisation
        env, flows_mod._NO_PAIR_FUSION_ENV = flows_mod._NO_PAIR_FUSION_ENV, not fused
        try:
            loss = -model.log_prob(x).mean()
            loss.backward()
        finally:
            flows_mod._NO_PAIR_FUSION_ENV = env
        results[fused] = (loss.detach(), x.grad, {n: p.grad for n, p in model.named_parameters()})
    assert abs(float(results[True][0]) - float(results[False][0])) <= 1e-6 * abs(float(results[False][0]))
    assert _err(results[True][1], results[False][1]) <= 5e-6
    for n, g in results[False][2].items():
        assert g is not None and results[True][2][n] is not None, n
        if n.endswith(".s"):
            # ActNorm's s right after its data-dependent initialisation: +1 from log_det and -(1 - 1/rows) from the rows
            # cancel to 1/rows, so the budget is the fp32 error of the O(1) terms, not of their difference
            # (4e-6: both passes add these sums with float atomics in whatever order the workgroups finish; 25 repeats of the
            #  8,192-row case gave up to 2.06e-6 between two runs' orders)
            assert float((results[True][2][n] - g).abs().max()) <= 4e-6, n
        else:
            assert _err(results[True][2][n], g) <= 2e-5, n


def test_graphed_training_step_of_a_spline_block_with_the_fused_pair(amd):
    """GraphedStep over 2 x [ActNormFlow, Glow, NSF_CL] at d = 32 -- the pair node's launches (operand-image gathers,
    mnf_glow_actnorm_inv / _bwd, the sums buffer) recorded in a hipGraph -- against the same steps run eagerly."""
    def build():
        torch.manual_seed(21)
        layers = []
        for _ in range(2):
            layers += [amd.ActNormFlow(32), amd.Glow(32), amd.NSF_CL(32, K=8, B=3, n_h=8)]
        model = amd.NormalizingFlowModel(amd.StandardNormal(32), layers).to(DEV)
        with torch.no_grad():
            model.log_prob(recipes.gaussian(300, 2048, 32).to(DEV))  # ActNorm's data-dependent initialisation
        return model, amd.FusedAdam(amd.FlatParameters(model), lr=1e-3, capturable=True)

    batches = [recipes.gaussian(300 + i, 2048, 32).to(DEV) for i in range(9)]
    model_e, opt_e = build()
    losses_e = []
    for x in [batches[0]] * 3 + batches[1:]:
        opt_e.zero_grad()
        loss = -model_e.log_prob(x).mean()
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
    del loss
    model_g, opt_g = build()
    step = amd.GraphedStep(opt_g, lambda x: -model_g.log_prob(x).mean(), batches[0])
    losses_g = [float(step(x)) for x in batches[1:]]
    for a, b in zip(losses_e[3:], losses_g):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (a, b)
    assert_close(opt_g.flat.data, opt_e.flat.data, 2e-3, "parameters after 11 steps")


@pytest.mark.parametrize("rows,dim", [(1, 32), (17, 32), (1000, 32), (70001, 32), (37, 16), (5000, 16), (37, 64), (5000, 64)])
def test_glow_actnorm_inverse_pair_logprob_kernels_vs_float64_oracle(amd, O, rows, dim):
    """mnf_glow_actnorm_inv_logprob / _bwd (the pair closing a density pass, standard-normal base) against the oracle's
    layers and base.log_prob composed in float64: log p per row, and with a row-dependent cotangent the gradients of
    u, M, s, t, Glow's log|det| and the running log_det.  Budgets as for the plain pair kernels."""
    import math
    lib = amd._lib.load()
    u, M, s, t, _ = _pair_case(500 + rows + dim, rows, dim)
    g = torch.Generator().manual_seed(9 + rows)
    ld_rows = torch.randn(rows, generator=g, dtype=torch.float64)
    g_lp = torch.randn(rows, generator=g, dtype=torch.float64) / rows
    u64, M64, s64, t64, ldr64 = (a.clone().requires_grad_(True) for a in (u, M, s, t, ld_rows))
    ldg64 = torch.tensor(0.75, dtype=torch.float64, requires_grad=True)
    z64, ld_an = O.affine_const(u64 @ M64, s64, t64, inverse=True)
    lp64 = ldr64 + ldg64 + ld_an + (-0.5 * z64 ** 2 - 0.5 * math.log(2 * math.pi)).sum(1)  # distributions: Normal(0, 1)
    lp64.backward(g_lp)
    ud, Md = u.float().to(DEV), M.float().to(DEV).contiguous()
    st = torch.cat((torch.zeros(1), s.float().reshape(-1), t.float().reshape(-1))).to(DEV)
    sd, td = st[1:1 + dim], st[1 + dim:1 + 2 * dim]
    ldg, ldr, gl = torch.full((1,), 0.75, device=DEV), ld_rows.float().to(DEV), g_lp.float().to(DEV)
    lp = torch.empty(rows, device=DEV)
    amd._lib.check("mnf_glow_actnorm_inv_logprob", lib.mnf_glow_actnorm_inv_logprob(
        ud.data_ptr(), Md.data_ptr(), sd.data_ptr(), td.data_ptr(), ldg.data_ptr(), ldr.data_ptr(), lp.data_ptr(), rows, dim,
        None))
    gu = torch.full_like(ud, float("nan"))
    gM, gs, gt, gld = (torch.zeros(n, device=DEV) for n in (dim * dim, dim, dim, 1))
    amd._lib.check("mnf_glow_actnorm_inv_logprob_bwd", lib.mnf_glow_actnorm_inv_logprob_bwd(
        ud.data_ptr(), gl.data_ptr(), Md.data_ptr(), sd.data_ptr(), td.data_ptr(), gu.data_ptr(), gM.data_ptr(),
        gs.data_ptr(), gt.data_ptr(), gld.data_ptr(), rows, dim, None))
    torch.cuda.synchronize()
    assert _err(lp.double(), lp64.detach()) <= 2e-6
    assert _err(gu.double(), u64.grad) <= 2e-6
    assert _err(gM.view(dim, dim).double(), M64.grad) <= 2e-5
    assert _err(gs.double(), s64.grad.reshape(-1)) <= 2e-5
    assert _err(gt.double(), t64.grad.reshape(-1)) <= 2e-5
    assert abs(float(gld) - float(ldg64.grad)) <= 2e-5 * max(1.0, float(g_lp.abs().sum()))
