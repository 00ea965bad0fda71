"""GPU parity tests added in round 2: deep stacks (more layers than one stack launch covers), full-size C4 / C5,
the seeded fuzz slice, the self-spawning multi-rank benchmark.

Same tolerance rule as tests/test_hip_parity.py (normwise 1e-5 per output tensor)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipes
from helpers import RTOL, assert_close, c2_layers, normwise_err

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


def ahf_stack(amd, dim, n_layers, base_seed=1000, s_last_gain=2.0, **kw):
    specs, flows = [], []
    for i in range(n_layers):
        sd = recipes.affine_half_params(base_seed + i, dim, s_last_gain=s_last_gain, **kw)
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2), **kw)
        f.load_state_dict(sd)
        flows.append(f)
        specs.append({"kind": "affine_half", "parity": bool(i % 2), "params": sd})
    return amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV), specs


@pytest.mark.parametrize("dim,n_layers", [(64, 33), (64, 70), (2, 40)])
def test_runs_longer_than_one_launch(amd, O, dim, n_layers):
    """The reference takes any depth (core.py:17-35).  One stack launch covers at most 32 layers (parities travel
    as a 32-bit word), so NormalizingFlow and FusedAffineStack cut longer runs into chunks: same tensors as the
    oracle, every intermediate kept."""
    model, specs = ahf_stack(amd, dim, n_layers, s_last_gain=0.5)
    runs = model._affine_runs()
    assert all(len(r.layers) <= 32 for r in runs.values()) and sum(len(r.layers) for r in runs.values()) >= n_layers - 1
    x = recipes.gaussian(77, 500, dim)
    with torch.no_grad():
        for inverse in (True, False):
            zs, ld = (model.inverse if inverse else model.forward)(x.to(DEV))
            ref_zs, ref_ld = O.flow_stack(x, specs, inverse)
            assert len(zs) == n_layers + 1
            for i in (1, 31, 32, 33, n_layers):
                assert_close(zs[i], ref_zs[i], RTOL, f"inverse={inverse} tensor {i}")
            assert_close(ld, ref_ld, RTOL, f"inverse={inverse} log_det")
        lp = model.log_prob(x.to(DEV))
        ref_mean, ref_lp = O.mean_log_prob(x, specs)
        assert_close(lp, ref_lp, RTOL, "log_prob")
        fused = amd.NormalizingFlowModel(model.base, [amd.FusedAffineStack(list(model.flows))]).to(DEV)
        z2, ld2 = fused.inverse(x.to(DEV))
        assert len(z2) == 2
        assert_close(z2[-1], ref_zs_inv(O, x, specs), RTOL, "FusedAffineStack z")
        lp2 = fused.log_prob(x.to(DEV))
        assert_close(lp2, ref_lp, RTOL, "FusedAffineStack log_prob")
    # gradients: every chunk is its own autograd node
    xg = x[:64].to(DEV)
    loss = -model.log_prob(xg).mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def ref_zs_inv(O, x, specs):
    return O.flow_stack(x, specs, True)[0][-1]


def run_bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(amd):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts torch.distributed.run as a child
    process (before touching the GPU) and relays rank 0's one JSON line.  On this one-GPU box both ranks share GPU 0
    and the 16-byte all-reduce runs on gloo (MNF_BENCH_BACKEND / MNF_BENCH_SHARE_GPU are test switches); on an
    8-GPU node the same command runs one rank per GPU over RCCL."""
    line = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--prime-ms", "5",
                     env={"MNF_BENCH_BACKEND": "gloo", "MNF_BENCH_SHARE_GPU": "1"})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["distributed"]["ranks"] == 2 and len(line["distributed"]["ms_per_step_per_rank"]) == 2
    assert line["distributed"]["collective_backend"] == "gloo" and line["distributed"]["rccl_ranks"] == 0
    assert line["config"]["total_rows"] == 2 * line["config"]["rows_per_gpu"]
    assert abs(line["value"] - 2 * line["config"]["rows_per_gpu"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


def test_bench_single_gpu_line(amd):
    """--gpus 1 is unchanged by the launcher logic, and the roofline object carries both fractions."""
    line = run_bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--prime-ms", "5")
    assert line["n_gpus"] == 1 and line["distributed"]["ranks"] == 1
    r = line["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "frac_physical" in r and "traffic_source" in r
    assert line["min_ms"] <= line["median_ms"] and r["min_kernel_us"] <= r["median_kernel_us"]
    assert "secondary" not in line


def test_bench_training_step_line(amd):
    """--workload c2t: one Adam step of the C2 model per step; the line carries the gradient kernel's roofline, the
    oracle's training step on the host as cpu_baseline and the gradients' parity against the float64 oracle."""
    line = run_bench("--workload", "c2t", "--steps", "3", "--warmup", "1", "--prime-ms", "5")
    r = line["roofline"]
    assert line["unit"] == "samples/s" and r["bound"] == "hbm" and "ahf_bwd_split_kernel" in r["kernel"]
    assert r["launches_timed"] == 3 * 9 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert line["loss_last_step"] < line["loss_first_step"]          # Adam is descending
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["parity"]["worst_parameter_gradient_normwise_err"] <= line["parity"]["tolerance"]
    assert line["parity"]["loss_gpu_vs_cpu_rel_err"] <= 1e-6


def test_bench_c3_training_step_line(amd):
    """--workload c3t: one Adam step of config 3's model per step; the NSF_CL row gradient kernel's roofline figures,
    the oracle's training step as cpu_baseline, the gradients' parity against the float64 oracle."""
    line = run_bench("--workload", "c3t", "--steps", "3", "--warmup", "1", "--prime-ms", "5")
    r = line["roofline"]
    assert line["unit"] == "samples/s" and "nsf_bwd_tile_kernel" in r["kernel"] and r["bound"] == "valu"
    assert r["launches_timed"] == 3 * 3 and abs(r.get("frac_plain_count", r["frac"]) - r["achieved"] / r["peak"]) < 1e-12
    assert line["loss_last_step"] < line["loss_first_step"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["parity"]["worst_parameter_gradient_normwise_err"] <= line["parity"]["tolerance"]
    assert line["parity"]["loss_gpu_vs_cpu_rel_err"] <= 1e-5


# ------------------------------------------------------------------ RNVP: the register-resident kernel
def rnvp_layer(amd, seed, dim=800, hid=50):
    sd = recipes.rnvp_params(seed, dim, hid)
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    return f.to(DEV), sd


class rnvp_kernel:
    """Which kernel a seeded d = 800, h = 50 launch runs, for the duration: "resident" (the default: one wave per 16-row
    tile, rows in the register file) or "streaming" (MNF_RNVP_RESIDENT=0)."""

    def __init__(self, which):
        self.env = {"resident": {}, "streaming": {"MNF_RNVP_RESIDENT": "0"}}[which]

    def __enter__(self):
        os.environ.update(self.env)

    def __exit__(self, *exc):
        for k in self.env:
            os.environ.pop(k, None)


def streaming_rnvp():
    return rnvp_kernel("streaming")


REGISTER_KERNELS = ["resident"]


@pytest.mark.parametrize("kernel", REGISTER_KERNELS)
@pytest.mark.parametrize("rows", [64, 100, 128, 64 * 300, 64 * 301, 64 * 300 + 17, 128 * 700 + 3, 3])
def test_rnvp_resident_kernel_vs_oracle_and_streaming(amd, O, rows, kernel):
    """d = 800, h = 50 with the in-kernel mask runs a register-resident kernel (every z read once): against the
    oracle with the mask the launch used, and against the streaming split kernel on the same inputs.  Row counts:
    whole 64- / 128-row groups, a short last group (fp32 body), more groups than CUs (in-place prefetch of the next
    rows), fewer rows than one group."""
    f, sd = rnvp_layer(amd, 901)
    z = recipes.gaussian(902, rows, 800, scale=1.3)
    zc = z.to(DEV)
    with rnvp_kernel(kernel):
        x, ld = f.forward(zc, seed=12345)
    mask = f.mask_for(12345, rows)
    ref_x, ref_ld = O.rnvp(z, sd, mask.cpu())
    assert_close(x, ref_x, RTOL, "x vs oracle")
    assert_close(ld, ref_ld, RTOL, "log_det vs oracle")
    with streaming_rnvp():
        x_s, ld_s = f.forward(zc, seed=12345)
    assert_close(x, x_s, 2e-6, "x vs streaming kernel")
    assert_close(ld, ld_s, 2e-6, "log_det vs streaming kernel")
    # accumulate: log_det += inside the kernel, twice
    acc = torch.full((rows,), 0.5, device=DEV)
    with rnvp_kernel(kernel):
        f._run(zc, False, acc, seed=12345)
        f._run(zc, False, acc, seed=12345)
    assert_close(acc, 0.5 + 2 * ref_ld, RTOL, "accumulated log_det")


def test_rnvp_resident_kernel_is_the_one_that_runs(amd):
    """The seeded d = 800 launch must not silently stay on the streaming kernel: results differ in the last bits
    (bias folded into the accumulator), and switching the kernel off changes them."""
    f, _ = rnvp_layer(amd, 903)
    z = recipes.gaussian(904, 64 * 40, 800).to(DEV)
    with torch.no_grad():
        x_r, _ = f.forward(z, seed=5)
        with streaming_rnvp():
            x_s, _ = f.forward(z, seed=5)
    assert not torch.equal(x_r, x_s) and normwise_err(x_r.cpu(), x_s.cpu()) < 2e-6


@pytest.mark.parametrize("kernel", REGISTER_KERNELS)
@pytest.mark.parametrize("case", ["big_inputs", "inf_input", "one_big_row", "big_weights"])
def test_rnvp_resident_range_guard(amd, O, case, kernel):
    """A 64-row group whose operands leave the f16 range is flagged and redone by the fp32 body at the end of the
    launch; the other groups stay on the split path."""
    hid, rows, dim = 50, 64 * 9, 800
    sd = recipes.rnvp_params(140, dim, hid)
    z = recipes.gaussian(141, rows, dim)
    if case == "big_inputs":
        z = z * 1.0e5
        sd = {k: (v * 1e-5 if k == "net.0.weight" else v) for k, v in sd.items()}
    elif case == "big_weights":
        sd = {k: (v * 1e5 if k == "net.0.weight" else v) for k, v in sd.items()}
        z = z * 1e-5
    elif case == "inf_input":
        z = z.clone()
        z[300, 17] = float("inf")
    elif case == "one_big_row":
        z = z.clone()
        z[5] *= 1e5
        z[450] *= 1e5
        sd = {k: (v * 1e-3 if k == "net.0.weight" else v) for k, v in sd.items()}
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    f.to(DEV)
    with rnvp_kernel(kernel):
        x, ld = f.forward(z.to(DEV), seed=31)
    ref_x, ref_ld = O.rnvp(z, sd, f.mask_for(31, rows).cpu())
    ok = torch.isfinite(ref_x).all(1)
    assert torch.equal(torch.isfinite(x).all(1).cpu(), ok)
    assert_close(x[ok.to(DEV)], ref_x[ok], RTOL, f"{case} x")
    assert_close(ld[ok.to(DEV)], ref_ld[ok], RTOL, f"{case} ld")


@pytest.mark.parametrize("kernel", REGISTER_KERNELS)
def test_sample_z_on_the_resident_kernel(amd, O, kernel):
    """MNFLinear(800, 50).sample_z: the prologue z0 = q0_mean + q0_std eps is formed when the first flow's rows are
    first used and kept in the register file for the gate epilogue; the second flow accumulates log_det."""
    layer = amd.MNFLinear(800, 50)
    for i, fl in enumerate(layer.flow_q.flows):
        fl.load_state_dict(recipes.rnvp_params(800 + i, 800, 50))
    layer.to(DEV)
    rows = 64 * 70 + 5
    eps = recipes.gaussian(77, rows, 800)
    torch.manual_seed(99)
    with torch.no_grad(), rnvp_kernel(kernel):
        z, ld = layer.sample_z(rows, eps=eps.to(DEV))
        torch.manual_seed(99)  # the same seeds again, masks materialised for the oracle
        seeds = [int(torch.empty((), dtype=torch.int64).random_().item()) for _ in range(2)]
        masks = [fl.mask_for(s & 0xFFFFFFFFFFFFFFFF, rows).cpu() for fl, s in zip(layer.flow_q.flows, seeds)]
    specs = [{"kind": "rnvp", "params": recipes.rnvp_params(800 + i, 800, 50), "mask": masks[i]} for i in range(2)]
    z_ref, ld_ref = O.sample_z(layer.q0_mean.detach().cpu(), layer.q0_log_var.detach().cpu(), eps, specs)
    assert_close(z, z_ref, RTOL, "z")
    assert_close(ld, ld_ref, RTOL, "log_det")


# ------------------------------------------------------------------ seeded fuzz slice (tools/fuzz_shapes.py, promoted)
def f64(sd):
    return {k: v.double() for k, v in sd.items()}


def rel(a, b):
    return normwise_err(a.detach().double().cpu().numpy(), b.numpy())


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_seeded_fuzz_slice_against_float64(amd, O, seed):
    """A fixed-seed slice of tools/fuzz_shapes.py inside the driver-run suite: random shapes on the MFMA kernels
    (AffineHalfFlow runs of 1-4 layers with any even d <= 256, any three hidden widths <= 32, NICE / no-shift; RNVP
    49 <= d <= 900, any hidden <= 50; NSF_CL d in {32, 64}, K in {5, 8}, any n_h <= 16), every draw against the oracle
    evaluated in FLOAT64 at the plain 1e-5 normwise bar -- no draw may exceed it (round 1 had one: a one-unit hidden
    layer on the split path, now routed to the fp32 MFMA kernels)."""
    rng = np.random.default_rng(1000 + seed)
    torch.manual_seed(seed)
    worst = 0.0
    nsf_widest = 0.0  # largest float64 head-room a spline draw claimed (printed; capped at 5e-5 per draw)
    for case in range(150):
        u = rng.random()
        if u < 0.2:
            dim, K, n_h = int(rng.choice([32, 64])), int(rng.choice([5, 8])), int(rng.integers(1, 17))
            if dim == 64 and K == 5 and n_h > 8:
                n_h = 8
            rows, inverse = int(rng.integers(1, 1500)), bool(rng.integers(0, 2))
            sd = recipes.nsf_cl_params(int(rng.integers(1 << 30)), dim, K, n_h)
            f = amd.NSF_CL(dim, K=K, B=3, n_h=n_h)
            f.load_state_dict(sd)
            f.to(DEV)
            x = torch.randn(rows, dim) * 1.5
            with torch.no_grad():
                y, ld = (f.inverse if inverse else f.forward)(x.to(DEV))
            y64, ld64 = O.nsf_cl(x.double(), f64(sd), K, 3.0, inverse)
            y32, ld32 = O.nsf_cl(x, sd, K, 3.0, inverse)
            # the spline's bin search is discontinuous in rounding: an element a few ulps from a knot lands in the
            # neighbouring bin in one fp32 evaluation and not in another.  Budget as in helpers.assert_parity: 1e-5 +
            # twice the fp32 oracle's own distance from float64 on this draw.
            wid_y, wid_ld = 2 * normwise_err(y32.numpy(), y64.numpy()), 2 * normwise_err(ld32.numpy(), ld64.numpy())
            # (tame nn.Linear-scale weights: the head-room itself may not exceed 5e-5, helpers.MAX_WIDENING)
            assert max(wid_y, wid_ld) <= 5e-5, (case, "nsf head-room", wid_y, wid_ld)
            budget_y, budget_ld = RTOL + wid_y, RTOL + wid_ld
            assert rel(y, y64) <= budget_y and rel(ld, ld64) <= budget_ld, (case, "nsf", dim, K, n_h, rows, inverse)
            nsf_widest = max(nsf_widest, wid_y, wid_ld)
            continue
        if u < 0.8:
            dim = int(rng.integers(1, 129)) * 2
            h = tuple(int(v) for v in rng.integers(1, 33, size=3)) if rng.random() < 0.6 else (24, 24, 24)
            if dim > 128 and (max(h) > 24 or max(h) <= 16):
                h = (24, 24, 24)
            kw = {}
            r = rng.random()
            if r < 0.15:
                kw["scale"] = False
            elif r < 0.3:
                kw["shift"] = False
            n_layers, rows, inverse = int(rng.integers(1, 5)), int(rng.integers(1, 3000)), bool(rng.integers(0, 2))
            flows, specs = [], []
            for i in range(n_layers):
                parity = bool(rng.integers(0, 2))
                sd = recipes.affine_half_params(int(rng.integers(1 << 30)), dim, h_sizes=h, s_last_gain=1.5, **kw)
                f = amd.AffineHalfFlow(dim, parity, h_sizes=h, **kw)
                f.load_state_dict(sd)
                flows.append(f)
                specs.append({"kind": "affine_half", "parity": parity, "params": f64(sd), **kw})
            model = amd.NormalizingFlow(flows).to(DEV)
            x = torch.randn(rows, dim) * float(rng.choice([0.1, 1.0, 3.0]))
            with torch.no_grad():
                zs, ld = (model.inverse if inverse else model.forward)(x.to(DEV))
            z64, ld64 = O.flow_stack(x.double(), specs, inverse)
            if not (torch.isfinite(z64[-1]).all() and torch.isfinite(ld64).all()):
                continue  # the draw overflows in float64 too
            e = max(rel(zs[-1], z64[-1]), rel(ld, ld64))
            worst = max(worst, e)
            assert e <= RTOL, (case, "ahf", dim, h, kw, n_layers, rows, inverse, e)
            continue
        dim, hid, rows = int(rng.integers(49, 901)), int(rng.integers(1, 51)), int(rng.integers(1, 2000))
        sd = recipes.rnvp_params(int(rng.integers(1 << 30)), dim, hid)
        f = amd.RNVP(dim, h_sizes=(hid,))
        f.load_state_dict(sd)
        f.to(DEV)
        z = torch.randn(rows, dim)
        s = int(rng.integers(1 << 40))
        with torch.no_grad():
            x1, l1 = f.forward(z.to(DEV), seed=s)
            mask = f.mask_for(s, rows).cpu()
        x64, l64 = O.rnvp(z.double(), f64(sd), mask.double())
        e = max(rel(x1, x64), rel(l1, l64))
        worst = max(worst, e)
        assert e <= RTOL, (case, "rnvp", dim, hid, rows, e)
    assert worst > 0.0
    print(f"fuzz seed {seed}: worst error outside the spline draws {worst:.2e}; widest spline head-room {nsf_widest:.2e}")


def test_narrow_hidden_layers_take_the_fp32_kernels(amd, O):
    """The round-1 fuzz draw that ended 1.2e-5 from float64 on the split path: three layers through a ONE-unit hidden
    layer, inputs scaled by 3.  Such conditioners now run on the fp32 MFMA kernels by themselves (_MIN_SPLIT_HIDDEN)."""
    dim, h = 64, (20, 1, 9)
    flows, specs = [], []
    for i in range(3):
        sd = recipes.affine_half_params(7000 + i, dim, h_sizes=h, s_last_gain=1.5)
        f = amd.AffineHalfFlow(dim, bool(i % 2), h_sizes=h)
        f.load_state_dict(sd)
        flows.append(f)
        specs.append({"kind": "affine_half", "parity": bool(i % 2), "params": f64(sd)})
    model = amd.NormalizingFlow(flows).to(DEV)
    assert all(f._split_image(torch.device(DEV, 0)) is None and f._packed(torch.device(DEV, 0))[1] is not None
               for f in flows)
    x = recipes.gaussian(7003, 2000, dim, scale=3.0)
    with torch.no_grad():
        zs, ld = model.forward(x.to(DEV))
    z64, ld64 = O.flow_stack(x.double(), specs, False)
    assert rel(zs[-1], z64[-1]) <= RTOL and rel(ld, ld64) <= RTOL


# ------------------------------------------------------------------ BASELINE configs[3] / [4] at full size
def test_c4_full_size_properties(amd, O):
    """BASELINE configs[3], one GPU's shard at full size (2^19 x 256, 9 layers, 4.8 GB of intermediates; the 8-wave
    wide stack kernel with a persistent grid that wraps): the same checks as test_c2_full_size_properties."""
    dim, rows = 256, 1 << 19
    model, specs = ahf_stack(amd, dim, 9)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(rows, dim, device=DEV, generator=g)
    with torch.no_grad():
        zs, ld = model.inverse(x)
        assert len(zs) == 10 and torch.isfinite(zs[-1]).all() and torch.isfinite(ld).all()
        z_last = zs[-1].clone()
        mids = [z[:: rows // 2048][:2048].cpu() for z in zs]  # a 2,048-row slice of every intermediate
        del zs
        xs, ld_f = model.forward(z_last)
        # forward(inverse(x)) == x ; the two log-dets cancel
        assert float((xs[-1] - x).abs().max()) <= 5e-4 * float(x.abs().max())
        assert float((ld + ld_f).abs().max()) <= 2e-4 * float(ld.abs().max())
        del xs
        # the slice against the oracle: every intermediate tensor and log_det
        sel = torch.arange(0, rows, rows // 2048, device=DEV)[:2048]
        ref_zs, ref_ld = O.flow_stack(x[sel].cpu(), specs, True)
        for i, (a, b) in enumerate(zip(mids, ref_zs)):
            assert_close(a, b, RTOL, f"intermediate {i}")
        assert_close(ld[sel], ref_ld, RTOL, "log_det slice")
        # log-prob: fused epilogue, fp64 sum identity over the whole shard, slice mean vs the oracle
        lp_all, total_all = model.log_prob(x, return_sum=True)
        assert model._logprob_done
        assert abs(float(total_all.item()) - float(lp_all.double().sum())) <= 1e-9 * abs(float(total_all.item()))
        ref_mean, ref_lp = O.mean_log_prob(x[sel].cpu(), specs)
        assert_close(lp_all[sel], ref_lp, RTOL, "log_prob slice")
        # the split stack kernel == layer-by-layer launches on the full shard (last tensor, log_det)
        model.fuse_affine_runs = False
        zs_u, ld_u = model.inverse(x)
        assert_close(zs_u[-1][sel], z_last[sel], 1e-6, "run fusion vs layer by layer")
        assert_close(ld_u[sel], ld[sel], 1e-6, "run fusion vs layer by layer (log_det)")


def test_c5_full_size_sample_z(amd, O):
    """BASELINE configs[4] at full size: MNFLinear(800, 50).sample_z on 512 x 500 = 256,000 rows with the in-kernel
    masks (the register-resident RNVP kernel, prologue fused): finite everywhere, a 4,096-row slice spread over the
    batch against the oracle (masks materialised from the same seeds), and MNFLinear(50, 10)'s ragged flow too."""
    rows = 512 * 500
    for n_in, n_out in ((800, 50), (50, 10)):
        layer = amd.MNFLinear(n_in, n_out)
        for i, fl in enumerate(layer.flow_q.flows):
            fl.load_state_dict(recipes.rnvp_params(800 + i, n_in, 50))
        layer.to(DEV)
        g = torch.Generator(device=DEV).manual_seed(5)
        eps = torch.randn(rows, n_in, device=DEV, generator=g)
        torch.manual_seed(4242)
        with torch.no_grad():
            z, ld = layer.sample_z(rows, eps=eps)
            assert z.shape == (rows, n_in) and ld.shape == (rows,)
            assert torch.isfinite(z).all() and torch.isfinite(ld).all()
            torch.manual_seed(4242)
            seeds = [int(torch.empty((), dtype=torch.int64).random_().item()) for _ in layer.flow_q.flows]
            sel = torch.arange(0, rows, rows // 4096, device=DEV)[:4096]
            masks = [fl.mask_for(s, rows)[sel].cpu() for fl, s in zip(layer.flow_q.flows, seeds)]
        specs = [{"kind": "rnvp", "params": recipes.rnvp_params(800 + i, n_in, 50), "mask": masks[i]} for i in range(2)]
        z_ref, ld_ref = O.sample_z(layer.q0_mean.detach().cpu(), layer.q0_log_var.detach().cpu(), eps[sel].cpu(), specs)
        assert_close(z[sel], z_ref, RTOL, f"z ({n_in})")
        assert_close(ld[sel], ld_ref, RTOL, f"log_det ({n_in})")
        del z, ld, eps


# ------------------------------------------------------------------ MNFLinear.forward behind the flow path
def g11_layer(amd, fx, tag):
    from test_oracle_golden import G11_CASES

    n_in, n_out, seed = G11_CASES[tag]
    layer = amd.MNFLinear(n_in, n_out)
    sd = {k: torch.from_numpy(fx[f"{tag}.{k}"]) for k in ("W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var")}
    layer.load_state_dict(sd, strict=False)
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(1100 + seed + i, n_in, 50))
    return layer.to(DEV), n_in, n_out


@pytest.mark.parametrize("tag", ["l800", "l50"])
def test_g11_mnf_linear_forward_vs_reference(amd, golden, tag):
    """Fixture G11: the reference's MNFLinear.forward with its three random draws captured.  Here: sample_z with the
    captured noise and masks, then mnf_mnf_linear_fwd (both products in split arithmetic + the noise epilogue, one
    launch) with the captured output noise."""
    from helpers import unpack_mask

    fx = golden("g11_mnf_linear_forward")
    layer, n_in, n_out = g11_layer(amd, fx, tag)
    x = torch.from_numpy(fx[f"{tag}.x"]).to(DEV)
    masks = [unpack_mask(fx[f"{tag}.mask{i}_bits"], n_in).to(DEV) for i in range(2)]
    eps_z, eps_out = torch.from_numpy(fx[f"{tag}.eps_z"]).to(DEV), torch.from_numpy(fx[f"{tag}.eps_out"]).to(DEV)
    with torch.no_grad():
        z, _ = layer.sample_z(x.shape[0], eps=eps_z, masks=masks)
        # forward() draws its own z; the kernel is the part under test: call it on the fixture's z
        real = layer.sample_z
        layer.sample_z = lambda n: (z, None)
        try:
            y = layer.forward(x, eps=eps_out)
        finally:
            layer.sample_z = real
    assert layer._forward_operands(torch.device(DEV, 0)) is not None
    assert_close(y, fx[f"{tag}.y"], RTOL, "y vs the reference")


@pytest.mark.parametrize("n_in,n_out,rows", [(800, 50, 1000), (800, 50, 128 * 40 + 3), (50, 10, 777), (784, 64, 300),
                                             (100, 33, 129), (30, 7, 50), (500, 1, 200)])
def test_mnf_linear_forward_kernel_vs_oracle(amd, O, n_in, n_out, rows):
    """mnf_mnf_linear_fwd on seeded inputs against the oracle (mnf_linear.py:46-56): widths with partial K-steps and
    output tiles, ragged rows, trained-like tiny variances, injected and in-kernel noise."""
    torch.manual_seed(n_in + n_out)
    layer = amd.MNFLinear(n_in, n_out).to(DEV)
    with torch.no_grad():
        layer.W_log_var.add_(-3.0)  # exp(-12) ~ 6e-6: below the f16 normal range before the pack-time scaling
        layer.b_mean.normal_(0, 0.3)
    x = recipes.gaussian(5, rows, n_in, scale=1.2).abs().to(DEV)
    z = (1.0 + 0.3 * recipes.gaussian(6, rows, n_in)).to(DEV)
    eps = recipes.gaussian(7, rows, n_out).to(DEV)
    real = layer.sample_z
    layer.sample_z = lambda n: (z, None)
    try:
        with torch.no_grad():
            y = layer.forward(x, eps=eps)
            torch.manual_seed(31)
            y_seeded = layer.forward(x)
            torch.manual_seed(31)
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
    finally:
        layer.sample_z = real
    p = {k: v.detach().cpu() for k, v in layer.state_dict().items()}
    ref = O.mnf_linear_forward(x.cpu(), z.cpu(), p["W_mean"], p["W_log_var"], p["b_mean"], p["b_log_var"], eps.cpu())
    assert_close(y, ref, RTOL, "y (injected noise)")
    e2 = layer.noise_for(seed, rows)
    ref2 = O.mnf_linear_forward(x.cpu(), z.cpu(), p["W_mean"], p["W_log_var"], p["b_mean"], p["b_log_var"], e2.cpu())
    assert_close(y_seeded, ref2, RTOL, "y (in-kernel noise)")
    # the in-kernel stream is standard normal
    if rows * n_out >= 20000:
        assert abs(float(e2.mean())) < 0.03 and abs(float(e2.std()) - 1.0) < 0.03


def test_mnf_linear_forward_range_guard_and_training_path(amd, O):
    """Activations whose squares leave the f16 range: the 128-row groups concerned are redone in fp32 by the fix-up
    launch.  With gradients wanted the layer runs the reference's composition under autograd."""
    torch.manual_seed(3)
    layer = amd.MNFLinear(800, 50).to(DEV)
    rows = 128 * 5
    x = recipes.gaussian(8, rows, 800).abs()
    x[130] *= 400.0   # x^2 ~ 1e6 in group 1
    x[600, 7] = 3.0e4
    z = 1.0 + 0.3 * recipes.gaussian(9, rows, 800)
    eps = recipes.gaussian(10, rows, 50)
    real = layer.sample_z
    layer.sample_z = lambda n: (z.to(DEV), None)
    try:
        with torch.no_grad():
            y = layer.forward(x.to(DEV), eps=eps.to(DEV))
        y_train = layer.forward(x.to(DEV), eps=eps.to(DEV))
    finally:
        layer.sample_z = real
    p = {k: v.detach().cpu() for k, v in layer.state_dict().items()}
    ref = O.mnf_linear_forward(x, z, p["W_mean"], p["W_log_var"], p["b_mean"], p["b_log_var"], eps)
    for lo in range(0, rows, 128):  # per group: the guarded rows dominate the norm otherwise
        assert_close(y[lo:lo + 128], ref[lo:lo + 128], RTOL, f"rows {lo}..")
    assert y_train.requires_grad
    assert_close(y_train, ref, RTOL, "autograd path")
    y_train.sum().backward()
    assert layer.W_mean.grad is not None and layer.W_log_var.grad is not None


# ------------------------------------------------------------------ NSF_AR on the spline device function
G12_CASES = {"d2_k8": (2, 8, 16, 1.0), "d6_k5": (6, 5, 8, 1.0), "d16_k8": (16, 8, 8, 1.5)}  # dim, K, n_h, gain


def nsf_ar_layer(amd, sd, dim, K, n_h):
    f = amd.NSF_AR(dim, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd)
    return f.to(DEV)


@pytest.mark.parametrize("tag", sorted(G12_CASES))
def test_g12_nsf_ar_vs_reference(amd, O, golden, tag):
    """Fixture G12: the reference's NSF_AR.forward / .inverse (spline_flow.py:182-235)."""
    from helpers import assert_parity

    fx = golden("g12_nsf_ar")
    dim, K, n_h, gain = G12_CASES[tag]
    sd = recipes.nsf_ar_params(1200 + dim + K, dim, K, n_h, gain=gain)
    f = nsf_ar_layer(amd, sd, dim, K, n_h)
    assert list(f.state_dict()) == list(sd)  # init_param, layers.{i}.{0,2,4,6}.{weight,bias}
    x = torch.from_numpy(fx[f"{tag}.x"]).to(DEV)
    with torch.no_grad():
        for direction, fn in (("fwd", f.forward), ("inv", f.inverse)):
            y, ld = fn(x)
            assert_parity(y, fx[f"{tag}.{direction}"], fx[f"{tag}.{direction}64"], f"g12 {tag} {direction} y",
                          max_widening=None if gain != 1.0 else 5e-5)
            # (the fixture carries the reference's fp32 log-det only -- its float64 run allocates a float32 log-det --: the
            # float64 head-room comes from the oracle in float64 on the same inputs; round 3: a flat 3e-5)
            _, ld64 = O.nsf_ar(x.cpu().double(), {k: v.double() for k, v in sd.items()}, K, 3.0, direction == "inv")
            assert_parity(ld, fx[f"{tag}.ld_{direction}"], ld64.numpy(), f"g12 {tag} {direction} log_det",
                          max_widening=None if gain != 1.0 else 5e-5)
        # inverse(forward(x)) == x and the log-dets cancel
        y, ld = f.forward(x)
        back, ld_b = f.inverse(y)
        assert float((back - x).abs().max()) <= 2e-4 * float(x.abs().max())
        assert float((ld + ld_b).abs().max()) <= 2e-4 * max(float(ld.abs().max()), 1.0)


@pytest.mark.parametrize("dim,K,n_h,rows", [(3, 5, 8, 1000), (12, 8, 4, 333), (32, 8, 8, 4099), (1, 5, 8, 17)])
def test_nsf_ar_vs_oracle_and_in_a_stack(amd, O, dim, K, n_h, rows):
    """Seeded shapes against the oracle (float64 budget as for every spline test), incl. dim = 1 (init_param only),
    ragged row counts, +-T and outside-interval inputs, and the layer inside a NormalizingFlow (log_det += in-kernel)."""
    from helpers import assert_parity

    sd = recipes.nsf_ar_params(50 + dim, dim, K, n_h)
    f = nsf_ar_layer(amd, sd, dim, K, n_h)
    x = recipes.gaussian(51 + dim, rows, dim, scale=1.5)
    x[0, 0] = 3.0
    x[1 % rows, dim - 1] = -3.0
    x[3 % rows, dim // 2] = 7.5  # outside the interval: identity, log-det 0
    # (no NaN rows: a NaN element reaches the conditioners of the later elements, and the reference then stops at
    #  its `assert discriminant >= 0`, spline_flow.py:143)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        for inverse in (False, True):
            y, ld = (f.inverse if inverse else f.forward)(x.to(DEV))
            r32 = O.nsf_ar(x, sd, K, 3.0, inverse)
            r64 = O.nsf_ar(x.double(), sd64, K, 3.0, inverse)
            assert_parity(y, r32[0].numpy(), r64[0].numpy(), f"inverse={inverse} y")
            ok = ~torch.isnan(r32[1])
            assert_parity(ld[ok.to(DEV)], r32[1][ok].numpy(), r64[1][ok].numpy(), f"inverse={inverse} log_det")
        flow = amd.NormalizingFlow([f, nsf_ar_layer(amd, recipes.nsf_ar_params(99, dim, K, n_h), dim, K, n_h)])
        xs = x[4:].contiguous()
        zs, ld = flow.inverse(xs.to(DEV))
        specs = [{"kind": "nsf_ar", "K": K, "B": 3.0, "params": sd},
                 {"kind": "nsf_ar", "K": K, "B": 3.0, "params": recipes.nsf_ar_params(99, dim, K, n_h)}]
        ref_zs, ref_ld = O.flow_stack(xs, specs, True)
        assert len(zs) == 3
        assert_close(zs[-1], ref_zs[-1], 1e-4, "stack z")  # (two splines deep: the reference's own fp32 noise)
        assert_close(ld, ref_ld, 1e-4, "stack log_det")


@pytest.mark.parametrize("dim,K,n_h", [(2, 8, 16), (5, 5, 8), (9, 8, 6)])
@pytest.mark.parametrize("inverse", [False, True])
def test_nsf_ar_gradients_vs_autograd_through_the_oracle(amd, O, dim, K, n_h, inverse):
    """mnf_nsf_ar_bwd (reverse mode through the sequential direction too) against torch.autograd through the oracle:
    gradient wrt the input and wrt every parameter tensor, loss = sum(w * y) + sum(v * log_det)."""
    from test_hip_autograd import OracleGrads, cot_loss

    sd = recipes.nsf_ar_params(70 + dim, dim, K, n_h)
    rows = 97
    x = recipes.gaussian(71 + dim, rows, dim, scale=1.2)
    w, v = recipes.gaussian(72, rows, dim), recipes.gaussian(73, rows, 1)[:, 0]
    # fp32 AND float64 autograd through the oracle: the budget is 1e-5 + twice the fp32 oracle's own distance from
    # float64 (round 3 held this test to a flat 2e-4)
    og = OracleGrads(cot_loss(lambda xx, p: O.nsf_ar(xx, p, K, 3.0, inverse), w, v), x, sd)
    f = nsf_ar_layer(amd, sd, dim, K, n_h)
    xg = x.clone().to(DEV).requires_grad_(True)
    y, ld = (f.inverse if inverse else f.forward)(xg)
    ((w.to(DEV) * y).sum() + (v.to(DEV) * ld).sum()).backward()
    got = {"x": xg.grad, **{name: prm.grad for name, prm in f.named_parameters()}}
    og.check_all(got, f"NSF_AR d={dim} K={K} inverse={inverse}")


# ------------------------------------------------------------------ training: flat parameters, one optimizer launch
def moons(n):
    sk = pytest.importorskip("sklearn.datasets")
    pts, _ = sk.make_moons(n, noise=0.05, random_state=0)
    return torch.as_tensor(pts).float()


@pytest.mark.parametrize("kind", ["ahf_d64", "ahf_d2", "mixed"])
def test_flat_parameters_and_fused_adam_match_torch_adam(amd, kind):
    """train.FlatParameters + train.FusedAdam (parameters re-homed in one buffer, gradients written in place by the
    run's kernels, ONE optimizer launch) against torch.optim.Adam on an identical copy: same losses step by step,
    same parameters at the end; state_dict keys and load_state_dict unaffected."""
    import copy

    torch.manual_seed(0)
    if kind == "mixed":
        dim = 2
        flows = [amd.ActNormFlow(2), amd.Glow(2), amd.NSF_CL(2, K=8, B=3, n_h=16), amd.AffineHalfFlow(2, False),
                 amd.AffineHalfFlow(2, True), amd.NSF_AR(2, K=5, B=3, n_h=8)]
        x = moons(256).to(DEV)
    else:
        dim = 64 if kind == "ahf_d64" else 2
        flows = []
        for i, sd in enumerate(recipes.c2_stack_params(dim, 5)):
            f = amd.AffineHalfFlow(dim, parity=bool(i % 2))
            f.load_state_dict(sd)
            flows.append(f)
        x = recipes.gaussian(3, 1000, dim).to(DEV) if dim == 64 else moons(512).to(DEV)
    ref = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(DEV)
    if kind == "mixed":
        with torch.no_grad():
            ref.inverse(x)  # ActNorm's data-dependent init, before the copy
    model = copy.deepcopy(ref)
    keys = list(model.state_dict())
    flat = amd.FlatParameters(model)
    assert list(model.state_dict()) == keys
    assert sum(p.numel() for p in model.parameters()) == flat.data.numel()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=2e-3)
    opt = amd.FusedAdam(flat, lr=2e-3)
    for step in range(6):
        opt_ref.zero_grad()
        loss_ref = -ref.log_prob(x).mean()
        loss_ref.backward()
        opt_ref.step()
        opt.zero_grad()
        loss = -model.log_prob(x).mean()
        loss.backward()
        opt.step()
        assert abs(float(loss) - float(loss_ref)) <= 2e-5 * abs(float(loss_ref)), (step, float(loss), float(loss_ref))
    # Two Adam implementations fed by the same HIP gradients: what separates them after six steps is the run-to-run
    # noise of the gradient sums (fp32 atomics: ~1e-7 relative) passed through m / (sqrt(v) + eps), which turns a
    # relative change of a near-zero gradient into a change of up to one step (lr) of that parameter.  Budget: 1 % of
    # the distance six steps can move a parameter (6 lr), relative to the largest parameter -- recorded for the audit.
    from helpers import budgeted, normwise_err
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        scale = float(p2.detach().abs().max())
        budgeted(normwise_err(p1.detach().cpu().numpy(), p2.detach().cpu().numpy()), 0.01 * 6 * 2e-3 / max(scale, 1e-30),
                 f"FusedAdam vs torch.optim.Adam, 6 steps: {n1}")
    # the parameters are still views of the one buffer, and loading a state_dict writes through them
    assert all(p.data_ptr() == flat.data.data_ptr() + 4 * flat.offset[id(p)] for p in flat.params)
    model.load_state_dict(ref.state_dict())
    flat.touch()
    with torch.no_grad():
        assert_close(model.log_prob(x), ref.log_prob(x), 1e-6, "after load_state_dict")


# ------------------------------------------------------------------ MNFConv2d drop-in (fixture G13)
@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_g13_mnf_conv2d_vs_reference(amd, golden, tag):
    """Fixture G13: the reference's MNFConv2d.forward and kl_div with every random draw captured.  The drop-in takes the
    reference's state_dict (same keys), runs flow_q / flow_r through the HIP RNVP kernels on the captured masks and
    the convolutions on stock PyTorch-ROCm."""
    from test_oracle_golden import G13_CASES, G13_KEYS

    fx = golden("g13_mnf_conv2d")
    n_in, n_out, k, seed = G13_CASES[tag]
    layer = amd.MNFConv2d(n_in, n_out, k)
    assert sorted(k_ for k_ in layer.state_dict() if "flow" not in k_) == sorted(G13_KEYS)  # b_mean: a plain tensor (:45)
    layer.load_state_dict({k_: torch.from_numpy(fx[f"{tag}.{k_}"]) for k_ in G13_KEYS}, strict=False)
    for which, flow in (("q", layer.flow_q), ("r", layer.flow_r)):
        for i, f in enumerate(flow.flows):
            f.load_state_dict(recipes.rnvp_params(1300 + seed + 10 * (which == "r") + i, n_out, 50))
    layer.to(DEV)
    dev = lambda name: torch.from_numpy(fx[f"{tag}.{name}"]).to(DEV)
    masks = lambda name: [m for m in dev(name)]
    with torch.no_grad():
        y = layer.forward(dev("x"), eps=dev("fwd.eps_out"), eps_z=dev("fwd.eps_z"), masks=masks("fwd.masks"))
        kl = layer.kl_div({"eps_z": dev("kl.eps_z"), "masks_q": masks("kl.masks_q"), "eps_w": dev("kl.eps_w"),
                           "eps_b": dev("kl.eps_b"), "masks_r": masks("kl.masks_r")})
    assert_close(y, fx[f"{tag}.y"], RTOL, "MNFConv2d.forward vs the reference")
    assert abs(float(kl) - float(fx[f"{tag}.kl"])) <= 1e-5 * abs(float(fx[f"{tag}.kl"])), (float(kl), float(fx[f"{tag}.kl"]))
    # default draws: runs, finite, differentiable (flow_q gradients through the HIP backward kernels)
    out = layer.forward(dev("x"))
    loss = out.pow(2).mean() + 1e-3 * layer.kl_div()
    loss.backward()
    assert torch.isfinite(out).all() and all(p.grad is not None and torch.isfinite(p.grad).all()
                                             for p in layer.parameters())


@pytest.mark.parametrize("dim,rows", [(2, 128), (64, 4096)])
def test_graphed_training_step_follows_the_eager_one(amd, dim, rows):
    """GraphedStep (zero_grad + log_prob loss + backward + FusedAdam captured in one hipGraph) against the same steps
    run eagerly: same parameters after 3 warm-up + 25 steps on changing batches (up to the order of the gradient
    atomics), the loss tensor follows, and an eager evaluation after replays sees the updated parameters."""
    def build():
        flows = []
        for i, sd in enumerate(recipes.c2_stack_params(dim)):
            f = amd.AffineHalfFlow(dim, parity=bool(i % 2))
            f.load_state_dict(sd)
            flows.append(f)
        model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to("cuda")
        return model, amd.FusedAdam(amd.FlatParameters(model), lr=1e-3, capturable=True)

    batches = [recipes.gaussian(900 + i, rows, dim).to("cuda") for i in range(26)]
    model_e, opt_e = build()
    losses_e = []
    for x in [batches[0]] * 3 + batches[1:]:
        opt_e.zero_grad()
        loss = -model_e.log_prob(x).mean()
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))

    model_g, opt_g = build()
    step = amd.GraphedStep(opt_g, lambda x: -model_g.log_prob(x).mean(), batches[0])
    losses_g = [float(step(x)) for x in batches[1:]]
    assert step.replays == 25 and float(opt_g.state[0]) == 28.0
    for a, b in zip(losses_e[3:], losses_g):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (a, b)
    assert_close(opt_g.flat.data, opt_e.flat.data, 2e-3, "parameters after 28 steps")
    with torch.no_grad():  # eager pass after replays: repacks the operand images from the updated parameters
        lp_g, lp_e = model_g.log_prob(batches[0]), model_e.log_prob(batches[0])
    assert_close(lp_g, lp_e, 2e-3, "log_prob with the trained parameters")
    with pytest.raises(ValueError):
        step(batches[1][:-1])
    with pytest.raises(ValueError):
        amd.GraphedStep(amd.FusedAdam(opt_e.flat), lambda x: x.sum(), batches[0])


def test_graphed_training_step_of_an_mnf_model(amd):
    """A step of an MNF classifier (MNFLinear layers: RNVP flows, torch.optim.Adam(capturable=True)) captured in a
    hipGraph: the RNVP masks are device draws there, so replays on the same batch differ (fresh masks and noise), the
    loss falls over replays, and an eager pass afterwards sees the trained parameters."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(amd.MNFLinear(64, 50), torch.nn.ReLU(), amd.MNFLinear(50, 10),
                              torch.nn.LogSoftmax(dim=-1)).to("cuda")
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, capturable=True)
    x = torch.rand(128, 64, device="cuda")
    y = (x[:, :10].argmax(1)).to(torch.int64)

    def loss_fn(xb, yb):
        kl = sum(m.kl_div() for m in net if hasattr(m, "kl_div"))
        return torch.nn.functional.nll_loss(net(xb), yb) + kl / 60000

    with pytest.raises(ValueError):
        amd.GraphedStep(torch.optim.Adam(net.parameters()), loss_fn, (x, y), model=net)
    step = amd.GraphedStep(opt, loss_fn, (x, y), model=net)
    losses = [float(step(x, y)) for _ in range(150)]
    assert all(l == l for l in losses)
    assert len({round(l, 6) for l in losses[:8]}) > 1, "replays must redraw masks and noise"
    assert sum(losses[-20:]) / 20 < sum(losses[:20]) / 20 - 0.05, (losses[:3], losses[-3:])
    with torch.no_grad():
        acc = float((net(x).argmax(1) == y).float().mean())
    assert acc > 0.5, acc  # (chance is 0.1; the eager pass repacked its operands from the replayed parameters)


def test_graphed_training_step_of_the_spline_block_model(amd):
    """ActNorm + Glow + NSF_CL blocks (config 3's model) captured in a hipGraph: Glow's permutation is copied to the
    device once (not per call) and its inverse has no host-side check, so the whole step records; the replayed
    losses follow the eager loop's."""
    import bench

    def build():
        model, _ = bench.build_c3(torch.device("cuda", 0))
        return model, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)

    batches = [recipes.gaussian(950 + i, 512, 32).to("cuda") for i in range(13)]
    model_e, opt_e = build()
    losses_e = []
    for x in [batches[0]] * 3 + batches[1:]:
        opt_e.zero_grad()
        loss = -model_e.log_prob(x).mean()
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))
    del loss
    model_g, opt_g = build()
    step = amd.GraphedStep(opt_g, lambda x: -model_g.log_prob(x).mean(), batches[0], model=model_g)
    losses_g = [float(step(x)) for x in batches[1:]]
    for a, b in zip(losses_e[3:], losses_g):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(a)), (a, b)
    assert losses_g[-1] < losses_g[0]
