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
    line = run_bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--prime-ms", "5")
    assert line["n_gpus"] == 1 and line["distributed"]["ranks"] == 1
    r = line["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "frac_physical" in r and "traffic_source" in r


# ------------------------------------------------------------------ RNVP: the register-resident kernel
def rnvp_layer(amd, seed, dim=800, hid=50):
    sd = recipes.rnvp_params(seed, dim, hid)
    f = amd.RNVP(dim, h_sizes=(hid,))
    f.load_state_dict(sd)
    return f.to(DEV), sd


class streaming_rnvp:
    """MNF_RNVP_RESIDENT=0 for the duration: the streaming split kernel instead of the register-resident one."""

    def __enter__(self):
        os.environ["MNF_RNVP_RESIDENT"] = "0"

    def __exit__(self, *exc):
        os.environ.pop("MNF_RNVP_RESIDENT", None)


@pytest.mark.parametrize("rows", [64, 100, 64 * 300, 64 * 300 + 17, 3])
def test_rnvp_resident_kernel_vs_oracle_and_streaming(amd, O, rows):
    """d = 800, h = 50 with the in-kernel mask runs the register-resident kernel (every z read once): against the
    oracle with the mask the launch used, and against the streaming split kernel on the same inputs.  Row counts:
    whole 64-row groups, a short last group (fp32 body), more groups than CUs (in-place prefetch of the next rows),
    fewer rows than one group."""
    f, sd = rnvp_layer(amd, 901)
    z = recipes.gaussian(902, rows, 800, scale=1.3)
    zc = z.to(DEV)
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
    f._run(zc, False, acc, seed=12345)
    f._run(zc, False, acc, seed=12345)
    assert_close(acc, 0.5 + 2 * ref_ld, RTOL, "accumulated log_det")


def test_rnvp_resident_kernel_is_the_one_that_runs(amd):
    """The seeded d = 800 launch must not silently stay on the streaming kernel: results differ in the last bits
    (bias folded into the accumulator), and switching the resident kernel off changes them."""
    f, _ = rnvp_layer(amd, 903)
    z = recipes.gaussian(904, 64 * 40, 800).to(DEV)
    with torch.no_grad():
        x, _ = f.forward(z, seed=5)
        with streaming_rnvp():
            x_s, _ = f.forward(z, seed=5)
    assert not torch.equal(x, x_s) and normwise_err(x.cpu(), x_s.cpu()) < 2e-6


@pytest.mark.parametrize("case", ["big_inputs", "inf_input", "one_big_row", "big_weights"])
def test_rnvp_resident_range_guard(amd, O, case):
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
    x, ld = f.forward(z.to(DEV), seed=31)
    ref_x, ref_ld = O.rnvp(z, sd, f.mask_for(31, rows).cpu())
    ok = torch.isfinite(ref_x).all(1)
    assert torch.equal(torch.isfinite(x).all(1).cpu(), ok)
    assert_close(x[ok.to(DEV)], ref_x[ok], RTOL, f"{case} x")
    assert_close(ld[ok.to(DEV)], ref_ld[ok], RTOL, f"{case} ld")


def test_sample_z_on_the_resident_kernel(amd, O):
    """MNFLinear(800, 50).sample_z: the prologue z0 = q0_mean + q0_std eps is formed when the first flow's rows are
    first used and kept in the register file for the gate epilogue; the second flow accumulates log_det."""
    layer = amd.MNFLinear(800, 50)
    for i, fl in enumerate(layer.flow_q.flows):
        fl.load_state_dict(recipes.rnvp_params(800 + i, 800, 50))
    layer.to(DEV)
    rows = 64 * 70 + 5
    eps = recipes.gaussian(77, rows, 800)
    torch.manual_seed(99)
    with torch.no_grad():
        z, ld = layer.sample_z(rows, eps=eps.to(DEV))
        torch.manual_seed(99)  # the same seeds again, masks materialised for the oracle
        seeds = [int(torch.empty((), dtype=torch.int64).random_().item()) for _ in range(2)]
        masks = [fl.mask_for(s & 0xFFFFFFFFFFFFFFFF, rows).cpu() for fl, s in zip(layer.flow_q.flows, seeds)]
    specs = [{"kind": "rnvp", "params": recipes.rnvp_params(800 + i, 800, 50), "mask": masks[i]} for i in range(2)]
    z_ref, ld_ref = O.sample_z(layer.q0_mean.detach().cpu(), layer.q0_log_var.detach().cpu(), eps, specs)
    assert_close(z, z_ref, RTOL, "z")
    assert_close(ld, ld_ref, RTOL, "log_det")
