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
