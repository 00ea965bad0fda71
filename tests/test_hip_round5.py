"""Round 5: the library's deterministic mode (MNF_DETERMINISTIC=1, include/mnf_hip.h mnf_deterministic).

The switch is read once per process, so these tests start child processes.  The reference's training loop repeats bit
for bit under its torch.manual_seed(0) (tests/test_flows.py:11); with the switch on so do the bench's three training
models (c2t: AffineHalfFlow, c3t: [ActNorm, Glow, NSF_CL] blocks, c5t: MNFLinear with its RNVP flows), and the
gradients the fixed-order reductions produce still pass the default mode's parity tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(args, timeout):
    env = dict(os.environ, MNF_DETERMINISTIC="1", SOAK_FIX_ROWS="4096")
    return subprocess.run([sys.executable, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_deterministic_mode_training_repeats_bit_for_bit():
    # (four models, then three of them again with a tenth of the input rows times 1e4: those rows' tiles / row groups go
    #  through the fp32 fix-up passes, whose lists are sorted on the device and which run as one workgroup under the switch)
    p = _child(["tools/soak_determinism_train.py", "6", "65536", "65536", "32768"], 900)
    lines = [ln for ln in p.stdout.splitlines() if "Adam steps twice" in ln]
    assert len(lines) == 7, p.stdout + p.stderr
    for ln in lines:
        assert " 0 of " in ln and "MNF_DETERMINISTIC=1" in ln, ln
    assert p.returncode == 0, p.stdout + p.stderr


@pytest.mark.gpu
def test_deterministic_mode_gradients_pass_the_default_modes_parity_tests():
    # the RNVP / MNFLinear / sample_z gradient launches (fixed-order sums through the workspace extension), the
    # [Glow, ActNorm] pair and the AffineHalfFlow fp32-MFMA gradient kernel (d = 128, padded halves) against the float64
    # oracle, in a process that runs them in deterministic mode
    p = _child(["-m", "pytest", "tests/test_hip_round3.py", "tests/test_hip_round4.py", "tests/test_hip_autograd.py", "-q",
                "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                "rnvp or mnf_linear or MNFLinear or glow_actnorm or sample_z or fp32_mfma_gradient_kernel"], 1200)
    tail = "\n".join(p.stdout.splitlines()[-15:])
    assert p.returncode == 0, tail + p.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail, tail


def test_deterministic_switch_is_exported_and_off_by_default():
    import torch_mnf_amd as amd

    if os.environ.get("MNF_DETERMINISTIC", "0") in ("", "0"):
        assert amd.deterministic() is False
    p = _child(["-c", "import torch_mnf_amd as amd; print(int(amd.deterministic()))"], 300)
    assert p.stdout.strip().endswith("1"), p.stdout + p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("dim,rows,family", [(50, 4096, "rnvp_bwd_rt"), (50, 1024, "rnvp_bwd_generic"), (50, 32768, "rnvp_bwd_mfma"),
                                              (100, 4096, "rnvp_bwd_mfma"), (100, 1024, "rnvp_bwd_generic"),
                                              (800, 4096, "rnvp_bwd_mfma"), (800, 128, "rnvp_bwd_few")])
def test_rnvp_gradient_kernel_choice_by_shape(dim, rows, family):
    """Narrow RNVP layers take the per-shape matrix-core gradient pass only where it measured faster than the one-launch
    kernels (_dispatch.rnvp_bwd_small): below, the run-time-shaped kernel from 2,048 rows on (_dispatch.RT_MIN_ROWS), the
    VALU kernel under that; `last_kernel()` tells which one ran.  Both families are
    held to the oracle by tests/test_hip_round3.py::test_rnvp_mfma_gradient_kernels."""
    import torch

    import torch_mnf_amd as amd

    f = amd.RNVP(dim, h_sizes=(50,)).to("cuda")
    x = torch.randn(rows, dim, device="cuda", requires_grad=True)
    y, ld = f.forward(x, seed=11)
    (y.sum() + ld.sum()).backward()
    torch.cuda.synchronize()
    if amd.deterministic():  # (the run-time-shaped gradient kernels add atomically and refuse under the switch)
        family = family.replace("_bwd_rt", "_bwd_generic")
    assert amd.last_kernel() == family
    assert torch.isfinite(x.grad).all() and all(torch.isfinite(p.grad).all() for p in f.parameters())


@pytest.mark.gpu
def test_padded_nsf_twin_follows_a_fused_optimizer(monkeypatch):
    """FusedAdam rewrites the flat parameter buffer through its raw pointer (no version bump): the padded twin of an
    NSF_CL(dim=2) -- the reference's shape, tests/test_flows.py:89-99 -- must pick the new values up (its refresh key
    includes the buffer's generation).  Three Adam steps with the twin against three with the run-time-shaped kernels."""
    import torch

    import torch_mnf_amd as amd
    import torch_mnf_amd.flows as fl

    def train(min_rows):
        monkeypatch.setattr(fl._dispatch, "NSF_PAD_MIN_ROWS", min_rows)
        torch.manual_seed(5)
        flows = [amd.NSF_CL(dim=2, K=8, B=3, n_h=16) for _ in range(2)]
        model = amd.NormalizingFlowModel(amd.StandardNormal(2), flows).to("cuda")
        opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-2)
        x = torch.randn(4096, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss = -model.log_prob(x).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses, [p.detach().clone() for p in model.parameters()], amd.last_kernel()

    l_twin, p_twin, k_twin = train(0)
    l_any, p_any, k_any = train(1 << 40)
    assert k_twin == "nsf_bwd_tile" and k_any == ("nsf_bwd_generic" if amd.deterministic() else "nsf_bwd_rt"), (k_twin, k_any)
    assert l_twin[2] < l_twin[0]
    for a, b in zip(l_twin, l_any):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (l_twin, l_any)
    for a, b in zip(p_twin, p_any):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4), float((a - b).abs().max())
