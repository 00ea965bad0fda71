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
    env = dict(os.environ, MNF_DETERMINISTIC="1")
    return subprocess.run([sys.executable, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_deterministic_mode_training_repeats_bit_for_bit():
    p = _child(["tools/soak_determinism_train.py", "6", "65536", "65536", "32768"], 600)
    lines = [ln for ln in p.stdout.splitlines() if "Adam steps twice" in ln]
    assert len(lines) == 3, p.stdout + p.stderr
    for ln in lines:
        assert " 0 of " in ln and "MNF_DETERMINISTIC=1" in ln, ln
    assert p.returncode == 0, p.stdout + p.stderr


@pytest.mark.gpu
def test_deterministic_mode_gradients_pass_the_default_modes_parity_tests():
    # the RNVP / MNFLinear / sample_z gradient launches (fixed-order sums through the workspace extension) and the
    # [Glow, ActNorm] pair against the float64 oracle, in a process that runs them in deterministic mode
    p = _child(["-m", "pytest", "tests/test_hip_round3.py", "tests/test_hip_round4.py", "-q", "-x", "-m", "gpu", "-p",
                "no:cacheprovider", "-k", "rnvp or mnf_linear or MNFLinear or glow_actnorm or sample_z"], 1200)
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
@pytest.mark.parametrize("dim,rows,family", [(50, 4096, "rnvp_bwd_generic"), (50, 32768, "rnvp_bwd_mfma"),
                                              (100, 4096, "rnvp_bwd_mfma"), (100, 1024, "rnvp_bwd_generic"),
                                              (800, 4096, "rnvp_bwd_mfma"), (800, 128, "rnvp_bwd_few")])
def test_rnvp_gradient_kernel_choice_by_shape(dim, rows, family):
    """Narrow RNVP layers take the matrix-core gradient pass only where it measured faster than the any-shape kernel
    (flows._rnvp_bwd_small, tools/time_rnvp_bwd_small_dim.py); `last_kernel()` tells which one ran.  Both families are
    held to the oracle by tests/test_hip_round3.py::test_rnvp_mfma_gradient_kernels."""
    import torch

    import torch_mnf_amd as amd

    f = amd.RNVP(dim, h_sizes=(50,)).to("cuda")
    x = torch.randn(rows, dim, device="cuda", requires_grad=True)
    y, ld = f.forward(x, seed=11)
    (y.sum() + ld.sum()).backward()
    torch.cuda.synchronize()
    assert amd.last_kernel() == family
    assert torch.isfinite(x.grad).all() and all(torch.isfinite(p.grad).all() for p in f.parameters())
