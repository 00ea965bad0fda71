"""Are training steps reproducible bit for bit?  For each of the bench's training workloads (c2t: 9 x AffineHalfFlow at
d = 64; c3t: 3 x [ActNorm, Glow, NSF_CL] at d = 32; c5t: MNFLinear(800, 50)) and for 9 x AffineHalfFlow at d = 256 (c4t:
the fp32-MFMA gradient kernel) -- and for the first three again with a tenth of the input rows times 1e4 (c2t_fix, c3t_fix,
c5t_fix: those rows leave the split format's range, so their tiles / row groups take the fp32 fix-up passes, whose
sums were the last atomics under MNF_DETERMINISTIC until round 6) -- the same N Adam steps are run twice from
identical parameters, inputs and seeds; the parameters after each run must be identical (torch.equal).  The reference's
loop is reproducible under its torch.manual_seed(0) (tests/test_flows.py:11).

usage: python3 tools/soak_determinism_train.py [steps] [rows_c2t rows_c3t rows_c5t]      (MNF_DETERMINISTIC=1: the
fixed-order reductions everywhere)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import torch_mnf_amd as amd
from torch_mnf_amd import synthetic as recipes

dev = torch.device("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else [1 << 18, 1 << 18, 64000]


def run_c2t():
    model, _ = bench.build_model(64, dev)
    opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-3)
    x = torch.randn(rows[0], 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    return model, opt, (lambda: -model.log_prob(x).mean())


def run_c4t():
    """BASELINE configs[3]'s shape in training: 9 x AffineHalfFlow at d = 256 (the fp32-MFMA gradient kernel)"""
    model, _ = bench.build_model(256, dev)
    opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-3)
    x = torch.randn(max(rows[0] // 4, 4096), 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    return model, opt, (lambda: -model.log_prob(x).mean())


def run_c3t():
    model, _ = bench.build_c3(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = torch.randn(rows[1], 32, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    return model, opt, (lambda: -model.log_prob(x).mean())


def run_c5t():
    torch.manual_seed(55)
    layer = amd.MNFLinear(800, 50)
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(800 + i, 800, 50))
    layer = layer.to(dev)
    opt = amd.FusedAdam(amd.FlatParameters(layer), lr=1e-3)
    x = torch.rand(rows[2], 800, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    return layer, opt, (lambda: layer.forward(x).pow(2).mean())


def with_fixup_rows(build, n_rows_small):
    """The same workload on fewer rows (the fix-up passes are fp32 VALU kernels), every tenth of them times 1e4."""
    def make():
        global rows
        keep, rows = rows, [n_rows_small] * 3
        try:
            model, opt, loss_fn = build()
        finally:
            rows = keep
        x = loss_fn.__closure__ and next(c.cell_contents for c in loss_fn.__closure__
                                         if isinstance(c.cell_contents, torch.Tensor) and c.cell_contents.dim() == 2)
        x[::10] *= 1e4
        # (AffineHalfFlow's exp(s) overflows on such rows with nn.Linear-scale last layers: shrink those, the rows still
        #  leave the split range in the hidden layers)
        with torch.no_grad():
            for f in getattr(model, "flows", []):
                if isinstance(f, amd.AffineHalfFlow):
                    for net in (f.s_net, f.t_net):
                        last = [m for m in net.modules() if isinstance(m, torch.nn.Linear)][-1]
                        last.weight *= 1e-5
                        last.bias *= 1e-5
        return model, opt, loss_fn
    return make


def trajectory(build):
    torch.manual_seed(0)  # (the host-drawn seeds of the in-kernel masks / noise)
    model, opt, loss_fn = build()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    return [p.detach().clone() for p in model.parameters()], float(loss), dt


def first_gradients(build):
    """the gradients of the FIRST step (before any update mixes them): which parameter's sum differs, by name"""
    torch.manual_seed(0)
    model, opt, loss_fn = build()
    opt.zero_grad()
    loss_fn().backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


bad_total = 0
fix_rows = int(os.environ.get("SOAK_FIX_ROWS", "16384"))
for name, build in (("c2t", run_c2t), ("c4t", run_c4t), ("c3t", run_c3t), ("c5t", run_c5t),
                    ("c2t_fix", with_fixup_rows(run_c2t, fix_rows)), ("c3t_fix", with_fixup_rows(run_c3t, fix_rows)),
                    ("c5t_fix", with_fixup_rows(run_c5t, fix_rows))):
    a, la, ta = trajectory(build)
    b, lb, tb = trajectory(build)
    bad = sum(int(not torch.equal(p, q)) for p, q in zip(a, b))
    worst = max(float((p - q).abs().max() / (p.abs().max() + 1e-30)) for p, q in zip(a, b))
    print(f"{name}: {steps} Adam steps twice: {bad} of {len(a)} parameter tensors differ (worst relative difference {worst:.2e}); "
          f"final loss {la:.6f} / {lb:.6f}; {min(ta, tb):.3f} ms per step; MNF_DETERMINISTIC={os.environ.get('MNF_DETERMINISTIC', '0')}")
    bad_total += bad
    ga, gb = first_gradients(build), first_gradients(build)
    diff = [n for n in ga if not torch.equal(ga[n], gb[n])]
    if diff:
        print(f"   first-step gradients that differ between two runs: {', '.join(diff[:12])}{' ...' if len(diff) > 12 else ''}")
sys.exit(1 if bad_total else 0)
