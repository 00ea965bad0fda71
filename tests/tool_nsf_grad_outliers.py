"""Where the NSF_CL row gradient kernel and the generic one disagree most, and whether the float64 oracle says the row is
ill-conditioned there (an element within rounding of a spline knot or a hidden pre-activation within rounding of the
LeakyReLU kink takes either one-sided derivative).  `python3 tools/nsf_grad_outliers.py [param_seed] [x_seed] [inverse]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import recipes
import torch_mnf_amd as amd
from oracle import flow_oracle as O

ps, xs, inverse = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
rows, K, n_h, T = 70001, 8, 8, 3.0
sd = recipes.nsf_cl_params(ps, 32, K, n_h)
x_cpu = recipes.gaussian(xs, rows, 32, scale=1.3)
w_y = recipes.gaussian(383, rows, 32)
w_l = recipes.gaussian(384, rows, 1)[:, 0]


def grads(generic):
    f = amd.NSF_CL(32, K=K, B=3, n_h=n_h)
    f.load_state_dict(sd); f.to("cuda"); f.force_generic = generic
    x = x_cpu.clone().to("cuda").requires_grad_(True)
    y, ld = (f.inverse if inverse else f.forward)(x)
    ((y * w_y.cuda()).sum() + (ld * w_l.cuda()).sum()).backward()
    return x.grad.cpu().double()


g, r = grads(False), grads(True)
xc = x_cpu.double().requires_grad_(True)
p = {k: v.double() for k, v in sd.items()}
y, ld = O.nsf_cl(xc, p, K, T, inverse)
((y * w_y.double()).sum() + (ld * w_l.double()).sum()).backward()
o = xc.grad
scale = o.abs().max().item()
print(f"max |grad_x| {scale:.3f}; normwise error rows kernel {(g - o).abs().max().item() / scale:.3e}, "
      f"generic kernel {(r - o).abs().max().item() / scale:.3e}")
bad = torch.unique(torch.topk((g - o).abs().flatten(), 8).indices // 32)
for row in bad.tolist():
    x1 = x_cpu[row:row + 1].double()
    lower, upper = x1[:, :16], x1[:, 16:]

    def report(cond, act, net):
        raw = O.mlp(cond, p, net).reshape(-1, 16, 3 * K - 1)
        W, H, _ = torch.split(raw, K, dim=2)
        _, xk = O._knots(2 * T * torch.softmax(W, 2), -T, T, 1e-3)
        _, yk = O._knots(2 * T * torch.softmax(H, 2), -T, T, 1e-3)
        knots = yk if inverse else xk
        h, pre_min = cond, 1e9
        for l in (0, 2, 4):
            pre = h @ p[f"{net}.{l}.weight"].T + p[f"{net}.{l}.bias"]
            pre_min = min(pre_min, pre.abs().min().item())
            h = torch.nn.functional.leaky_relu(pre, 0.2)
        return (act[..., None] - knots).abs().min().item(), pre_min

    if inverse:
        lo1, _ = O._nsf_half_step(upper, lower, p, "f2", K, T, True)
        a, b = report(upper, lower, "f2"), report(lo1, upper, "f1")
    else:
        up1, _ = O._nsf_half_step(lower, upper, p, "f1", K, T, False)
        a, b = report(lower, upper, "f1"), report(up1, lower, "f2")
    err = (g[row] - o[row]).abs().max().item() / scale
    print(f"row {row}: error {err:.2e}; first half-step: knot distance {a[0]:.2e}, |pre-activation| {a[1]:.2e}; "
          f"second: knot distance {b[0]:.2e}, |pre-activation| {b[1]:.2e}")
