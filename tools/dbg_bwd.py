import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, recipes
import torch_mnf_amd as amd
from oracle import flow_oracle as O
def nerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
for dim, hs, rows, inverse in [(64, (24,), 16, False), (64, (24,), 300, False), (64, (24, 24), 300, False), (64, (24,), 300, True)]:
    sd = recipes.affine_half_params(31 + dim, dim, h_sizes=hs, s_last_gain=2.0)
    x_cpu = recipes.gaussian(32 + dim, rows, dim)
    w_y, w_l = recipes.gaussian(33, rows, dim), recipes.gaussian(34, rows, 1)[:, 0]
    xx = x_cpu.double().requires_grad_(True)
    p = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    y, ld = O.affine_half(xx, p, False, inverse)
    ((y * w_y.double()).sum() + (ld * w_l.double()).sum()).backward()
    f = amd.AffineHalfFlow(dim, False, h_sizes=hs); f.load_state_dict(sd); f.to("cuda"); f.force_generic = 2
    x = x_cpu.cuda().requires_grad_(True)
    yg, ldg = f.forward(x, inverse=inverse)
    ((yg * w_y.cuda()).sum() + (ldg * w_l.cuda()).sum()).backward()
    print(dim, hs, rows, inverse, amd.last_kernel())
    print("   x cond half", nerr(x.grad[:, :dim//2], xx.grad[:, :dim//2]), " x act half", nerr(x.grad[:, dim//2:], xx.grad[:, dim//2:]))
    for n, q in f.named_parameters():
        g, r = q.grad.double().cpu(), p[n].grad
        print(f"   {n:18s} err {nerr(q.grad, p[n].grad):.3e}  ratio(median) {float((g / r).flatten().median()):.4f}  |ref| {float(r.abs().max()):.3e}")
    if rows == 16:
        for n in ("s_net.2.weight",):
            g, r = dict(f.named_parameters())[n].grad.double().cpu(), p[n].grad
            e = ((g - r).abs() / r.abs().max())
            print(n, tuple(g.shape))
            for o0 in range(0, g.shape[0], 4):
                print("   rows %2d-%2d:" % (o0, o0 + 3), " ".join("%7.1e" % float(e[o0:o0+4, k0:k0+4].max()) for k0 in range(0, g.shape[1], 4)))

        if os.environ.get("MNF_RT_DBG"):
            gw = dict(f.named_parameters())["s_net.2.weight"].grad.cpu(); gb = dict(f.named_parameters())["s_net.2.bias"].grad.cpu()
            print("dW_out rows 0..3:", gw[:4, :8].tolist()); print("db_out 0..3:", gb[:4].tolist())
        gw = dict(f.named_parameters())["s_net.2.weight"].grad.cpu(); r = p["s_net.2.weight"].grad
        print("GPU dW_out[0,:8]", [round(v, 4) for v in gw[0, :8].tolist()]); print("ref dW_out[0,:8]", [round(v, 4) for v in r[0, :8].tolist()])
        print("GPU dW_out[:8,0]", [round(v, 4) for v in gw[:8, 0].tolist()]); print("ref dW_out[:8,0]", [round(v, 4) for v in r[:8, 0].tolist()])
