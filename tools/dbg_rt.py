import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import torch, recipes
import torch_mnf_amd as amd
from oracle import flow_oracle as O
for dim, hs in [(6, (5, 9)), (2, (24, 24)), (50, (17, 30)), (8, (24,24)), (40, (24, 24))]:
    sd = recipes.affine_half_params(11 + dim, dim, h_sizes=hs, s_last_gain=3.0)
    x = recipes.gaussian(5, 3, dim)
    f = amd.AffineHalfFlow(dim, parity=False, h_sizes=hs); f.load_state_dict(sd); f.to("cuda"); f.force_generic = 2
    ry, rld = O.affine_half(x, sd, False, False)
    with torch.no_grad():
        y, ld = f.forward(x.cuda())
    print(dim, hs, amd.last_kernel())
    print(" y  ", y.cpu()[0, :8].tolist()); print(" ref", ry[0, :8].tolist()); print(" ld", ld.cpu().tolist(), rld.tolist())
    flat = f._packed3(torch.device("cuda"))[0]
    print(" wmax", float(flat.abs().max()), flat.numel())
