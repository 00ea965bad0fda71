"""Debug aid: per-element error of the HIP spline against the G4 fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch_mnf_amd as amd
fx = dict(np.load(os.path.join(ROOT, "tests/golden/g4_rqs_direct.npz")))
for K in (5, 8):
    v, W, H, D = (torch.from_numpy(fx[f"K{K}.{n}"]).cuda() for n in "vWHD")
    for inv, name in ((False, "fwd"), (True, "inv")):
        out, lad = amd.rqs(v, W, H, D, inverse=inv, tail_bound=3.0)
        ro, rl = fx[f"K{K}.out_{name}"], fx[f"K{K}.lad_{name}"]
        eo = np.abs(out.cpu().numpy() - ro); el = np.abs(lad.cpu().numpy() - rl)
        eo[np.isnan(eo)] = 0; el[np.isnan(el)] = 0
        print(K, name, "out err", eo.max(), "at", eo.argmax(), "scale", np.nanmax(np.abs(ro)),
              "| lad err", el.max(), "at", el.argmax(), "scale", np.nanmax(np.abs(rl)))
        i = int(el.argmax())
        print("   worst lad: v", v[i].item(), "ref", rl[i], "got", lad[i].item(), "out ref", ro[i], "got", out[i].item())
        print("   # elements with lad err > 1e-5*scale:", int((el > 1e-5 * np.nanmax(np.abs(rl))).sum()), "of", len(el))
