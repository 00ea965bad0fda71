"""The streaming split RNVP forward kernel (explicit masks; shapes without a register-resident kernel) at 256,000 rows:
`python3 tools/time_rnvp_streaming.py` (A/B of MNF_RNVP_SPLIT_OCC variants through MNF_LIB_PATH)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch_mnf_amd as amd

R = 256000
for dim, explicit in ((800, True), (784, False), (784, True), (100, False)):
    f = amd.RNVP(dim, h_sizes=(50,)).to("cuda")
    z = torch.randn(R, dim, device="cuda")
    mask = (torch.rand(R, dim, device="cuda") < 0.5).float() if explicit else None
    with torch.no_grad():
        for _ in range(5):
            f.forward(z, mask=mask, seed=None if explicit else 3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f.forward(z, mask=mask, seed=None if explicit else 3)
        e1.record(); torch.cuda.synchronize()
    print(f"d={dim} {'explicit mask' if explicit else 'in-kernel mask'}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us", flush=True)
