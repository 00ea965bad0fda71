import sys, cProfile, pstats, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import recipes, torch_mnf_amd as amd
flows = [amd.AffineHalfFlow(2, bool(i % 2)) for i in range(9)]
model = amd.NormalizingFlowModel(amd.StandardNormal(2), flows).to("cuda")
x = torch.randn(4096, 2, device="cuda")
with torch.no_grad():
    for _ in range(20): model.log_prob(x, return_sum=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): model.log_prob(x, return_sum=True)
    torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
