"""Config 5 harness: one MNF-LeNet forward on 512 images x 500 MC repeats (256,000 rows).

MNFLeNet is a caller, not the path: convolution and pooling run on stock
PyTorch-ROCm here (torch_mnf_amd.MNFConv2d: the reference's layer with its flows on the HIP kernels), the
multiplicative-noise flows (MNFLinear.sample_z at (R, 800) and (R, 50); MNFConv2d's at (1, 20) / (1, 50)) and
MNFLinear.forward run in libmnf_hip.so.
The point of the measurement: how much of the end-to-end forward the flow path still is.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn
import torch_mnf_amd as amd


def main():
    images, repeats = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 500
    dev = "cuda"
    torch.manual_seed(0)
    net = nn.Sequential(amd.MNFConv2d(1, 20, 5), nn.ReLU(), nn.MaxPool2d(2), amd.MNFConv2d(20, 50, 5), nn.ReLU(),
                        nn.MaxPool2d(2), nn.Flatten(), amd.MNFLinear(800, 50), nn.ReLU(), amd.MNFLinear(50, 10),
                        nn.LogSoftmax(dim=-1)).to(dev)
    x = torch.rand(images, 1, 28, 28, device=dev).repeat_interleave(repeats, dim=0)  # each image `repeats` times
    chunk = 32000  # rows per pass through the convolutions (activation memory)
    def forward():
        outs = []
        for r0 in range(0, x.shape[0], chunk):
            outs.append(net(x[r0:r0 + chunk]))
        return torch.cat(outs)
    with torch.no_grad():
        out = forward(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = forward(); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        lin1, lin2 = net[7], net[9]
        t0 = time.perf_counter()
        for r0 in range(0, x.shape[0], chunk):
            n = min(chunk, x.shape[0] - r0)
            lin1.sample_z(n); lin2.sample_z(n)
        torch.cuda.synchronize(); t_flow = time.perf_counter() - t0
    assert out.shape == (images * repeats, 10) and torch.isfinite(out).all()
    probs = out.exp().view(images, repeats, 10).mean(1)
    print(f"MNF-LeNet forward, {images} images x {repeats} MC repeats = {x.shape[0]} rows: {t_all*1e3:.1f} ms end to end "
          f"({x.shape[0]/t_all:.3e} rows/s); of which MNFLinear flows (sample_z, HIP path) {t_flow*1e3:.2f} ms = "
          f"{100*t_flow/t_all:.1f} %; predictive entropy mean {float(-(probs*probs.clamp_min(1e-12).log()).sum(1).mean()):.3f}")

main()
