import torch, time
n = 9 * (1 << 20) * 64
buf = torch.empty(n, device="cuda")
src = torch.randn(1 << 26, device="cuda")
def t(fn, k=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
w = t(lambda: buf.fill_(1.0))
print(f"fill {n*4/1e9:.2f} GB: {w*1e6:.0f} us -> {n*4/w/1e12:.2f} TB/s")
w = t(lambda: buf.zero_())
print(f"zero {n*4/1e9:.2f} GB: {w*1e6:.0f} us -> {n*4/w/1e12:.2f} TB/s")
dst = torch.empty_like(src)
w = t(lambda: dst.copy_(src))
print(f"copy {src.numel()*4/1e9:.2f} GB each way: {w*1e6:.0f} us -> {2*src.numel()*4/w/1e12:.2f} TB/s")
w = t(lambda: torch.sum(buf))
print(f"read {n*4/1e9:.2f} GB: {w*1e6:.0f} us -> {n*4/w/1e12:.2f} TB/s")
