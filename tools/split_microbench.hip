// Ablation microbenchmark of the split AffineHalfFlow kernel (d = 64): which of {f16 MFMAs, operand
// splitting, HBM traffic, range guard, prefetch} bounds it.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize \
//       tools/split_microbench.hip -o /tmp/split_mb && /tmp/split_mb
#include "../torch_mnf_amd/csrc/mnf_ahf_split.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace mnf {
thread_local int g_last_hip_error = 0;
int check_launch() { return hipGetLastError() == hipSuccess ? 0 : -3; }
int64_t fill_net(NetDesc& nd, int n_sizes, const int* sizes, int64_t base) {
  nd.n_lin = n_sizes - 1;
  int64_t off = base;
  for (int i = 0; i < n_sizes; ++i) nd.sizes[i] = sizes[i];
  for (int l = 0; l < nd.n_lin; ++l) {
    nd.w_off[l] = (int)off; off += (int64_t)sizes[l] * sizes[l + 1];
    nd.b_off[l] = (int)off; off += sizes[l + 1];
  }
  return off - base;
}
bool hidden_ok(int, const int*) { return true; }
}  // namespace mnf

using namespace mnf;

template <int ABL>
static float run(const float* x, float* y, float* ld, const uint32_t* simg, const float* img, int64_t rows,
                 int blocks_per_cu, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((ahf_split_kernel<32, 24, true, ABL>), dim3(blocks), dim3(kSplitWaves * 64), 0, 0, x, y, ld,
                       nullptr, simg, img, rows, 0, 1);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((ahf_split_kernel<32, 24, true, ABL>), dim3(blocks), dim3(kSplitWaves * 64), 0, 0, x, y, ld,
                       nullptr, simg, img, rows, 0, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}

int main() {
  const int64_t rows = 1 << 20;
  const int dim = 64;
  float *x, *y, *ld, *img;
  uint32_t* simg;
  hipMalloc(&x, rows * dim * 4); hipMalloc(&y, rows * dim * 4); hipMalloc(&ld, rows * 4);
  const int nimg = AhfShape<32, 24>::IMAGE_FLOATS, nsimg = SplitShape<32, 24>::IMAGE_WORDS;
  hipMalloc(&img, nimg * 4); hipMalloc(&simg, nsimg * 4);
  std::vector<float> h(rows * dim), hi(nimg);
  std::vector<_Float16> hs(nsimg * 2);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  for (auto& v : hi) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.3f);
  for (int i = 0; i < 2 * (SplitShape<32, 24>::PLAIN_WORDS + kSplitTailWords); ++i) hs[hs.size() - 1 - i] = (_Float16)0.f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(simg, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
  hipMemset(ld, 0, rows * 4);
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ahf_split_kernel<32, 24, true, 0>, kSplitWaves * 64, 0);
  printf("resident workgroups/CU (full kernel): %d\n", per_cu);
  // interleaved rounds in one process (rule: never rank variants from separate or one-shot runs)
  struct V { const char* name; float (*fn)(const float*, float*, float*, const uint32_t*, const float*, int64_t, int, int); };
  const V vs[] = {
      {"full", run<0>},          {"no-mfma", run<1>},   {"compute-only", run<2>}, {"no-split-valu", run<3>},
      {"no-guard", run<5>},      {"copy-only", run<6>}, {"no-prefetch", run<7>},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  for (int bpc : {2, 4, 8, 16, 32}) {
    float best[32], sum[32];
    for (int v = 0; v < nv; ++v) { best[v] = 1e9f; sum[v] = 0.f; }
    const int rounds = 5;
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < nv; ++v) {
        const float t = vs[v].fn(x, y, ld, simg, img, rows, bpc, 10);
        best[v] = t < best[v] ? t : best[v];
        sum[v] += t;
      }
    printf("== %d workgroups/CU (us per launch: min / mean of %d interleaved rounds)\n", bpc, rounds);
    for (int v = 0; v < nv; ++v) printf("  %-16s %7.1f / %7.1f\n", vs[v].name, best[v], sum[v] / rounds);
  }
  return 0;
}
