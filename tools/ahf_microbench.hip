// Ablation microbenchmark of the AffineHalfFlow MFMA kernel (d = 64): which of {MFMA chain,
// HBM traffic, exp/divide, LDS operand reads} bounds it.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ahf_microbench.hip -o /tmp/ahf_mb && /tmp/ahf_mb
#include "../torch_mnf_amd/csrc/mnf_ahf_mfma.hip"

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

template <bool PF, int ABL, int WAVES_CAP>
static float run(const float* x, float* y, float* ld, const float* img, int64_t rows, int blocks_per_cu, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((ahf_mfma_kernel<32, 24, true, PF, ABL>), dim3(blocks), dim3(kAhfWaves * 64), 0, 0, x, y, ld,
                       nullptr, img, rows, 0, 1);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((ahf_mfma_kernel<32, 24, true, PF, ABL>), dim3(blocks), dim3(kAhfWaves * 64), 0, 0, x, y, ld,
                       nullptr, img, rows, 0, 1);
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
  hipMalloc(&x, rows * dim * 4); hipMalloc(&y, rows * dim * 4); hipMalloc(&ld, rows * 4);
  const int nimg = AhfShape<32, 24>::IMAGE_FLOATS;
  hipMalloc(&img, nimg * 4);
  std::vector<float> h(rows * dim), hi(nimg);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  for (auto& v : hi) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
  hipMemset(ld, 0, rows * 4);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ahf_mfma_kernel<32, 24, true, true, 0>, kAhfWaves * 64, 0);
  printf("occupancy (prefetch build): %d blocks/CU of %d waves\n", occ, kAhfWaves);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ahf_mfma_kernel<32, 24, true, false, 0>, kAhfWaves * 64, 0);
  printf("occupancy (no-prefetch build): %d blocks/CU\n", occ);
  for (int bpc : {2, 3, 4, 5, 6, 8}) {
    printf("blocks/CU %d | full pf %.1f  nopf %.1f | noMFMA pf %.1f nopf %.1f | noHBM pf %.1f nopf %.1f | noExpDiv pf %.1f | noLDS pf %.1f  [us]\n", bpc,
           run<true, 0, 0>(x, y, ld, img, rows, bpc, 20), run<false, 0, 0>(x, y, ld, img, rows, bpc, 20),
           run<true, 1, 0>(x, y, ld, img, rows, bpc, 20), run<false, 1, 0>(x, y, ld, img, rows, bpc, 20),
           run<true, 2, 0>(x, y, ld, img, rows, bpc, 20), run<false, 2, 0>(x, y, ld, img, rows, bpc, 20),
           run<true, 3, 0>(x, y, ld, img, rows, bpc, 20), run<true, 4, 0>(x, y, ld, img, rows, bpc, 20));
  }
  return 0;
}
