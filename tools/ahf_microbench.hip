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
  const int acc = WAVES_CAP ? 0 : 1;  // (template slot reused: 1 = run with accumulate = 0)
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((ahf_mfma_kernel<32, 24, true, PF, ABL>), dim3(blocks), dim3(ahf_waves<32>() * 64), 0, 0, x, y, ld,
                       nullptr, img, rows, 0, acc);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((ahf_mfma_kernel<32, 24, true, PF, ABL>), dim3(blocks), dim3(ahf_waves<32>() * 64), 0, 0, x, y, ld,
                       nullptr, img, rows, 0, acc);
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
  // interleaved rounds in one process (rule: never rank variants from separate or one-shot runs)
  struct V { const char* name; float (*fn)(const float*, float*, float*, const float*, int64_t, int, int); };
  const V vs[] = {
      {"full", run<false, 0, 0>},          {"full+prefetch", run<true, 0, 0>},
      {"accumulate=0", run<false, 0, 1>},  {"no-logdet-rmw", run<false, 7, 0>},
      {"no-cnd-store", run<false, 8, 0>},  {"no-act-store", run<false, 9, 0>},
      {"no-stores", run<false, 6, 0>},     {"compute-only", run<false, 2, 0>},
      {"copy-only", run<false, 5, 0>},     {"no-mfma", run<false, 1, 0>},
      {"no-exp", run<false, 3, 0>},        {"no-lds-reads", run<false, 4, 0>},
      {"wide-stores", run<false, 12, 0>},   {"nt-stores", run<false, 13, 0>},
      {"nt-loads+stores", run<false, 14, 0>}, {"nt-loads", run<false, 15, 0>},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  for (int bpc : {4, 6}) {
    float best[32], sum[32];
    for (int v = 0; v < nv; ++v) { best[v] = 1e9f; sum[v] = 0.f; }
    const int rounds = 5;
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < nv; ++v) {
        const float t = vs[v].fn(x, y, ld, img, rows, bpc, 10);
        best[v] = t < best[v] ? t : best[v];
        sum[v] += t;
      }
    printf("== %d workgroups/CU (us per launch: min / mean of %d interleaved rounds)\n", bpc, rounds);
    for (int v = 0; v < nv; ++v) printf("  %-16s %7.1f / %7.1f\n", vs[v].name, best[v], sum[v] / rounds);
  }
  return 0;
}
