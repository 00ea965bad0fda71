// Throughput of float atomics from every CU onto one shared parameter-gradient vector (what the run-time-shaped gradient
// kernels do once per row block): n_params addresses, `rounds` additions to each from each of the 256 workgroups.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o tools/micro/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: a wave-instruction covers 64 consecutive floats; MODE 1: 4 runs of 16 consecutive floats `stride` apart (a
// 16 x 16 weight-gradient tile: lane (i, q) register r -> row 4 q + r, column i)
template <int MODE>
__global__ void __launch_bounds__(512) k(float* g, int n_params, int rounds, int stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  for (int it = 0; it < rounds; ++it) {
    if (MODE == 0) {
      for (int p = threadIdx.x; p < n_params; p += blockDim.x) atomicAdd(g + p, 1.f);
    } else {
      const int tiles = n_params / 256;
      for (int t = wave; t < tiles; t += 8) {
        const int tr = t / (stride / 16), tc = t % (stride / 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(g + (size_t)(16 * tr + 4 * q + r) * stride + 16 * tc + i, 1.f);
      }
    }
  }
}

template <int MODE>
void run(const char* name, int blocks, int n_params, int rounds, int stride, int replicas) {
  float* g;
  (void)hipMalloc(&g, (size_t)n_params * 4 * replicas + 4096);
  (void)hipMemset(g, 0, (size_t)n_params * 4 * replicas);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, g, n_params, 1, stride);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, g, n_params, rounds, stride);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double n = (double)blocks * rounds * n_params;
  printf("%-34s %4d workgroups x %3d rounds x %6d addresses: %8.3f ms, %6.1f G atomics/s, %6.2f us per round\n", name, blocks,
         rounds, n_params, ms, n / ms * 1e-6, ms * 1e3 / rounds);
  (void)hipFree(g);
}

int main() {
  for (int n : {4096, 24576, 131072}) {
    run<0>("64 consecutive floats / instr", 256, n, 8, 64, 1);
    run<1>("16 x 16 tiles, row stride 64", 256, n, 8, 64, 1);
  }
  run<0>("one workgroup alone", 1, 24576, 64, 64, 1);
  run<0>("32 workgroups", 32, 24576, 16, 64, 1);
  return 0;
}
