// Write-only bandwidth of the store shapes a 16-row MFMA tile can produce, on the C4 intermediates
// (9 tensors of 2^19 x 256 fp32 = 4.8 GB per pass):
//   0  linear: a wave instruction writes 1 KB contiguous (the plain-fill ceiling)
//   1  tile64: lane (j, q) writes 16 B at row j, column 16 g + 4 q   -> 16 rows x 64 B per instruction
//   2  tile128: after a lane exchange, 8 rows x 128 B per instruction
//   3  tile256: 4 rows x 256 B per instruction
// build: hipcc -O3 --offload-arch=gfx950 tools/write_pattern_bench.hip -o tools/bin/write_pattern_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int64_t rows, int dim, int n_tensors) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = rows / 16;
  const f32x4 v = {1.f + lane, 2.f, 3.f, 4.f};
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 8) {
    for (int t = 0; t < n_tensors; ++t) {
      float* base = out + ((int64_t)t * rows + tile * 16) * dim;
      const int G = dim / 16;
      if (MODE == 0) {
        for (int i = 0; i < G; ++i) *reinterpret_cast<f32x4*>(base + (i * 64 + lane) * 4) = v;
      } else if (MODE == 1) {
        const int j = lane & 15, q = lane >> 4;
        for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(base + j * dim + 16 * g + 4 * q) = v;
      } else if (MODE == 2) {
        const int j = lane & 7, p = lane >> 3;  // 8 rows x 8 pieces of 16 B
        for (int g = 0; g < G; g += 2)
          for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(base + (j + 8 * h) * dim + 16 * g + 4 * p) = v;
      } else {
        const int j = lane & 3, p = lane >> 2;  // 4 rows x 16 pieces
        for (int g = 0; g < G; g += 4)
          for (int h = 0; h < 4; ++h) *reinterpret_cast<f32x4*>(base + (j + 4 * h) * dim + 16 * g + 4 * p) = v;
      }
    }
  }
}

int main() {
  const int64_t rows = 1 << 19;
  const int dim = 256, nt = 9;
  float* out;
  const size_t bytes = (size_t)nt * rows * dim * 4;
  if (hipMalloc(&out, bytes) != hipSuccess) return 1;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 4; ++mode)
    for (int blocks : {256, 512, 1024, 2048}) {
      auto run = [&] {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, rows, dim, nt);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, rows, dim, nt);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, rows, dim, nt);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, rows, dim, nt);
      };
      for (int i = 0; i < 3; ++i) run();
      hipEventRecord(a);
      for (int i = 0; i < 10; ++i) run();
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d blocks %4d: %.1f us per pass -> %.2f TB/s\n", mode, blocks, ms * 100.f, bytes / (ms * 1e-4) / 1e12);
    }
  return 0;
}
