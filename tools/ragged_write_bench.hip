// Write traffic of a 16-row tile of a (rows, 50) fp32 array -- 200-byte rows -- by store shape:
//   0  scalar: lane (j, q) writes 4 separate dwords at row j, columns 16 g + 4 q + r (what row_store4<RAG> does when dim % 4 != 0)
//   1  pair:   the same as two 8-byte stores (rows of an even dim are 8-byte aligned)
//   2  linear: the tile's 3,200 contiguous bytes as 16-byte stores, lane after lane
// run under `rocprofv3 --pmc WRITE_SIZE` / `--pmc FETCH_SIZE`; prints the time per pass.
// build: hipcc -O3 --offload-arch=gfx950 tools/ragged_write_bench.hip -o tools/bin/ragged_write_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int64_t rows, int dim) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = rows / 16;
  const int j = lane & 15, q = lane >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 8) {
    float* base = out + tile * 16 * dim;
    if (MODE == 2) {
      for (int i = lane; i < 4 * dim; i += 64) *reinterpret_cast<f32x4*>(base + 4 * i) = f32x4{1.f, 2.f, 3.f, (float)i};
    } else {
      for (int g = 0; g < (dim + 15) / 16; ++g) {
        const int c = 16 * g + 4 * q;
        float* p = base + j * dim + c;
        if (MODE == 0) {
          for (int r = 0; r < 4; ++r)
            if (c + r < dim) p[r] = (float)(r + g);
        } else {
          if (c + 1 < dim) *reinterpret_cast<f32x2*>(p) = f32x2{1.f, (float)g};
          if (c + 3 < dim) *reinterpret_cast<f32x2*>(p + 2) = f32x2{3.f, (float)g};
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int64_t rows = 256000;
  const int dim = 50;
  float* buf;
  hipMalloc(&buf, rows * dim * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 12; ++rep) {
      if (rep == 2) hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(512), 0, 0, buf, rows, dim);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(512), dim3(512), 0, 0, buf, rows, dim);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(512), dim3(512), 0, 0, buf, rows, dim);
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("mode %d: %.1f us per pass, %.0f GB/s of the array's %.1f MB\n", mode, ms / 10 * 1e3,
           rows * dim * 4 / (ms / 10 * 1e-3) / 1e9, rows * dim * 4 / 1e6);
  }
  return 0;
}
