// Does the 16-rows x 64-B access shape of the MFMA B-operand layout cost HBM bandwidth against a
// fully contiguous 1-KiB-per-instruction copy?  (d = 64 rows of 256 B, 2^20 rows, out-of-place.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) copy_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ ld, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  for (int tile = blockIdx.x * WAVES + wave; tile < n_tiles; tile += gridDim.x * WAVES) {
    f32x4 v[4];
    if (MODE == 0) {  // operand layout: lane (j,q): row j, float4s at 16g + 4q
      const float* xr = x + ((size_t)tile * 16 + j) * 64 + 4 * q;
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(xr + 16 * g);
      float* yr = y + ((size_t)tile * 16 + j) * 64 + 4 * q;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(yr + 16 * g) = v[g];
    } else {  // contiguous: instruction c covers rows 4c..4c+3 entirely
      const float* xr = x + (size_t)tile * 1024 + lane * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const f32x4*>(xr + 256 * c);
      float* yr = y + (size_t)tile * 1024 + lane * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(yr + 256 * c) = v[c];
    }
    if (MODE != 2 && q == 0) ld[(size_t)tile * 16 + j] += v[0][0];
  }
}

template <int MODE, int WAVES>
float run(const float* x, float* y, float* ld, int n_tiles, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, x, y, ld, n_tiles);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((copy_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, x, y, ld, n_tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 50.f;
}

int main() {
  const size_t rows = 1 << 20;
  float *x, *y, *ld;
  hipMalloc(&x, rows * 256); hipMalloc(&y, rows * 256); hipMalloc(&ld, rows * 4);
  hipMemset(x, 0, rows * 256); hipMemset(ld, 0, rows * 4);
  const int n_tiles = rows / 16;
  for (int bpc : {2, 4, 6, 8, 16, 64}) {
    const int blocks = bpc == 64 ? n_tiles / 4 : 256 * bpc;
    printf("blocks/CU %2d: operand-layout %.1f us | contiguous %.1f us | contiguous no-logdet %.1f us\n", bpc,
           run<0, 4>(x, y, ld, n_tiles, blocks), run<1, 4>(x, y, ld, n_tiles, blocks), run<2, 4>(x, y, ld, n_tiles, blocks));
  }
  return 0;
}
