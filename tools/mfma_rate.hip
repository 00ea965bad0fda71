// Cycles per MFMA, one wave per SIMD, independent accumulators: v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16
// (hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x4 a4 = {(_Float16)(threadIdx.x * 0.001f), 1, 2, 3}, b4 = {1, (_Float16)0.5f, 2, 1};
  f16x8 a8 = {1, 2, 3, 4, 5, 6, 7, (_Float16)(threadIdx.x * 0.001f)}, b8 = {1, 1, 2, 2, 1, 1, 2, 2};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int kind : {16, 32}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("16x16x%d f16: %.2f cycles per MFMA (one wave per SIMD, 8 independent accumulators)\n", kind, (double)c / (iters * 8.0));
  }
  return 0;
}
