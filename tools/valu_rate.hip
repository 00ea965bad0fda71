// Vector-instruction throughput on gfx950: v_fma_f32 vs v_pk_fma_f32 (two fp32 lanes' worth per instruction) and
// v_exp_f32, at one and two waves per SIMD (hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/bin/valu_rate).
// Answers whether packing the spline's arithmetic into v_pk_* instructions can raise a VALU-bound kernel's rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
  float a[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; }
  const float c = 0.999f, d = 1e-3f;
  const f32x2 c2 = {0.999f, 0.998f}, d2 = {1e-3f, 2e-3f};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
  const int iters = 20000;
  const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_pk_mul_f32"};
  for (int waves = 1; waves <= 4; waves *= 2)
    for (int kind = 0; kind < 4; ++kind) {
      long long c = 0;
      for (int rep = 0; rep < 2; ++rep) {
        dim3 g(256), b(256 * waves);
        if (kind == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc, iters);
        if (kind == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc, iters);
        if (kind == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc, iters);
        if (kind == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc, iters);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
      }
      if (hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
      printf("%d wave(s)/SIMD  %-13s %.2f ticks per instruction per wave (8 independent chains)\n", waves, names[kind],
             (double)c / (iters * 8.0));
    }
  return 0;
}
