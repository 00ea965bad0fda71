// Issue rate of packed fp32 VALU arithmetic on gfx950 in a pure vector stream (no MFMA beside it): the question behind
// packing the two spline axes of NSF_CL's forward kernel.  One wave per SIMD and four, dependent chains of 8.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 0.9999f, c = 1e-4f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 v = {a[i], a[i + 1]};
        const f32x2 mm = {m, m}, cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
        a[i] = v[0]; a[i + 1] = v[1];
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 v = {a[i], a[i + 1]};
        const f32x2 cc = {c, c};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(cc));
        a[i] = v[0]; a[i + 1] = v[1];
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (MODE == 5) {  // 8 exp + 8 fma interleaved
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i + 1]) : "v"(m), "v"(c));
      }
    } else if (MODE == 6) {  // 8 exp + 4 pk_fma interleaved
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i + 1]));
        f32x2 v = {a[i + 2], a[i + 3]};
        const f32x2 mm = {m, m}, cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
        a[i + 2] = v[0]; a[i + 3] = v[1];
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int n_instr_per_iter, int elems) {
  float* out;
  hipMalloc(&out, 256 * 1024 * 16 * 4);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD of a CU
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  // cycles per instruction per SIMD at 2.4 GHz: time * clock / (iters * instr * waves_per_simd)
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * n_instr_per_iter * waves_per_simd);
  printf("%-28s %d wave(s)/SIMD: %.2f ms, %.2f cycles per instruction (2.4 GHz assumed), %.2f cycles per fp32 result\n", name,
         waves_per_simd, ms, cyc, cyc * n_instr_per_iter / elems);
  hipFree(out);
}

int main() {
  for (int w : {1, 4}) {
    run<0>("v_fma_f32 x16", w, 16, 16);
    run<1>("v_pk_fma_f32 x8", w, 8, 16);
    run<2>("v_add_f32 x16", w, 16, 16);
    run<3>("v_pk_add_f32 x8", w, 8, 16);
    run<4>("v_exp_f32 x16", w, 16, 16);
    run<5>("8 v_exp + 8 v_fma", w, 16, 16);
    run<6>("8 v_exp + 4 v_pk_fma", w, 12, 16);
  }
  return 0;
}
