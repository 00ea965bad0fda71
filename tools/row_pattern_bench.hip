// How fast can 16-row tiles of a (rows x 800) fp32 matrix be streamed with the MFMA operand access shape
// (lane (j, q): row j, 16 B at 64 g + 16 q -> 16 rows x 64 B per instruction, rows 3,200 B apart) against
// row-contiguous shapes?  Read-only sweep (GEMM-1 like) and read+write sweep (GEMM-2 like).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: operand shape, G groups per step.  MODE 1: one row per instruction quarter: lane l -> row l / 16 ... see below
template <int MODE, bool WRITE>
__global__ void __launch_bounds__(512) sweep(const float* __restrict__ z, float* __restrict__ x, float* __restrict__ sink,
                                             int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n_groups = rows / 128;
  f32x4 acc = {0, 0, 0, 0};
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const size_t row0 = (size_t)grp * 128 + wave * 16;
    if (MODE == 0) {  // 16 rows x 64 B per instruction
      const float* zr = z + (row0 + j) * d + 4 * q;
      float* xr = x + (row0 + j) * d + 4 * q;
      for (int g = 0; g < d / 16; g += 4) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(zr + 16 * (g + i < d / 16 ? g + i : 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc += v[i];
          if (WRITE && g + i < d / 16) *reinterpret_cast<f32x4*>(xr + 16 * (g + i)) = v[i] * 2.f;
        }
      }
    } else if (MODE == 1) {  // 4 rows x 256 B per instruction (lane l: row l / 16, 16 B at 16 (l % 16))
      const int r = lane >> 4, c = lane & 15;
      for (int col = 0; col < d; col += 64) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cc = col + 4 * c < d ? col + 4 * c : 0;
          v[i] = *reinterpret_cast<const f32x4*>(z + (row0 + 4 * i + r) * d + cc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc += v[i];
          if (WRITE && col + 4 * c < d) *reinterpret_cast<f32x4*>(x + (row0 + 4 * i + r) * d + col + 4 * c) = v[i] * 2.f;
        }
      }
    } else {  // 1 row x 1 KiB per instruction
      for (int rr = 0; rr < 16; rr += 4) {
        for (int col = 0; col < d; col += 256) {
          f32x4 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cc = col + 4 * lane < d ? col + 4 * lane : 0;
            v[i] = *reinterpret_cast<const f32x4*>(z + (row0 + rr + i) * d + cc);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc += v[i];
            if (WRITE && col + 4 * lane < d) *reinterpret_cast<f32x4*>(x + (row0 + rr + i) * d + col + 4 * lane) = v[i] * 2.f;
          }
        }
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) sink[0] = acc[0];
}

template <int MODE, bool WRITE>
float run(const float* z, float* x, float* sink, int rows, int d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((sweep<MODE, WRITE>), dim3(blocks), dim3(512), 0, 0, z, x, sink, rows, d);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((sweep<MODE, WRITE>), dim3(blocks), dim3(512), 0, 0, z, x, sink, rows, d);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.f;
}

int main() {
  const int rows = 256000, d = 800;
  float *z, *x, *sink;
  hipMalloc(&z, (size_t)rows * d * 4); hipMalloc(&x, (size_t)rows * d * 4); hipMalloc(&sink, 4);
  hipMemset(z, 0, (size_t)rows * d * 4);
  const double gb = (double)rows * d * 4 / 1e9;
  for (int bpc : {1, 2, 4}) {
    const int blocks = 256 * bpc;
    float t;
    printf("blocks/CU %d\n", bpc);
    t = run<0, false>(z, x, sink, rows, d, blocks); printf("  operand shape   read       %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
    t = run<1, false>(z, x, sink, rows, d, blocks); printf("  4 rows x 256 B  read       %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
    t = run<2, false>(z, x, sink, rows, d, blocks); printf("  1 row x 1 KiB   read       %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
    t = run<0, true>(z, x, sink, rows, d, blocks); printf("  operand shape   read+write %7.1f us  %.2f TB/s\n", t, 2 * gb / t * 1e3);
    t = run<1, true>(z, x, sink, rows, d, blocks); printf("  4 rows x 256 B  read+write %7.1f us  %.2f TB/s\n", t, 2 * gb / t * 1e3);
    t = run<2, true>(z, x, sink, rows, d, blocks); printf("  1 row x 1 KiB   read+write %7.1f us  %.2f TB/s\n", t, 2 * gb / t * 1e3);
  }
  return 0;
}
