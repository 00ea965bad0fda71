// y = x @ W for Glow's d x d matrix on the fp32 matrix cores (d in {16, 32, 64, 128}).
//
// Same transposed scheme as the coupling kernels: one wave owns 16 rows; lane (j, q) loads the
// row as float4s (element 16 g + 4 q + e is the k = q operand of K-step 4 g + e); output tile m
// leaves dims 16 m + 4 q + r in register r of lane (j, q), i.e. a float4 of the output row.
// W is pre-arranged into A-operand order (image) and copied to LDS once per workgroup.
// HBM-bound: 8 d bytes per row, 2 d^2 flops per row (d = 32: 8 flop/B).
#include <hip/hip_runtime.h>

#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLinWaves = 4;

template <int D>
__global__ void __launch_bounds__(kLinWaves * 64)
linear_rows_mfma_kernel(const float* __restrict__ x, const float* __restrict__ image, float* __restrict__ y,
                        int64_t rows) {
  constexpr int G = D / 16, NK = D / 4;
  __shared__ __attribute__((aligned(16))) float lds[D * D];
  {
    const float4* src = reinterpret_cast<const float4*>(image);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < D * D / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n_tiles = (int)((rows + 15) >> 4);
  for (int tile = (int)blockIdx.x * kLinWaves + wave; tile < n_tiles; tile += (int)gridDim.x * kLinWaves) {
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* xr = x + rowc * D + 4 * q;
    f32x4 xv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) xv[g] = *reinterpret_cast<const f32x4*>(xr + 16 * g);
    int a_off = lane * 4;
    asm volatile("" : "+v"(a_off));  // keep the operand reads in the loop (see mnf_ahf_mfma.hip)
    const f32x4* A4 = reinterpret_cast<const f32x4*>(lds + a_off);
    f32x4 acc[G];
#pragma unroll
    for (int m = 0; m < G; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    int n = 0;
    f32x4 a4;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
      for (int m = 0; m < G; ++m) {
        if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], xv[kk >> 2][kk & 3], acc[m], 0, 0, 0);
        ++n;
      }
    if (live) {
      float* yr = y + rowc * D + 4 * q;
#pragma unroll
      for (int m = 0; m < G; ++m) *reinterpret_cast<f32x4*>(yr + 16 * m) = acc[m];
    }
  }
}

template <int D>
static void build_index(int32_t* idx) {
  constexpr int G = D / 16, NK = D / 4;
  int n = 0;
  for (int kk = 0; kk < NK; ++kk) {
    const int g = kk >> 2, e = kk & 3;
    for (int m = 0; m < G; ++m, ++n)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4;
        idx[(n >> 2) * 256 + lane * 4 + (n & 3)] = (16 * g + 4 * kq + e) * D + 16 * m + i;  // W[k][out]
      }
  }
}

template <int D>
static int launch(const float* x, const float* image, float* y, int64_t rows, hipStream_t stream) {
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + kLinWaves - 1) / kLinWaves;
  if (blocks > 256 * 8) blocks = 256 * 8;
  tag_kernel("linear_rows_mfma");
  hipLaunchKernelGGL((linear_rows_mfma_kernel<D>), dim3((unsigned)blocks), dim3(kLinWaves * 64), 0, stream, x,
                     image, y, rows);
  return check_launch();
}

}  // namespace mnf

extern "C" {

int64_t mnf_linear_rows_image_floats(int dim) {
  return (dim == 16 || dim == 32 || dim == 64 || dim == 128) ? (int64_t)dim * dim : 0;
}

int mnf_linear_rows_image_index(int dim, int32_t* idx_host) {
  if (!idx_host) return MNF_ERR_INVALID_ARG;
  switch (dim) {
    case 16: mnf::build_index<16>(idx_host); return MNF_OK;
    case 32: mnf::build_index<32>(idx_host); return MNF_OK;
    case 64: mnf::build_index<64>(idx_host); return MNF_OK;
    case 128: mnf::build_index<128>(idx_host); return MNF_OK;
  }
  return MNF_ERR_UNSUPPORTED;
}

int mnf_linear_rows_img(const float* x, const float* image, float* y, int64_t rows, int dim, void* stream) {
  if (!x || !image || !y || x == y || rows < 0) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(image)) & 15)
    return MNF_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (dim) {
    case 16: return mnf::launch<16>(x, image, y, rows, st);
    case 32: return mnf::launch<32>(x, image, y, rows, st);
    case 64: return mnf::launch<64>(x, image, y, rows, st);
    case 128: return mnf::launch<128>(x, image, y, rows, st);
  }
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
