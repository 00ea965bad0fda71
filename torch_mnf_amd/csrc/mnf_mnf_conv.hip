// The elementwise parts of MNFConv2d.forward around its two convolutions (torch_mnf/layers/mnf_conv.py:67-88), each
// as one launch forward and one backward -- at the reference's batch size these are launches, not arithmetic:
//   operands   W_mean * z.view(-1, 1, 1, 1),  exp(W_log_var),  exp(b_log_var)          (:69-72: the conv weights / bias)
//   noise      mean + sqrt(var) * epsilon                                             (:86-88)
// The convolutions themselves are the caller's (MIOpen).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnf_hip.h"
#include "mnf_host.h"

namespace mnf {

__global__ void __launch_bounds__(256) conv_operands_kernel(const float* __restrict__ W_mean, const float* __restrict__ W_log_var,
                                                            const float* __restrict__ b_log_var, const float* __restrict__ z,
                                                            float* __restrict__ Wz, float* __restrict__ Wvar,
                                                            float* __restrict__ bvar, int n_out, int per_out) {
  const int64_t n = (int64_t)n_out * per_out, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    Wz[i] = W_mean[i] * z[i / per_out];
    Wvar[i] = expf(W_log_var[i]);
    if (i < n_out) bvar[i] = expf(b_log_var[i]);
  }
}

// a wave per output channel: gW_mean = gWz z[o];  gz[o] = sum_m gWz[o][m] W_mean[o][m];  gW_log_var = gWvar exp(W_log_var);
// lane 0 also gb_log_var[o] = gbvar[o] exp(b_log_var[o])
__global__ void __launch_bounds__(256) conv_operands_bwd_kernel(const float* __restrict__ W_mean, const float* __restrict__ W_log_var,
                                                                const float* __restrict__ b_log_var, const float* __restrict__ z,
                                                                const float* __restrict__ gWz, const float* __restrict__ gWvar,
                                                                const float* __restrict__ gbvar, float* __restrict__ gW_mean,
                                                                float* __restrict__ gW_log_var, float* __restrict__ gb_log_var,
                                                                float* __restrict__ gz, int n_out, int per_out, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = blockIdx.x * 4 + wave; o < n_out; o += gridDim.x * 4) {
    const float zo = z[o];
    float acc = 0.f;
    for (int m = lane; m < per_out; m += 64) {
      const int64_t i = (int64_t)o * per_out + m;
      const float g1 = gWz ? gWz[i] : 0.f, g2 = gWvar ? gWvar[i] : 0.f;
      const float a = g1 * zo, b = g2 * expf(W_log_var[i]);
      gW_mean[i] = accumulate ? gW_mean[i] + a : a;
      gW_log_var[i] = accumulate ? gW_log_var[i] + b : b;
      acc = fmaf(g1, W_mean[i], acc);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) {
      gz[o] = acc;
      const float c = gbvar ? gbvar[o] * expf(b_log_var[o]) : 0.f;
      gb_log_var[o] = accumulate ? gb_log_var[o] + c : c;
    }
  }
}

__global__ void __launch_bounds__(256) noise_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                    const float* __restrict__ eps, float* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = fmaf(sqrtf(var[i]), eps[i], mean[i]);
}

// d out / d var = eps / (2 sqrt(var));  (d out / d mean = 1: the caller passes the cotangent through)
__global__ void __launch_bounds__(256) noise_bwd_kernel(const float* __restrict__ var, const float* __restrict__ eps,
                                                        const float* __restrict__ g, float* __restrict__ g_var, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    g_var[i] = g[i] * eps[i] * 0.5f / sqrtf(var[i]);
}

static inline unsigned conv_grid(int64_t n) {
  const int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : g > 2048 ? 2048 : g);
}

}  // namespace mnf

using namespace mnf;

extern "C" {

int mnf_mnf_conv_operands(const float* W_mean, const float* W_log_var, const float* b_log_var, const float* z, float* Wz,
                          float* W_var, float* b_var, int n_out, int per_out, void* stream) {
  if (!W_mean || !W_log_var || !b_log_var || !z || !Wz || !W_var || !b_var || n_out < 1 || per_out < 1)
    return MNF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(conv_operands_kernel, dim3(conv_grid((int64_t)n_out * per_out)), dim3(256), 0, (hipStream_t)stream,
                     W_mean, W_log_var, b_log_var, z, Wz, W_var, b_var, n_out, per_out);
  return check_launch();
}

int mnf_mnf_conv_operands_bwd(const float* W_mean, const float* W_log_var, const float* b_log_var, const float* z,
                              const float* grad_Wz, const float* grad_W_var, const float* grad_b_var, float* grad_W_mean,
                              float* grad_W_log_var, float* grad_b_log_var, float* grad_z, int n_out, int per_out,
                              int accumulate, void* stream) {
  if (!W_mean || !W_log_var || !b_log_var || !z || !grad_W_mean || !grad_W_log_var || !grad_b_log_var || !grad_z ||
      n_out < 1 || per_out < 1)
    return MNF_ERR_INVALID_ARG;
  const unsigned grid = (unsigned)((n_out + 3) / 4);
  hipLaunchKernelGGL(conv_operands_bwd_kernel, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, (hipStream_t)stream, W_mean,
                     W_log_var, b_log_var, z, grad_Wz, grad_W_var, grad_b_var, grad_W_mean, grad_W_log_var, grad_b_log_var,
                     grad_z, n_out, per_out, accumulate != 0);
  return check_launch();
}

int mnf_mnf_noise(const float* mean, const float* var, const float* eps, float* out, int64_t n, void* stream) {
  if (!mean || !var || !eps || !out || n < 0) return MNF_ERR_INVALID_ARG;
  if (n == 0) return MNF_OK;
  hipLaunchKernelGGL(noise_kernel, dim3(conv_grid(n)), dim3(256), 0, (hipStream_t)stream, mean, var, eps, out, n);
  return check_launch();
}

int mnf_mnf_noise_bwd(const float* var, const float* eps, const float* grad_out, float* grad_var, int64_t n, void* stream) {
  if (!var || !eps || !grad_out || !grad_var || n < 0) return MNF_ERR_INVALID_ARG;
  if (n == 0) return MNF_OK;
  hipLaunchKernelGGL(noise_bwd_kernel, dim3(conv_grid(n)), dim3(256), 0, (hipStream_t)stream, var, eps, grad_out, grad_var, n);
  return check_launch();
}

}  // extern "C"
