// Backward (gradient) kernels: what torch.autograd needs so the Flow modules train like the
// reference's (tests/test_flows.py trains through the layers).
//
// Generic, any-shape kernels in the style of mnf_generic.hip: one 256-thread workgroup owns R
// rows, recomputes the conditioner's forward pass keeping every layer's activations in LDS,
// back-propagates through the MLPs and adds the parameter gradients into `grad_flat` (same
// layout as `flat`) with one fp32 atomic per parameter per workgroup.  Atomic order is not fixed,
// so parameter gradients are reproducible only to fp32 rounding.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {

constexpr int kBwdThreads = 256;
constexpr int kBwdLdsFloats = 15 * 1024;

extern __shared__ __attribute__((aligned(16))) float bsmem[];

// forward of one MLP keeping every layer output: acts[l] is [R][sizes[l+1]] at offset act_off[l]
__device__ __forceinline__ void mlp_forward_keep(const float* __restrict__ flat, const NetDesc& nd,
                                                 const float* in, int ld_in, float* acts, const int* act_off,
                                                 int R) {
  const float* cur = in;
  int ld_cur = ld_in;
  for (int l = 0; l < nd.n_lin; ++l) {
    const int n_in = nd.sizes[l], n_out = nd.sizes[l + 1];
    float* out = acts + act_off[l] * R;
    const float* W = flat + nd.w_off[l];
    const float* b = flat + nd.b_off[l];
    const bool last = l == nd.n_lin - 1;
    for (int idx = threadIdx.x; idx < R * n_out; idx += blockDim.x) {
      const int r = idx / n_out, o = idx - r * n_out;
      float acc = b[o];
      for (int k = 0; k < n_in; ++k) acc = fmaf(W[(size_t)o * n_in + k], cur[r * ld_cur + k], acc);
      out[r * n_out + o] = last ? acc : leaky(acc);
    }
    __syncthreads();
    cur = out;
    ld_cur = n_out;
  }
}

// backward of one MLP: delta (in dA, [R][n_last]) is the gradient wrt the net's output.
// Adds dW, db into grad_flat, and the gradient wrt the net's input into g_in [R][sizes[0]].
__device__ __forceinline__ void mlp_backward(const float* __restrict__ flat, float* __restrict__ grad_flat,
                                             const NetDesc& nd, const float* in, int ld_in, const float* acts,
                                             const int* act_off, float* dA, float* dB, float* g_in, int R) {
  float* delta = dA;
  float* other = dB;
  for (int l = nd.n_lin - 1; l >= 0; --l) {
    const int n_in = nd.sizes[l], n_out = nd.sizes[l + 1];
    const float* hin = (l == 0) ? in : acts + act_off[l - 1] * R;
    const int ld_h = (l == 0) ? ld_in : n_in;
    const float* W = flat + nd.w_off[l];
    if (grad_flat) {
      for (int idx = threadIdx.x; idx < n_out * n_in; idx += blockDim.x) {
        const int o = idx / n_in, k = idx - o * n_in;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(delta[r * n_out + o], hin[r * ld_h + k], acc);
        atomicAdd(grad_flat + nd.w_off[l] + idx, acc);
      }
      for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += delta[r * n_out + o];
        atomicAdd(grad_flat + nd.b_off[l] + o, acc);
      }
    }
    for (int idx = threadIdx.x; idx < R * n_in; idx += blockDim.x) {
      const int r = idx / n_in, k = idx - r * n_in;
      float acc = 0.f;
      for (int o = 0; o < n_out; ++o) acc = fmaf(W[(size_t)o * n_in + k], delta[r * n_out + o], acc);
      if (l == 0) {
        g_in[idx] += acc;
      } else {
        const float h = hin[r * ld_h + k];  // LeakyReLU keeps the sign: h > 0 <=> pre-activation > 0
        other[idx] = h > 0.f ? acc : kLeakySlope * acc;
      }
    }
    __syncthreads();
    float* t = delta;
    delta = other;
    other = t;
  }
}

struct AhfBwdArgs {
  const float* x;
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  int64_t rows;
  int dim, parity, inverse, has_scale, has_shift;
  int R, maxw, act_floats;  // act_floats = sum of layer output widths of one net
  int act_off[MNF_MAX_LINEAR];
  NetDesc s_net, t_net;
};

__global__ void __launch_bounds__(kBwdThreads) ahf_bwd_kernel(AhfBwdArgs a) {
  const int H = a.dim / 2;
  const int64_t row0 = (int64_t)blockIdx.x * a.R;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* cond = bsmem;                       // [R][H]
  float* g_cond = cond + a.R * H;            // [R][H]
  float* acts_s = g_cond + a.R * H;          // [act_floats][R]-sized region
  float* acts_t = acts_s + a.R * a.act_floats;
  float* dA = acts_t + a.R * a.act_floats;   // [R][maxw]
  float* dB = dA + a.R * a.maxw;             // [R][maxw]
  const int cond_off = a.parity ? H : 0, act_off = a.parity ? 0 : H;
  const int last = a.s_net.n_lin > 0 ? a.s_net.n_lin - 1 : a.t_net.n_lin - 1;

  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    cond[idx] = a.x[(row0 + r) * a.dim + cond_off + j];
    g_cond[idx] = a.grad_y ? a.grad_y[(row0 + r) * a.dim + cond_off + j] : 0.f;
  }
  __syncthreads();
  if (a.has_scale) mlp_forward_keep(a.flat, a.s_net, cond, H, acts_s, a.act_off, R);
  if (a.has_shift) mlp_forward_keep(a.flat, a.t_net, cond, H, acts_t, a.act_off, R);
  const float* s_out = acts_s + a.act_off[last] * R;  // [R][H]
  const float* t_out = acts_t + a.act_off[last] * R;

  // gradient wrt the transformed half, and the deltas at the nets' outputs
  //   forward: y = e^s v + t          g_v = g e^s      g_s = g e^s v + g_ld      g_t = g
  //   inverse: y = (v - t) e^-s       g_v = g e^-s     g_s = -g y - g_ld         g_t = -g e^-s
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const int64_t gi = (row0 + r) * a.dim + act_off + j;
    const float g = a.grad_y ? a.grad_y[gi] : 0.f;
    const float gl = a.grad_ld ? a.grad_ld[row0 + r] : 0.f;
    const float v = a.x[gi];
    const float s = a.has_scale ? s_out[idx] : 0.f, t = a.has_shift ? t_out[idx] : 0.f;
    float gv, gs, gt;
    if (a.inverse) {
      const float e = expf(-s);
      gv = g * e;
      gs = -g * ((v - t) * e) - gl;
      gt = -g * e;
    } else {
      const float e = expf(s);
      gv = g * e;
      gs = g * e * v + gl;
      gt = g;
    }
    a.grad_x[gi] = gv;
    dA[idx] = gs;            // dA as [R][H] for the s net
    dB[r * a.maxw + j] = gt; // parked; copied into dA before the t net runs
  }
  __syncthreads();
  if (a.has_scale) {
    // keep g_t safe: mlp_backward ping-pongs dA/dB, so move it out of the way first
    float* park = acts_s + a.act_off[last] * R;  // s_out is no longer needed
    for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) park[idx] = dB[(idx / H) * a.maxw + idx % H];
    __syncthreads();
    mlp_backward(a.flat, a.grad_flat, a.s_net, cond, H, acts_s, a.act_off, dA, dB, g_cond, R);
    for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) dA[idx] = park[idx];
    __syncthreads();
  } else {
    for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) dA[idx] = dB[(idx / H) * a.maxw + idx % H];
    __syncthreads();
  }
  if (a.has_shift) mlp_backward(a.flat, a.grad_flat, a.t_net, cond, H, acts_t, a.act_off, dA, dB, g_cond, R);
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    a.grad_x[(row0 + r) * a.dim + cond_off + j] = g_cond[idx];
  }
}

// column sums over rows: out[j] += sum_r a[r][j] * (b ? b[r][j] : 1)
__global__ void col_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                               int64_t rows, int dim, float scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dim) return;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
  float acc = 0.f;
  for (int64_t r = r0; r < r1; ++r) acc += b ? a[r * dim + j] * b[r * dim + j] : a[r * dim + j];
  atomicAdd(out + j, scale * acc);
}

// out (dim x dim) += x^T g  (Glow: grad wrt W)
__global__ void xtg_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out,
                           int64_t rows, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= dim * dim) return;
  const int i = idx / dim, j = idx - i * dim;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
  float acc = 0.f;
  for (int64_t r = r0; r < r1; ++r) acc = fmaf(x[r * dim + i], g[r * dim + j], acc);
  atomicAdd(out + idx, acc);
}

// grad_x = grad_y * exp(+-s) (AffineConstantFlow), elementwise
__global__ void affine_const_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ s,
                                        float* __restrict__ gx, int64_t n, int dim, int inverse) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int j = (int)(i % dim);
    gx[i] = gy[i] * expf(inverse ? -s[j] : s[j]);
  }
}

}  // namespace mnf

using namespace mnf;

extern "C" {

int mnf_affine_half_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                        float* grad_flat, const float* flat, int64_t rows, int dim, int parity, int inverse,
                        int n_hidden, const int* hidden, int has_scale, int has_shift, void* stream) {
  if (!x || !grad_x || rows < 0 || dim < 2 || (dim & 1) || !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if ((has_scale || has_shift) && !flat) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  AhfBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.rows = rows; a.dim = dim; a.parity = parity != 0; a.inverse = inverse != 0;
  a.has_scale = has_scale != 0; a.has_shift = has_shift != 0;
  const int H = dim / 2;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = H;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  sizes[n_hidden + 1] = H;
  int64_t off = 0;
  if (has_scale) off += fill_net(a.s_net, n_hidden + 2, sizes, off);
  if (has_shift) off += fill_net(a.t_net, n_hidden + 2, sizes, off);
  int act = 0, maxw = H;
  for (int l = 0; l <= n_hidden; ++l) {
    a.act_off[l] = act;
    act += sizes[l + 1];
    if (sizes[l + 1] > maxw) maxw = sizes[l + 1];
  }
  a.act_floats = act;
  a.maxw = maxw;
  const int per_row = 2 * H + 2 * act + 2 * maxw;
  int R = kBwdLdsFloats / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  if (R > 64) R = 64;
  a.R = R;
  const int64_t blocks = (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ahf_bwd_kernel, dim3((unsigned)blocks), dim3(kBwdThreads),
                     (size_t)R * per_row * sizeof(float), (hipStream_t)stream, a);
  return check_launch();
}

int mnf_affine_const_bwd(const float* x, const float* y, const float* grad_y, const float* s, float* grad_x,
                         float* grad_s, float* grad_t, int64_t rows, int dim, int inverse, void* stream) {
  // forward: y = x e^s + t   -> gx = gy e^s ; gs_j = sum_r gy x e^s = sum_r gy (y - t) ; gt_j = sum_r gy
  // inverse: y = (x - t) e^-s -> gx = gy e^-s ; gs_j = -sum_r gy y ; gt_j = -sum_r gy e^-s = -sum_r gx
  if (!x || !y || !grad_y || !s || !grad_x || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = rows * dim;
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(affine_const_bwd_kernel, dim3((unsigned)g), dim3(256), 0, st, grad_y, s, grad_x, n, dim,
                     inverse != 0);
  if (int rc = check_launch()) return rc;
  const dim3 grid((dim + 63) / 64, (unsigned)(rows > 4096 ? 256 : 1));
  if (grad_s) {
    // forward: sum gy * (x e^s) = sum gx * x ; inverse: -sum gy * y
    hipLaunchKernelGGL(col_sum_kernel, grid, dim3(64), 0, st, inverse ? grad_y : grad_x, inverse ? y : x, grad_s,
                       rows, dim, inverse ? -1.f : 1.f);
    if (int rc = check_launch()) return rc;
  }
  if (grad_t) {
    hipLaunchKernelGGL(col_sum_kernel, grid, dim3(64), 0, st, inverse ? grad_x : grad_y, (const float*)nullptr,
                       grad_t, rows, dim, inverse ? -1.f : 1.f);
    if (int rc = check_launch()) return rc;
  }
  return MNF_OK;
}

int mnf_linear_rows_bwd_weight(const float* x, const float* grad_y, float* grad_W, int64_t rows, int dim,
                               void* stream) {
  if (!x || !grad_y || !grad_W || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  const dim3 grid((dim * dim + 255) / 256, (unsigned)(rows > 4096 ? 128 : 1));
  hipLaunchKernelGGL(xtg_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, grad_y, grad_W, rows, dim);
  return check_launch();
}

}  // extern "C"
