// Backward (gradient) kernels: what torch.autograd needs so the Flow modules train like the
// reference's (tests/test_flows.py trains through the layers).
//
// (Specialised gradient kernels live next door: mnf_ahf_bwd_split.hip / mnf_ahf_bwd_mfma.hip for AffineHalfFlow,
// mnf_nsf_bwd_rows.hip for NSF_CL at d = 32; here also Glow's x^T g on MFMAs at d = 32 and the one-pass ActNorm kernel.)
// Generic, any-shape kernels in the style of mnf_generic.hip: one 256-thread workgroup owns R
// rows, recomputes the conditioner's forward pass keeping every layer's activations in LDS,
// back-propagates through the MLPs and adds the parameter gradients into `grad_flat` (same
// layout as `flat`) with one fp32 atomic per parameter per workgroup.  Atomic order is not fixed,
// so parameter gradients are reproducible only to fp32 rounding.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_generic_gemm.h"
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
    staged_linear(W, b, cur, ld_cur, out, n_out, n_in, n_out, R, !last);  // (mnf_generic_gemm.h; ends with a barrier)
    cur = out;
    ld_cur = n_out;
  }
}

// backward of one MLP: delta (in dA, [R][n_last]) is the gradient wrt the net's output.
// Adds dW, db into grad_flat, and the gradient wrt the net's input into g_in [R][sizes[0]].
// `grad_dst` has the flat layout.  lds_acc: it is this workgroup's private LDS accumulator (each
// (layer, unit) slot is always visited by the same thread, so a plain += is race free) and is
// flushed to global memory once per workgroup; otherwise it is the global buffer itself.
__device__ __forceinline__ void grad_add(float* dst, float v, bool lds_acc) {
  if (lds_acc) *dst += v;
  else atomicAdd(dst, v);
}

__device__ __forceinline__ void mlp_backward(const float* __restrict__ flat, float* grad_flat, bool lds_acc,
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
        grad_add(grad_flat + nd.w_off[l] + idx, acc, lds_acc);
      }
      for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += delta[r * n_out + o];
        grad_add(grad_flat + nd.b_off[l] + o, acc, lds_acc);
      }
    }
    const float* Wk = staged_weights(W, n_in, n_out);  // (lanes along k: conflict-free LDS reads when it fits)
    for (int idx = threadIdx.x; idx < R * n_in; idx += blockDim.x) {
      const int r = idx / n_in, k = idx - r * n_in;
      float acc = 0.f;
      for (int o = 0; o < n_out; ++o) acc = fmaf(Wk[(size_t)o * n_in + k], delta[r * n_out + o], acc);
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
  int n_params;             // > 0: parameter gradients are accumulated in LDS and flushed once
  int act_off[MNF_MAX_LINEAR];
  NetDesc s_net, t_net;
};

__global__ void __launch_bounds__(kBwdThreads) ahf_bwd_kernel(AhfBwdArgs a) {
  const int H = a.dim / 2;
  const bool lds_acc = a.n_params > 0 && a.grad_flat;
  float* gacc = bsmem + (size_t)a.R * (2 * H + 2 * a.act_floats + 2 * a.maxw);
  if (lds_acc) {
    for (int i = threadIdx.x; i < a.n_params; i += blockDim.x) gacc[i] = 0.f;
    __syncthreads();
  }
  float* gdst = lds_acc ? gacc : a.grad_flat;
  const int64_t n_groups = (a.rows + a.R - 1) / a.R;
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
  const int64_t row0 = grp * a.R;
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
    mlp_backward(a.flat, gdst, lds_acc, a.s_net, cond, H, acts_s, a.act_off, dA, dB, g_cond, R);
    for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) dA[idx] = park[idx];
    __syncthreads();
  } else {
    for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) dA[idx] = dB[(idx / H) * a.maxw + idx % H];
    __syncthreads();
  }
  if (a.has_shift) mlp_backward(a.flat, gdst, lds_acc, a.t_net, cond, H, acts_t, a.act_off, dA, dB, g_cond, R);
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    a.grad_x[(row0 + r) * a.dim + cond_off + j] = g_cond[idx];
  }
  __syncthreads();  // LDS rows are reused by the next group
  }
  if (lds_acc) {
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_params; i += blockDim.x) atomicAdd(a.grad_flat + i, gacc[i]);
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

// x^T g for dim = 32 on the matrix cores: a wave sums 4 rows per v_mfma_f32_16x16x4_f32 (rows on the K axis; lane
// (c = lane & 15, k = lane >> 4) loads x[row k][16 mi + c] and g[row k][16 nj + c] -- whole 128-byte rows, each
// read once), 2 x 2 output tiles in registers for the whole launch, one LDS sum over the waves and one atomic per
// output element per workgroup.  HBM bound (8 bytes per row element).
__global__ void __launch_bounds__(256) xtg32_mfma_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         float* __restrict__ out, int64_t rows) {
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, k = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t n_groups = (rows + 3) >> 2, stride = (int64_t)gridDim.x * 4;
  constexpr int U = 4;  // 4-row groups in flight per trip
  for (int64_t grp = ((int64_t)blockIdx.x * 4 + wave) * U; grp < n_groups; grp += stride * U) {
    float xa[U][2], ga[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = (grp + u) * 4 + k;
      const bool live = row < rows;
      const int64_t off = (live ? row : 0) * 32 + c;
      xa[u][0] = live ? x[off] : 0.f;
      xa[u][1] = live ? x[off + 16] : 0.f;
      ga[u][0] = live ? g[off] : 0.f;
      ga[u][1] = live ? g[off + 16] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][i], ga[u][j], acc[i][j], 0, 0, 0);
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4* p = reinterpret_cast<f32x4*>(red + t * 256 + lane * 4);
        *p = w == 0 ? acc[t >> 1][t & 1] : *p + acc[t >> 1][t & 1];
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
    const int t = e >> 8, l = (e >> 2) & 63, reg = e & 3;
    const int i = 16 * (t >> 1) + 4 * (l >> 4) + reg, j = 16 * (t & 1) + (l & 15);
    atomicAdd(out + i * 32 + j, red[e]);
  }
}

// AffineConstantFlow / ActNorm gradients in one pass: grad_x = grad_y e^(+-s), and the column sums for grad_s,
// grad_t.  A workgroup runs a multiple of `dim` threads, so a thread stays on one column while it strides over the
// rows; the per-thread sums meet in LDS and leave as one atomic per column per workgroup.
//   forward: y = x e^s + t    -> gs_j = sum_r gx x,   gt_j = sum_r gy
//   inverse: y = (x - t) e^-s -> gs_j = -sum_r gy y,  gt_j = -sum_r gx
__global__ void __launch_bounds__(256) affine_const_bwd_fused_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ y,
                                                                     const float* __restrict__ gy,
                                                                     const float* __restrict__ s, float* __restrict__ gx,
                                                                     float* __restrict__ grad_s,
                                                                     float* __restrict__ grad_t, int64_t n, int dim,
                                                                     int inverse) {
  __shared__ float red_s[256], red_t[256];
  const int col = threadIdx.x % dim;
  const float e = expf(inverse ? -s[col] : s[col]);
  float acc_s = 0.f, acc_t = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;  // a multiple of dim
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = gy[i], v = g * e;
    gx[i] = v;
    acc_s += inverse ? -g * y[i] : v * x[i];
    acc_t += inverse ? -v : g;
  }
  red_s[threadIdx.x] = acc_s;
  red_t[threadIdx.x] = acc_t;
  __syncthreads();
  if ((int)threadIdx.x < dim) {
    float a = 0.f, b = 0.f;
    for (int t = threadIdx.x; t < (int)blockDim.x; t += dim) {
      a += red_s[t];
      b += red_t[t];
    }
    if (grad_s) atomicAdd(grad_s + threadIdx.x, a);
    if (grad_t) atomicAdd(grad_t + threadIdx.x, b);
  }
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
  // parameter-gradient accumulator in LDS when it leaves room for >= 8 rows
  a.n_params = (grad_flat && off + 8 * per_row <= kBwdLdsFloats) ? (int)off : 0;
  int R = (kBwdLdsFloats - a.n_params) / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  if (R > 64) R = 64;
  a.R = R;
  int64_t blocks = (rows + R - 1) / R;
  if (a.n_params > 0 && blocks > 1024) blocks = 1024;  // persistent: 4 workgroups per CU, one flush each
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("ahf_bwd_generic");
  hipLaunchKernelGGL(ahf_bwd_kernel, dim3((unsigned)blocks), dim3(kBwdThreads),
                     ((size_t)R * per_row + a.n_params) * sizeof(float), (hipStream_t)stream, a);
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
  if (dim <= 256) {  // one pass: a workgroup of (256 / dim) dim threads keeps every thread on its column
    const int threads = (256 / dim) * dim;
    int64_t blocks = (n + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(affine_const_bwd_fused_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, x, y, grad_y, s,
                       grad_x, grad_s, grad_t, n, dim, inverse != 0);
    return check_launch();
  }
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
  if (dim == 32) {
    int64_t blocks = (rows + 63) / 64;  // 4 waves x 4 groups x 4 rows per trip
    const int cap = 2 * device_cus(current_device());
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(xtg32_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, grad_y, grad_W,
                       rows);
    return check_launch();
  }
  const dim3 grid((dim * dim + 255) / 256, (unsigned)(rows > 4096 ? 128 : 1));
  hipLaunchKernelGGL(xtg_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, grad_y, grad_W, rows, dim);
  return check_launch();
}

}  // extern "C"

// =====================================================================================
// NSF_CL and RNVP gradients (generic kernels)
// =====================================================================================
namespace mnf {

constexpr int kMaxBins = 32;  // spline backward keeps per-bin arrays in private memory

__device__ __forceinline__ float sigmoid_of_softplus_arg(float x) {  // d softplus / dx, threshold 20
  return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
}

// forward of one spline axis keeping both softmax levels: p1, p2, knots[0..K]
__device__ __forceinline__ void axis_forward(const float* u, int K, float T, float* p1, float* p2, float* knot) {
  const float twoT = 2.f * T, c1 = 1.f - kMinBin * (float)K;
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, u[k]);
  float s = 0.f;
  for (int k = 0; k < K; ++k) { p1[k] = expf(u[k] - m); s += p1[k]; }
  const float r = 1.f / s;
  float m2 = -INFINITY;
  for (int k = 0; k < K; ++k) { p1[k] *= r; m2 = fmaxf(m2, twoT * p1[k]); }
  float s2 = 0.f;
  for (int k = 0; k < K; ++k) { p2[k] = expf(twoT * p1[k] - m2); s2 += p2[k]; }
  const float r2 = 1.f / s2;
  float c = 0.f;
  knot[0] = -T;
  for (int k = 0; k < K; ++k) {
    p2[k] *= r2;
    c += kMinBin + c1 * p2[k];
    knot[k + 1] = (k == K - 1) ? T : twoT * c - T;
  }
}

// gradient wrt the K raw parameters of an axis, given the gradients of knot_b and knot_{b+1}
__device__ __forceinline__ void axis_backward(const float* p1, const float* p2, int K, float T, int b, float g_lo,
                                              float g_hi, float* g_u) {
  const float twoT = 2.f * T, c1 = 1.f - kMinBin * (float)K;
  // knot_i = 2T cum_{i-1} - T for 1 <= i <= K-1 (knot_0, knot_K are constants)
  float dot2 = 0.f;
  for (int k = 0; k < K; ++k) {
    float gf = 0.f;
    if (b >= 1 && k < b) gf += g_lo;
    if (b + 1 <= K - 1 && k <= b) gf += g_hi;
    g_u[k] = c1 * twoT * gf;  // g_p2
    dot2 += p2[k] * g_u[k];
  }
  float dot1 = 0.f;
  for (int k = 0; k < K; ++k) {
    g_u[k] = twoT * (p2[k] * (g_u[k] - dot2));  // g_p1 = 2T g_w1
    dot1 += p1[k] * g_u[k];
  }
  for (int k = 0; k < K; ++k) g_u[k] = p1[k] * (g_u[k] - dot1);
}

// Reverse-mode derivative of rqs_element<true> for one element.  p: 3K-1 raw parameters;
// g_o, g_l: cotangents of (output, log-derivative); writes g_v and g_p[0..3K-1).
__device__ void rqs_element_bwd(float v, int K, float T, bool inverse, const float* p, float g_o, float g_l,
                                float& g_v, float* g_p) {
  const int P = 3 * K - 1;
  for (int i = 0; i < P; ++i) g_p[i] = 0.f;
  if (!((v >= -T) && (v <= T))) {  // identity tails
    g_v = g_o;
    return;
  }
  float uw[kMaxBins], uh[kMaxBins], p1w[kMaxBins], p2w[kMaxBins], p1h[kMaxBins], p2h[kMaxBins];
  float xk[kMaxBins + 1], yk[kMaxBins + 1];
  for (int k = 0; k < K; ++k) { uw[k] = p[k]; uh[k] = p[K + k]; }
  axis_forward(uw, K, T, p1w, p2w, xk);
  axis_forward(uh, K, T, p1h, p2h, yk);
  int b = 0;
  for (int k = 1; k < K; ++k)
    if (v >= (inverse ? yk[k] : xk[k])) b = k;
  const float x0 = xk[b], x1 = xk[b + 1], y0 = yk[b], y1 = yk[b + 1];
  const float raw0 = (b == 0) ? 0.f : p[2 * K + b - 1], raw1 = (b == K - 1) ? 0.f : p[2 * K + b];
  const float pad0 = (b == 0) ? kEdgeDerivConst : softplus(raw0);
  const float pad1 = (b == K - 1) ? kEdgeDerivConst : softplus(raw1);
  const float d0 = kMinDeriv + softplus(pad0), d1 = kMinDeriv + softplus(pad1);
  const float w = x1 - x0, h = y1 - y0, delta = h / w;
  float g_x0 = 0.f, g_x1 = 0.f, g_y0 = 0.f, g_y1 = 0.f, g_d0 = 0.f, g_d1 = 0.f;
  float g_w = 0.f, g_h = 0.f, g_delta = 0.f;
  if (!inverse) {
    const float th = (v - x0) / w, t1 = th * (1.f - th), omt = 1.f - th;
    const float B = delta * th * th + d0 * t1, N = h * B;
    const float cv = d0 + d1 - 2.f * delta, Dn = delta + cv * t1;
    const float A = d1 * th * th + 2.f * delta * t1 + d0 * omt * omt, dn = delta * delta * A;
    const float gN = g_o / Dn, gDn = -g_o * N / (Dn * Dn) - 2.f * g_l / Dn, g_dn = g_l / dn;
    g_y0 += g_o;
    float g_th = 0.f, g_t1 = 0.f;
    g_delta += g_dn * (2.f * delta * A + delta * delta * 2.f * t1);
    const float gA = g_dn * delta * delta;
    g_d1 += gA * th * th; g_d0 += gA * omt * omt; g_th += gA * (2.f * d1 * th - 2.f * d0 * omt); g_t1 += gA * 2.f * delta;
    g_delta += gDn * (1.f - 2.f * t1); g_d0 += gDn * t1; g_d1 += gDn * t1; g_t1 += gDn * cv;
    g_h += gN * B;
    const float gB = gN * h;
    g_delta += gB * th * th; g_th += gB * 2.f * delta * th; g_d0 += gB * t1; g_t1 += gB * d0;
    g_th += g_t1 * (1.f - 2.f * th);
    g_v = g_th / w; g_x0 -= g_th / w; g_w -= g_th * th / w;
  } else {
    const float dy = v - y0, cv = d0 + d1 - 2.f * delta;
    const float a = dy * cv + h * (delta - d0), bb = h * d0 - dy * cv, c = -delta * dy;
    const float disc = bb * bb - 4.f * a * c, sq = sqrtf(disc), den = -bb - sq, xi = 2.f * c / den;
    const float t1 = xi * (1.f - xi), omx = 1.f - xi, Dn = delta + cv * t1;
    const float A = d1 * xi * xi + 2.f * delta * t1 + d0 * omx * omx, dn = delta * delta * A;
    float g_xi = g_o * w, g_t1 = 0.f, g_cv = 0.f;
    g_w += g_o * xi; g_x0 += g_o;
    const float gDn = 2.f * g_l / Dn, g_dn = -g_l / dn;
    g_delta += g_dn * (2.f * delta * A + 2.f * delta * delta * t1);
    const float gA = g_dn * delta * delta;
    g_d1 += gA * xi * xi; g_d0 += gA * omx * omx; g_xi += gA * (2.f * d1 * xi - 2.f * d0 * omx); g_t1 += gA * 2.f * delta;
    g_delta += gDn; g_cv += gDn * t1; g_t1 += gDn * cv;
    g_xi += g_t1 * (1.f - 2.f * xi);
    float g_c = 2.f * g_xi / den;
    const float g_den = -g_xi * xi / den;
    float g_b = -g_den;
    const float g_disc = -g_den / (2.f * sq);
    g_b += 2.f * bb * g_disc;
    const float g_a = -4.f * c * g_disc;
    g_c += -4.f * a * g_disc;
    float g_dy = 0.f;
    g_delta += -dy * g_c; g_dy += -delta * g_c;
    g_h += d0 * g_b; g_d0 += h * g_b; g_dy += -cv * g_b; g_cv += -dy * g_b;
    g_dy += cv * g_a; g_cv += dy * g_a; g_h += (delta - d0) * g_a; g_delta += h * g_a; g_d0 += -h * g_a;
    g_d0 += g_cv; g_d1 += g_cv; g_delta += -2.f * g_cv;
    g_v = g_dy; g_y0 -= g_dy;
  }
  g_h += g_delta / w; g_w -= g_delta * delta / w;
  g_y1 += g_h; g_y0 -= g_h; g_x1 += g_w; g_x0 -= g_w;
  // derivative parameters
  if (b > 0) g_p[2 * K + b - 1] = g_d0 * sigmoid_of_softplus_arg(pad0) * sigmoid_of_softplus_arg(raw0);
  if (b < K - 1) g_p[2 * K + b] = g_d1 * sigmoid_of_softplus_arg(pad1) * sigmoid_of_softplus_arg(raw1);
  axis_backward(p1w, p2w, K, T, b, g_x0, g_x1, g_p);
  axis_backward(p1h, p2h, K, T, b, g_y0, g_y1, g_p + K);
}

struct NsfBwdArgs {
  const float* x;
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  int64_t rows;
  int dim, K, inverse;
  float T;
  int R, maxw, act_floats, ldp;
  int act_off[MNF_MAX_LINEAR];
  NetDesc f1, f2;
  // non-null: the fix-up pass of mnf_nsf_cl_bwd_tile -- only the 16-row tiles cold[2 .. 2 + cold[0]) (every tile when
  // cold[0] < 0 or cold[1] != 0), by a fixed grid; R then divides 16
  const int32_t* cold;
  int cold_capacity;
};

// one half-step forward on LDS rows: vals <- spline(vals; net(cond)); acts keeps the net's layers
__device__ __forceinline__ void nsf_half_forward(const float* flat, const NetDesc& nd, const float* cond, float* vals,
                                                 float* acts, const int* act_off, int H, int K, float T, bool inverse,
                                                 int R) {
  mlp_forward_keep(flat, nd, cond, H, acts, act_off, R);
  const float* params = acts + act_off[nd.n_lin - 1] * R;  // [R][(3K-1) H]
  const int P = 3 * K - 1;
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const float* p = params + r * (P * H) + j * P;
    float o, l;
    rqs_element<true>(vals[idx], K, T, inverse, [&](int k) { return p[k]; }, [&](int k) { return p[K + k]; },
                      [&](int k) { return p[2 * K + k]; }, o, l);
    vals[idx] = o;
  }
  __syncthreads();
}

// backward of one half-step.  vals_in: the half BEFORE the spline; g_vals: cotangent of the half
// AFTER it (in/out: becomes the cotangent of vals_in); g_cond += through the net.
__device__ __forceinline__ void nsf_half_backward(const float* flat, float* grad_flat, const NetDesc& nd,
                                                  const float* cond, const float* vals_in, float* g_vals, float* g_cond,
                                                  const float* g_ld_rows, float* acts, const int* act_off, float* dA,
                                                  float* dB, int H, int K, float T, bool inverse, int R) {
  const int P = 3 * K - 1;
  const float* params = acts + act_off[nd.n_lin - 1] * R;
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    float gv;
    rqs_element_bwd(vals_in[idx], K, T, inverse, params + r * (P * H) + j * P, g_vals[idx], g_ld_rows[r], gv,
                    dA + r * (P * H) + j * P);
    g_vals[idx] = gv;
  }
  __syncthreads();
  mlp_backward(flat, grad_flat, false, nd, cond, H, acts, act_off, dA, dB, g_cond, R);
}

__device__ __forceinline__ void nsf_bwd_block(const NsfBwdArgs& a, const int64_t row0) {
  const int H = a.dim / 2;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* lo0 = bsmem;                 // lower half, input
  float* up0 = lo0 + a.R * H;         // upper half, input
  float* mid = up0 + a.R * H;         // the half after the first half-step
  float* g_lo = mid + a.R * H;
  float* g_up = g_lo + a.R * H;
  float* g_ldr = g_up + a.R * H;      // [R]
  float* acts = g_ldr + a.R;          // [act_floats][R]
  float* dA = acts + a.R * a.act_floats;
  float* dB = dA + a.R * a.maxw;
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const int64_t g = (row0 + r) * a.dim;
    lo0[idx] = a.x[g + j];
    up0[idx] = a.x[g + H + j];
    mid[idx] = a.inverse ? a.x[g + j] : a.x[g + H + j];
    g_lo[idx] = a.grad_y ? a.grad_y[g + j] : 0.f;
    g_up[idx] = a.grad_y ? a.grad_y[g + H + j] : 0.f;
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) g_ldr[r] = a.grad_ld ? a.grad_ld[row0 + r] : 0.f;
  __syncthreads();
  if (!a.inverse) {
    // forward: up1 = S(up0; f1(lo0)); lo1 = S(lo0; f2(up1)).  Reverse: f2 step first.
    nsf_half_forward(a.flat, a.f1, lo0, mid, acts, a.act_off, H, a.K, a.T, false, R);        // mid = up1
    mlp_forward_keep(a.flat, a.f2, mid, H, acts, a.act_off, R);
    nsf_half_backward(a.flat, a.grad_flat, a.f2, mid, lo0, g_lo, g_up, g_ldr, acts, a.act_off, dA, dB, H, a.K, a.T,
                      false, R);                                                              // g_lo -> wrt lo0 (direct); g_up += via f2
    mlp_forward_keep(a.flat, a.f1, lo0, H, acts, a.act_off, R);
    nsf_half_backward(a.flat, a.grad_flat, a.f1, lo0, up0, g_up, g_lo, g_ldr, acts, a.act_off, dA, dB, H, a.K, a.T,
                      false, R);
  } else {
    // inverse: lo1 = S^-1(lo0; f2(up0)); up1 = S^-1(up0; f1(lo1)).  Reverse: f1 step first.
    nsf_half_forward(a.flat, a.f2, up0, mid, acts, a.act_off, H, a.K, a.T, true, R);         // mid = lo1
    mlp_forward_keep(a.flat, a.f1, mid, H, acts, a.act_off, R);
    nsf_half_backward(a.flat, a.grad_flat, a.f1, mid, up0, g_up, g_lo, g_ldr, acts, a.act_off, dA, dB, H, a.K, a.T,
                      true, R);
    mlp_forward_keep(a.flat, a.f2, up0, H, acts, a.act_off, R);
    nsf_half_backward(a.flat, a.grad_flat, a.f2, up0, lo0, g_lo, g_up, g_ldr, acts, a.act_off, dA, dB, H, a.K, a.T,
                      true, R);
  }
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const int64_t g = (row0 + r) * a.dim;
    a.grad_x[g + j] = g_lo[idx];
    a.grad_x[g + H + j] = g_up[idx];
  }
}

__global__ void __launch_bounds__(kBwdThreads) nsf_bwd_kernel(NsfBwdArgs a) {
  if (a.cold == nullptr) {
    nsf_bwd_block(a, (int64_t)blockIdx.x * a.R);
    return;
  }
  const int64_t n_tiles = (a.rows + 15) >> 4;
  const bool all = a.cold[0] < 0 || a.cold[1] != 0;
  const int64_t n = all ? n_tiles : min((int64_t)a.cold[0], (int64_t)a.cold_capacity);
  const int per = 16 / a.R;
  for (int64_t item = blockIdx.x; item < n * per; item += gridDim.x) {
    const int64_t tile = all ? item / per : a.cold[2 + item / per];
    const int64_t row0 = tile * 16 + (item % per) * a.R;
    if (row0 < a.rows) nsf_bwd_block(a, row0);
    __syncthreads();  // the block's LDS rows are reused by the next item
  }
}

struct RnvpBwdArgs {
  const float* z;
  const float* mask;
  const float* grad_x;
  const float* grad_ld;
  float* grad_z;
  float* grad_flat;
  const float* flat;
  int64_t rows;
  int dim, R, maxw, act_floats;
  int t_w, t_b, s_w, s_b;
  uint64_t seed;
  int act_off[MNF_MAX_LINEAR];
  NetDesc net;
  const int32_t* list;  // non-null: only the row groups list[1 .. list[0]] (rows_per_group rows each, a multiple of R)
  int rows_per_group;
};

__device__ __forceinline__ void rnvp_bwd_block(const RnvpBwdArgs& a, const int64_t row0) {
  const int d = a.dim, hl = a.net.sizes[a.net.n_lin];
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* kept = bsmem;                 // [R][d]  m z
  float* g_t = kept + a.R * d;         // [R][d]
  float* g_s = g_t + a.R * d;          // [R][d]
  float* g_kept = g_s + a.R * d;       // [R][d]
  float* acts = g_kept + a.R * d;      // [act_floats][R]
  float* dA = acts + a.R * a.act_floats;
  float* dB = dA + a.R * a.maxw;
  auto mask_of = [&](int r, int j) {
    return a.mask ? a.mask[(row0 + r) * d + j] : rnvp_mask_bit(a.seed, row0 + r, j);
  };
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) {
    const int r = idx / d, j = idx - r * d;
    kept[idx] = mask_of(r, j) * a.z[(row0 + r) * d + j];
    g_kept[idx] = 0.f;
  }
  __syncthreads();
  mlp_forward_keep(a.flat, a.net, kept, d, acts, a.act_off, R);
  const float* y = acts + a.act_off[a.net.n_lin - 1] * R;  // [R][hl]
  const float* Wt = a.flat + a.t_w;
  const float* Ws = a.flat + a.s_w;
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) {
    const int r = idx / d, j = idx - r * d;
    const int64_t gi = (row0 + r) * d + j;
    float shift = a.flat[a.t_b + j], scale = a.flat[a.s_b + j];
    for (int k = 0; k < hl; ++k) {
      shift = fmaf(Wt[(size_t)j * hl + k], y[r * hl + k], shift);
      scale = fmaf(Ws[(size_t)j * hl + k], y[r * hl + k], scale);
    }
    const float m = mask_of(r, j), zz = a.z[gi], gate = sigmoidf(scale);
    const float G = a.grad_x ? a.grad_x[gi] : 0.f, gl = a.grad_ld ? a.grad_ld[row0 + r] : 0.f;
    // x = (1-m) z g + (1-g) t + m z ; ld = sum (1-m) log g
    g_t[idx] = G * (1.f - gate);
    g_s[idx] = (G * ((1.f - m) * zz - shift) * gate + gl * (1.f - m)) * (1.f - gate);
    a.grad_z[gi] = G * ((1.f - m) * gate + m);
  }
  __syncthreads();
  if (a.grad_flat) {
    for (int idx = threadIdx.x; idx < d * hl; idx += blockDim.x) {
      const int j = idx / hl, k = idx - j * hl;
      float at = 0.f, as = 0.f;
      for (int r = 0; r < R; ++r) {
        at = fmaf(g_t[r * d + j], y[r * hl + k], at);
        as = fmaf(g_s[r * d + j], y[r * hl + k], as);
      }
      atomicAdd(a.grad_flat + a.t_w + idx, at);
      atomicAdd(a.grad_flat + a.s_w + idx, as);
    }
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
      float at = 0.f, as = 0.f;
      for (int r = 0; r < R; ++r) { at += g_t[r * d + j]; as += g_s[r * d + j]; }
      atomicAdd(a.grad_flat + a.t_b + j, at);
      atomicAdd(a.grad_flat + a.s_b + j, as);
    }
  }
  for (int idx = threadIdx.x; idx < R * hl; idx += blockDim.x) {  // g_y -> dA [R][hl]
    const int r = idx / hl, k = idx - r * hl;
    float acc = 0.f;
    for (int j = 0; j < d; ++j) {
      acc = fmaf(Wt[(size_t)j * hl + k], g_t[r * d + j], acc);
      acc = fmaf(Ws[(size_t)j * hl + k], g_s[r * d + j], acc);
    }
    dA[idx] = acc;
  }
  __syncthreads();
  mlp_backward(a.flat, a.grad_flat, false, a.net, kept, d, acts, a.act_off, dA, dB, g_kept, R);
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) {
    const int r = idx / d, j = idx - r * d;
    a.grad_z[(row0 + r) * d + j] += mask_of(r, j) * g_kept[idx];
  }
}

__global__ void __launch_bounds__(kBwdThreads) rnvp_bwd_kernel(RnvpBwdArgs a) {
  if (a.list == nullptr) {
    rnvp_bwd_block(a, (int64_t)blockIdx.x * a.R);
    return;
  }
  // fix-up pass: a fixed grid walks the listed row groups (an empty list costs one load per workgroup)
  const int n = a.list[0];
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int64_t g0 = (int64_t)a.list[1 + e] * a.rows_per_group;
    for (int64_t row0 = g0; row0 < g0 + a.rows_per_group && row0 < a.rows; row0 += a.R) {
      __syncthreads();  // the previous rows' LDS contents are no longer read
      rnvp_bwd_block(a, row0);
    }
  }
}

// -------------------------------------------------------------------------- NSF_AR
struct NsfArBwdArgs {
  const float* x;
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  int64_t rows;
  int dim, K, inverse;
  float T;
  int R, maxw, act_floats, n_hidden;
  int hidden[MNF_MAX_LINEAR];
  int act_off[MNF_MAX_LINEAR];
};

// the conditioner of element i >= 1 inside flat (same layout arithmetic as mnf_generic.hip's nsf_ar_net)
__device__ __forceinline__ void nsf_ar_net_bwd(NetDesc& nd, int i, int n_hidden, const int* hidden, int P) {
  int64_t C = 0;
  {
    int prev = 0;
    for (int l = 0; l < n_hidden; ++l) {
      C += (int64_t)prev * hidden[l] + hidden[l];
      prev = hidden[l];
    }
    C += (int64_t)prev * P + P;
  }
  const int h0 = n_hidden > 0 ? hidden[0] : P;
  int64_t off = P + (int64_t)h0 * ((int64_t)(i - 1) * i / 2) + C * (i - 1);
  nd.n_lin = n_hidden + 1;
  nd.sizes[0] = i;
  for (int l = 0; l < n_hidden; ++l) nd.sizes[1 + l] = hidden[l];
  nd.sizes[n_hidden + 1] = P;
  nd.max_width = 0;
  for (int l = 0; l < nd.n_lin; ++l) {
    nd.w_off[l] = (int)off;
    off += (int64_t)nd.sizes[l] * nd.sizes[l + 1];
    nd.b_off[l] = (int)off;
    off += nd.sizes[l + 1];
  }
}

// Reverse mode through the autoregressive layer.  `cond` = what the conditioners saw: the layer's input (inverse
// direction: elements are independent given it) or its output (forward direction: recomputed first, then walked
// backwards, G[:, j] collecting the cotangent of output element j from the elements conditioned on it).
__global__ void __launch_bounds__(kBwdThreads) nsf_ar_bwd_kernel(NsfArBwdArgs a) {
  const int d = a.dim, P = 3 * a.K - 1;
  const int64_t row0 = (int64_t)blockIdx.x * a.R;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* src = bsmem;                  // [R][d]  the layer's input
  float* out = src + a.R * d;          // [R][d]  its output (forward direction: recomputed)
  float* G = out + a.R * d;            // [R][d]  cotangent of the conditioning tensor's elements
  float* g_in = G + a.R * d;           // [R][d]  gradient wrt the layer's input
  float* g_ldr = g_in + a.R * d;       // [R]
  float* acts = g_ldr + a.R;           // [act_floats][R]
  float* dA = acts + a.R * a.act_floats;
  float* dB = dA + a.R * a.maxw;
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) {
    src[idx] = a.x[row0 * d + idx];
    G[idx] = a.grad_y ? a.grad_y[row0 * d + idx] : 0.f;
    g_in[idx] = 0.f;
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) g_ldr[r] = a.grad_ld ? a.grad_ld[row0 + r] : 0.f;
  __syncthreads();
  const bool rqs_inv = !a.inverse;  // forward() runs the spline inverted (spline_flow.py:213-215)
  const float* cond = a.inverse ? src : out;
  NetDesc nd;
  auto params_of = [&](int i) -> const float* {  // [R][P] spline parameters of element i (acts keeps the net's layers)
    if (i == 0) {
      float* p0 = acts + a.act_off[a.n_hidden] * R;
      for (int idx = threadIdx.x; idx < R * P; idx += blockDim.x) p0[idx] = a.flat[idx % P];
      __syncthreads();
      return p0;
    }
    nsf_ar_net_bwd(nd, i, a.n_hidden, a.hidden, P);
    mlp_forward_keep(a.flat, nd, cond, d, acts, a.act_off, R);
    return acts + a.act_off[a.n_hidden] * R;
  };
  if (!a.inverse) {  // recompute the outputs, element by element
    for (int i = 0; i < d; ++i) {
      const float* params = params_of(i);
      for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const float* p = params + r * P;
        float o, l;
        rqs_element<true>(src[r * d + i], a.K, a.T, rqs_inv, [&](int k) { return p[k]; },
                          [&](int k) { return p[a.K + k]; }, [&](int k) { return p[2 * a.K + k]; }, o, l);
        out[r * d + i] = o;
      }
      __syncthreads();
    }
  }
  // inverse direction: the cotangent of output element i is grad_y[:, i] alone, and what flows back through the
  // conditioners lands on the INPUT (g_in); forward direction: it lands on earlier OUTPUT elements (G), so walk down
  for (int i = d - 1; i >= 0; --i) {
    const float* params = params_of(i);
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      float gv;
      rqs_element_bwd(src[r * d + i], a.K, a.T, rqs_inv, params + r * P, G[r * d + i], g_ldr[r], gv, dA + r * P);
      g_in[r * d + i] += gv;
    }
    __syncthreads();
    if (i == 0) {  // init_param: the same P values for every row
      if (a.grad_flat)
        for (int k = threadIdx.x; k < P; k += blockDim.x) {
          float acc = 0.f;
          for (int r = 0; r < R; ++r) acc += dA[r * P + k];
          atomicAdd(a.grad_flat + k, acc);
        }
      __syncthreads();
    } else {
      // gradient wrt the conditioner's input [R][i] -> dense scratch, then onto G (forward) or g_in (inverse)
      float* g_cond = dB + a.R * a.maxw;  // [R][d]
      for (int idx = threadIdx.x; idx < R * i; idx += blockDim.x) g_cond[idx] = 0.f;
      __syncthreads();
      mlp_backward(a.flat, a.grad_flat, false, nd, cond, d, acts, a.act_off, dA, dB, g_cond, R);
      float* sink = a.inverse ? g_in : G;
      for (int idx = threadIdx.x; idx < R * i; idx += blockDim.x) {
        const int r = idx / i, k = idx - r * i;
        sink[r * d + k] += g_cond[idx];
      }
      __syncthreads();
    }
  }
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) a.grad_x[row0 * d + idx] = g_in[idx];
}

}  // namespace mnf

extern "C" {

int mnf_nsf_ar_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                   const float* flat, int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden,
                   const int* hidden, void* stream) {
  if (!x || !grad_x || !flat || rows < 0 || dim < 1 || K < 2 || K > kMaxBins || !(tail_bound > 0.f) ||
      !hidden_ok(n_hidden, hidden))
    return K > kMaxBins ? MNF_ERR_UNSUPPORTED : MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  NsfArBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.rows = rows; a.dim = dim; a.K = K; a.inverse = inverse != 0; a.T = tail_bound; a.n_hidden = n_hidden;
  const int P = 3 * K - 1;
  int act = 0, maxw = dim > P ? dim : P;
  for (int l = 0; l < n_hidden; ++l) {
    a.hidden[l] = hidden[l];
    a.act_off[l] = act;
    act += hidden[l];
    if (hidden[l] > maxw) maxw = hidden[l];
  }
  a.act_off[n_hidden] = act;
  act += P;
  a.act_floats = act; a.maxw = maxw;
  const int per_row = 5 * dim + 1 + act + 2 * maxw;  // src, out, G, g_in, g_cond; g_ldr; acts; dA, dB
  int R = kBwdLdsFloats / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  if (R > 32) R = 32;
  a.R = R;
  const int64_t blocks = (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("nsf_ar_bwd_generic");
  hipLaunchKernelGGL(nsf_ar_bwd_kernel, dim3((unsigned)blocks), dim3(kBwdThreads), (size_t)R * per_row * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch();
}

static int nsf_cl_bwd_launch(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                             const float* flat, int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden,
                             const int* hidden, const int32_t* cold, int cold_capacity, void* stream) {
  if (!x || !grad_x || !flat || rows < 0 || dim < 2 || (dim & 1) || K < 2 || K > kMaxBins || !(tail_bound > 0.f) ||
      !hidden_ok(n_hidden, hidden))
    return K > kMaxBins ? MNF_ERR_UNSUPPORTED : MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  NsfBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.rows = rows; a.dim = dim; a.K = K; a.inverse = inverse != 0; a.T = tail_bound;
  const int H = dim / 2;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = H;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  sizes[n_hidden + 1] = ((3 * K - 1) * dim) / 2;
  int64_t off = fill_net(a.f1, n_hidden + 2, sizes, 0);
  fill_net(a.f2, n_hidden + 2, sizes, off);
  int act = 0, maxw = H;
  for (int l = 0; l <= n_hidden; ++l) {
    a.act_off[l] = act;
    act += sizes[l + 1];
    if (sizes[l + 1] > maxw) maxw = sizes[l + 1];
  }
  a.act_floats = act; a.maxw = maxw; a.ldp = sizes[n_hidden + 1];
  const int per_row = 5 * H + 1 + act + 2 * maxw;
  int R = kBwdLdsFloats / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  if (R > 32) R = 32;
  int64_t blocks = (rows + R - 1) / R;
  if (cold) {  // the listed 16-row tiles only: R divides 16, a fixed grid walks the list
    R = R >= 16 ? 16 : (R >= 8 ? 8 : (R >= 4 ? 4 : (R >= 2 ? 2 : 1)));
    a.cold = cold;
    a.cold_capacity = cold_capacity;
    blocks = ((rows + 15) / 16) * (16 / R);
    const int64_t cap = 8 * (int64_t)device_cus(current_device());
    if (blocks > cap) blocks = cap;
    // MNF_DETERMINISTIC: the list in ascending order and ONE workgroup -- a parameter's additions then follow the list
    // (mnf_host.h det_sort_ids_async; a slot of the sums is always visited by the same thread)
    if (deterministic() && det_sort_ids_async(const_cast<int32_t*>(cold) + 2, cold, cold_capacity, (rows + 15) / 16,
                                              (hipStream_t)stream) == MNF_OK)
      blocks = 1;
  }
  a.R = R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  if (!cold) tag_kernel("nsf_bwd_generic");  // (as the tile kernel's fix-up pass it keeps that kernel's name)
  hipLaunchKernelGGL(nsf_bwd_kernel, dim3((unsigned)blocks), dim3(kBwdThreads), (size_t)R * per_row * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch();
}

int mnf_nsf_cl_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                   const float* flat, int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden,
                   const int* hidden, void* stream) {
  return nsf_cl_bwd_launch(x, grad_y, grad_ld, grad_x, grad_flat, flat, rows, dim, K, tail_bound, inverse, n_hidden,
                           hidden, nullptr, 0, stream);
}

int mnf_nsf_cl_bwd_tile_fixup(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                              const float* flat, int64_t rows, int dim, int K, float tail_bound, int inverse,
                              int n_hidden, const int* hidden, const int32_t* cold, int cold_capacity, void* stream) {
  if (!cold || cold_capacity < (rows + 15) / 16) return MNF_ERR_INVALID_ARG;
  return nsf_cl_bwd_launch(x, grad_y, grad_ld, grad_x, grad_flat, flat, rows, dim, K, tail_bound, inverse, n_hidden,
                           hidden, cold, cold_capacity, stream);
}

}  // extern "C"

namespace mnf {
// list != nullptr: the fix-up pass of mnf_rnvp_bwd_mfma -- only the row groups list[1 .. list[0]] (device memory) of
// `rows_per_group` rows each are computed, by a fixed grid; rows per workgroup then divide rows_per_group
int rnvp_bwd_generic_launch(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                            float* grad_z, float* grad_flat, const float* flat, int64_t rows, int dim, int n_hidden,
                            const int* hidden, const int32_t* list, int rows_per_group, hipStream_t stream) {
  if (!z || !grad_z || !flat || rows < 0 || dim < 1 || n_hidden < 1 || !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  RnvpBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.z = z; a.mask = mask; a.seed = seed; a.grad_x = grad_x; a.grad_ld = grad_ld; a.grad_z = grad_z;
  a.grad_flat = grad_flat; a.flat = flat; a.rows = rows; a.dim = dim;
  a.list = list; a.rows_per_group = rows_per_group;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = dim;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  int64_t off = fill_net(a.net, n_hidden + 1, sizes, 0);
  const int hl = hidden[n_hidden - 1];
  a.t_w = (int)off; off += (int64_t)hl * dim;
  a.t_b = (int)off; off += dim;
  a.s_w = (int)off; off += (int64_t)hl * dim;
  a.s_b = (int)off;
  int act = 0, maxw = dim;
  for (int l = 0; l < n_hidden; ++l) {
    a.act_off[l] = act;
    act += sizes[l + 1];
    if (sizes[l + 1] > maxw) maxw = sizes[l + 1];
  }
  a.act_floats = act; a.maxw = maxw;
  const int per_row = 4 * dim + act + 2 * maxw;
  int R = kBwdLdsFloats / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  if (R > 32) R = 32;
  if (list) {  // a workgroup must not straddle two groups
    int r2 = 1;
    while (2 * r2 <= R && rows_per_group % (2 * r2) == 0) r2 *= 2;
    R = r2;
  }
  a.R = R;
  int64_t blocks = list ? 256 : (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  if (list && deterministic()) {  // the groups in ascending order, one workgroup: mnf_host.h det_sort_ids_async
    const int64_t n_groups = (rows + rows_per_group - 1) / rows_per_group;
    if (det_sort_ids_async(const_cast<int32_t*>(list) + 1, list, (int)n_groups, n_groups, stream) == MNF_OK) blocks = 1;
  }
  if (!list) tag_kernel("rnvp_bwd_generic");  // (as the matrix-core pass's fix-up it keeps that pass's name)
  hipLaunchKernelGGL(rnvp_bwd_kernel, dim3((unsigned)blocks), dim3(kBwdThreads), (size_t)R * per_row * sizeof(float),
                     stream, a);
  return check_launch();
}
}  // namespace mnf

extern "C" {

int mnf_rnvp_bwd(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                 float* grad_z, float* grad_flat, const float* flat, int64_t rows, int dim, int n_hidden,
                 const int* hidden, void* stream) {
  if (z && grad_z && flat && hidden && mnf::rnvp_few_ok(rows, dim, n_hidden, hidden))
    return mnf::rnvp_few_bwd_launch(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, flat, nullptr, rows, dim, hidden[0],
                                    (hipStream_t)stream);
  return mnf::rnvp_bwd_generic_launch(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, flat, rows, dim, n_hidden, hidden,
                                      nullptr, 0, (hipStream_t)stream);
}

}  // extern "C"
