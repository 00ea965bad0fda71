// The KL term of the two MNF layers behind their flows: MNFLinear.kl_div (torch_mnf/layers/mnf_linear.py:66-90) and
// MNFConv2d.kl_div (torch_mnf/layers/mnf_conv.py:100-133) as ONE launch forward and ONE backward.
//
// What bounds these is not arithmetic (a weight matrix of 1 k .. 40 k elements) but launches: composed from stock
// elementwise ops a layer's term is ~70 kernels forward and ~100 backward of ~3 us each, and MNF-LeNet's training step
// at batch 128 spends 680 of its 1,200 launches there.  Both layers' terms have the same shape once the weight tensor is
// seen as a (rows, cols) matrix whose rows feed the auxiliary activation `act`:
//
//   linear:  rows = n_out, cols = n_in; z scales COLUMN j;                noise eps[r][j] per weight;  act = tanh(pre)
//   conv:    rows = n_in k k, cols = n_out (the reference's `.view(-1, len(r0_c))` of the flat (n_out, n_in, k, k)
//            tensor, mnf_conv.py:117-118); z scales flat element idx by z[idx / rows] (the output channel);
//            noise eps[r] per row (eq. 12's W_std @ c is multiplied by one draw per row) and a scalar draw eps_b for
//            the bias part;  act = pre + bias_term  (linear activation, mnf_conv.py:119-123)
//
//   pre[r]   = sum_j c[j] (W_mean z + sqrt(exp(W_log_var)) eps)[r][j]
//   abar     = mean_r act[r];  mean_r[i] = b1[i] abar;  log_var_r[i] = b2[i] abar           (outer(..).mean(1))
//   out      = 0.5 sum(-W_log_var + exp(W_log_var) + (W_mean z)^2 - 1) + kl_b - log_det_q - 0.5 sum q0_log_var
//              - log_det_r - 0.5 sum_i(-exp(log_var_r) (z_r - mean_r)^2 + log_var_r)
//
// One workgroup of 1,024 threads each way (the tensors are a few tens of kB: L2-resident after the first touch, and
// a grid would only add a cross-workgroup reduction); every reduction is a fixed-order tree, so results repeat bit
// for bit from run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnf_hip.h"
#include "mnf_host.h"

namespace mnf {

constexpr int kKlThreads = 1024;
constexpr int kKlWaves = kKlThreads / 64;

struct KlArgs {
  const float* W_mean;
  const float* W_log_var;
  const float* eps;
  const float* eps_b;
  const float* z;
  const float* z_r;
  const float* log_det_q;
  const float* log_det_r;
  const float* b_mean;
  const float* b_log_var;
  const float* q0_log_var;
  const float* c;
  const float* b1;
  const float* b2;
  int64_t rows;
  int cols, n_bias;
  float* out;           // forward: the term
  float* saved;         // act[rows], abar, sqrt(bias variance)
  const float* grad_out;
  float* grads;         // backward: dz | dz_r | dlog_det_q, dlog_det_r
  float* param_grads;   // backward: the layer's parameter gradients in the module's parameter order (mnf_hip.h)
  int accumulate;       // param_grads: 0 written, 1 added to
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// fixed-order sum over the workgroup, returned to every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();  // red may still be read by the previous call
  if (lane == 0) red[wave] = v;
  __syncthreads();
  v = lane < kKlWaves ? red[lane] : 0.f;
  return wave_sum(v);
}

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  v = lane < kKlWaves ? red[lane] : 0.0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <bool CONV>
__global__ void __launch_bounds__(kKlThreads) kl_fwd_kernel(const KlArgs a) {
  __shared__ float red[kKlWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cols = a.cols;
  const int64_t rows = a.rows;
  float bias_term = 0.f, bias_sd = 0.f;
  if (CONV) {  // mnf_conv.py:125-128
    float bs = 0.f, bv = 0.f;
    for (int o = tid; o < a.n_bias; o += kKlThreads) {
      const float cj = a.c[o];
      bs += (a.b_mean ? a.b_mean[o] : 0.f) * a.z[o] * cj;
      bv += expf(a.b_log_var[o]) * cj * cj;
    }
    bs = block_sum(bs, red);
    bv = block_sum(bv, red);
    bias_sd = sqrtf(bv);
    bias_term = bs + bias_sd * a.eps_b[0];
  }
  float klw = 0.f, asum = 0.f;
  for (int64_t r = wave; r < rows; r += kKlWaves) {
    float pre = 0.f;
    const float er = CONV ? a.eps[r] : 0.f;
    for (int j = lane; j < cols; j += 64) {
      const int64_t idx = r * cols + j;
      const float zv = a.z[CONV ? idx / rows : j];
      const float wm = a.W_mean[idx] * zv, lv = a.W_log_var[idx], wv = expf(lv);
      klw += -lv + wv + wm * wm - 1.f;
      pre = fmaf(a.c[j], fmaf(sqrtf(wv), CONV ? er : a.eps[idx], wm), pre);
    }
    pre = wave_sum(pre);
    const float act = CONV ? pre + bias_term : tanhf(pre);
    if (lane == 0) {
      a.saved[r] = act;
      asum += act;
    }
  }
  klw = block_sum(klw, red);
  const float abar = block_sum(asum, red) / (float)rows;
  float t = 0.f, q0 = 0.f, klb = 0.f;
  for (int i = tid; i < cols; i += kKlThreads) {
    const float lvr = a.b2[i] * abar, dz = a.z_r[i] - a.b1[i] * abar;
    t += -expf(lvr) * dz * dz + lvr;
    q0 += a.q0_log_var[i];
  }
  for (int o = tid; o < a.n_bias; o += kKlThreads) {
    const float lv = a.b_log_var[o];
    const float bm = (a.b_mean ? a.b_mean[o] : 0.f) * (CONV ? a.z[o] : 1.f);
    klb += -lv + expf(lv) + bm * bm - 1.f;
  }
  t = block_sum(t, red);
  q0 = block_sum(q0, red);
  klb = block_sum(klb, red);
  if (tid == 0) {
    a.saved[rows] = abar;
    a.saved[rows + 1] = bias_sd;
    a.out[0] = 0.5f * klw + 0.5f * klb - a.log_det_q[0] - 0.5f * q0 - a.log_det_r[0] - 0.5f * t;
  }
}

// grads: dz | dz_r | dlog_det_q, dlog_det_r;  param_grads: dW_mean | dW_log_var | db_mean (linear only) | db_log_var |
// dq0_mean (no direct dependence: zero) | dq0_log_var | dr0_c | dr0_b1 | dr0_b2 -- the order both reference modules
// register their parameters in, so that a flat gradient buffer's slice can be handed in and added to
template <bool CONV>
__global__ void __launch_bounds__(kKlThreads) kl_bwd_kernel(const KlArgs a) {
  extern __shared__ float klds[];
  __shared__ float red[kKlWaves];
  __shared__ double red64[kKlWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cols = a.cols, nb = a.n_bias;
  const int64_t rows = a.rows, n = rows * cols;
  float* p = klds;                  // [rows]  d out / d pre[r], times grad_out
  float* part_c = p + rows;         // [1024]  column sums, one slot per (row group, column)
  float* part_z = part_c + kKlThreads;
  float* g_z = a.grads;
  float* g_zr = g_z + cols;
  float* g_ld = g_zr + cols;
  float* gWm = a.param_grads;
  float* gWlv = gWm + n;
  float* g_bm = CONV ? nullptr : gWlv + n;
  float* g_blv = gWlv + n + (CONV ? 0 : nb);
  float* g_q0m = g_blv + nb;
  float* g_q0 = g_q0m + cols;
  float* g_c = g_q0 + cols;
  float* g_b1 = g_c + cols;
  float* g_b2 = g_b1 + cols;
  const bool acc = a.accumulate != 0;
  auto put = [acc](float* dst, float v) { *dst = acc ? *dst + v : v; };
  const float g = a.grad_out[0], abar = a.saved[rows], bias_sd = a.saved[rows + 1];

  // d out / d abar: a sum of `cols` terms of both signs that every activation-side gradient is proportional to -- formed
  // in double (cols <= a few hundred elements; in float its cancellation showed as 4e-5 on d r0_c)
  double dab = 0.0;
  for (int i = tid; i < cols; i += kKlThreads) {
    const double b1 = a.b1[i], b2 = a.b2[i];
    const double e = exp(b2 * (double)abar), dz = (double)a.z_r[i] - b1 * (double)abar;
    const double dmr = -e * dz, dlv = -0.5 * (1.0 - e * dz * dz);
    g_zr[i] = g * (float)(e * dz);
    put(g_b1 + i, g * (float)(dmr * (double)abar));
    put(g_b2 + i, g * (float)(dlv * (double)abar));
    put(g_q0 + i, -0.5f * g);
    if (!acc) g_q0m[i] = 0.f;
    dab += dmr * b1 + dlv * b2;
  }
  dab = block_sum_f64(dab, red64);
  const float q = g * (float)(dab / (double)rows);
  float psum = 0.f;
  for (int64_t r = tid; r < rows; r += kKlThreads) {
    const float act = a.saved[r];
    const float pr = CONV ? q : q * (1.f - act * act);
    p[r] = pr;
    psum += pr;
  }
  psum = block_sum(psum, red);  // (its barriers also publish p)

  // pass A: thread (row group, column); elementwise weight gradients and the sums over a column's rows
  for (int jb = 0; jb < cols; jb += kKlThreads) {
    const int width = min(cols - jb, kKlThreads), groups = kKlThreads / width;
    const int grp = tid / width, jj = tid - grp * width, j = jb + jj;
    float gc = 0.f, gz = 0.f;
    if (grp < groups) {
      const float cj = a.c[j];
      for (int64_t r = grp; r < rows; r += groups) {
        const int64_t idx = r * cols + j;
        const float zv = a.z[CONV ? idx / rows : j];
        const float w = a.W_mean[idx], wm = w * zv, wv = expf(a.W_log_var[idx]), sde = sqrtf(wv) * a.eps[CONV ? r : idx];
        const float pr = p[r], pc = pr * cj;
        const float t = fmaf(g, wm, pc);  // d / d (W_mean z)
        put(gWm + idx, t * zv);
        put(gWlv + idx, 0.5f * fmaf(g, wv - 1.f, pc * sde));
        gc = fmaf(pr, wm + sde, gc);
        if (!CONV) gz = fmaf(t, w, gz);
      }
      part_c[grp * width + jj] = gc;
      part_z[grp * width + jj] = gz;
    }
    __syncthreads();
    if (tid < width) {
      float sc = 0.f, sz = 0.f;
      for (int gi = 0; gi < groups; ++gi) {
        sc += part_c[gi * width + tid];
        sz += part_z[gi * width + tid];
      }
      if (CONV) {  // the bias part of act (mnf_conv.py:125-128) also depends on c
        const float bm = a.b_mean ? a.b_mean[j] : 0.f;
        sc += psum * (bm * a.z[j] + a.eps_b[0] * a.c[j] * expf(a.b_log_var[j]) / bias_sd);
      } else {
        g_z[j] = sz;
      }
      put(g_c + j, sc);
    }
    __syncthreads();
  }
  if (CONV) {
    // pass B: a wave per output channel o = the contiguous chunk [o rows, (o + 1) rows) of the flat tensor
    for (int o = wave; o < cols; o += kKlWaves) {
      const float zv = a.z[o];
      float gz = 0.f;
      for (int64_t m = lane; m < rows; m += 64) {
        const int64_t idx = (int64_t)o * rows + m;
        const int64_t r = idx / cols;
        const float w = a.W_mean[idx];
        gz = fmaf(fmaf(g, w * zv, p[r] * a.c[idx - r * cols]), w, gz);
      }
      gz = wave_sum(gz);
      if (lane == 0) {
        const float bm = a.b_mean ? a.b_mean[o] : 0.f;
        g_z[o] = gz + g * bm * bm * zv + psum * bm * a.c[o];
      }
    }
    for (int o = tid; o < nb; o += kKlThreads) {
      const float cj = a.c[o], bv = expf(a.b_log_var[o]);
      put(g_blv + o, 0.5f * g * (bv - 1.f) + psum * a.eps_b[0] * 0.5f * bv * cj * cj / bias_sd);
    }
  } else {
    for (int o = tid; o < nb; o += kKlThreads) {
      put(g_bm + o, g * a.b_mean[o]);
      put(g_blv + o, 0.5f * g * (expf(a.b_log_var[o]) - 1.f));
    }
  }
  if (tid == 0) {
    g_ld[0] = -g;
    g_ld[1] = -g;
  }
}

// rows are kept in LDS by the backward kernel (p[rows]); beyond this the layer has no kernel
constexpr int64_t kKlMaxRows = 28 * 1024;

static bool kl_args_ok(const KlArgs& a, int conv) {
  if (!a.W_mean || !a.W_log_var || !a.eps || !a.z || !a.z_r || !a.log_det_q || !a.log_det_r || !a.b_log_var ||
      !a.q0_log_var || !a.c || !a.b1 || !a.b2 || !a.saved || a.rows < 1 || a.cols < 1 || a.n_bias < 1)
    return false;
  if (conv ? (!a.eps_b || a.n_bias != a.cols) : (!a.b_mean || (int64_t)a.n_bias != a.rows)) return false;
  return true;
}

}  // namespace mnf

using namespace mnf;

extern "C" {

int64_t mnf_mnf_kl_saved_floats(int64_t rows) { return rows < 0 ? 0 : rows + 2; }

int64_t mnf_mnf_kl_grad_floats(int cols) { return cols < 1 ? 0 : 2 * (int64_t)cols + 2; }

int64_t mnf_mnf_kl_param_grad_floats(int conv, int64_t rows, int cols, int n_bias) {
  if (rows < 1 || cols < 1 || n_bias < 1) return 0;
  return 2 * rows * cols + (conv ? 1 : 2) * (int64_t)n_bias + 5 * (int64_t)cols;
}

int mnf_mnf_kl_fwd(const float* W_mean, const float* W_log_var, const float* eps, const float* eps_b, const float* z,
                   const float* z_r, const float* log_det_q, const float* log_det_r, const float* b_mean,
                   const float* b_log_var, const float* q0_log_var, const float* r0_c, const float* r0_b1,
                   const float* r0_b2, int conv, int64_t rows, int cols, int n_bias, float* out, float* saved,
                   void* stream) {
  KlArgs a{W_mean, W_log_var, eps,   eps_b, z,    z_r,  log_det_q, log_det_r, b_mean, b_log_var, q0_log_var,
           r0_c,   r0_b1,     r0_b2, rows,  cols, n_bias, out,     saved,     nullptr, nullptr, nullptr, 0};
  if (!kl_args_ok(a, conv) || !out) return MNF_ERR_INVALID_ARG;
  if (rows > kKlMaxRows) return MNF_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (conv)
    hipLaunchKernelGGL(kl_fwd_kernel<true>, dim3(1), dim3(kKlThreads), 0, s, a);
  else
    hipLaunchKernelGGL(kl_fwd_kernel<false>, dim3(1), dim3(kKlThreads), 0, s, a);
  return check_launch();
}

int mnf_mnf_kl_bwd(const float* W_mean, const float* W_log_var, const float* eps, const float* eps_b, const float* z,
                   const float* z_r, const float* b_mean, const float* b_log_var, const float* r0_c, const float* r0_b1,
                   const float* r0_b2, const float* saved, const float* grad_out, int conv, int64_t rows, int cols,
                   int n_bias, float* grads, float* param_grads, int accumulate, void* stream) {
  // (log_det_q, log_det_r and q0_log_var enter the term linearly: their values are not needed here)
  KlArgs a{W_mean, W_log_var, eps,   eps_b, z,    z_r,    z,       z,     b_mean,   b_log_var, z,
           r0_c,   r0_b1,     r0_b2, rows,  cols, n_bias, nullptr, const_cast<float*>(saved), grad_out, grads,
           param_grads, accumulate};
  if (!kl_args_ok(a, conv) || !grad_out || !grads || !param_grads) return MNF_ERR_INVALID_ARG;
  if (rows > kKlMaxRows) return MNF_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(rows + 2 * kKlThreads) * sizeof(float);
  static DeviceMemo attr_set;
  const int ok = attr_set.get([&](int) {
    const int big = (int)((kKlMaxRows + 2 * kKlThreads) * sizeof(float));
    return hipFuncSetAttribute((const void*)kl_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big) ==
                       hipSuccess &&
                   hipFuncSetAttribute((const void*)kl_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       big) == hipSuccess
               ? 1
               : -1;
  });
  if (ok < 0) return MNF_ERR_LAUNCH;
  if (conv)
    hipLaunchKernelGGL(kl_bwd_kernel<true>, dim3(1), dim3(kKlThreads), lds, s, a);
  else
    hipLaunchKernelGGL(kl_bwd_kernel<false>, dim3(1), dim3(kKlThreads), lds, s, a);
  return check_launch();
}

}  // extern "C"
