// MAF / IAF on the generic path (torch_mnf/flows/maf.py:21-72 over the MADE network of torch_mnf/layers/made.py:11-94).
//
// One thread per row; the MASKED weights (W[o][k] * mask[k][o], made.py:24-25) sit in LDS, where every lane reads the same
// word per multiply (a broadcast), and a thread's activations sit in LDS as act[unit][thread] (conflict-free).  Both
// directions of the layer are here:
//   one pass    (MAF.inverse, IAF.forward; maf.py:54-62):  s, t = net(x);  y = x exp(s) + t, flipped along the features
//               when `parity`;  log_det = sum(s)
//   sequential  (MAF.forward, IAF.inverse; maf.py:39-52):  the input is flipped first when `parity`; starting from zeros,
//               element i becomes (z_i - t_i) exp(-s_i) with s, t from the net on the elements decoded so far (dim net
//               evaluations per row; only outputs i and dim + i of the last layer are computed);  log_det = -sum(s_i)
// and their gradients.  The autoregressive masks make the sequential direction's reverse mode simple: the activations at
// the FINAL output serve every step (s_i, t_i only see units connected to elements < i, and those have their final
// values), so it is one net evaluation and dim back-propagations of a one-hot pair of output cotangents, walking i down
// while the input cotangents of later steps land on earlier elements.
// Weight gradients: per weight, a DPP sum over the wave's 64 rows, one LDS add per wave, one global atomic per
// workgroup and parameter (contiguous).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mnf_hip.h"
#include "mnf_host.h"

namespace mnf {

constexpr int kMafLdsFloats = 36 * 1024;  // 144 KB

struct MafArgs {
  const float* x;
  const float* y;        // gradients, sequential direction: the forward call's output
  float* out;
  float* log_det;
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  const uint8_t* masks;
  int64_t rows;
  int dim, parity, sequential, accumulate;
  int n_lin;                       // linear layers: hidden + 1
  int sizes[MNF_MAX_LINEAR + 1];   // dim, hidden..., 2 dim
  int w_off[MNF_MAX_LINEAR], b_off[MNF_MAX_LINEAR], m_off[MNF_MAX_LINEAR], a_off[MNF_MAX_LINEAR + 1];
  int n_par, act_floats, maxw;
  int rows_per_block;  // rows (= LDS activation slots) per workgroup; blockDim is this rounded up to whole waves
  int lds_grads;       // the workgroup sums the parameter gradients in LDS first (else: one global atomic per wave)
};

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float maf_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the wave; the total is valid in lane 63
__device__ __forceinline__ float maf_wave_sum63(float v) {
  v += maf_dpp<0xB1>(v);
  v += maf_dpp<0x4E>(v);
  v += maf_dpp<0x141>(v);
  v += maf_dpp<0x140>(v);
  v += maf_dpp<0x142, 0xA>(v);
  v += maf_dpp<0x143, 0xC>(v);
  return v;
}

// masked weights -> LDS (wm has the flat layout; biases copied as they are)
__device__ __forceinline__ void maf_stage(const MafArgs& a, float* wm) {
  for (int i = threadIdx.x; i < a.n_par; i += blockDim.x) wm[i] = a.flat[i];
  __syncthreads();
  for (int l = 0; l < a.n_lin; ++l) {
    const int n_in = a.sizes[l], n_out = a.sizes[l + 1];
    for (int i = threadIdx.x; i < n_in * n_out; i += blockDim.x) {
      const int o = i / n_in, k = i - o * n_in;
      if (!a.masks[a.m_off[l] + k * n_out + o]) wm[a.w_off[l] + i] = 0.f;  // mask is (n_in, n_out), made.py:19
    }
  }
  __syncthreads();
}

// act[(a_off[l] + unit) * T + tid]: layer l's input (l = 0: the net's input), post-ReLU for hidden layers.
// Evaluates layers [0, upto); of the LAST layer only outputs {only0, only1} when only0 >= 0.
__device__ __forceinline__ void maf_eval(const MafArgs& a, const float* wm, float* act, int T, int only0, int only1) {
  const int tid = threadIdx.x;
  for (int l = 0; l < a.n_lin; ++l) {
    const int n_in = a.sizes[l], n_out = a.sizes[l + 1];
    const float* W = wm + a.w_off[l];
    const float* b = wm + a.b_off[l];
    const float* in = act + (size_t)a.a_off[l] * T + tid;
    float* out = act + (size_t)a.a_off[l + 1] * T + tid;
    const bool last = l == a.n_lin - 1;
    for (int o = 0; o < n_out; ++o) {
      if (last && only0 >= 0 && o != only0 && o != only1) continue;
      float acc = b[o];
      for (int k = 0; k < n_in; ++k) acc = fmaf(W[o * n_in + k], in[(size_t)k * T], acc);
      out[(size_t)o * T] = last ? acc : fmaxf(acc, 0.f);
    }
  }
}

__global__ void __launch_bounds__(256) maf_fwd_kernel(const MafArgs a) {
  extern __shared__ float maf_lds[];
  const int T = a.rows_per_block, tid = threadIdx.x, d = a.dim;  // T rows per workgroup (<= blockDim: wide nets leave lanes idle)
  float* wm = maf_lds;
  float* act = wm + a.n_par;
  maf_stage(a, wm);
  const int64_t row = (int64_t)blockIdx.x * T + tid;
  if (tid >= T || row >= a.rows) return;
  const float* xr = a.x + row * d;
  float* yr = a.out + row * d;
  float* in0 = act + tid;
  const float* st = act + (size_t)a.a_off[a.n_lin] * T + tid;
  float ld = 0.f;
  if (!a.sequential) {
    for (int j = 0; j < d; ++j) in0[(size_t)j * T] = xr[j];
    maf_eval(a, wm, act, T, -1, -1);
    for (int j = 0; j < d; ++j) {
      const float s = st[(size_t)j * T], t = st[(size_t)(d + j) * T];
      yr[a.parity ? d - 1 - j : j] = fmaf(xr[j], expf(s), t);
      ld += s;
    }
  } else {
    for (int j = 0; j < d; ++j) in0[(size_t)j * T] = 0.f;
    for (int i = 0; i < d; ++i) {
      maf_eval(a, wm, act, T, i, d + i);
      const float s = st[(size_t)i * T], t = st[(size_t)(d + i) * T];
      const float v = (xr[a.parity ? d - 1 - i : i] - t) * expf(-s);
      in0[(size_t)i * T] = v;
      yr[i] = v;
      ld -= s;
    }
  }
  if (a.log_det) a.log_det[row] = a.accumulate ? a.log_det[row] + ld : ld;
}

// One back-propagation through the net for this thread's row.  delta (layer L's output cotangents) lives in
// dl[unit * T + tid] with `n_live` semantics left to the caller (zeros elsewhere); the input cotangent is ADDED to
// g_in[unit * T + tid].  Weight / bias gradients: wave sums into gacc (LDS, flat layout).
__device__ __forceinline__ void maf_backprop(const MafArgs& a, const float* wm, const float* act, float* dl, float* dl2,
                                             float* g_in, float* gacc, int T, bool live) {
  // live: this lane has a row (and an LDS slot); lanes without one only take part in the wave sums
  const int tid = threadIdx.x, lane = tid & 63;
  float* cur = dl + tid;
  float* nxt = dl2 + tid;
  for (int l = a.n_lin - 1; l >= 0; --l) {
    const int n_in = a.sizes[l], n_out = a.sizes[l + 1];
    const float* W = wm + a.w_off[l];
    const float* in = act + (size_t)a.a_off[l] * T + tid;
    if (gacc) {
      for (int o = 0; o < n_out; ++o) {
        const float dv = live ? cur[(size_t)o * T] : 0.f;
        if (__ballot(dv != 0.f) == 0) continue;  // (a one-hot cotangent: most outputs)
        const float sb = maf_wave_sum63(dv);
        if (lane == 63) atomicAdd(gacc + a.b_off[l] + o, sb);
        for (int k = 0; k < n_in; ++k) {
          if (W[o * n_in + k] == 0.f && !a.masks[a.m_off[l] + k * n_out + o]) continue;  // masked out (uniform)
          const float sw = maf_wave_sum63(dv * (live ? in[(size_t)k * T] : 0.f));
          if (lane == 63) atomicAdd(gacc + a.w_off[l] + o * n_in + k, sw);
        }
      }
    }
    // cotangent of this layer's input (through the ReLU of the layer below, whose output it is)
    if (live) {
      for (int k = 0; k < n_in; ++k) {
        float acc = 0.f;
        for (int o = 0; o < n_out; ++o) acc = fmaf(W[o * n_in + k], cur[(size_t)o * T], acc);
        if (l > 0)
          nxt[(size_t)k * T] = in[(size_t)k * T] > 0.f ? acc : 0.f;
        else
          g_in[(size_t)k * T + tid] += acc;
      }
    }
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
}

__global__ void __launch_bounds__(256) maf_bwd_kernel(const MafArgs a) {
  extern __shared__ float maf_lds[];
  const int T = a.rows_per_block, tid = threadIdx.x, d = a.dim;  // T rows (LDS slots) per workgroup, <= blockDim
  float* wm = maf_lds;
  const bool lds_grads = a.grad_flat && a.lds_grads;
  float* gacc = lds_grads ? wm + a.n_par : a.grad_flat;  // (a net too large for a second LDS copy: straight to memory)
  float* act = wm + (lds_grads ? 2 : 1) * a.n_par;
  float* dl = act + (size_t)a.act_floats * T;
  float* dl2 = dl + (size_t)a.maxw * T;
  float* g_in = dl2 + (size_t)a.maxw * T;  // [dim][T]
  if (lds_grads)
    for (int i = tid; i < a.n_par; i += blockDim.x) gacc[i] = 0.f;
  maf_stage(a, wm);
  const int64_t row = (int64_t)blockIdx.x * T + tid;
  const bool live = tid < T && row < a.rows;  // lanes without a row keep out of LDS and add zeros to the wave sums
  const int64_t rc = live ? row : 0;
  const float* xr = a.x + rc * d;
  float* in0 = act + tid;
  const float* st = act + (size_t)a.a_off[a.n_lin] * T + tid;
  const float gl = a.grad_ld && live ? a.grad_ld[rc] : 0.f;
  const int n_last = 2 * d;
  if (!a.sequential) {
    if (live) {
      for (int j = 0; j < d; ++j) in0[(size_t)j * T] = xr[j];
      maf_eval(a, wm, act, T, -1, -1);
      for (int j = 0; j < d; ++j) {
        const float gy = a.grad_y ? a.grad_y[rc * d + (a.parity ? d - 1 - j : j)] : 0.f;
        const float e = expf(st[(size_t)j * T]);
        dl[(size_t)j * T + tid] = fmaf(gy * xr[j], e, gl);   // y = x e^s + t; log_det = sum s
        dl[(size_t)(d + j) * T + tid] = gy;
        g_in[(size_t)j * T + tid] = gy * e;
      }
    }
    maf_backprop(a, wm, act, dl, dl2, g_in, gacc, T, live);
    if (live)
      for (int j = 0; j < d; ++j) a.grad_x[row * d + j] = g_in[(size_t)j * T + tid];
  } else {
    const float* yr = a.y + rc * d;
    if (live) {
      for (int j = 0; j < d; ++j) {
        in0[(size_t)j * T] = yr[j];
        g_in[(size_t)j * T + tid] = a.grad_y ? a.grad_y[rc * d + j] : 0.f;  // G: cotangent of element j
      }
      maf_eval(a, wm, act, T, -1, -1);
    }
    for (int i = d - 1; i >= 0; --i) {
      if (live) {
        const float G = g_in[(size_t)i * T + tid];
        const float e = expf(-st[(size_t)i * T]);
        for (int o = 0; o < n_last; ++o) dl[(size_t)o * T + tid] = 0.f;
        // x_i = (z_i - t_i) e^{-s_i}; log_det = -sum s_i
        dl[(size_t)i * T + tid] = -(G * yr[i]) - gl;
        dl[(size_t)(d + i) * T + tid] = -(G * e);
        a.grad_x[row * d + (a.parity ? d - 1 - i : i)] = G * e;
      }
      maf_backprop(a, wm, act, dl, dl2, g_in, gacc, T, live);
    }
  }
  if (lds_grads) {
    __syncthreads();
    for (int i = tid; i < a.n_par; i += blockDim.x)
      if (gacc[i] != 0.f) atomicAdd(a.grad_flat + i, gacc[i]);
  }
}

static int maf_fill(MafArgs& a, int dim, int n_hidden, const int* hidden) {
  if (dim < 1 || n_hidden < 1 || n_hidden + 1 > MNF_MAX_LINEAR || !hidden) return MNF_ERR_INVALID_ARG;
  a.dim = dim;
  a.n_lin = n_hidden + 1;
  a.sizes[0] = dim;
  for (int i = 0; i < n_hidden; ++i) {
    if (hidden[i] < 1) return MNF_ERR_INVALID_ARG;
    a.sizes[1 + i] = hidden[i];
  }
  a.sizes[n_hidden + 1] = 2 * dim;
  int64_t off = 0, moff = 0, aoff = 0;
  a.maxw = 0;
  for (int l = 0; l < a.n_lin; ++l) {
    const int64_t n = (int64_t)a.sizes[l] * a.sizes[l + 1];
    a.w_off[l] = (int)off;
    off += n;
    a.b_off[l] = (int)off;
    off += a.sizes[l + 1];
    a.m_off[l] = (int)moff;
    moff += n;
    a.a_off[l] = (int)aoff;
    aoff += a.sizes[l];
    if (a.sizes[l + 1] > a.maxw) a.maxw = a.sizes[l + 1];
  }
  a.a_off[a.n_lin] = (int)aoff;
  aoff += a.sizes[a.n_lin];
  if (off > kMafLdsFloats) return MNF_ERR_UNSUPPORTED;
  a.n_par = (int)off;
  a.act_floats = (int)aoff;
  return MNF_OK;
}

// rows (LDS activation slots) per workgroup: as many as fit, at most 256 (0: not even one)
static int maf_rows_per_block(int64_t fixed_floats, int64_t per_row_floats) {
  const int64_t r = (kMafLdsFloats - fixed_floats) / per_row_floats;
  return r < 1 ? 0 : r >= 256 ? 256 : r >= 128 ? 128 : r >= 64 ? 64 : (int)r;
}

}  // namespace mnf

using namespace mnf;

extern "C" {

int64_t mnf_maf_flat_floats(int dim, int n_hidden, const int* hidden_host) {
  MafArgs a;
  memset(&a, 0, sizeof(a));
  const int rc = maf_fill(a, dim, n_hidden, hidden_host);
  return rc == MNF_OK ? a.n_par : -1;
}

int64_t mnf_maf_mask_bytes(int dim, int n_hidden, const int* hidden_host) {
  MafArgs a;
  memset(&a, 0, sizeof(a));
  if (maf_fill(a, dim, n_hidden, hidden_host) != MNF_OK) return -1;
  return (int64_t)a.m_off[a.n_lin - 1] + (int64_t)a.sizes[a.n_lin - 1] * a.sizes[a.n_lin];
}

int mnf_maf(const float* x, float* y, float* log_det, int accumulate, const float* flat, const uint8_t* masks,
            int64_t rows, int dim, int parity, int sequential, int n_hidden, const int* hidden_host, void* stream) {
  if (!x || !y || x == y || !flat || !masks || rows < 0) return MNF_ERR_INVALID_ARG;
  MafArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = maf_fill(a, dim, n_hidden, hidden_host)) return rc;
  if (rows == 0) return MNF_OK;
  a.x = x; a.out = y; a.log_det = log_det; a.accumulate = accumulate != 0; a.flat = flat; a.masks = masks;
  a.rows = rows; a.parity = parity != 0; a.sequential = sequential != 0;
  const int T = maf_rows_per_block(a.n_par, a.act_floats);
  if (!T) return MNF_ERR_UNSUPPORTED;
  a.rows_per_block = T;
  const size_t lds = ((size_t)a.n_par + (size_t)a.act_floats * T) * sizeof(float);
  static DeviceMemo attr;
  if (attr.get([&](int) {
        return hipFuncSetAttribute((const void*)maf_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kMafLdsFloats * (int)sizeof(float)) == hipSuccess ? 1 : -1;
      }) < 0)
    return MNF_ERR_LAUNCH;
  const int64_t blocks = (rows + T - 1) / T;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("maf_generic");
  hipLaunchKernelGGL(maf_fwd_kernel, dim3((unsigned)blocks), dim3((T + 63) / 64 * 64), lds, (hipStream_t)stream, a);
  return check_launch();
}

int mnf_maf_bwd(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                const float* flat, const uint8_t* masks, int64_t rows, int dim, int parity, int sequential, int n_hidden,
                const int* hidden_host, void* stream) {
  if (!x || !grad_x || !flat || !masks || rows < 0 || (sequential && !y)) return MNF_ERR_INVALID_ARG;
  MafArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = maf_fill(a, dim, n_hidden, hidden_host)) return rc;
  if (rows == 0) return MNF_OK;
  a.x = x; a.y = y; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.masks = masks; a.rows = rows; a.parity = parity != 0; a.sequential = sequential != 0;
  const int64_t per_thread = (int64_t)a.act_floats + 2 * (int64_t)a.maxw + dim;
  int64_t fixed = (int64_t)a.n_par * (grad_flat ? 2 : 1);
  a.lds_grads = grad_flat != nullptr;
  int T = maf_rows_per_block(fixed, per_thread);
  if (T < 16 && grad_flat) {  // no room for the gradient copy next to a useful number of rows
    fixed = a.n_par;
    a.lds_grads = 0;
    T = maf_rows_per_block(fixed, per_thread);
  }
  if (!T) return MNF_ERR_UNSUPPORTED;
  a.rows_per_block = T;
  const size_t lds = ((size_t)fixed + (size_t)per_thread * T) * sizeof(float);
  static DeviceMemo attr;
  if (attr.get([&](int) {
        return hipFuncSetAttribute((const void*)maf_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kMafLdsFloats * (int)sizeof(float)) == hipSuccess ? 1 : -1;
      }) < 0)
    return MNF_ERR_LAUNCH;
  const int64_t blocks = (rows + T - 1) / T;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("maf_bwd_generic");
  hipLaunchKernelGGL(maf_bwd_kernel, dim3((unsigned)blocks), dim3((T + 63) / 64 * 64), lds, (hipStream_t)stream, a);
  return check_launch();
}

}  // extern "C"
