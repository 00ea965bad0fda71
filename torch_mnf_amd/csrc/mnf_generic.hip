// Generic (any-shape) kernels for every layer on the hot path, plus the C-ABI entry points.
//
// These cover every configuration the reference accepts (odd hidden sizes, d = 2, NICE
// variants, any K); the specialised MFMA kernels in mnf_ahf_mfma.hip / mnf_nsf_mfma.hip /
// mnf_rnvp_mfma.hip take over for the shapes that matter for throughput.
//
// Structure of a generic coupling kernel: one 256-thread workgroup owns R consecutive rows
// (R chosen on the host so that everything fits in 64 KiB of LDS); the conditioner's
// activations live in LDS as [R][width]; every thread computes (row, unit) pairs of a
// Linear layer with an fmaf chain in k order; the transform and the per-row log|det J|
// reduction run out of LDS.  Global traffic is coalesced: rows are read and written once.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_generic_gemm.h"
#include "mnf_split.h"

namespace mnf {

constexpr int kThreads = 256;
constexpr int kLdsBudgetFloats = 15 * 1024;  // 60 KiB of dynamic LDS per workgroup

extern __shared__ __attribute__((aligned(16))) float smem[];

// One Linear (+ optional LeakyReLU) over R rows held in LDS.
// in: [R][ld_in], out: [R][ld_out]; weights (n_out, n_in) row-major in global memory (mnf_generic_gemm.h: staged
// through LDS tile by tile in the orientation the dot products read them in).
__device__ __forceinline__ void block_linear(const float* __restrict__ W, const float* __restrict__ b,
                                             const float* in, int ld_in, float* out, int ld_out,
                                             int n_in, int n_out, int R, bool act) {
  staged_linear(W, b, in, ld_in, out, ld_out, n_in, n_out, R, act);
}

// Whole MLP.  `in` holds the input [R][ld_in]; bufA/bufB are ping-pong scratch [R][ldw];
// the last layer writes to `dst` [R][ld_dst].
__device__ __forceinline__ void block_mlp(const float* __restrict__ flat, const NetDesc& nd,
                                          const float* in, int ld_in, float* bufA, float* bufB,
                                          int ldw, float* dst, int ld_dst, int R) {
  const float* cur = in;
  int ld_cur = ld_in;
  for (int l = 0; l < nd.n_lin; ++l) {
    const bool last = (l == nd.n_lin - 1);
    float* o = last ? dst : ((l & 1) ? bufB : bufA);
    const int ld_o = last ? ld_dst : ldw;
    block_linear(flat + nd.w_off[l], flat + nd.b_off[l], cur, ld_cur, o, ld_o, nd.sizes[l],
                 nd.sizes[l + 1], R, !last);
    cur = o;
    ld_cur = ld_o;
  }
}

// Small parameter sets (<= kParamLdsFloats) are copied into LDS once per workgroup so the fmaf
// chains read weights at LDS latency instead of L2 latency; `w_lds` = floats to stage (0 = none).
constexpr int kParamLdsFloats = 8192;

__device__ __forceinline__ const float* stage_params(const float* __restrict__ flat, float* lds_dst, int n) {
  if (n <= 0) return flat;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds_dst[i] = flat[i];
  __syncthreads();
  return lds_dst;
}

// rows per workgroup: fill the chip first (>= 2 workgroups per CU), then grow up to `cap`
static int pick_rows_per_group(int64_t rows, int cap) {
  int64_t r = (rows + 511) / 512;
  if (r < 4) r = 4;
  if (r > cap) r = cap;
  return (int)r;
}

// ------------------------------------------------------------------ AffineHalfFlow
struct AhfArgs {
  const float* x;
  float* y;
  float* log_det;
  const float* flat;
  int64_t rows;
  int dim, parity, inverse, accumulate, has_scale, has_shift;
  int R;    // rows per workgroup
  int ldw;  // scratch row stride (max hidden width)
  int w_lds;
  NetDesc s_net, t_net;
};

__global__ void __launch_bounds__(kThreads) ahf_generic_kernel(AhfArgs a) {
  const int H = a.dim / 2;
  const int64_t row0 = (int64_t)blockIdx.x * a.R;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* cond = smem;                 // [R][H]
  float* s_out = cond + a.R * H;      // [R][H]
  float* t_out = s_out + a.R * H;     // [R][H]
  float* bufA = t_out + a.R * H;      // [R][ldw]
  float* bufB = bufA + a.R * a.ldw;   // [R][ldw]
  const int cond_off = a.parity ? H : 0, act_off = a.parity ? 0 : H;
  const float* flat = stage_params(a.flat, bufB + a.R * a.ldw, a.w_lds);

  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const float c = a.x[(row0 + r) * a.dim + cond_off + j];
    cond[idx] = c;
    a.y[(row0 + r) * a.dim + cond_off + j] = c;  // untouched half keeps its place
    s_out[idx] = 0.f;                            // scale=False / shift=False -> zeros (:38)
    t_out[idx] = 0.f;
  }
  __syncthreads();
  if (a.has_scale) block_mlp(flat, a.s_net, cond, H, bufA, bufB, a.ldw, s_out, H, R);
  if (a.has_shift) block_mlp(flat, a.t_net, cond, H, bufA, bufB, a.ldw, t_out, H, R);

  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const float v = a.x[(row0 + r) * a.dim + act_off + j];
    const float s = s_out[idx], t = t_out[idx];
    a.y[(row0 + r) * a.dim + act_off + j] = a.inverse ? (v - t) / expf(s) : expf(s) * v + t;
  }
  if (a.log_det) {
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      float acc = 0.f;
      for (int j = 0; j < H; ++j) acc += a.inverse ? -s_out[r * H + j] : s_out[r * H + j];
      float* p = a.log_det + row0 + r;
      *p = a.accumulate ? *p + acc : acc;
    }
  }
}

// -------------------------------------------------------------------------- NSF_CL
struct NsfArgs {
  const float* x;
  float* y;
  float* log_det;
  const float* flat;
  int64_t rows;
  int dim, K, inverse, accumulate;
  float T;
  int R, ldw, ldp;  // ldp = (3K-1)*H, the spline-parameter row stride
  int w_lds;
  NetDesc f1, f2;
};

// spline for all (row, element) pairs of one half; params [R][ldp] in LDS
__device__ __forceinline__ void block_spline(const float* params, int ldp, float* vals /*[R][H] in/out*/,
                                             float* lad_sum /*[R]*/, float* lad_tmp /*[R][H]*/, int H,
                                             int K, float T, bool inverse, int R) {
  const int P = 3 * K - 1;
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    const float* p = params + r * ldp + j * P;
    float out, lad;
    rqs_element<true>(
        vals[idx], K, T, inverse, [&](int k) { return p[k]; }, [&](int k) { return p[K + k]; },
        [&](int k) { return p[2 * K + k]; }, out, lad);
    vals[idx] = out;
    lad_tmp[idx] = lad;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < H; ++j) acc += lad_tmp[r * H + j];
    lad_sum[r] += acc;  // log_det += sum(ld, dim=1) per half-step (spline_flow.py:258,265)
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kThreads) nsf_generic_kernel(NsfArgs a) {
  const int H = a.dim / 2;
  const int64_t row0 = (int64_t)blockIdx.x * a.R;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* lower = smem;                  // [R][H]
  float* upper = lower + a.R * H;       // [R][H]
  float* lad_tmp = upper + a.R * H;     // [R][H]
  float* lad_sum = lad_tmp + a.R * H;   // [R]
  float* bufA = lad_sum + a.R;          // [R][ldw]
  float* bufB = bufA + a.R * a.ldw;     // [R][ldw]
  float* params = bufB + a.R * a.ldw;   // [R][ldp]
  const float* flat = stage_params(a.flat, params + a.R * a.ldp, a.w_lds);

  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    lower[idx] = a.x[(row0 + r) * a.dim + j];
    upper[idx] = a.x[(row0 + r) * a.dim + H + j];
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) lad_sum[r] = 0.f;
  __syncthreads();

  if (!a.inverse) {  // f1(lower) moves upper, then f2(upper') moves lower (:249-266)
    block_mlp(flat, a.f1, lower, H, bufA, bufB, a.ldw, params, a.ldp, R);
    block_spline(params, a.ldp, upper, lad_sum, lad_tmp, H, a.K, a.T, false, R);
    block_mlp(flat, a.f2, upper, H, bufA, bufB, a.ldw, params, a.ldp, R);
    block_spline(params, a.ldp, lower, lad_sum, lad_tmp, H, a.K, a.T, false, R);
  } else {  // (:268-285)
    block_mlp(flat, a.f2, upper, H, bufA, bufB, a.ldw, params, a.ldp, R);
    block_spline(params, a.ldp, lower, lad_sum, lad_tmp, H, a.K, a.T, true, R);
    block_mlp(flat, a.f1, lower, H, bufA, bufB, a.ldw, params, a.ldp, R);
    block_spline(params, a.ldp, upper, lad_sum, lad_tmp, H, a.K, a.T, true, R);
  }
  for (int idx = threadIdx.x; idx < R * H; idx += blockDim.x) {
    const int r = idx / H, j = idx - r * H;
    a.y[(row0 + r) * a.dim + j] = lower[idx];
    a.y[(row0 + r) * a.dim + H + j] = upper[idx];
  }
  if (a.log_det) {
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      float* p = a.log_det + row0 + r;
      *p = a.accumulate ? *p + lad_sum[r] : lad_sum[r];
    }
  }
}

// -------------------------------------------------------------------------- NSF_AR
// Autoregressive spline layer (flows/spline_flow.py:182-235): element i is transformed by a spline whose 3K-1
// parameters come from MLP_i(first i elements) -- of the layer's OUTPUT in `forward` (sequential in i, :201-218) and of
// its INPUT in `inverse` (:220-235); element 0 uses the learned `init_param`.  forward runs the spline inverted
// (unconstrained_RQS(..., inverse=True)), inverse runs it forward.  Same double normalisation as NSF_CL.
// flat: init_param (3K-1), then for i = 1 .. dim-1 the state_dict tensors of layers[i-1] = MLP(i, hidden..., 3K-1).
struct NsfArArgs {
  const float* x;
  float* y;
  float* log_det;
  const float* flat;
  int64_t rows;
  int dim, K, inverse, accumulate;
  float T;
  int R, ldw;
  int n_hidden;
  int hidden[MNF_MAX_LINEAR];
};

// the conditioner of element i >= 1: MLP(i, hidden..., 3K-1) at its offset inside flat
__host__ __device__ inline int64_t nsf_ar_net_floats(int i, int n_hidden, const int* hidden, int P) {
  int64_t n = 0;
  int prev = i;
  for (int l = 0; l < n_hidden; ++l) {
    n += (int64_t)prev * hidden[l] + hidden[l];
    prev = hidden[l];
  }
  return n + (int64_t)prev * P + P;
}
__host__ __device__ inline void nsf_ar_net(NetDesc& nd, int i, int n_hidden, const int* hidden, int P) {
  // offset of layers[i-1]: P + sum_{k=1}^{i-1} floats(k); floats(k) = k h0 + C  (h0 = first width after the input)
  const int h0 = n_hidden > 0 ? hidden[0] : P;
  const int64_t C = nsf_ar_net_floats(0, n_hidden, hidden, P);
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

__global__ void __launch_bounds__(kThreads) nsf_ar_generic_kernel(NsfArArgs a) {
  const int d = a.dim, P = 3 * a.K - 1;
  const int64_t row0 = (int64_t)blockIdx.x * a.R;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* src = smem;                  // [R][d]  the layer's input
  float* dst = src + a.R * d;         // [R][d]  its output, element by element
  float* lad_sum = dst + a.R * d;     // [R]
  float* bufA = lad_sum + a.R;        // [R][ldw]
  float* bufB = bufA + a.R * a.ldw;   // [R][ldw]
  float* params = bufB + a.R * a.ldw; // [R][P]
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) src[idx] = a.x[row0 * d + idx];
  for (int r = threadIdx.x; r < R; r += blockDim.x) lad_sum[r] = 0.f;
  __syncthreads();
  const float* cond = a.inverse ? src : dst;  // what the conditioners see (:209 / :228)
  for (int i = 0; i < d; ++i) {
    if (i == 0) {
      for (int idx = threadIdx.x; idx < R * P; idx += blockDim.x) params[idx] = a.flat[idx % P];  // init_param
      __syncthreads();
    } else {
      NetDesc nd;
      nsf_ar_net(nd, i, a.n_hidden, a.hidden, P);
      block_mlp(a.flat, nd, cond, d, bufA, bufB, a.ldw, params, P, R);
    }
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      const float* p = params + r * P;
      float out, lad;
      rqs_element<true>(
          src[r * d + i], a.K, a.T, !a.inverse, [&](int k) { return p[k]; }, [&](int k) { return p[a.K + k]; },
          [&](int k) { return p[2 * a.K + k]; }, out, lad);
      dst[r * d + i] = out;
      lad_sum[r] += lad;
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) a.y[row0 * d + idx] = dst[idx];
  if (a.log_det) {
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      float* p = a.log_det + row0 + r;
      *p = a.accumulate ? *p + lad_sum[r] : lad_sum[r];
    }
  }
}

// elementwise unconstrained_RQS on caller-supplied (W, H, D)
__global__ void rqs_kernel(const float* __restrict__ v, const float* __restrict__ W,
                           const float* __restrict__ Hh, const float* __restrict__ D,
                           float* __restrict__ out, float* __restrict__ lad, int64_t n, int K, float T,
                           int inverse) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* w = W + i * K;
  const float* h = Hh + i * K;
  const float* d = D + i * (K - 1);
  float o, l;
  rqs_element<false>(
      v[i], K, T, inverse != 0, [&](int k) { return w[k]; }, [&](int k) { return h[k]; },
      [&](int k) { return d[k]; }, o, l);
  out[i] = o;
  lad[i] = l;
}

// ------------------------------------------------------------------- RNVP (gated)
struct RnvpArgs {
  const float* z;
  const float* mask;
  float* x;
  float* log_det;
  const float* flat;
  int64_t rows;
  int dim, accumulate;
  int R, ldw;
  int t_w, t_b, s_w, s_b;  // float offsets of t.weight, t.bias, s.weight, s.bias
  uint64_t seed;           // used when mask == nullptr
  int w_lds;
  NetDesc net;
};

__global__ void __launch_bounds__(kThreads) rnvp_generic_kernel(RnvpArgs a) {
  const int d = a.dim, hl = a.net.sizes[a.net.n_lin];
  const int64_t row0 = (int64_t)blockIdx.x * a.R;
  const int R = (int)min((int64_t)a.R, a.rows - row0);
  float* kept = smem;                // [R][d]  mask * z
  float* y = kept + a.R * d;         // [R][hl]
  float* lad = y + a.R * hl;         // [R][d]
  float* bufA = lad + a.R * d;       // [R][ldw]
  float* bufB = bufA + a.R * a.ldw;  // [R][ldw]
  const float* flat = stage_params(a.flat, bufB + a.R * a.ldw, a.w_lds);

  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) {
    const int r = idx / d, j = idx - r * d;
    const int64_t g = (row0 + r) * d + j;
    const float m = a.mask ? a.mask[g] : rnvp_mask_bit(a.seed, row0 + r, j);
    kept[idx] = m * a.z[g];
  }
  __syncthreads();
  block_mlp(flat, a.net, kept, d, bufA, bufB, a.ldw, y, hl, R);

  const float* Wt = flat + a.t_w;
  const float* bt = flat + a.t_b;
  const float* Ws = flat + a.s_w;
  const float* bs = flat + a.s_b;
  for (int idx = threadIdx.x; idx < R * d; idx += blockDim.x) {
    const int r = idx / d, j = idx - r * d;
    const int64_t g = (row0 + r) * d + j;
    const float* yr = y + r * hl;
    float shift = bt[j], scale = bs[j];
    for (int k = 0; k < hl; ++k) {
      shift = fmaf(Wt[(size_t)j * hl + k], yr[k], shift);
      scale = fmaf(Ws[(size_t)j * hl + k], yr[k], scale);
    }
    const float m = a.mask ? a.mask[g] : rnvp_mask_bit(a.seed, row0 + r, j), zz = a.z[g];
    const float gate = sigmoidf(scale);
    // x = z1*gate + (1-gate)*shift + z2, every position (rnvp.py:37)
    a.x[g] = ((1.f - m) * zz * gate + (1.f - gate) * shift) + m * zz;
    lad[idx] = (1.f - m) * logf(gate);  // 0 * -inf = NaN, as in the reference (:36)
  }
  __syncthreads();
  if (a.log_det) {
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      float acc = 0.f;
      for (int j = 0; j < d; ++j) acc += lad[r * d + j];
      float* p = a.log_det + row0 + r;
      *p = a.accumulate ? *p + acc : acc;
    }
  }
}

// ----------------------------------------------------------- data-independent layers
__global__ void sum_vec_kernel(const float* __restrict__ s, int dim, int negate, float* __restrict__ out) {
  // one wave; sequential-per-lane then shuffle tree (order fixed -> deterministic)
  float acc = 0.f;
  for (int j = threadIdx.x; j < dim; j += 64) acc += s[j];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) *out = negate ? -acc : acc;
}

__global__ void affine_const_kernel(const float* __restrict__ x, float* __restrict__ y,
                                    const float* __restrict__ s, const float* __restrict__ t,
                                    int64_t n, int dim, int inverse) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int j = (int)(i % dim);
    y[i] = inverse ? (x[i] - t[j]) * expf(-s[j]) : x[i] * expf(s[j]) + t[j];
  }
}

// float4 variant (dim % 4 == 0, 16-byte aligned rows): each thread owns 4 consecutive dims
__global__ void affine_const_kernel_v4(const float4* __restrict__ x, float4* __restrict__ y,
                                       const float* __restrict__ s, const float* __restrict__ t, int64_t n4,
                                       int dim4, int inverse) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int j = (int)(i % dim4) * 4;
    const float4 v = x[i];
    const float4 sv = *reinterpret_cast<const float4*>(s + j), tv = *reinterpret_cast<const float4*>(t + j);
    float4 o;
    if (inverse) {
      o.x = (v.x - tv.x) * expf(-sv.x); o.y = (v.y - tv.y) * expf(-sv.y);
      o.z = (v.z - tv.z) * expf(-sv.z); o.w = (v.w - tv.w) * expf(-sv.w);
    } else {
      o.x = v.x * expf(sv.x) + tv.x; o.y = v.y * expf(sv.y) + tv.y;
      o.z = v.z * expf(sv.z) + tv.z; o.w = v.w * expf(sv.w) + tv.w;
    }
    y[i] = o;
  }
}

__global__ void add_scalar_rows_kernel(float* __restrict__ log_det, const float* __restrict__ scalar,
                                       int64_t rows, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float v = *scalar;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride)
    log_det[i] = accumulate ? log_det[i] + v : v;
}

// y = x @ W for small dim: W in LDS, R rows per workgroup, thread per (row, column)
__global__ void __launch_bounds__(kThreads) linear_rows_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ W,
                                                               float* __restrict__ y, int64_t rows,
                                                               int dim, int R, int w_lds) {
  // W is staged in LDS when it fits (w_lds != 0); otherwise it is read through L1/L2
  const float* Wl = W;
  float* xl = smem;
  if (w_lds) {
    float* Ws = smem;           // [dim][dim]
    xl = Ws + dim * dim;        // [R][dim]
    for (int i = threadIdx.x; i < dim * dim; i += blockDim.x) Ws[i] = W[i];
    Wl = Ws;
  }
  const int64_t n_groups = (rows + R - 1) / R;
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t row0 = grp * R;
    const int Rn = (int)min((int64_t)R, rows - row0);
    __syncthreads();
    for (int i = threadIdx.x; i < Rn * dim; i += blockDim.x) xl[i] = x[row0 * dim + i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < Rn * dim; idx += blockDim.x) {
      const int r = idx / dim, j = idx - r * dim;
      float acc = 0.f;
      for (int k = 0; k < dim; ++k) acc = fmaf(xl[r * dim + k], Wl[k * dim + j], acc);
      y[row0 * dim + idx] = acc;
    }
  }
}

// ------------------------------------------------------------- base log-prob epilogue
// One wave per row group; lanes stride the row, shuffle-reduce |z|^2, fp64 block sum.
__global__ void __launch_bounds__(kThreads) gauss_logprob_kernel(const float* __restrict__ z,
                                                                 const float* __restrict__ log_det,
                                                                 float* __restrict__ log_prob,
                                                                 double* __restrict__ sum_out,
                                                                 int64_t rows, int dim) {
  __shared__ double wave_sums[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * (kThreads / 64);
  const float cst = (float)dim * kHalfLog2Pi;
  double local = 0.0;
  // lanes_per_row = smallest power of two >= dim/4 capped at 64; rows_per_wave = 64 / lanes_per_row
  int lpr = 1;
  while (lpr < 64 && lpr * 4 < dim) lpr <<= 1;
  const int rpw = 64 / lpr;
  const int sub = lane / lpr, l = lane % lpr;
  for (int64_t base = ((int64_t)blockIdx.x * (kThreads / 64) + wave) * rpw; base < rows;
       base += n_waves * rpw) {
    const int64_t row = base + sub;
    float acc = 0.f;
    if (row < rows) {
      const float* zr = z + row * dim;
      for (int j = l; j < dim; j += lpr) acc = fmaf(zr[j], zr[j], acc);
    }
    for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (row < rows && l == 0) {
      const float lp = (log_det ? log_det[row] : 0.f) + (-0.5f * acc - cst);
      if (log_prob) log_prob[row] = lp;
      local += (double)lp;
    }
  }
  if (sum_out) {
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) wave_sums[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) tot += wave_sums[w];
      atomicAdd(sum_out, tot);
    }
  }
}

// log-prob from the per-row |z|^2 the last coupling kernel already produced
__global__ void __launch_bounds__(kThreads) gauss_logprob_sq_kernel(const float* __restrict__ zsq,
                                                                    const float* __restrict__ log_det,
                                                                    float* __restrict__ log_prob,
                                                                    double* __restrict__ sum_out,
                                                                    int64_t rows, int dim) {
  __shared__ double wave_sums[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float cst = (float)dim * kHalfLog2Pi;
  double local = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
    const float lp = (log_det ? log_det[r] : 0.f) + (-0.5f * zsq[r] - cst);
    if (log_prob) log_prob[r] = lp;
    local += (double)lp;
  }
  if (sum_out) {
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) wave_sums[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) tot += wave_sums[w];
      atomicAdd(sum_out, tot);
    }
  }
}

// torch.optim.Adam's update (no amsgrad, L2 weight decay folded into the gradient) over one flat buffer
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float step_size, float beta1, float beta2, float eps,
                            float weight_decay, float inv_sqrt_bc2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
    const float mi = fmaf(beta1, m[i], (1.f - beta1) * gi);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(beta2, v[i], (1.f - beta2) * gi * gi);     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);   // denom = sqrt(v) / sqrt(bc2) + eps
  }
}

// the step counter of a graph-captured optimizer lives on the device: state = {step, 1 / (1 - beta1^step),
// 1 / sqrt(1 - beta2^step)}; one thread advances it, the update kernel behind it reads the two factors
__global__ void adam_advance_kernel(float* __restrict__ state, float beta1, float beta2) {
  const double step = (double)state[0] + 1.0;
  state[0] = (float)step;
  state[1] = (float)(1.0 / (1.0 - pow((double)beta1, step)));
  state[2] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, step)));
}
__global__ void adam_state_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, const float* __restrict__ state) {
  const float step_size = lr * state[1], inv_sqrt_bc2 = state[2];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
    const float mi = fmaf(beta1, m[i], (1.f - beta1) * gi);
    const float vi = fmaf(beta2, v[i], (1.f - beta2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

__global__ void sample_z0_kernel(const float* __restrict__ mean, const float* __restrict__ log_var,
                                 const float* __restrict__ eps, float* __restrict__ z0, int64_t n,
                                 int dim) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int j = (int)(i % dim);
    z0[i] = mean[j] + sqrtf(expf(log_var[j])) * eps[i];  // mnf_linear.py:59-62
  }
}
// the same on 16-byte pieces (dim % 4 == 0, 16-byte aligned buffers): 1.64 GB of eps in, z0 out at 256,000 x 800
__global__ void sample_z0_kernel_v4(const float* __restrict__ mean, const float* __restrict__ log_var,
                                    const float4* __restrict__ eps, float4* __restrict__ z0, int64_t n4, int dim4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int j = 4 * (int)(i % dim4);
    const float4 e = eps[i];
    float4 o;
    o.x = fmaf(sqrtf(expf(log_var[j])), e.x, mean[j]);
    o.y = fmaf(sqrtf(expf(log_var[j + 1])), e.y, mean[j + 1]);
    o.z = fmaf(sqrtf(expf(log_var[j + 2])), e.z, mean[j + 2]);
    o.w = fmaf(sqrtf(expf(log_var[j + 3])), e.w, mean[j + 3]);
    z0[i] = o;
  }
}

// the prologue with its noise generated in place: eps[r][j] = z0_normal(seed, r, j) (mnf_device.h; what
// mnf_sample_z0_noise(seed, ., rows, dim) writes) -- no (rows, dim) noise tensor is drawn, stored or read back: 3.3 GB less
// traffic per training step of MNFLinear(800, .) at 256,000 rows (the draw, this launch's read, the gradient launch's read)
// blockDim = (64 lanes, 4 row lanes), like the gradient kernel below: a lane owns VEC consecutive dims -- their mean and
// standard deviation are formed once -- and walks the rows blockIdx.y * 4 + threadIdx.y, + 4 gridDim.y, ...: no index
// division and no exp / sqrt inside the loop (the first version, one flat index per piece, took 0.30 ms for a 0.14 ms store)
template <int VEC>
__global__ void __launch_bounds__(256) sample_z0_seeded_kernel(const float* __restrict__ mean, const float* __restrict__ log_var,
                                                               uint64_t seed, float* __restrict__ z0, int64_t rows, int dim) {
  const int j0 = (blockIdx.x * 64 + threadIdx.x) * VEC;
  if (j0 >= dim) return;  // (VEC = 4: dim % 4 == 0, so the whole piece is inside)
  float mu[VEC], sd[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    mu[v] = mean[j0 + v];
    sd[v] = sqrtf(expf(log_var[j0 + v]));  // mnf_linear.py:60
  }
  for (int64_t r = (int64_t)blockIdx.y * 4 + threadIdx.y; r < rows; r += (int64_t)gridDim.y * 4) {
    if (VEC == 4) {  // (j0 % 4 == 0: the piece is two whole column pairs)
      float e[4];
      const uint32_t rh = z0_row_hash(seed, r);
      z0_normal_pair(rh, (uint32_t)seed, j0 >> 1, e[0], e[1]);
      z0_normal_pair(rh, (uint32_t)seed, (j0 >> 1) + 1, e[2], e[3]);
      float4 o;
      o.x = fmaf(sd[0], e[0], mu[0]);
      o.y = fmaf(sd[VEC > 1 ? 1 : 0], e[1], mu[VEC > 1 ? 1 : 0]);
      o.z = fmaf(sd[VEC > 2 ? 2 : 0], e[2], mu[VEC > 2 ? 2 : 0]);
      o.w = fmaf(sd[VEC > 3 ? 3 : 0], e[3], mu[VEC > 3 ? 3 : 0]);
      *reinterpret_cast<float4*>(z0 + r * dim + j0) = o;
    } else {
      z0[r * dim + j0] = fmaf(sd[0], z0_normal(seed, r, j0), mu[0]);
    }
  }
}
__global__ void sample_z0_noise_kernel(uint64_t seed, float* __restrict__ eps, int64_t rows, int dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * dim) eps[i] = z0_normal(seed, i / dim, (int)(i % dim));
}

// gradients of the prologue: d mean[j] = sum_r g[r][j];  d log_var[j] = sum_r g[r][j] eps[r][j] * 0.5 sqrt(exp(log_var[j])).
// blockDim = (64 lanes, 4 row lanes); a lane owns VEC consecutive dims (VEC = 4: 16-byte loads, a wave reads 1 KB of a
// row at a time -- with 4-byte loads the 1.64 GB pass ran at half the HBM rate); a workgroup takes 64 VEC dims and the
// rows blockIdx.y, blockIdx.y + gridDim.y, ... in steps of 4, sums in registers, then over its 4 row lanes in LDS, and
// adds one value per dim to the outputs.
// eps == nullptr: the noise is regenerated from `seed` (the forward launch was mnf_sample_z0_seeded)
template <int VEC>
__global__ void __launch_bounds__(256) sample_z0_bwd_kernel(const float* __restrict__ g, const float* __restrict__ eps,
                                                            const float* __restrict__ log_var,
                                                            float* __restrict__ g_mean, float* __restrict__ g_log_var,
                                                            int64_t rows, int dim, int atomic, uint64_t seed,
                                                            float* __restrict__ blocks) {
  __shared__ float part[2][4][64 * VEC];
  const int j0 = (blockIdx.x * 64 + threadIdx.x) * VEC;
  float sm[VEC], sv[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) sm[v] = sv[v] = 0.f;
  if (j0 < dim) {  // (VEC = 4: dim % 4 == 0, so the whole piece is inside)
    for (int64_t r = (int64_t)blockIdx.y * 4 + threadIdx.y; r < rows; r += (int64_t)gridDim.y * 4) {
      float gv[VEC], ev[VEC];
      if (VEC == 4) {
        const float4 a = *reinterpret_cast<const float4*>(g + r * dim + j0);
        gv[0] = a.x, gv[VEC > 1 ? 1 : 0] = a.y, gv[VEC > 2 ? 2 : 0] = a.z, gv[VEC > 3 ? 3 : 0] = a.w;
        if (eps) {
          const float4 b = *reinterpret_cast<const float4*>(eps + r * dim + j0);
          ev[0] = b.x, ev[VEC > 1 ? 1 : 0] = b.y, ev[VEC > 2 ? 2 : 0] = b.z, ev[VEC > 3 ? 3 : 0] = b.w;
        }
      } else {
        gv[0] = g[r * dim + j0];
        if (eps) ev[0] = eps[r * dim + j0];
      }
      if (!eps) {
        if (VEC == 4) {
          const uint32_t rh = z0_row_hash(seed, r);
          z0_normal_pair(rh, (uint32_t)seed, j0 >> 1, ev[0], ev[VEC > 1 ? 1 : 0]);
          z0_normal_pair(rh, (uint32_t)seed, (j0 >> 1) + 1, ev[VEC > 2 ? 2 : 0], ev[VEC > 3 ? 3 : 0]);
        } else {
          ev[0] = z0_normal(seed, r, j0);
        }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        sm[v] += gv[v];
        sv[v] = fmaf(gv[v], ev[v], sv[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    part[0][threadIdx.y][threadIdx.x * VEC + v] = sm[v];
    part[1][threadIdx.y][threadIdx.x * VEC + v] = sv[v];
  }
  __syncthreads();
  for (int c = threadIdx.y * 64 + threadIdx.x; c < 64 * VEC; c += 256) {
    const int j = blockIdx.x * 64 * VEC + c;
    if (j >= dim) continue;
    const float m = (part[0][0][c] + part[0][1][c]) + (part[0][2][c] + part[0][3][c]);
    float w = (part[1][0][c] + part[1][1][c]) + (part[1][2][c] + part[1][3][c]);
    w *= 0.5f * sqrtf(expf(log_var[j]));
    if (blocks) {  // the row block's sums as its own block [mean | log_var]: det_reduce_async adds them in order
      blocks[(int64_t)blockIdx.y * 2 * dim + j] = m;
      blocks[(int64_t)blockIdx.y * 2 * dim + dim + j] = w;
    } else if (atomic) {
      atomicAdd(g_mean + j, m);
      atomicAdd(g_log_var + j, w);
    } else {  // one workgroup per dim block: plain adds, results repeat bit for bit
      g_mean[j] += m;
      g_log_var[j] += w;
    }
  }
}

// blockIdx.y = image number: the k-th image is gathered from flat + k * flat_stride (layers of one shape whose
// parameters sit back to back share the index table)
__global__ void pack_gather_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx,
                                   float* __restrict__ image, int64_t n, int64_t flat_stride) {
  flat += (int64_t)blockIdx.y * flat_stride;
  image += (int64_t)blockIdx.y * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t s = idx[i];
    image[i] = s >= 0 ? flat[s] : (s == kPackBigBias ? kPackBigBiasValue : 0.f);
  }
}

// ---------------------------------------------------------------- pack: flat fp32 -> split image
__global__ void __launch_bounds__(1024) pack_gather_split_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx,
                                         uint32_t* __restrict__ image, int64_t n_split, int64_t n_plain,
                                         int64_t flat_stride) {
  flat += (int64_t)blockIdx.y * flat_stride;
  image += (int64_t)blockIdx.y * (n_split + n_plain + MNF_SPLIT_TAIL_WORDS);
  float mx = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_split + n_plain; w += stride) {
    if (w < n_split) {
      uint32_t word = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int32_t e = idx[2 * w + h];
        if (e < 0) continue;
        const float v = flat[e & (kSplitLoBit - 1)];
        const _Float16 hi = (_Float16)v;
        const _Float16 part = (e & kSplitLoBit) ? (_Float16)((v - (float)hi) * kSplitScale) : hi;
        word |= (uint32_t)__builtin_bit_cast(uint16_t, part) << (16 * h);
        mx = fmaxf(mx, fabsf(v));  // NaN weights: fmaxf drops them here, the MFMAs propagate them
        if (!(fabsf(v) <= 3.0e38f)) mx = __builtin_inff();  // inf or NaN weight: always take the fp32 path
      }
      image[w] = word;
    } else {
      const int32_t e = idx[2 * n_split + (w - n_split)];
      image[w] = __builtin_bit_cast(uint32_t, e >= 0 ? flat[e] : (e == kPackBigBias ? kPackBigBiasValue : 0.f));
    }
  }
  // ONE atomic per workgroup: the maxima all go to one address and same-address atomics serialise at the memory side
  // (~7 ns each: with one per wave, 1,600 of them were 11 of the launch's 16 us at a 100 k-word image; with one per
  // thread no better).  Non-negative floats order like their bit patterns.
  __shared__ float wave_max[16];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) mx = fmaxf(mx, wave_max[wv]);
    if (mx > 0.f) atomicMax(image + n_split + n_plain, __builtin_bit_cast(uint32_t, mx));
  }
}

}  // namespace mnf

// =====================================================================================
// host side: argument checking, descriptors, launches
// =====================================================================================
using namespace mnf;

namespace mnf {
__global__ void zero_word_kernel(uint32_t* w) { *w = 0u; }
__global__ void zero_tails_kernel(uint32_t* tail0, int64_t image_words) {
  if (threadIdx.x < MNF_SPLIT_TAIL_WORDS) tail0[(int64_t)blockIdx.x * image_words + threadIdx.x] = 0u;
}
thread_local int g_last_hip_error = 0;
std::atomic<const char*> g_last_kernel{""};

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return MNF_ERR_LAUNCH;
  }
  return MNF_OK;
}

int zero_word_async(void* word, hipStream_t stream) {
  hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(1), 0, stream, static_cast<uint32_t*>(word));
  return check_launch();
}

bool deterministic() {
  static const bool on = [] {
    const char* e = getenv("MNF_DETERMINISTIC");
    return e != nullptr && e[0] != '\0' && strcmp(e, "0") != 0;
  }();
  return on;
}

__global__ void __launch_bounds__(256) zero_floats_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}
int zero_floats_async(float* p, int64_t n, hipStream_t stream) {
  if (n <= 0) return MNF_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p, n);
  return check_launch();
}

// one thread per parameter, the row parts in order: every run adds the same numbers in the same order
__global__ void __launch_bounds__(256)
det_reduce_kernel(const float* __restrict__ part, int n_rows, int64_t stride, int64_t count, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int r = 0; r < n_rows; ++r) s += part[(int64_t)r * stride + i];
  dst[i] += s;
}
int det_reduce_async(const float* part, int n_rows, int64_t stride, int64_t count, float* dst, hipStream_t stream) {
  if (count <= 0 || n_rows <= 0) return MNF_OK;
  hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, part, n_rows, stride,
                     count, dst);
  return check_launch();
}

// det_sort_ids_async: ONE workgroup; a bit per possible id in LDS (ids are distinct), then the set bits written back in
// ascending order.
constexpr int kDetSortThreads = 1024;
constexpr int64_t kDetSortMaxIds = 64 * 1024 * 8;  // 64 KB of bits
__global__ void __launch_bounds__(kDetSortThreads)
det_sort_ids_kernel(int32_t* __restrict__ ids, const int32_t* __restrict__ count, int capacity, int n_words) {
  extern __shared__ uint32_t sort_bits[];
  __shared__ int chunk_sum[kDetSortThreads];
  int n = count[0];
  if (n > capacity) n = capacity;
  if (n < 2) return;  // (negative: the "every item" marker of the tile lists)
  for (int i = threadIdx.x; i < n_words; i += blockDim.x) sort_bits[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t id = (uint32_t)ids[i];
    if ((int)(id >> 5) < n_words) atomicOr(&sort_bits[id >> 5], 1u << (id & 31));
  }
  __syncthreads();
  // thread t owns the words [t * per, (t + 1) * per): its ids land behind those of the threads before it
  const int per = (n_words + (int)blockDim.x - 1) / (int)blockDim.x;
  const int w0 = threadIdx.x * per, w1 = w0 + per < n_words ? w0 + per : n_words;
  int mine = 0;
  for (int w = w0; w < w1; ++w) mine += __popc(sort_bits[w]);
  chunk_sum[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < (int)blockDim.x; ++t) {
      const int c = chunk_sum[t];
      chunk_sum[t] = run;
      run += c;
    }
  }
  __syncthreads();
  int at = chunk_sum[threadIdx.x];
  for (int w = w0; w < w1; ++w) {
    uint32_t bits = sort_bits[w];
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      ids[at++] = w * 32 + b;
    }
  }
}

int det_sort_ids_async(int32_t* ids, const int32_t* count, int capacity, int64_t n_items, hipStream_t stream) {
  if (capacity < 2 || n_items < 2) return MNF_OK;
  if (n_items > kDetSortMaxIds) return MNF_ERR_UNSUPPORTED;
  const int n_words = (int)((n_items + 31) / 32);
  static DeviceMemo attr;
  attr.get([&](int) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(det_sort_ids_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              64 * 1024);
    return 1;
  });
  hipLaunchKernelGGL(det_sort_ids_kernel, dim3(1), dim3(kDetSortThreads), (size_t)n_words * 4, stream, ids, count, capacity,
                     n_words);
  return check_launch();
}

// Fill a NetDesc for MLP(sizes...) whose parameters start at float offset `base` of the flat
// buffer (weight then bias per Linear, state_dict order).  Returns floats consumed.
int64_t fill_net(NetDesc& nd, int n_sizes, const int* sizes, int64_t base) {
  nd.n_lin = n_sizes - 1;
  nd.max_width = 0;
  int64_t off = base;
  for (int i = 0; i < n_sizes; ++i) {
    nd.sizes[i] = sizes[i];
    if (sizes[i] > nd.max_width) nd.max_width = sizes[i];
  }
  for (int l = 0; l < nd.n_lin; ++l) {
    nd.w_off[l] = (int)off;
    off += (int64_t)sizes[l] * sizes[l + 1];
    nd.b_off[l] = (int)off;
    off += sizes[l + 1];
  }
  return off - base;
}

bool hidden_ok(int n_hidden, const int* hidden) {
  if (n_hidden < 0 || n_hidden + 1 > MNF_MAX_LINEAR) return false;
  if (n_hidden > 0 && !hidden) return false;
  for (int i = 0; i < n_hidden; ++i)
    if (hidden[i] <= 0) return false;
  return true;
}

static int grid_for(int64_t n, int threads, int cap = 256 * 8) {
  int64_t g = (n + threads - 1) / threads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
}  // namespace mnf

extern "C" {

int mnf_abi_version(void) { return MNF_ABI_VERSION; }

const char* mnf_error_string(int code) {
  switch (code) {
    case MNF_OK: return "ok";
    case MNF_ERR_INVALID_ARG: return "invalid argument";
    case MNF_ERR_UNSUPPORTED: return "shape not supported by the HIP kernels";
    case MNF_ERR_LAUNCH: return "HIP kernel launch failed";
    case MNF_ERR_NO_DEVICE: return "no gfx950 device";
    case MNF_ERR_DOMAIN: return "minimal bin width/height too large for the number of bins";
    default: return "unknown error";
  }
}

int mnf_last_hip_error(void) { return g_last_hip_error; }
const char* mnf_last_kernel(void) { return g_last_kernel.load(std::memory_order_relaxed); }
int mnf_deterministic(void) { return deterministic() ? 1 : 0; }

int mnf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

// ------------------------------------------------------------------ AffineHalfFlow
int64_t mnf_affine_half_flat_floats(int dim, int n_hidden, const int* hidden, int has_scale,
                                    int has_shift) {
  if (dim < 2 || (dim & 1) || !hidden_ok(n_hidden, hidden)) return -1;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = dim / 2;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  sizes[n_hidden + 1] = dim / 2;
  NetDesc nd;
  const int64_t one = fill_net(nd, n_hidden + 2, sizes, 0);
  return one * ((has_scale ? 1 : 0) + (has_shift ? 1 : 0));
}

int mnf_affine_half(const float* x, float* y, float* log_det, int accumulate, const float* flat,
                    const float* image, const void* split_image, int64_t rows, int dim, int parity, int inverse,
                    int n_hidden, const int* hidden, int has_scale, int has_shift, int force_generic,
                    void* stream) {
  return mnf_affine_half_sq(x, y, log_det, nullptr, accumulate, flat, image, split_image, rows, dim, parity, inverse,
                            n_hidden, hidden, has_scale, has_shift, force_generic, stream);
}

int mnf_affine_half_sq(const float* x, float* y, float* log_det, float* y_sqnorm, int accumulate,
                       const float* flat, const float* image, const void* split_image, int64_t rows, int dim,
                       int parity, int inverse, int n_hidden, const int* hidden, int has_scale, int has_shift,
                       int force_generic, void* stream) {
  if (!x || !y || x == y || rows < 0 || dim < 2 || (dim & 1) || !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if ((has_scale || has_shift) && !flat && !image) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (image && split_image && !force_generic) {
    const int rc = ahf_split_launch(x, y, log_det, y_sqnorm, accumulate, split_image, image, rows, dim, parity,
                                    inverse, n_hidden, hidden, has_scale, has_shift, (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
    if (dim / 2 != ahf_padded_half(dim / 2) && (has_scale || has_shift) && ahf_padded_hidden(n_hidden, hidden)) {
      // a half narrower than its MFMA tile (d = 2, 6, 50 ...): the stack kernel's ragged variant, one layer
      const int rc1 = ahf_split_stack_launch(x, y, nullptr, log_det, y_sqnorm, accumulate, split_image, image,
                                             parity ? 1u : 0u, 1, rows, dim, inverse, ahf_padded_hidden(n_hidden, hidden), nullptr, nullptr,
                                             (hipStream_t)stream);
      if (rc1 != MNF_ERR_UNSUPPORTED) return rc1;
    }
  }
  if (image && !force_generic) {
    const int rc = ahf_mfma_launch(x, y, log_det, y_sqnorm, accumulate, image, rows, dim, parity, inverse,
                                   n_hidden, hidden, has_scale, has_shift, (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if ((has_scale || has_shift) && !flat) return y_sqnorm ? MNF_ERR_UNSUPPORTED : MNF_ERR_INVALID_ARG;
  // any other shape: the run-time-shaped matrix-core kernel (mnf_ahf_rt.hip) from kRtMinRows rows on (force_generic == 2:
  // at any row count, whatever the shape's specialised kernels -- tests and the coverage map compare the two).  An fp32
  // request -- the fp32 operand image without the split one -- does not take it: its arithmetic is split-f16; the VALU
  // kernel below is fp32.
  if (force_generic == 2 || (!force_generic && !(image && !split_image) && rows >= kRtMinRows)) {
    const int rc = ahf_rt_launch(x, y, log_det, y_sqnorm, accumulate, flat, rows, dim, parity, inverse, n_hidden, hidden,
                                 has_scale, has_shift, (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if (y_sqnorm) return MNF_ERR_UNSUPPORTED;  // the VALU kernel does not emit |y|^2

  AhfArgs a;
  a.x = x; a.y = y; a.log_det = log_det; a.flat = flat; a.rows = rows; a.dim = dim;
  a.parity = parity != 0; a.inverse = inverse != 0; a.accumulate = accumulate != 0;
  a.has_scale = has_scale != 0; a.has_shift = has_shift != 0;
  const int H = dim / 2;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = H;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  sizes[n_hidden + 1] = H;
  int64_t off = 0;
  memset(&a.s_net, 0, sizeof(NetDesc));
  memset(&a.t_net, 0, sizeof(NetDesc));
  if (has_scale) off += fill_net(a.s_net, n_hidden + 2, sizes, off);
  if (has_shift) off += fill_net(a.t_net, n_hidden + 2, sizes, off);
  int ldw = 1;
  for (int i = 0; i < n_hidden; ++i) ldw = hidden[i] > ldw ? hidden[i] : ldw;
  a.ldw = ldw;
  const int per_row = 3 * H + 2 * ldw;
  a.w_lds = (off <= kParamLdsFloats && off + 4 * per_row <= kLdsBudgetFloats) ? (int)off : 0;
  int R = (kLdsBudgetFloats - a.w_lds) / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  const int want = pick_rows_per_group(rows, 64);
  if (R > want) R = want;
  a.R = R;
  const int64_t blocks = (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("ahf_generic");
  hipLaunchKernelGGL(ahf_generic_kernel, dim3((unsigned)blocks), dim3(kThreads),
                     ((size_t)R * per_row + a.w_lds) * sizeof(float), (hipStream_t)stream, a);
  return check_launch();
}

int mnf_pack_gather_split_batch(const float* flat, const int32_t* idx, void* images, int64_t n_split_words,
                                int64_t n_plain_words, int n_images, int64_t flat_stride, void* stream) {
  if (!flat || !idx || !images || n_split_words < 0 || n_plain_words < 0 || n_images < 1 || n_images > 65535 ||
      flat_stride < 0)
    return MNF_ERR_INVALID_ARG;
  uint32_t* img = static_cast<uint32_t*>(images);
  const int64_t n = n_split_words + n_plain_words, image_words = n + MNF_SPLIT_TAIL_WORDS;
  // the tail words collect max |weight| by atomicMax: zero them first.  A KERNEL, not hipMemsetAsync: recorded in a
  // hipGraph, the 16-byte memset node in front of the pack kernel was not reliably in effect when the pack kernel ran --
  // one captured MNF-LeNet step in three then replayed with stale maxima in the tails, every row group took the fp32
  // fix-up path, and a replay cost 60 ms instead of 4 (results unchanged: the fix-up path is exact)
  hipLaunchKernelGGL(zero_tails_kernel, dim3(n_images), dim3(64), 0, (hipStream_t)stream, img + n, image_words);
  if (int rc = check_launch()) return rc;
  if (n == 0) return MNF_OK;
  hipLaunchKernelGGL(pack_gather_split_kernel, dim3(grid_for(n, 1024), n_images), dim3(1024), 0, (hipStream_t)stream,
                     flat, idx, img, n_split_words, n_plain_words, flat_stride);
  return check_launch();
}

int mnf_pack_gather_split(const float* flat, const int32_t* idx, void* image, int64_t n_split_words,
                          int64_t n_plain_words, void* stream) {
  return mnf_pack_gather_split_batch(flat, idx, image, n_split_words, n_plain_words, 1, 0, stream);
}

int mnf_pack_gather_batch(const float* flat, const int32_t* idx, float* images, int64_t n, int n_images,
                          int64_t flat_stride, void* stream) {
  if (!flat || !idx || !images || n < 0 || n_images < 1 || n_images > 65535 || flat_stride < 0)
    return MNF_ERR_INVALID_ARG;
  if (n == 0) return MNF_OK;
  hipLaunchKernelGGL(pack_gather_kernel, dim3(grid_for(n, 256), n_images), dim3(256), 0, (hipStream_t)stream,
                     flat, idx, images, n, flat_stride);
  return check_launch();
}

int mnf_pack_gather(const float* flat, const int32_t* idx, float* image, int64_t n, void* stream) {
  return mnf_pack_gather_batch(flat, idx, image, n, 1, 0, stream);
}

// -------------------------------------------------------------------------- fused Adam on one flat buffer
int mnf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f))
    return MNF_ERR_INVALID_ARG;
  if (n == 0) return MNF_OK;
  // bias corrections in double on the host (torch.optim.Adam: step_size = lr / (1 - beta1^t); denom uses sqrt(1 - beta2^t))
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)));
  return check_launch();
}

int mnf_adam_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, float* state_dev, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state_dev || n < 0 || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f))
    return MNF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, beta1, beta2);
  if (int rc = check_launch()) return rc;
  if (n == 0) return MNF_OK;
  hipLaunchKernelGGL(adam_state_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state_dev);
  return check_launch();
}

// -------------------------------------------------------------------------- NSF_AR
int64_t mnf_nsf_ar_flat_floats(int dim, int K, int n_hidden, const int* hidden) {
  if (dim < 1 || K < 1 || !hidden_ok(n_hidden, hidden)) return -1;
  const int P = 3 * K - 1;
  int64_t n = P;
  for (int i = 1; i < dim; ++i) n += mnf::nsf_ar_net_floats(i, n_hidden, hidden, P);
  return n;
}

int mnf_nsf_ar(const float* x, float* y, float* log_det, int accumulate, const float* flat, int64_t rows, int dim, int K,
               float tail_bound, int inverse, int n_hidden, const int* hidden, void* stream) {
  if (!x || !y || x == y || !flat || rows < 0 || dim < 1 || K < 1 || !(tail_bound > 0.f) ||
      !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (1e-3 * K > 1.0) return MNF_ERR_DOMAIN;  // spline_flow.py:90-93
  if (rows == 0) return MNF_OK;
  if (mnf_nsf_ar_flat_floats(dim, K, n_hidden, hidden) >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;
  NsfArArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.log_det = log_det; a.flat = flat; a.rows = rows; a.dim = dim; a.K = K;
  a.inverse = inverse != 0; a.accumulate = accumulate != 0; a.T = tail_bound; a.n_hidden = n_hidden;
  int ldw = 1;
  for (int i = 0; i < n_hidden; ++i) {
    a.hidden[i] = hidden[i];
    if (hidden[i] > ldw) ldw = hidden[i];
  }
  a.ldw = ldw;
  const int per_row = 2 * dim + 1 + 2 * ldw + (3 * K - 1);
  int R = kLdsBudgetFloats / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  const int cap = pick_rows_per_group(rows, 64);
  if (R > cap) R = cap;
  a.R = R;
  const int64_t blocks = (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("nsf_ar_generic");
  hipLaunchKernelGGL(nsf_ar_generic_kernel, dim3((unsigned)blocks), dim3(kThreads), (size_t)R * per_row * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch();
}

// -------------------------------------------------------------------------- NSF_CL
static int nsf_sizes(int dim, int K, int n_hidden, const int* hidden, int* sizes) {
  sizes[0] = dim / 2;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  sizes[n_hidden + 1] = ((3 * K - 1) * dim) / 2;  // spline_flow.py:246
  return n_hidden + 2;
}

int64_t mnf_nsf_cl_flat_floats(int dim, int K, int n_hidden, const int* hidden) {
  if (dim < 2 || (dim & 1) || K < 1 || !hidden_ok(n_hidden, hidden)) return -1;
  int sizes[MNF_MAX_LINEAR + 1];
  const int n = nsf_sizes(dim, K, n_hidden, hidden, sizes);
  NetDesc nd;
  return 2 * fill_net(nd, n, sizes, 0);
}

int mnf_nsf_cl(const float* x, float* y, float* log_det, int accumulate, const float* flat,
               const float* image, const void* split_image, int64_t rows, int dim, int K, float tail_bound,
               int inverse, int n_hidden, const int* hidden, int force_generic, void* stream) {
  if (!x || !y || x == y || rows < 0 || dim < 2 || (dim & 1) || K < 1 || !(tail_bound > 0.f) ||
      !hidden_ok(n_hidden, hidden) || (!flat && !image))
    return MNF_ERR_INVALID_ARG;
  if (1e-3 * K > 1.0) return MNF_ERR_DOMAIN;  // spline_flow.py:90-93
  if (rows == 0) return MNF_OK;
  if (image && !force_generic) {
    const int rc = nsf_mfma_launch(x, y, log_det, accumulate, image, split_image, rows, dim, K, tail_bound, inverse,
                                   n_hidden, hidden, (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if (!flat) return MNF_ERR_INVALID_ARG;
  // any other shape: the run-time-shaped matrix-core kernel (mnf_nsf_rt.hip; force_generic == 2: at any row count, whatever
  // the shape's specialised kernels)
  if (force_generic == 2 || (!force_generic && !(image && !split_image) && rows >= kRtMinRows)) {
    const int rc = nsf_rt_launch(x, y, log_det, accumulate, flat, rows, dim, K, tail_bound, inverse, n_hidden, hidden,
                                 (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  NsfArgs a;
  a.x = x; a.y = y; a.log_det = log_det; a.flat = flat; a.rows = rows; a.dim = dim; a.K = K;
  a.inverse = inverse != 0; a.accumulate = accumulate != 0; a.T = tail_bound;
  const int H = dim / 2;
  int sizes[MNF_MAX_LINEAR + 1];
  const int n = nsf_sizes(dim, K, n_hidden, hidden, sizes);
  int64_t off = fill_net(a.f1, n, sizes, 0);
  fill_net(a.f2, n, sizes, off);
  int ldw = 1;
  for (int i = 0; i < n_hidden; ++i) ldw = hidden[i] > ldw ? hidden[i] : ldw;
  a.ldw = ldw;
  a.ldp = sizes[n - 1];
  const int per_row = 3 * H + 1 + 2 * ldw + a.ldp;
  const int64_t n_par = mnf_nsf_cl_flat_floats(dim, K, n_hidden, hidden);
  a.w_lds = (n_par <= kParamLdsFloats && n_par + 4 * per_row <= kLdsBudgetFloats) ? (int)n_par : 0;
  int R = (kLdsBudgetFloats - a.w_lds) / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  const int want = pick_rows_per_group(rows, 64);
  if (R > want) R = want;
  a.R = R;
  const int64_t blocks = (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("nsf_generic");
  hipLaunchKernelGGL(nsf_generic_kernel, dim3((unsigned)blocks), dim3(kThreads),
                     ((size_t)R * per_row + a.w_lds) * sizeof(float), (hipStream_t)stream, a);
  return check_launch();
}

int mnf_rqs(const float* inputs, const float* W, const float* H, const float* D, float* outputs,
            float* logabsdet, int64_t n, int K, float tail_bound, int inverse, void* stream) {
  if (!inputs || !W || !H || (!D && K > 1) || !outputs || !logabsdet || n < 0 || K < 1 ||
      !(tail_bound > 0.f))
    return MNF_ERR_INVALID_ARG;
  if (1e-3 * K > 1.0) return MNF_ERR_DOMAIN;
  if (n == 0) return MNF_OK;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(rqs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, inputs, W, H,
                     D, outputs, logabsdet, n, K, tail_bound, inverse);
  return check_launch();
}

// ---------------------------------------------------------------------------- RNVP
int64_t mnf_rnvp_flat_floats(int dim, int n_hidden, const int* hidden) {
  if (dim < 1 || n_hidden < 1 || !hidden_ok(n_hidden, hidden)) return -1;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = dim;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  NetDesc nd;
  const int64_t net = fill_net(nd, n_hidden + 1, sizes, 0);
  return net + 2 * ((int64_t)hidden[n_hidden - 1] * dim + dim);
}

int mnf_rnvp(const float* z, const float* mask, float* x, float* log_det, int accumulate,
             const float* flat, const float* image, const void* split_image, int64_t rows, int dim,
             int n_hidden, const int* hidden, int force_generic, void* stream) {
  if (!mask) return MNF_ERR_INVALID_ARG;
  return mnf_rnvp_seeded(z, mask, 0, x, log_det, accumulate, flat, image, split_image, rows, dim, n_hidden, hidden,
                         force_generic, stream);
}

// MNFLinear.sample_z with its prologue fused into the first flow (mnf_linear.py:58-64): z0 = q0_mean +
// sqrt(exp(q0_log_var)) * eps is formed in the RNVP kernel's loads, so z0 is neither written nor re-read.
int mnf_rnvp_sample(const float* eps, const float* q0_mean, const float* q0_log_var, const float* mask,
                    uint64_t seed, float* x, float* log_det, int accumulate, const float* image,
                    const void* split_image, int64_t rows, int dim, int n_hidden, const int* hidden, void* stream) {
  if (!eps || !q0_mean || !q0_log_var || !x || eps == x || rows < 0 || dim < 1 || n_hidden < 1 ||
      !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (!image || !split_image) return MNF_ERR_UNSUPPORTED;  // only the split MFMA kernel has the fused prologue
  if (rows == 0) return MNF_OK;
  return rnvp_mfma_launch(eps, mask, x, log_det, accumulate, image, split_image, rows, dim, n_hidden, hidden, seed,
                          (hipStream_t)stream, q0_mean, q0_log_var);
}

__global__ void rnvp_mask_kernel(uint64_t seed, float* __restrict__ mask, int64_t rows, int dim) {
  const int64_t n = rows * dim, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    mask[i] = rnvp_mask_bit(seed, i / dim, (int)(i % dim));
}

int mnf_rnvp_mask(uint64_t seed, float* mask, int64_t rows, int dim, void* stream) {
  if (!mask || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  hipLaunchKernelGGL(rnvp_mask_kernel, dim3(grid_for(rows * dim, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     mask, rows, dim);
  return check_launch();
}

int mnf_rnvp_seeded(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                    const float* flat, const float* image, const void* split_image, int64_t rows, int dim,
                    int n_hidden, const int* hidden, int force_generic, void* stream) {
  return mnf_rnvp_seeded_train(z, mask, seed, x, log_det, accumulate, flat, image, split_image, rows, dim, n_hidden,
                               hidden, force_generic, nullptr, nullptr, stream);
}

int mnf_rnvp_seeded_train(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                          const float* flat, const float* image, const void* split_image, int64_t rows, int dim,
                          int n_hidden, const int* hidden, int force_generic, float* y_out, int* y_written_host,
                          void* stream) {
  if (y_written_host) *y_written_host = 0;
  if (!z || !x || z == x || rows < 0 || dim < 1 || n_hidden < 1 || !hidden_ok(n_hidden, hidden) ||
      (!flat && !image))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (flat && !force_generic && rnvp_few_fwd_ok(rows, dim, n_hidden, hidden, mask != nullptr))  // few rows: the latency kernel
    return rnvp_few_fwd_launch(z, mask, seed, x, log_det, accumulate, flat, rows, dim, hidden[0], (hipStream_t)stream);
  if (image && !force_generic) {
    const int rc = rnvp_mfma_launch(z, mask, x, log_det, accumulate, image, split_image, rows, dim, n_hidden, hidden,
                                    seed, (hipStream_t)stream, nullptr, nullptr, y_out, y_written_host);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if (!flat) return MNF_ERR_INVALID_ARG;
  // any other shape: the run-time-shaped matrix-core kernel (mnf_rnvp_rt.hip; force_generic == 2: at any row count,
  // whatever the shape's specialised kernels)
  if (force_generic == 2 || (!force_generic && !(image && !split_image) && rows >= kRtMinRows)) {
    const int rc = rnvp_rt_launch(z, mask, seed, x, log_det, accumulate, flat, rows, dim, n_hidden, hidden, (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  RnvpArgs a;
  a.z = z; a.mask = mask; a.x = x; a.log_det = log_det; a.flat = flat; a.rows = rows; a.dim = dim;
  a.accumulate = accumulate != 0; a.seed = seed;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = dim;
  for (int i = 0; i < n_hidden; ++i) sizes[1 + i] = hidden[i];
  int64_t off = fill_net(a.net, n_hidden + 1, sizes, 0);
  const int hl = hidden[n_hidden - 1];
  a.t_w = (int)off; off += (int64_t)hl * dim;
  a.t_b = (int)off; off += dim;
  a.s_w = (int)off; off += (int64_t)hl * dim;
  a.s_b = (int)off;
  int ldw = 1;
  for (int i = 0; i + 1 < n_hidden; ++i) ldw = hidden[i] > ldw ? hidden[i] : ldw;
  a.ldw = ldw;
  const int per_row = 2 * dim + hl + 2 * ldw;
  const int64_t n_par = (int64_t)a.s_b + dim;
  a.w_lds = (n_par <= kParamLdsFloats && n_par + 4 * per_row <= kLdsBudgetFloats) ? (int)n_par : 0;
  int R = (kLdsBudgetFloats - a.w_lds) / per_row;
  if (R < 1) return MNF_ERR_UNSUPPORTED;
  const int want = pick_rows_per_group(rows, 32);
  if (R > want) R = want;
  a.R = R;
  const int64_t blocks = (rows + R - 1) / R;
  if (blocks > 0x7fffffff) return MNF_ERR_UNSUPPORTED;
  tag_kernel("rnvp_generic");
  hipLaunchKernelGGL(rnvp_generic_kernel, dim3((unsigned)blocks), dim3(kThreads),
                     ((size_t)R * per_row + a.w_lds) * sizeof(float), (hipStream_t)stream, a);
  return check_launch();
}

// ----------------------------------------------------------- data-independent layers
int mnf_affine_const(const float* x, float* y, const float* s, const float* t, float* log_det,
                     int accumulate, float* ld_scalar, int64_t rows, int dim, int inverse, void* stream) {
  if (!x || !y || !s || !t || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (log_det && !ld_scalar) return MNF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (ld_scalar) {
    hipLaunchKernelGGL(sum_vec_kernel, dim3(1), dim3(64), 0, st, s, dim, inverse != 0, ld_scalar);
    if (int rc = check_launch()) return rc;
  }
  if (rows == 0) return MNF_OK;
  const int64_t n = rows * dim;
  const bool vec = (dim % 4 == 0) && !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                        reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(t)) & 15);
  if (vec)
    hipLaunchKernelGGL(affine_const_kernel_v4, dim3(grid_for(n / 4, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), s, t, n / 4, dim / 4,
                       inverse != 0);
  else
    hipLaunchKernelGGL(affine_const_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, x, y, s, t, n, dim,
                       inverse != 0);
  if (int rc = check_launch()) return rc;
  if (log_det) {
    hipLaunchKernelGGL(add_scalar_rows_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, st, log_det,
                       ld_scalar, rows, accumulate != 0);
    return check_launch();
  }
  return MNF_OK;
}

int mnf_linear_rows(const float* x, const float* W, float* y, int64_t rows, int dim, void* stream) {
  if (!x || !W || !y || x == y || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  int64_t wf = (int64_t)dim * dim;
  const int w_lds = wf + 8 * (int64_t)dim <= kLdsBudgetFloats;
  if (!w_lds) wf = 0;
  if (dim > kLdsBudgetFloats) return MNF_ERR_UNSUPPORTED;
  int R = (int)((kLdsBudgetFloats - wf) / dim);
  if (R > 256) R = 256;
  const int64_t groups = (rows + R - 1) / R;
  const int grid = (int)(groups < 2048 ? groups : 2048);
  tag_kernel("linear_rows_generic");
  hipLaunchKernelGGL(linear_rows_kernel, dim3(grid), dim3(kThreads), (size_t)(wf + (int64_t)R * dim) * 4,
                     (hipStream_t)stream, x, W, y, rows, dim, R, w_lds);
  return check_launch();
}

int mnf_gauss_logprob(const float* z, const float* log_det, float* log_prob, double* sum_out,
                      int64_t rows, int dim, void* stream) {
  if (!z || rows < 0 || dim < 1 || (!log_prob && !sum_out)) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  hipLaunchKernelGGL(gauss_logprob_kernel, dim3(grid_for(rows, 16, 2048)), dim3(kThreads), 0,
                     (hipStream_t)stream, z, log_det, log_prob, sum_out, rows, dim);
  return check_launch();
}

int mnf_gauss_logprob_sq(const float* z_sqnorm, const float* log_det, float* log_prob, double* sum_out,
                         int64_t rows, int dim, void* stream) {
  if (!z_sqnorm || rows < 0 || dim < 1 || (!log_prob && !sum_out)) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  // 8 MB in, 4 MB out: one workgroup per CU is enough, and keeps the fp64 atomics on sum_out to 256
  hipLaunchKernelGGL(gauss_logprob_sq_kernel, dim3(grid_for(rows, 4 * kThreads, 256)), dim3(kThreads), 0,
                     (hipStream_t)stream, z_sqnorm, log_det, log_prob, sum_out, rows, dim);
  return check_launch();
}

int mnf_sample_z0(const float* q0_mean, const float* q0_log_var, const float* eps, float* z0, int64_t rows,
                  int dim, void* stream) {
  if (!q0_mean || !q0_log_var || !eps || !z0 || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  const int64_t n = rows * dim;
  if (dim % 4 == 0 && !((reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(z0)) & 15))
    hipLaunchKernelGGL(sample_z0_kernel_v4, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, q0_mean,
                       q0_log_var, reinterpret_cast<const float4*>(eps), reinterpret_cast<float4*>(z0), n / 4, dim / 4);
  else
    hipLaunchKernelGGL(sample_z0_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, q0_mean,
                       q0_log_var, eps, z0, n, dim);
  return check_launch();
}

static void sample_z0_bwd_plan(int64_t rows, int dim, bool vec, int& dim_blocks, int64_t& row_blocks) {
  const int per_block = vec ? 256 : 64;
  dim_blocks = (dim + per_block - 1) / per_block;
  row_blocks = (rows * dim) / (64 * 1024);
  row_blocks = row_blocks < 1 ? 1 : row_blocks > 2048 / dim_blocks + 1 ? 2048 / dim_blocks + 1 : row_blocks;
}

static int sample_z0_bwd_launch(const float* grad_z0, const float* eps, uint64_t seed, const float* q0_log_var,
                                float* grad_mean, float* grad_log_var, int64_t rows, int dim, void* stream,
                                float* blocks = nullptr) {
  if (rows == 0) return MNF_OK;
  // enough workgroups to fill the chip once rows x dim is large; a single row block (no atomics) while it is small
  const bool vec = dim % 4 == 0 && !((reinterpret_cast<uintptr_t>(grad_z0) | reinterpret_cast<uintptr_t>(eps)) & 15);
  int dim_blocks;
  int64_t row_blocks;
  sample_z0_bwd_plan(rows, dim, vec, dim_blocks, row_blocks);
  if (row_blocks == 1) blocks = nullptr;  // (one workgroup per dim block adds in place: nothing to order)
  if (vec)
    hipLaunchKernelGGL(sample_z0_bwd_kernel<4>, dim3(dim_blocks, (unsigned)row_blocks), dim3(64, 4), 0, (hipStream_t)stream,
                       grad_z0, eps, q0_log_var, grad_mean, grad_log_var, rows, dim, row_blocks > 1 ? 1 : 0, seed, blocks);
  else
    hipLaunchKernelGGL(sample_z0_bwd_kernel<1>, dim3(dim_blocks, (unsigned)row_blocks), dim3(64, 4), 0, (hipStream_t)stream,
                       grad_z0, eps, q0_log_var, grad_mean, grad_log_var, rows, dim, row_blocks > 1 ? 1 : 0, seed, blocks);
  if (int rc = check_launch()) return rc;
  if (!blocks) return MNF_OK;
  if (int rc = det_reduce_async(blocks, (int)row_blocks, 2 * (int64_t)dim, dim, grad_mean, (hipStream_t)stream)) return rc;
  return det_reduce_async(blocks + dim, (int)row_blocks, 2 * (int64_t)dim, dim, grad_log_var, (hipStream_t)stream);
}

int64_t mnf_sample_z0_bwd_workspace(int64_t rows, int dim) {
  if (rows < 1 || dim < 1) return 0;
  int dim_blocks;
  int64_t row_blocks;
  sample_z0_bwd_plan(rows, dim, false, dim_blocks, row_blocks);  // (the scalar plan never has fewer row blocks)
  int db4;
  int64_t rb4;
  sample_z0_bwd_plan(rows, dim, true, db4, rb4);
  return (row_blocks > rb4 ? row_blocks : rb4) * 2 * dim;
}

int mnf_sample_z0_bwd_det(const float* grad_z0, const float* eps, uint64_t seed, const float* q0_log_var, float* grad_mean,
                          float* grad_log_var, int64_t rows, int dim, float* workspace, int64_t workspace_floats,
                          void* stream) {
  if (!grad_z0 || !q0_log_var || !grad_mean || !grad_log_var || !workspace || rows < 0 || dim < 1 ||
      workspace_floats < mnf_sample_z0_bwd_workspace(rows, dim))
    return MNF_ERR_INVALID_ARG;
  return sample_z0_bwd_launch(grad_z0, eps, seed, q0_log_var, grad_mean, grad_log_var, rows, dim, stream, workspace);
}

int mnf_sample_z0_bwd(const float* grad_z0, const float* eps, const float* q0_log_var, float* grad_mean,
                      float* grad_log_var, int64_t rows, int dim, void* stream) {
  if (!grad_z0 || !eps || !q0_log_var || !grad_mean || !grad_log_var || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  return sample_z0_bwd_launch(grad_z0, eps, 0, q0_log_var, grad_mean, grad_log_var, rows, dim, stream);
}

int mnf_sample_z0_seeded(const float* q0_mean, const float* q0_log_var, uint64_t seed, float* z0, int64_t rows, int dim,
                         void* stream) {
  if (!q0_mean || !q0_log_var || !z0 || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  const bool vec = dim % 4 == 0 && !(reinterpret_cast<uintptr_t>(z0) & 15);
  const int per_block = vec ? 256 : 64;
  const int dim_blocks = (dim + per_block - 1) / per_block;
  int64_t row_blocks = (rows + 3) / 4;  // at most one row per thread row; enough workgroups to fill the chip otherwise
  row_blocks = row_blocks > 4096 / dim_blocks + 1 ? 4096 / dim_blocks + 1 : row_blocks;
  if (vec)
    hipLaunchKernelGGL(sample_z0_seeded_kernel<4>, dim3(dim_blocks, (unsigned)row_blocks), dim3(64, 4), 0, (hipStream_t)stream,
                       q0_mean, q0_log_var, seed, z0, rows, dim);
  else
    hipLaunchKernelGGL(sample_z0_seeded_kernel<1>, dim3(dim_blocks, (unsigned)row_blocks), dim3(64, 4), 0, (hipStream_t)stream,
                       q0_mean, q0_log_var, seed, z0, rows, dim);
  return check_launch();
}

int mnf_sample_z0_noise(uint64_t seed, float* eps, int64_t rows, int dim, void* stream) {
  if (!eps || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  hipLaunchKernelGGL(sample_z0_noise_kernel, dim3((unsigned)((rows * dim + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     eps, rows, dim);
  return check_launch();
}

int mnf_sample_z0_seeded_bwd(const float* grad_z0, uint64_t seed, const float* q0_log_var, float* grad_mean,
                             float* grad_log_var, int64_t rows, int dim, void* stream) {
  if (!grad_z0 || !q0_log_var || !grad_mean || !grad_log_var || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  return sample_z0_bwd_launch(grad_z0, nullptr, seed, q0_log_var, grad_mean, grad_log_var, rows, dim, stream);
}

}  // extern "C"
