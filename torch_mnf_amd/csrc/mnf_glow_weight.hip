// Glow's d x d parameter preparation and its gradients (torch_mnf/flows/glow.py:20-37) in one launch each way:
//
//   Lm = tril(L, -1) + I;   Um = triu(U, 1) + diag(S);   W = P Lm Um;   log_det = sum log|S|
//   inverse direction:      W^-1 = Um^-1 Lm^-1 P^T  -- two triangular inverses by substitution: the matrix is GIVEN in
//                           PLU form, no factorisation (the reference calls torch.inverse on the assembled W, :34)
//
// Composed from stock ops this is ~15 launches forward (eye, tril, triu, diag, two matmuls, abs / log / sum, an LU
// factorisation and a triangular solve for the inverse: rocsolver getf2 alone is 52 us at d = 32) and ~30 backward per
// Glow layer per training step -- a tenth of config 3's training step.  One workgroup; the matrices live in LDS.
// Gradients, with G the cotangent of the matrix that was handed out:
//   inverse:  G_W = -W^-T G W^-T          (d W^-1 = -W^-1 dW W^-1)
//   A = P^T G_W;   dLm = A Um^T  -> strictly lower part to dL;   dUm = Lm^T A  -> strictly upper part to dU, diagonal
//   to dS;   dS += g_ld * sign / S  (log_det = +-sum log|S|)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnf_hip.h"
#include "mnf_host.h"

namespace mnf {

constexpr int kGlowThreads = 256;
constexpr int kGlowMaxDim = MNF_GLOW_WEIGHT_MAX_DIM;

struct GlowArgs {
  const float* P;
  const float* L;
  const float* S;
  const float* U;
  float* out;       // forward: W or W^-1 (d x d)
  float* log_det;   // forward: +-sum log|S| (1)
  const float* G;   // backward: cotangent of `out`
  const float* g_ld;  // backward: cotangent of log_det (1) or nullptr
  float* gL;
  float* gS;
  float* gU;
  int d, inverse, accumulate;
  const float* out_fwd;  // backward, inverse direction: the W^-1 the forward launch handed out (or nullptr: recomputed)
};

// The matrices live in LDS padded to DP x DP (DP = 8, 16, 32 or 64; identity padding: the padded W is diag(W, I), whose
// inverse and gradients are the unpadded ones in the top-left corner), so every loop bound is a compile-time constant.
// C = A B, optionally with A and / or B read transposed; a thread computes four adjacent outputs of a row
template <int DP, bool TA, bool TB>
__device__ __forceinline__ void glow_mm(const float* A, const float* B, float* C, float scale = 1.f) {
  for (int idx = threadIdx.x; idx < DP * DP / 4; idx += kGlowThreads) {
    const int i = idx / (DP / 4), j0 = 4 * (idx - i * (DP / 4));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < DP; ++k) {
      const float av = TA ? A[k * DP + i] : A[i * DP + k];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = fmaf(av, TB ? B[(j0 + u) * DP + k] : B[k * DP + j0 + u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) C[i * DP + j0 + u] = acc[u] * scale;
  }
  __syncthreads();
}

// Lm, Um, P into LDS
template <int DP>
__device__ __forceinline__ void glow_load(const GlowArgs& a, float* Lm, float* Um, float* Pm) {
  const int d = a.d;
  for (int idx = threadIdx.x; idx < DP * DP; idx += kGlowThreads) {
    const int i = idx / DP, j = idx - i * DP;
    const bool in = i < d && j < d;
    const float eye = i == j ? 1.f : 0.f;
    Lm[idx] = in && i > j ? a.L[i * d + j] : eye;
    Um[idx] = in && i < j ? a.U[i * d + j] : (i == j ? (i < d ? a.S[i] : 1.f) : 0.f);
    Pm[idx] = in ? a.P[i * d + j] : eye;
  }
  __syncthreads();
}

// Inverses of the unit lower triangular Lm and the upper triangular Um: thread j solves column j by substitution with
// the column in REGISTERS (fully unrolled: DP (DP - 1) / 2 multiply-adds, the coefficients broadcast reads from LDS;
// with the column in LDS and thread-dependent loop bounds the same work waited on every read: 45 of the launch's 50 us).
// The sums run over the full range -- the entries before the unit entry are zeros by themselves.
template <int DP>
__device__ __forceinline__ void glow_tri_inverses(const float* Lm, const float* Um, float* Li, float* Ui) {
  const int t = threadIdx.x;
  if (t < DP) {  // Lm x = e_t, forward substitution
    float x[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      float v = i == t ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < i; ++k) v = fmaf(-Lm[i * DP + k], x[k], v);
      x[i] = v;
    }
#pragma unroll
    for (int i = 0; i < DP; ++i) Li[i * DP + t] = x[i];
  } else if (t >= 64 && t < 64 + DP) {  // Um x = e_c, back substitution (its own wave)
    const int c = t - 64;
    float x[DP];
#pragma unroll
    for (int i = DP - 1; i >= 0; --i) {
      float v = i == c ? 1.f : 0.f;
#pragma unroll
      for (int k = i + 1; k < DP; ++k) v = fmaf(-Um[i * DP + k], x[k], v);
      x[i] = v / Um[i * DP + i];
    }
#pragma unroll
    for (int i = 0; i < DP; ++i) Ui[i * DP + c] = x[i];
  }
  __syncthreads();
}

template <int DP>
__global__ void __launch_bounds__(kGlowThreads) glow_weight_kernel(const GlowArgs a) {
  extern __shared__ float glds[];
  constexpr int n = DP * DP;
  const int d = a.d;
  float *Lm = glds, *Um = Lm + n, *Pm = Um + n, *T1 = Pm + n, *T2 = T1 + n;
  glow_load<DP>(a, Lm, Um, Pm);
  if (!a.inverse) {
    glow_mm<DP, false, false>(Lm, Um, T1);
    glow_mm<DP, false, false>(Pm, T1, T2);  // W = P (Lm Um)
  } else {
    glow_tri_inverses<DP>(Lm, Um, T1, T2);  // T1 = Lm^-1, T2 = Um^-1
    glow_mm<DP, false, true>(T1, Pm, Lm);   // Lm := Lm^-1 P^T (the originals are no longer needed)
    glow_mm<DP, false, false>(T2, Lm, Um);  // Um := Um^-1 Lm^-1 P^T = W^-1
    T2 = Um;
  }
  for (int idx = threadIdx.x; idx < d * d; idx += kGlowThreads) a.out[idx] = T2[(idx / d) * DP + idx % d];
  if (threadIdx.x < 64) {  // log_det = +-sum log|S| (glow.py:29, :35)
    float s = 0.f;
    for (int i = threadIdx.x; i < d; i += 64) s += logf(fabsf(a.S[i]));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (threadIdx.x == 0) a.log_det[0] = a.inverse ? -s : s;
  }
}

template <int DP>
__global__ void __launch_bounds__(kGlowThreads) glow_weight_bwd_kernel(const GlowArgs a) {
  extern __shared__ float glds[];
  constexpr int n = DP * DP;
  const int d = a.d;
  float *Lm = glds, *Um = Lm + n, *Pm = Um + n, *T1 = Pm + n, *T2 = T1 + n, *Gm = T2 + n, *T3 = Gm + n;
  glow_load<DP>(a, Lm, Um, Pm);
  for (int idx = threadIdx.x; idx < n; idx += kGlowThreads) {
    const int i = idx / DP, j = idx - i * DP;
    Gm[idx] = a.G && i < d && j < d ? a.G[i * d + j] : 0.f;
  }
  __syncthreads();
  if (a.inverse) {
    if (a.out_fwd) {  // W^-1 as the forward launch computed it (identity padding): two substitutions and two products less
      for (int idx = threadIdx.x; idx < n; idx += kGlowThreads) {
        const int i = idx / DP, j = idx - i * DP;
        T1[idx] = i < d && j < d ? a.out_fwd[i * d + j] : (i == j ? 1.f : 0.f);
      }
      __syncthreads();
    } else {
      glow_tri_inverses<DP>(Lm, Um, T1, T2);
      glow_mm<DP, false, true>(T1, Pm, T3);      // T3 = Lm^-1 P^T
      glow_mm<DP, false, false>(T2, T3, T1);     // T1 = W^-1
    }
    glow_mm<DP, true, false>(T1, Gm, T2);        // T2 = W^-T G
    glow_mm<DP, false, true>(T2, T1, Gm, -1.f);  // Gm = -W^-T G W^-T = cotangent of W
  }
  glow_mm<DP, true, false>(Pm, Gm, T1);   // A = P^T G_W
  glow_mm<DP, false, true>(T1, Um, T2);   // dLm = A Um^T
  glow_mm<DP, true, false>(Lm, T1, T3);   // dUm = Lm^T A
  const float gld = a.g_ld ? (a.inverse ? -a.g_ld[0] : a.g_ld[0]) : 0.f;
  for (int idx = threadIdx.x; idx < d * d; idx += kGlowThreads) {
    const int i = idx / d, j = idx - i * d, p = i * DP + j;
    const float vl = i > j ? T2[p] : 0.f, vu = i < j ? T3[p] : 0.f;
    a.gL[idx] = a.accumulate ? a.gL[idx] + vl : vl;
    a.gU[idx] = a.accumulate ? a.gU[idx] + vu : vu;
    if (i == j) {
      const float vs = T3[p] + gld / a.S[i];
      a.gS[i] = a.accumulate ? a.gS[i] + vs : vs;
    }
  }
}

template <typename K>
static int glow_attr(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 1
                                                                                                                     : -1;
}

}  // namespace mnf

using namespace mnf;

static inline int glow_pad(int dim) { return dim <= 8 ? 8 : dim <= 16 ? 16 : dim <= 32 ? 32 : 64; }

template <int DP>
static int glow_launch(const GlowArgs& a, bool bwd, hipStream_t stream) {
  const size_t lds = (bwd ? 7 : 5) * (size_t)DP * DP * sizeof(float);
  static DeviceMemo attr;
  if (attr.get([](int) {
        return glow_attr(glow_weight_kernel<DP>, 5 * (size_t)DP * DP * sizeof(float)) > 0 &&
                       glow_attr(glow_weight_bwd_kernel<DP>, 7 * (size_t)DP * DP * sizeof(float)) > 0
                   ? 1
                   : -1;
      }) < 0)
    return MNF_ERR_LAUNCH;
  if (bwd)
    hipLaunchKernelGGL(glow_weight_bwd_kernel<DP>, dim3(1), dim3(kGlowThreads), lds, stream, a);
  else
    hipLaunchKernelGGL(glow_weight_kernel<DP>, dim3(1), dim3(kGlowThreads), lds, stream, a);
  return check_launch();
}

static int glow_dispatch(const GlowArgs& a, bool bwd, hipStream_t stream) {
  switch (glow_pad(a.d)) {
    case 8: return glow_launch<8>(a, bwd, stream);
    case 16: return glow_launch<16>(a, bwd, stream);
    case 32: return glow_launch<32>(a, bwd, stream);
    default: return glow_launch<64>(a, bwd, stream);
  }
}

extern "C" {

int mnf_glow_weight(const float* P, const float* L, const float* S, const float* U, float* out, float* log_det, int dim,
                    int inverse, void* stream) {
  if (!P || !L || !S || !U || !out || !log_det || dim < 1) return MNF_ERR_INVALID_ARG;
  if (dim > kGlowMaxDim) return MNF_ERR_UNSUPPORTED;
  GlowArgs a{P, L, S, U, out, log_det, nullptr, nullptr, nullptr, nullptr, nullptr, dim, inverse != 0, 0, nullptr};
  return glow_dispatch(a, false, (hipStream_t)stream);
}

int mnf_glow_weight_bwd(const float* P, const float* L, const float* S, const float* U, const float* grad_out,
                        const float* grad_log_det, float* grad_L, float* grad_S, float* grad_U, int dim, int inverse,
                        int accumulate, const float* out_fwd, void* stream) {
  if (!P || !L || !S || !U || !grad_L || !grad_S || !grad_U || dim < 1) return MNF_ERR_INVALID_ARG;
  if (dim > kGlowMaxDim) return MNF_ERR_UNSUPPORTED;
  GlowArgs a{P, L, S, U, nullptr, nullptr, grad_out, grad_log_det, grad_L, grad_S, grad_U, dim, inverse != 0, accumulate != 0,
             inverse ? out_fwd : nullptr};
  return glow_dispatch(a, true, (hipStream_t)stream);
}

}  // extern "C"
