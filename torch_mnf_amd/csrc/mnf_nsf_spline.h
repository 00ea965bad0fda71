// The rational-quadratic spline of NSF_CL for one element whose 3K-1 raw parameters sit in registers
// (torch_mnf/flows/spline_flow.py:71-179 behind the double normalisation of :252-256), shared by the per-shape kernels
// (mnf_nsf_mfma.hip) and the run-time-shaped one (mnf_nsf_rt.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "mnf_device.h"

namespace mnf {

// exp for softmax terms (argument <= 0 after the max is subtracted): v_exp_f32 on x*log2(e).
// Relative error ~ |x| * 4e-8, and a term's weight in the sum is exp(x) itself, so the error
// it contributes to a normalised fraction is far below one ulp.
__device__ __forceinline__ float exp_sm(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float log_fast(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
// softplus(x) = max(x, 0) + log(1 + exp(-|x|)).  F.softplus switches to the identity above 20;
// there this form differs from x by log1p(e^-20) = 2e-9 < ulp(20)/2, i.e. it rounds to x as well.
__device__ __forceinline__ float softplus_fast(float x) {
  return __builtin_fmaxf(x, 0.f) + log_fast(1.f + exp_sm(-__builtin_fabsf(x)));
}
// the same for x >= 0 (second level: its argument is a softplus value or the edge constant)
__device__ __forceinline__ float softplus_pos(float x) { return x + log_fast(1.f + exp_sm(-x)); }

// K+1 knots on [-T, T] from K raw (first-level) parameters: 2T*softmax -> softmax again ->
// 1e-3 + (1 - 1e-3 K) p -> sequential cumsum -> 2T c - T -> ends forced (spline_flow.py:254-255,
// :95-101).  u[] is consumed.
template <int K>
__device__ __forceinline__ void knots_from_raw(const float (&u)[K], float T, float (&knot)[K + 1]) {
  // Instruction diet (this kernel is VALU bound): every exp argument is one fma with the scale
  // and the subtracted maximum folded into constants.  A rounding error in a folded constant
  // multiplies all K terms alike and cancels in the normalisation.
  constexpr float L2E = 1.44269504088896341f;
  const float twoT = 2.f * T;
  const float c1 = 1.f - kMinBin * (float)K;
  float m = u[0];
#pragma unroll
  for (int k = 1; k < K; ++k) m = __builtin_fmaxf(m, u[k]);
  const float mb = m * L2E;
  float e[K];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    e[k] = __builtin_amdgcn_exp2f(__builtin_fmaf(u[k], L2E, -mb));  // exp(u - max u)
    s += e[k];
  }
  const float r = rcp_fast(s);
  // second level: exp(2T p_k - 2T p_max), p_k = e_k r, p_max = r (the max element has e = 1)
  const float a2 = (twoT * r) * L2E;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    e[k] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[k], a2, -a2));
    s2 += e[k];
  }
  // knot_{k+1} = knot_k + 2T (1e-3 + (1 - 1e-3 K) p2_k): the cumulative sum carried in knot units
  const float g = (twoT * c1) * rcp_fast(s2), w0 = twoT * kMinBin;
  knot[0] = -T;
#pragma unroll
  for (int k = 0; k < K - 1; ++k) knot[k + 1] = knot[k] + __builtin_fmaf(e[k], g, w0);
  knot[K] = T;
}

// Spline for one element with all 3K-1 raw parameters in registers: p[0..K) widths, p[K..2K)
// heights, p[2K..3K-1) derivatives.  Same maths as rqs_element<true> (mnf_device.h).
template <int K, bool INV, int NP>
__device__ __forceinline__ void rqs_regs(float v, float T, const float (&p)[NP], float& out, float& lad) {
  const bool inside = (v >= -T) && (v <= T);  // NaN -> outside -> identity
  float uw[K], uh[K], xk[K + 1], yk[K + 1];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    uw[k] = p[k];
    uh[k] = p[K + k];
  }
  knots_from_raw<K>(uw, T, xk);
  knots_from_raw<K>(uh, T, yk);
  // bin = count(v >= knot) - 1 over the K+1 knots (last one nudged by 1e-6): the last k with
  // v >= knot_k.  Select the bin's quantities with compile-time indices.
  const float vs = inside ? v : 0.f;
  float x_k = xk[0], x_k1 = xk[1], y_k = yk[0], y_k1 = yk[1];
  // raw derivative parameters at the bin's two knots: D[bin-1] and D[bin]; the outermost knots
  // carry the edge constant instead (spline_flow.py:46-49)
  float r0 = 0.f, r1 = p[2 * K];
  bool bin0 = true, binlast = false;
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool hit = vs >= (INV ? yk[k] : xk[k]);  // knots increase: hits are a prefix of k
    x_k = hit ? xk[k] : x_k;
    x_k1 = hit ? xk[k + 1] : x_k1;
    y_k = hit ? yk[k] : y_k;
    y_k1 = hit ? yk[k + 1] : y_k1;
    r0 = hit ? p[2 * K + k - 1] : r0;
    if (k < K - 1) r1 = hit ? p[2 * K + k] : r1;
    if (k == 1) bin0 = !hit;
    if (k == K - 1) binlast = hit;
  }
  // first-level softplus for interior knots (:256), then 1e-3 + softplus(padded value) (:104)
  const float dk_in = bin0 ? kEdgeDerivConst : softplus_fast(r0);
  const float dk1_in = binlast ? kEdgeDerivConst : softplus_fast(r1);
  const float d_k = kMinDeriv + softplus_pos(dk_in);
  const float d_k1 = kMinDeriv + softplus_pos(dk1_in);

  const float w_k = x_k1 - x_k, h_k = y_k1 - y_k;
  const float rw = rcp_fast(w_k);
  const float delta = h_k * rw;
  float o, l;
  if (INV) {
    const float dy = vs - y_k;
    const float curv = d_k + d_k1 - 2.f * delta;
    const float a = dy * curv + h_k * (delta - d_k);
    const float b = h_k * d_k - dy * curv;
    const float c = -delta * dy;
    const float disc = b * b - 4.f * a * c;
    const float root = (2.f * c) * rcp_fast(-b - __builtin_amdgcn_sqrtf(disc));
    o = root * w_k + x_k;
    const float tomt = root * (1.f - root);
    const float denom = delta + curv * tomt;
    const float omr = 1.f - root;
    const float dnum = (delta * delta) * (d_k1 * (root * root) + 2.f * delta * tomt + d_k * (omr * omr));
    l = 2.f * log_fast(denom) - log_fast(dnum);
  } else {
    const float theta = (vs - x_k) * rw;
    const float tomt = theta * (1.f - theta);
    const float numer = h_k * (delta * (theta * theta) + d_k * tomt);
    const float denom = delta + (d_k + d_k1 - 2.f * delta) * tomt;
    o = y_k + numer * rcp_fast(denom);
    const float omt = 1.f - theta;
    const float dnum = (delta * delta) * (d_k1 * (theta * theta) + 2.f * delta * tomt + d_k * (omt * omt));
    l = log_fast(dnum) - 2.f * log_fast(denom);
  }
  out = inside ? o : v;
  lad = inside ? l : 0.f;
}

}  // namespace mnf
