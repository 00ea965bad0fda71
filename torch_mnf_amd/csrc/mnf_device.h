// Device-side helpers shared by the generic and the MFMA kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnf_hip.h"

namespace mnf {

constexpr float kLeakySlope = 0.2f;  // models/mlp.py:7
constexpr float kMinBin = 1e-3f;     // spline_flow.py:17-18 (width and height share it)
constexpr float kMinDeriv = 1e-3f;   // spline_flow.py:19
// log(exp(1 - 1e-3) - 1) evaluated in float64 then rounded (spline_flow.py:47-49)
constexpr float kEdgeDerivConst = 0.53974241439865964f;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
// Index-table entries of the pack kernels: >= 0 a flat-parameter offset, -1 zero, kPackBigBias the constant
// kPackBigBiasValue (a padded RNVP output dim's scale bias: sigmoid = 1 and its log = 0 exactly in fp32).
constexpr int32_t kPackBigBias = -2;
constexpr float kPackBigBiasValue = 80.f;

// Shapes of one conditioner net: n_lin Linear layers, sizes[0..n_lin], and the float
// offset of each weight / bias inside the flat parameter buffer.
struct NetDesc {
  int n_lin;
  int sizes[MNF_MAX_LINEAR + 1];
  int w_off[MNF_MAX_LINEAR];
  int b_off[MNF_MAX_LINEAR];
  int max_width;  // max over sizes
};

__device__ __forceinline__ float leaky(float v) { return fmaxf(v, kLeakySlope * v); }

// F.softplus with beta=1, threshold=20 (linear above the threshold).
__device__ __forceinline__ float softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// ------------------------------------------------------------------------------------
// Rational-quadratic spline for ONE element, streaming over the K bins so that no
// per-bin array is needed: `get(k)` returns the k-th raw value of a parameter group.
//
//   DOUBLE = true  : raw net outputs; NSF_CL's first normalisation (2T*softmax, softplus,
//                    spline_flow.py:254-256) is applied and then RQS's own (:95-113)
//   DOUBLE = false : inputs are what unconstrained_RQS receives (mnf_rqs entry point)
//
// Knots follow the reference: fractions 1e-3 + (1 - 1e-3 K) softmax, sequential cumsum,
// 2T*c - T, first/last knot forced to -T / +T, bin sizes re-derived by differencing.
// The bin is count(v >= knot) - 1 over K+1 knots with the last one nudged by 1e-6; for an
// inside element that is the last k < K with v >= knot_k.
// ------------------------------------------------------------------------------------
template <bool DOUBLE, typename GetW, typename GetH, typename GetD>
__device__ __forceinline__ void rqs_element(float v, int K, float T, bool inverse, GetW getW,
                                            GetH getH, GetD getD, float& out, float& lad) {
// ATen evaluates the reference's formulas one op at a time, each rounded to fp32; keep the
// same roundings (no fused multiply-add) so that ill-conditioned elements near a knot land
// on the same side as the reference as often as possible.
#pragma clang fp contract(off)
  const bool inside = (v >= -T) && (v <= T);  // NaN -> outside -> identity
  if (!inside) {
    out = v;
    lad = 0.f;
    return;
  }
  const float twoT = 2.f * T;
  const float c1 = 1.f - kMinBin * (float)K;  // (1 - min_bin * K)

  // softmax normalisers.  First level (DOUBLE): p = exp(u - max u) / sum; u' = 2T p.
  float mW = -INFINITY, mH = -INFINITY;
  for (int k = 0; k < K; ++k) {
    mW = fmaxf(mW, getW(k));
    mH = fmaxf(mH, getH(k));
  }
  float sW = 0.f, sH = 0.f;
  for (int k = 0; k < K; ++k) {
    sW += expf(getW(k) - mW);
    sH += expf(getH(k) - mH);
  }
  // Second level max/sum (or the only level when !DOUBLE: m2 = max, s2 = sum above).
  float m2W, m2H, s2W, s2H, rW = 1.f, rH = 1.f;
  if (DOUBLE) {
    rW = 1.f / sW;  // ATen's softmax multiplies by the reciprocal of the sum
    rH = 1.f / sH;
    m2W = twoT * rW;  // the max element has exp(0) = 1
    m2H = twoT * rH;
    s2W = 0.f;
    s2H = 0.f;
    for (int k = 0; k < K; ++k) {
      s2W += expf(twoT * (expf(getW(k) - mW) * rW) - m2W);
      s2H += expf(twoT * (expf(getH(k) - mH) * rH) - m2H);
    }
  } else {
    m2W = mW;
    m2H = mH;
    s2W = sW;
    s2H = sH;
  }

  const float r2W = 1.f / s2W, r2H = 1.f / s2H;
  // stream over the bins: running cumsum for both axes, remember the selected bin
  float cw = 0.f, ch = 0.f;      // cumulative fractions
  float xk_prev = -T, yk_prev = -T;
  float x_k = -T, w_k = 1.f, y_k = -T, h_k = 1.f;
  int bin = 0;
  for (int k = 0; k < K; ++k) {
    float uW = getW(k), uH = getH(k);
    if (DOUBLE) {
      uW = twoT * (expf(uW - mW) * rW);
      uH = twoT * (expf(uH - mH) * rH);
    }
    const float fw = kMinBin + c1 * (expf(uW - m2W) * r2W);
    const float fh = kMinBin + c1 * (expf(uH - m2H) * r2H);
    cw += fw;
    ch += fh;
    const float xk_next = (k == K - 1) ? T : twoT * cw + (-T);
    const float yk_next = (k == K - 1) ? T : twoT * ch + (-T);
    const float probe = inverse ? yk_prev : xk_prev;
    if (v >= probe) {  // knots increase, so the last hit is the bin
      bin = k;
      x_k = xk_prev;
      w_k = xk_next - xk_prev;
      y_k = yk_prev;
      h_k = yk_next - yk_prev;
    }
    xk_prev = xk_next;
    yk_prev = yk_next;
  }
  // derivatives at the two knots of the bin: index 0 and K carry the edge constant
  float d_k, d_k1;
  {
    float r0 = (bin == 0) ? kEdgeDerivConst : (DOUBLE ? softplus(getD(bin - 1)) : getD(bin - 1));
    float r1 = (bin == K - 1) ? kEdgeDerivConst : (DOUBLE ? softplus(getD(bin)) : getD(bin));
    d_k = kMinDeriv + softplus(r0);
    d_k1 = kMinDeriv + softplus(r1);
  }
  const float delta = h_k / w_k;
  if (inverse) {
    const float dy = v - y_k;
    const float curv = d_k + d_k1 - 2.f * delta;
    const float a = dy * curv + h_k * (delta - d_k);
    const float b = h_k * d_k - dy * curv;
    const float c = -delta * dy;
    const float disc = b * b - 4.f * a * c;
    const float root = (2.f * c) / (-b - sqrtf(disc));
    out = root * w_k + x_k;
    const float tomt = root * (1.f - root);
    const float denom = delta + curv * tomt;
    const float omr = 1.f - root;
    const float dnum = (delta * delta) * (d_k1 * (root * root) + 2.f * delta * tomt + d_k * (omr * omr));
    lad = -(logf(dnum) - 2.f * logf(denom));
  } else {
    const float theta = (v - x_k) / w_k;
    const float tomt = theta * (1.f - theta);
    const float numer = h_k * (delta * (theta * theta) + d_k * tomt);
    const float denom = delta + (d_k + d_k1 - 2.f * delta) * tomt;
    out = y_k + numer / denom;
    const float omt = 1.f - theta;
    const float dnum = (delta * delta) * (d_k1 * (theta * theta) + 2.f * delta * tomt + d_k * (omt * omt));
    lad = logf(dnum) - 2.f * logf(denom);
  }
}

// Counter-based Bernoulli(0.5) mask for RNVP (the reference draws torch.bernoulli per call,
// rnvp.py:28): bit (dim & 31) of a 32-bit hash of (seed, row, dim >> 5).  Stateless, so the
// GEMM-1 operand pass and the epilogue of the same kernel regenerate identical bits, and
// mnf_rnvp_mask() can materialise exactly the mask a seeded call used.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t rnvp_mask_word(uint64_t seed, int64_t row, int word) {
  const uint32_t a = mix32((uint32_t)row * 0x9e3779b1u + (uint32_t)((uint64_t)row >> 32) + (uint32_t)(seed >> 32));
  return mix32(a ^ ((uint32_t)word * 0x85ebca77u + (uint32_t)seed));
}
__device__ __forceinline__ float rnvp_mask_bit(uint64_t seed, int64_t row, int dim) {
  return (float)((rnvp_mask_word(seed, row, dim >> 5) >> (dim & 31)) & 1u);
}

// counter-based N(0, 1) for the in-kernel noise: Box-Muller on two hashes of (seed, row, column).  Stateless, so
// mnf_mnf_linear_noise() materialises exactly the numbers a seeded call used.
__device__ __forceinline__ float ml_normal(uint64_t seed, int64_t row, int col) {
  const uint32_t a = mix32((uint32_t)row * 0x9e3779b1u + (uint32_t)((uint64_t)row >> 32) + (uint32_t)(seed >> 32));
  const uint32_t h1 = mix32(a ^ ((uint32_t)col * 0x85ebca77u + (uint32_t)seed));
  const uint32_t h2 = mix32(h1 ^ 0x68bc21ebu);
  const float u1 = ((float)(h1 >> 8) + 0.5f) * (1.f / 16777216.f);  // (0, 1)
  const float u2 = ((float)(h2 >> 8) + 0.5f) * (1.f / 16777216.f);
  // hardware transcendentals (v_log_f32 = log2, v_cos_f32 takes revolutions): the stream is defined by these
  // instructions, and mnf_mnf_linear_noise() reproduces it with the same ones
  return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}

// The sample_z prologue's noise (mnf_sample_z0_seeded): both Box-Muller outputs of one (u1, u2) serve the column pair
// (2 c, 2 c + 1) of a row, and the row's hash is shared by its columns -- one hash pair, one log, one sqrt per TWO
// numbers (ml_normal spends three hashes per number: 0.38 ms per 256,000 x 800 prologue launch against 0.14 ms of
// stores).  mnf_sample_z0_noise() materialises exactly these numbers.
__device__ __forceinline__ uint32_t z0_row_hash(uint64_t seed, int64_t row) {
  return mix32((uint32_t)row * 0x9e3779b1u + (uint32_t)((uint64_t)row >> 32) + (uint32_t)(seed >> 32));
}
__device__ __forceinline__ void z0_normal_pair(uint32_t row_hash, uint32_t seed_lo, int pair, float& n0, float& n1) {
  const uint32_t h1 = mix32(row_hash ^ ((uint32_t)pair * 0x85ebca77u + (seed_lo ^ 0x5bd1e995u)));
  const uint32_t h2 = mix32(h1 ^ 0x68bc21ebu);
  const float u1 = ((float)(h1 >> 8) + 0.5f) * (1.f / 16777216.f);  // (0, 1)
  const float u2 = ((float)(h2 >> 8) + 0.5f) * (1.f / 16777216.f);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // (v_log_f32 = log2)
  n0 = r * __builtin_amdgcn_cosf(u2);                                                           // (revolutions)
  n1 = r * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ float z0_normal(uint64_t seed, int64_t row, int col) {
  float n0, n1;
  z0_normal_pair(z0_row_hash(seed, row), (uint32_t)seed, col >> 1, n0, n1);
  return (col & 1) ? n1 : n0;
}

// sum over the 4 lanes {j, j+16, j+32, j+48} that share a sample in the 16x16 MFMA layout
__device__ __forceinline__ float sum_over_q(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

}  // namespace mnf
