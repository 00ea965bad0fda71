// Rational-quadratic spline of NSF_CL for ONE element with all 3K-1 raw parameters in registers: value and reverse-mode
// derivative (spline_flow.py:22-179, :254-256), shared by the NSF_CL gradient kernels (mnf_nsf_bwd_rows.hip: a lane per
// (row, element); mnf_nsf_bwd_tile.hip: 16-row tiles on the matrix cores).  Every index is a compile-time constant.
#pragma once
#include <hip/hip_runtime.h>

#include "mnf_device.h"

namespace mnf {
namespace nsfgrad {

// one-instruction reciprocal / square root / exp / log (1 ulp): gradients are compared at 1e-4 and summed over rows
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float log_f(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
// softplus(x) = max(x, 0) + log(1 + exp(-|x|)); above F.softplus's threshold (20) this rounds to x as well
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log_f(1.f + exp_f(-fabsf(x))); }
__device__ __forceinline__ float softplus_slope(float x) { return x > 20.f ? 1.f : rcp(1.f + exp_f(-x)); }

// one spline axis keeping both softmax levels: p1, p2 and the K+1 knots (spline_flow.py:254-255, :95-101)
template <int K, int OFF, int NP>
__device__ __forceinline__ void axis_keep(const float (&p)[NP], float T, float (&p1)[K], float (&p2)[K],
                                          float (&knot)[K + 1]) {
  const float twoT = 2.f * T, c1 = 1.f - kMinBin * (float)K;
  float m = p[OFF];
#pragma unroll
  for (int k = 1; k < K; ++k) m = fmaxf(m, p[OFF + k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    p1[k] = exp_f(p[OFF + k] - m);
    s += p1[k];
  }
  const float r = rcp(s);
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    p1[k] *= r;
    p2[k] = exp_f(twoT * p1[k] - twoT * r);  // the largest p1 is r
    s2 += p2[k];
  }
  const float r2 = rcp(s2);
  float c = 0.f;
  knot[0] = -T;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    p2[k] *= r2;
    c += kMinBin + c1 * p2[k];
    knot[k + 1] = (k == K - 1) ? T : twoT * c - T;
  }
}

// gradient wrt the K raw parameters of an axis given the gradients of knot_b and knot_{b+1}
template <int K, int OFF, int NP>
__device__ __forceinline__ void axis_grad(const float (&p1)[K], const float (&p2)[K], float T, int b, float g_lo,
                                          float g_hi, float (&g_p)[NP]) {
  const float twoT = 2.f * T, c1 = 1.f - kMinBin * (float)K;
  // knot_i = 2T cum_{i-1} - T for 1 <= i <= K-1 (knot_0, knot_K are constants)
  float g[K];
  float dot2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float gf = 0.f;
    gf += (b >= 1 && k < b) ? g_lo : 0.f;
    gf += (b + 1 <= K - 1 && k <= b) ? g_hi : 0.f;
    g[k] = c1 * twoT * gf;
    dot2 += p2[k] * g[k];
  }
  float dot1 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    g[k] = twoT * (p2[k] * (g[k] - dot2));
    dot1 += p1[k] * g[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) g_p[OFF + k] = p1[k] * (g[k] - dot1);
}


// the bin of v and its knots / raw derivative parameters, all indices compile-time
template <int K>
struct BinOf {
  float x0, x1, y0, y1, raw0, raw1;
  int b;
};
template <int K, bool INV, int NP>
__device__ __forceinline__ BinOf<K> find_bin(float vs, const float (&xk)[K + 1], const float (&yk)[K + 1],
                                             const float (&p)[NP]) {
  BinOf<K> o;
  o.b = 0;
  o.x0 = xk[0]; o.x1 = xk[1]; o.y0 = yk[0]; o.y1 = yk[1];
  o.raw0 = 0.f;
  o.raw1 = p[2 * K];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool hit = vs >= (INV ? yk[k] : xk[k]);  // knots increase: hits are a prefix
    o.b = hit ? k : o.b;
    o.x0 = hit ? xk[k] : o.x0;
    o.x1 = hit ? xk[k + 1] : o.x1;
    o.y0 = hit ? yk[k] : o.y0;
    o.y1 = hit ? yk[k + 1] : o.y1;
    o.raw0 = hit ? p[2 * K + k - 1] : o.raw0;
    if (k < K - 1) o.raw1 = hit ? p[2 * K + k] : o.raw1;
  }
  return o;
}

// spline value only (the half-step whose output conditions the other net)
template <int K, bool INV>
__device__ __forceinline__ float rqs_value(float v, float T, const float (&p)[3 * K - 1]) {
  const bool inside = (v >= -T) && (v <= T);
  const float vs = inside ? v : 0.f;
  float p1[K], p2[K], xk[K + 1], yk[K + 1];
  axis_keep<K, 0>(p, T, p1, p2, xk);
  axis_keep<K, K>(p, T, p1, p2, yk);
  const BinOf<K> bin = find_bin<K, INV>(vs, xk, yk, p);
  const float pad0 = bin.b == 0 ? kEdgeDerivConst : softplus_f(bin.raw0);
  const float pad1 = bin.b == K - 1 ? kEdgeDerivConst : softplus_f(bin.raw1);
  const float d0 = kMinDeriv + softplus_f(pad0), d1 = kMinDeriv + softplus_f(pad1);
  const float w = bin.x1 - bin.x0, h = bin.y1 - bin.y0, rw = rcp(w), delta = h * rw;
  float o;
  if (INV) {
    const float dy = vs - bin.y0, cv = d0 + d1 - 2.f * delta;
    const float a = dy * cv + h * (delta - d0), bb = h * d0 - dy * cv, c = -delta * dy;
    const float root = (2.f * c) * rcp(-bb - __builtin_amdgcn_sqrtf(bb * bb - 4.f * a * c));
    o = root * w + bin.x0;
  } else {
    const float th = (vs - bin.x0) * rw, t1 = th * (1.f - th);
    o = bin.y0 + h * (delta * th * th + d0 * t1) * rcp(delta + (d0 + d1 - 2.f * delta) * t1);
  }
  return inside ? o : v;
}

// reverse-mode derivative of the spline for one element: g_o, g_l are the cotangents of (output, log-derivative);
// returns the gradient wrt v and the 3K-1 raw parameters (the maths of rqs_element_bwd in mnf_backward.hip)
template <int K, bool INV>
__device__ __forceinline__ void rqs_grad(float v, float T, const float (&p)[3 * K - 1], float g_out, float g_ld,
                                         float& g_v, float (&g_p)[3 * K - 1]) {
  const bool inside = (v >= -T) && (v <= T);  // identity tails: g_v = g_out, no parameter gradient
  const float vs = inside ? v : 0.f, g_o = inside ? g_out : 0.f, g_l = inside ? g_ld : 0.f;
  float p1w[K], p2w[K], p1h[K], p2h[K], xk[K + 1], yk[K + 1];
  axis_keep<K, 0>(p, T, p1w, p2w, xk);
  axis_keep<K, K>(p, T, p1h, p2h, yk);
  const BinOf<K> bin = find_bin<K, INV>(vs, xk, yk, p);
  const int b = bin.b;
  const float x0 = bin.x0, x1 = bin.x1, y0 = bin.y0, y1 = bin.y1, raw0 = bin.raw0, raw1 = bin.raw1;
  const float pad0 = b == 0 ? kEdgeDerivConst : softplus_f(raw0);
  const float pad1 = b == K - 1 ? kEdgeDerivConst : softplus_f(raw1);
  const float d0 = kMinDeriv + softplus_f(pad0), d1 = kMinDeriv + softplus_f(pad1);
  const float w = x1 - x0, h = y1 - y0, rw = rcp(w), delta = h * rw;
  float g_x0 = 0.f, g_x1 = 0.f, g_y0 = 0.f, g_y1 = 0.f, g_d0 = 0.f, g_d1 = 0.f;
  float g_w = 0.f, g_h = 0.f, g_delta = 0.f, gv = 0.f;
  if (!INV) {
    const float th = (vs - x0) * rw, t1 = th * (1.f - th), omt = 1.f - th;
    const float B = delta * th * th + d0 * t1, N = h * B;
    const float cv = d0 + d1 - 2.f * delta, Dn = delta + cv * t1;
    const float A = d1 * th * th + 2.f * delta * t1 + d0 * omt * omt, dn = delta * delta * A;
    const float rDn = rcp(Dn);
    const float gN = g_o * rDn, gDn = -g_o * N * (rDn * rDn) - 2.f * g_l * rDn, g_dn = g_l * rcp(dn);
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
    gv = g_th * rw; g_x0 -= g_th * rw; g_w -= g_th * th * rw;
  } else {
    const float dy = vs - y0, cv = d0 + d1 - 2.f * delta;
    const float a = dy * cv + h * (delta - d0), bb = h * d0 - dy * cv, c = -delta * dy;
    const float disc = bb * bb - 4.f * a * c, sq = __builtin_amdgcn_sqrtf(disc), den = -bb - sq, rden = rcp(den);
    const float xi = 2.f * c * rden;
    const float t1 = xi * (1.f - xi), omx = 1.f - xi, Dn = delta + cv * t1;
    const float A = d1 * xi * xi + 2.f * delta * t1 + d0 * omx * omx, dn = delta * delta * A;
    float g_xi = g_o * w, g_t1 = 0.f, g_cv = 0.f;
    g_w += g_o * xi; g_x0 += g_o;
    const float gDn = 2.f * g_l * rcp(Dn), g_dn = -g_l * rcp(dn);
    g_delta += g_dn * (2.f * delta * A + 2.f * delta * delta * t1);
    const float gA = g_dn * delta * delta;
    g_d1 += gA * xi * xi; g_d0 += gA * omx * omx; g_xi += gA * (2.f * d1 * xi - 2.f * d0 * omx); g_t1 += gA * 2.f * delta;
    g_delta += gDn; g_cv += gDn * t1; g_t1 += gDn * cv;
    g_xi += g_t1 * (1.f - 2.f * xi);
    float g_c = 2.f * g_xi * rden;
    const float g_den = -g_xi * xi * rden;
    float g_b = -g_den;
    const float g_disc = -g_den * (0.5f * rcp(sq));
    g_b += 2.f * bb * g_disc;
    const float g_a = -4.f * c * g_disc;
    g_c += -4.f * a * g_disc;
    float g_dy = 0.f;
    g_delta += -dy * g_c; g_dy += -delta * g_c;
    g_h += d0 * g_b; g_d0 += h * g_b; g_dy += -cv * g_b; g_cv += -dy * g_b;
    g_dy += cv * g_a; g_cv += dy * g_a; g_h += (delta - d0) * g_a; g_delta += h * g_a; g_d0 += -h * g_a;
    g_d0 += g_cv; g_d1 += g_cv; g_delta += -2.f * g_cv;
    gv = g_dy; g_y0 -= g_dy;
  }
  g_h += g_delta * rw; g_w -= g_delta * delta * rw;
  g_y1 += g_h; g_y0 -= g_h; g_x1 += g_w; g_x0 -= g_w;
  g_v = inside ? gv : g_out;
  // derivative parameters: knot b is raw parameter b - 1, knot b + 1 is raw parameter b (the outermost are constants)
  const float gd0 = b == 0 ? 0.f : g_d0 * softplus_slope(pad0) * softplus_slope(raw0);
  const float gd1 = b == K - 1 ? 0.f : g_d1 * softplus_slope(pad1) * softplus_slope(raw1);
#pragma unroll
  for (int i = 0; i < K - 1; ++i) g_p[2 * K + i] = (i == b - 1 ? gd0 : 0.f) + (i == b ? gd1 : 0.f);
  axis_grad<K, 0>(p1w, p2w, T, b, g_x0, g_x1, g_p);
  axis_grad<K, K>(p1h, p2h, T, b, g_y0, g_y1, g_p);
}

}  // namespace nsfgrad
}  // namespace mnf
