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

// One spline axis (spline_flow.py:254-255, :95-101: 2T softmax, softmax again, 1e-3 + (1 - 1e-3 K) p, cumulative sum)
// with every exp argument one fma (as mnf_nsf_mfma.hip knots_from_raw), keeping what the derivative needs: the UNNORMALISED exponentials of both levels (e1, e2) and the two reciprocals (p1 = e1 r, p2 = e2 r2).
template <int K, int OFF, int NP>
__device__ __forceinline__ void axis_keep(const float (&p)[NP], float T, float (&e1)[K], float (&e2)[K], float& r,
                                          float& r2, float (&knot)[K + 1]) {
  constexpr float L2E = 1.44269504088896341f;
  const float twoT = 2.f * T, c1 = 1.f - kMinBin * (float)K;
  float m = p[OFF];
#pragma unroll
  for (int k = 1; k < K; ++k) m = fmaxf(m, p[OFF + k]);
  const float mb = m * L2E;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    e1[k] = __builtin_amdgcn_exp2f(fmaf(p[OFF + k], L2E, -mb));  // exp(u - max u)
    s += e1[k];
  }
  r = rcp(s);
  // second level: exp(2T p1_k - 2T p1_max), p1_k = e1_k r, p1_max = r (the max element has e1 = 1)
  const float a2 = (twoT * r) * L2E;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    e2[k] = __builtin_amdgcn_exp2f(fmaf(e1[k], a2, -a2));
    s2 += e2[k];
  }
  r2 = rcp(s2);
  // the knots in the reference's own order of operations -- normalise, 1e-3 + c1 p2, running sum, 2T c - T, ends forced
  // (spline_flow.py:95-101) --: the folded form knot_{k+1} = knot_k + e2_k (2T c1 r2) + 2T 1e-3 the forward kernel uses
  // rounds the interior knots differently, and d log|dy/dx| / dx jumps at a knot: an element within rounding of one
  // would take the other side's derivative than the reference does (one such element in tests' 32 k: grad_x 6e-2 off)
  float c = 0.f;
  knot[0] = -T;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    c += kMinBin + c1 * (e2[k] * r2);
    knot[k + 1] = (k == K - 1) ? T : twoT * c - T;
  }
}

// Gradient wrt the K raw parameters of an axis given the cotangents of the bin's two knots (g_lo: knot_b, g_hi:
// knot_{b+1}; knot_0 and knot_K are constants).  knot_i = -T + 2T sum_{k < i} (1e-3 + c1 p2_k), so the cotangent of p2_k
// is A = 2T c1 (g_lo + g_hi) below the bin, B = 2T c1 g_hi in it, 0 above -- selected with the bin search's own
// comparison masks (hit[i]: v >= knot_i; a prefix) -- and sum_k p2_k g_k = A C_b + B p2_b needs no loop: the cumulative
// fraction C_b and the bin's own fraction are read off the bin's knot and width (k0, wbin).  Both softmax Jacobians act
// on the unnormalised exponentials with the reciprocals folded into scalars.
template <int K, int OFF, int NP>
__device__ __forceinline__ void axis_grad(const float (&e1)[K], const float (&e2)[K], float r, float r2, float T,
                                          const bool (&hit)[K + 1], float fb, float k0, float wbin, float g_lo,
                                          float g_hi, float (&g_p)[NP]) {
  const float twoT = 2.f * T, c1 = 1.f - kMinBin * (float)K, sc = twoT * c1, inv = rcp(sc);
  const float gl = hit[1] ? g_lo : 0.f;       // b >= 1
  const float gh = hit[K - 1] ? 0.f : g_hi;   // b + 1 <= K - 1
  const float B = sc * gh, A = sc * (gl + gh);
  const float Cb = ((k0 + T) - fb * (twoT * kMinBin)) * inv;  // ((knot_b + T) / 2T - b 1e-3) / c1
  const float pb = (wbin - twoT * kMinBin) * inv;             // (width / 2T - 1e-3) / c1
  const float dot2 = A * Cb + B * pb;
  // g2_k = 2T p2_k (g_k - dot2) = e2_k (2T r2) (g_k - dot2)
  const float q2 = twoT * r2;
  const float tA = q2 * (A - dot2), tB = q2 * (B - dot2), tZ = -q2 * dot2;
  float g2[K];
  float d1 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float t = hit[k + 1] ? tA : (hit[k] ? tB : tZ);
    g2[k] = e2[k] * t;
    d1 = fmaf(e1[k], g2[k], d1);
  }
  // p1_k (g2_k - sum_j p1_j g2_j) = e1_k (r g2_k - r^2 d1)
  const float d1rr = d1 * r * r;
#pragma unroll
  for (int k = 0; k < K; ++k) g_p[OFF + k] = e1[k] * fmaf(g2[k], r, -d1rr);
}

// the bin of v, its knots / raw derivative parameters and the search's comparison masks (hit[0] = true, hit[K] = false:
// hit[i] <=> i <= bin), all indices compile-time
template <int K>
struct BinOf {
  float x0, x1, y0, y1, raw0, raw1;
  int b;
  bool hit[K + 1];
};
template <int K, bool INV, int NP>
__device__ __forceinline__ BinOf<K> find_bin(float vs, const float (&xk)[K + 1], const float (&yk)[K + 1],
                                             const float (&p)[NP]) {
  BinOf<K> o;
  o.b = 0;
  o.x0 = xk[0]; o.x1 = xk[1]; o.y0 = yk[0]; o.y1 = yk[1];
  o.raw0 = 0.f;
  o.raw1 = p[2 * K];
  o.hit[0] = true;
  o.hit[K] = false;
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool hit = vs >= (INV ? yk[k] : xk[k]);  // knots increase: hits are a prefix
    o.hit[k] = hit;
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

// reverse-mode derivative of the spline for one element: g_o, g_l are the cotangents of (output, log-derivative);
// returns the gradient wrt v and the 3K-1 raw parameters (the maths of rqs_element_bwd in mnf_backward.hip)
template <int K, bool INV>
__device__ __forceinline__ void rqs_grad(float v, float T, const float (&p)[3 * K - 1], float g_out, float g_ld,
                                         float& g_v, float (&g_p)[3 * K - 1]) {
  const bool inside = (v >= -T) && (v <= T);  // identity tails: g_v = g_out, no parameter gradient
  const float vs = inside ? v : 0.f, g_o = inside ? g_out : 0.f, g_l = inside ? g_ld : 0.f;
  float e1w[K], e2w[K], e1h[K], e2h[K], xk[K + 1], yk[K + 1], rw1, rw2, rh1, rh2;
  axis_keep<K, 0>(p, T, e1w, e2w, rw1, rw2, xk);
  axis_keep<K, K>(p, T, e1h, e2h, rh1, rh2, yk);
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
  const float fb = (float)b;
  axis_grad<K, 0>(e1w, e2w, rw1, rw2, T, bin.hit, fb, x0, w, g_x0, g_x1, g_p);
  axis_grad<K, K>(e1h, e2h, rh1, rh2, T, bin.hit, fb, y0, h, g_y0, g_y1, g_p);
}

}  // namespace nsfgrad
}  // namespace mnf
