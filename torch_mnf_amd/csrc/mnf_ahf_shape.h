// Shapes and in-register helpers shared by the single-layer and the whole-stack AffineHalfFlow kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "mnf_device.h"

namespace mnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// On gfx950 the fp32 MFMA runs at the fp32 VALU rate and does NOT overlap with VALU work of
// other waves on the same SIMD (measured: kernel time = 32 cycles per MFMA + 4 cycles per VALU
// instruction, tools/ahf_microbench.hip), so every vector instruction in the tile loop costs
// ~0.1 % of the layer.  The two helpers below exist to keep that count down.

// LeakyReLU(0.2) = max(v, 0.2 v) in two instructions.  fmaxf() would add a canonicalising
// v_max per MFMA output; the median of (v, 0.2 v, +inf) is the same value without it.
__device__ __forceinline__ float leaky2(float v) {
  return __builtin_amdgcn_fmed3f(v, kLeakySlope * v, __builtin_inff());
}

// exp(x) in six instructions, <= 1.5 ulp: x*log2(e) as a two-term product, v_exp_f32 on the
// head, first-order correction for the tail.  Overflow gives +inf, underflow 0, NaN stays NaN
// (same as expf; ocml's expf is about twice as long because of its explicit range checks).
__device__ __forceinline__ float exp6(float x) {
  const float c_hi = 1.44269502162933349609375f;    // fl32(log2 e)
  const float c_lo = 1.925963033500011e-08f;        // log2 e - c_hi
  const float ln2 = 0.693147182464599609375f;
  const float t = x * c_hi;
  const float err = __builtin_fmaf(x, c_hi, -t);
  const float tl = __builtin_fmaf(x, c_lo, err);
  const float e1 = __builtin_amdgcn_exp2f(t);
  return e1 * __builtin_fmaf(tl, ln2, 1.0f);
}


// exp(x) in TWO instructions for the split kernels' hot path: exp2(fl32(x log2 e)).  What it gives up against exp6 is the
// rounding of x log2 e, a relative error of |x| 4e-8 in the result -- below half an ulp for |x| <= 1.4 (there the
// two agree bit for bit: exp6's correction factor rounds to 1) and 4e-7 at |s| = 10, against a parity bar of 1e-5.
// Same limits: overflow gives +inf, underflow 0, NaN stays NaN.  Worth 3.1 % of the 9-layer C2 launch (784 -> 759 us,
// same box; tools/split_accuracy.py: the distance from float64 does not move).
#ifndef MNF_EXP_FAST
#define MNF_EXP_FAST 1  // 0: exp6 everywhere (A/B builds)
#endif
__device__ __forceinline__ float exp2x(float x) {
  return MNF_EXP_FAST ? __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f) : exp6(x);
}

template <int H, int HID>
struct AhfShape {
  static_assert(H % 16 == 0, "conditioner width must be a multiple of 16");
  static_assert(HID % 4 == 0, "hidden width must be a multiple of 4");
  static constexpr int G = H / 16;             // float4 groups per half row == output tiles per net
  static constexpr int QN = HID / 4;           // quads (K-steps) per net
  static constexpr int NQ = 2 * QN;            // quads of the concatenated hidden vector
  static constexpr int NT = (NQ + 3) / 4;      // 16-row tiles of the concatenated hidden vector
  static constexpr bool tile_has_net(int m, int net) {
    for (int c = 4 * m; c < 4 * m + 4 && c < NQ; ++c)
      if ((c >= QN) == (net == 1)) return true;
    return false;
  }
  static constexpr int hidden_mfmas() {
    int n = 0;
    for (int c = 0; c < NQ; ++c)
      for (int m = 0; m < NT; ++m)
        if (tile_has_net(m, c >= QN ? 1 : 0)) ++n;
    return n;
  }
  static constexpr int N_MFMA = (H / 4) * NT + 2 * hidden_mfmas() + G * 2 * QN;
  static constexpr int A_FLOATS = ((N_MFMA + 3) / 4) * 256;
  static constexpr int N_BIAS_TILES = 3 * NT + 2 * G;
  static constexpr int IMAGE_FLOATS = A_FLOATS + N_BIAS_TILES * 16;
};

// The two conditioner nets on fp32 MFMAs (16x16x4): cnd (accumulator layout: lane (j, q) reg r <-> dim
// 16 g + 4 q + r) -> raw s and t in the same layout.  img is the fp32 operand image, in LDS (hot path
// of the fp32 kernels) or in global memory (cold path of the split kernels).
template <int H, int HID>
__device__ __forceinline__ void ahf_cond_f32(const float* img, int lane, int q, const f32x4 (&cnd)[H / 16],
                                             f32x4 (&s4)[H / 16], f32x4 (&t4)[H / 16]) {
  using S = AhfShape<H, HID>;
  constexpr int G = S::G, QN = S::QN, NQ = S::NQ, NT = S::NT;
  int a_off = lane * 4, b_off = S::A_FLOATS + q * 4;
  asm volatile("" : "+v"(a_off), "+v"(b_off));
  const f32x4* A4 = reinterpret_cast<const f32x4*>(img + a_off);
  const f32x4* B4 = reinterpret_cast<const f32x4*>(img + b_off);
  int n = 0, bt = 0;
  f32x4 a4;
  f32x4 h1[NT], h2[NT], h3[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) h1[m] = B4[4 * (bt++)];
#pragma unroll
  for (int c1 = 0; c1 < H / 4; ++c1)
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      h1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], cnd[c1 >> 2][c1 & 3], h1[m], 0, 0, 0);
      ++n;
    }
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h1[m][r] = leaky2(h1[m][r]);
    h2[m] = B4[4 * (bt++)];
  }
#pragma unroll
  for (int c = 0; c < NQ; ++c)
#pragma unroll
    for (int m = 0; m < NT; ++m)
      if (S::tile_has_net(m, c >= QN ? 1 : 0)) {
        if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
        h2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h1[c >> 2][c & 3], h2[m], 0, 0, 0);
        ++n;
      }
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[m][r] = leaky2(h2[m][r]);
    h3[m] = B4[4 * (bt++)];
  }
#pragma unroll
  for (int c = 0; c < NQ; ++c)
#pragma unroll
    for (int m = 0; m < NT; ++m)
      if (S::tile_has_net(m, c >= QN ? 1 : 0)) {
        if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
        h3[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h2[c >> 2][c & 3], h3[m], 0, 0, 0);
        ++n;
      }
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) h3[m][r] = leaky2(h3[m][r]);
#pragma unroll
  for (int m = 0; m < G; ++m) {
    s4[m] = B4[4 * (bt++)];
    t4[m] = B4[4 * (bt++)];
#pragma unroll
    for (int c = 0; c < QN; ++c) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      s4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h3[c >> 2][c & 3], s4[m], 0, 0, 0);
      ++n;
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      t4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h3[(QN + c) >> 2][(QN + c) & 3], t4[m], 0, 0, 0);
      ++n;
    }
  }
}

// act <- exp(s) act + t, or its inverse (act - t) exp(-s): one multiply instead of a ~10-instruction
// IEEE divide, same limits (0, inf, NaN), <= 2 ulp from the quotient.  Returns this lane's sum of s.
// FAST: the two-instruction exp (the split kernels' hot path); the fp32-MFMA kernels and the range-guard path keep exp6
template <int H, bool INV, bool FAST = false>
__device__ __forceinline__ float ahf_transform(const f32x4 (&s4)[H / 16], const f32x4 (&t4)[H / 16],
                                               f32x4 (&act)[H / 16]) {
  float ld = 0.f;
#pragma unroll
  for (int m = 0; m < H / 16; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = FAST ? exp2x(INV ? -s4[m][r] : s4[m][r]) : exp6(INV ? -s4[m][r] : s4[m][r]);
      act[m][r] = INV ? (act[m][r] - t4[m][r]) * e : __builtin_fmaf(e, act[m][r], t4[m][r]);
      ld += s4[m][r];
    }
  return ld;
}

}  // namespace mnf
