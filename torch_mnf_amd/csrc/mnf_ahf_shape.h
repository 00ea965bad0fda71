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

}  // namespace mnf
