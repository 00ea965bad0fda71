// Gradient-direction building blocks of the run-time-shaped kernels (mnf_rt.h): what a coupling layer's backward pass
// needs beyond the forward machinery.
//
// A workgroup of NW waves owns a block of 16 NW rows; a wave owns one 16-row tile of it.  The row-parallel work -- the
// forward recompute, the cotangents at the conditioner's outputs, the delta chain  delta_{l-1} = (W_l^T delta_l) *
// act'(H_{l-1})  -- runs per wave exactly like the forward kernels, with TRANSPOSED weight blocks staged from `flat`.
// The weight gradients  dW_l = sum_rows delta_l (x) H_{l-1}  are sums over rows, which want the rows on an MFMA's K
// axis where the chain has them along the lanes: every wave turns its tiles (one f16 MFMA against the identity per tile
// and part: exact) and leaves them in an LDS exchange area as [tile][unit][row] f16 planes (head and scaled residual);
// after a barrier the (delta tile, H tile) products of the layer are dealt out over the waves -- K = 16 rows per
// product, three f16 MFMAs for the split form, one wave-tile after the other with its power-of-two scale -- and each
// finished 16 x 16 block of dW is added to grad_flat with float atomics: ONE flush per row block instead of per tile,
// no accumulator registers, any layer width.  (Atomic sums: not bit-reproducible run to run; MNF_DETERMINISTIC=1 keeps
// these shapes on the VALU kernels.)
#pragma once
#include "mnf_rt.h"

namespace mnf {
namespace rt {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int kExPad = 8;       // halves of padding per unit row of an exchange plane (bank spread)
constexpr int kMaxBwdLayers = 4;  // hidden vectors per net the gradient kernels keep track of

__device__ __forceinline__ f32x4 mfma16(const f16x4& a, const f16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}

// The exchange area: tiles of [2 planes (hi, lo)][16 units][R + kExPad rows] halves
struct Exchange {
  uint16_t* base;
  int R;  // rows of the block (16 NW)
  __device__ __forceinline__ int unit_stride() const { return R + kExPad; }
  __device__ __forceinline__ int tile_halves() const { return 2 * 16 * unit_stride(); }
  __device__ __forceinline__ uint16_t* tile(int t) const { return base + (size_t)t * tile_halves(); }
};

// identity B operand of the transposing MFMA for this lane: B[k = 4 q + e][n = j] = (k == j)
__device__ __forceinline__ f16x4 identity_operand(int j, int q) {
  f16x4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = (4 * q + e == j) ? (_Float16)1.f : (_Float16)0.f;
  return b;
}

// one split tile (accumulator layout: lane (row j, q) holds units 4 q + r) -> both planes of exchange tile `dst`,
// rows row0 .. row0 + 15 (row0 = 16 * wave): lane (unit n, q) ends up with rows 4 q + r of unit n and stores 8 bytes
__device__ __forceinline__ void transpose_store(const u32x2& hi, const u32x2& lo, uint16_t* dst, int unit_stride, int row0,
                                                int lane, const f16x4& ident) {
  const int n = lane & 15, q = lane >> 4;
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 th = mfma16(__builtin_bit_cast(f16x4, hi), ident, zero);
  const f32x4 tl = mfma16(__builtin_bit_cast(f16x4, lo), ident, zero);
  typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
  u32x2 oh, ol;
  oh[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{th[0], th[1]}, f16x2v));
  oh[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{th[2], th[3]}, f16x2v));
  ol[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{tl[0], tl[1]}, f16x2v));
  ol[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{tl[2], tl[3]}, f16x2v));
  uint16_t* p = dst + n * unit_stride + row0 + 4 * q;
  *reinterpret_cast<u32x2*>(p) = oh;
  *reinterpret_cast<u32x2*>(p + 16 * unit_stride) = ol;
}

// fp32 tiles (accumulator layout) -> exchange tiles with ONE power-of-two scale for the wave's 16 rows (the rows of a
// tile share an MFMA's K axis): returns 2^e, the factor the tile's products are multiplied back by.  MT tiles used.
template <int MT_MAX>
__device__ __forceinline__ float exchange_store(const f32x4 (&v)[MT_MAX], int MT, const Exchange& ex, int tile0, int row0,
                                                int lane, const f16x4& ident) {
  float mx = 0.f;
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = __builtin_fmaxf(mx, finite_abs(v[m][r]));
  mx = wave_max(mx);
  const int e = down_exponent(mx, 13);
  const float down = pow2f(-e);
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m)
    if (m < MT) {
      u32x2 hi, lo;
      float unused = 0.f;
      split_tile(v[m] * down, hi, lo, unused);
      transpose_store(hi, lo, ex.tile(tile0 + m), ex.unit_stride(), row0, lane, ident);
    }
  return pow2f(e);
}

// Weight-gradient products of one layer phase: dW[16 (m0 + m) + i][16 (n0 + n) + j] += out_scale * sum over the block's
// wave-tiles w of sa[w] sb[w] (A tile m)^T-product (B tile n), A = the delta tiles at exchange tile a0 .. a0 + MA - 1, B
// = the input tiles at b0 .. b0 + MB - 1; bias: db[16 (m0 + m) + i] += the row sums of A (gb != nullptr).  The (m, n)
// pairs are dealt out over the workgroup's waves.  gW: the layer's weight gradient (n_out x n_in, row-major).
__device__ __forceinline__ void dw_phase(const Exchange& exa, int a0, int MA, const Exchange& exb, int b0, int MB,
                                         const float* sa, const float* sb,
                                         int nw, float out_scale, float* __restrict__ gW, float* __restrict__ gb, int n_out,
                                         int n_in, int m0, int n0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int us = exa.unit_stride();
  f16x4 ones;
#pragma unroll
  for (int e = 0; e < 4; ++e) ones[e] = (_Float16)1.f;
  bool same = true;
  for (int w = 0; w < nw; ++w) same = same && sa[w] == 1.f && sb[w] == 1.f;
  int m = 0, n = wave;  // pair index t = m * MB + n, dealt out round robin
  for (int t = wave; t < MA * MB; t += nw) {
    while (n >= MB) {
      n -= MB;
      ++m;
    }
    const uint16_t* pa = exa.tile(a0 + m) + i * us + 4 * q;
    const uint16_t* pb = exb.tile(b0 + n) + i * us + 4 * q;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, bacc = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    if (same) {  // (uniform; the usual case) every wave-tile at scale 1: the products add up in the MFMA accumulators
      f32x4 main = zero, corr = zero, bm = zero, bc = zero;
      for (int w = 0; w < nw; ++w) {
        const f16x4 ah = *reinterpret_cast<const f16x4*>(pa + 16 * w), al = *reinterpret_cast<const f16x4*>(pa + 16 * us + 16 * w);
        const f16x4 bh = *reinterpret_cast<const f16x4*>(pb + 16 * w), bl = *reinterpret_cast<const f16x4*>(pb + 16 * us + 16 * w);
        main = mfma16(ah, bh, main);
        corr = mfma16(ah, bl, corr);
        corr = mfma16(al, bh, corr);
        if (gb && n == 0) {  // (uniform)
          bm = mfma16(ah, ones, bm);
          bc = mfma16(al, ones, bc);
        }
      }
      acc = corr * kSplitInvScale + main;
      bacc = bc * kSplitInvScale + bm;
    } else {
      for (int w = 0; w < nw; ++w) {
        const f16x4 ah = *reinterpret_cast<const f16x4*>(pa + 16 * w), al = *reinterpret_cast<const f16x4*>(pa + 16 * us + 16 * w);
        const f16x4 bh = *reinterpret_cast<const f16x4*>(pb + 16 * w), bl = *reinterpret_cast<const f16x4*>(pb + 16 * us + 16 * w);
        const f32x4 main = mfma16(ah, bh, zero);
        f32x4 corr = mfma16(ah, bl, zero);
        corr = mfma16(al, bh, corr);
        const float s = sa[w] * sb[w];
        acc += (corr * kSplitInvScale + main) * s;
        if (gb && n == 0) {  // (uniform)
          const f32x4 bm = mfma16(ah, ones, zero), bc = mfma16(al, ones, zero);
          bacc += (bc * kSplitInvScale + bm) * sa[w];
        }
      }
    }
    // lane (j, q) register r = dW[delta unit 4 q + r of tile m][input unit j of tile n]
    const int k = 16 * (n0 + n) + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * (m0 + m) + 4 * q + r;
      if (o < n_out && k < n_in) atomicAdd(gW + (size_t)o * n_in + k, acc[r] * out_scale);
      if (gb && n == 0 && i == 0 && o < n_out) atomicAdd(gb + o, bacc[r] * out_scale);
    }
    n += nw;
  }
}

// W^T of a dense Linear W (n_out x n_in) as A blocks: block row i = INPUT unit 16 mi + i, K index = OUTPUT unit.
// Walked [K-step over outputs][input tile] from K-step ks0: digits (mi, ks - ks0)
struct DenseTKMajor {
  const float* W;
  int n_in, n_out, R0, ks0;  // R0 = input tiles
  static constexpr int R1 = 1 << 30;
  __device__ __forceinline__ void load(int mi, int ksl, int, int i, int q, f32x4& va, f32x4& vb) const {
    const int u = 16 * mi + i, o0 = 32 * (ks0 + ksl) + 4 * q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int oa = o0 + e, ob = oa + 16;
      const bool oka = u < n_in && oa < n_out, okb = u < n_in && ob < n_out;
      const float xa = W[oka ? (int64_t)oa * n_in + u : 0], xb = W[okb ? (int64_t)ob * n_in + u : 0];
      va[e] = oka ? xa : 0.f;
      vb[e] = okb ? xb : 0.f;
    }
  }
};
// the same walked [input tile][K-step over outputs] from input tile m0: digits (ks, mi - m0)
struct DenseTMMajor {
  const float* W;
  int n_in, n_out, R0, m0;  // R0 = KS over the outputs
  static constexpr int R1 = 1 << 30;
  __device__ __forceinline__ void load(int ks, int ml, int, int i, int q, f32x4& va, f32x4& vb) const {
    DenseTKMajor{W, n_in, n_out, 1, ks}.load(m0 + ml, 0, 0, i, q, va, vb);
  }
};
struct NoBias {
  __device__ __forceinline__ float operator()(int, int) const { return 0.f; }
};

// the LeakyReLU derivative of a hidden vector from its (split) values: sign of the head part, per unit
template <int MT_MAX>
__device__ __forceinline__ void leaky_gate(const Hidden<MT_MAX, 1>& h, f32x4 (&g)[MT_MAX]) {
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m) {
    // packed f16 pairs: element r of the tile sits in half (r & 1) of word (r >> 1); positive <=> sign bit clear and non-zero
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t half = (h.hi[0][m][r >> 1] >> (16 * (r & 1))) & 0xffffu;
      const bool pos = half != 0u && (half & 0x8000u) == 0u;
      g[m][r] = pos ? 1.f : kLeakySlope;
    }
  }
}

}  // namespace rt
}  // namespace mnf
