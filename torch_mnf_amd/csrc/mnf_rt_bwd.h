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
// rows of the weight matrix behind delta tile m: unit i' of tile m <-> row rows(m, i') (< 0: padding)
struct ContigRows {
  int m0, n_out;
  __device__ __forceinline__ int operator()(int m, int u) const {
    const int o = 16 * (m0 + m) + u;
    return o < n_out ? o : -1;
  }
};
template <typename RowMap>
__device__ __forceinline__ void dw_phase_rows(const Exchange& exa, int a0, int MA, const Exchange& exb, int b0, int MB,
                                              const float* sa, const float* sb, int nw, float out_scale,
                                              float* __restrict__ gW, float* __restrict__ gb, const RowMap& rows, int n_in,
                                              int n0) {
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
      const int o = rows(m, 4 * q + r);
      if (o >= 0 && k < n_in) atomicAdd(gW + (size_t)o * n_in + k, acc[r] * out_scale);
      if (gb && n == 0 && i == 0 && o >= 0) atomicAdd(gb + o, bacc[r] * out_scale);
    }
    n += nw;
  }
}
__device__ __forceinline__ void dw_phase(const Exchange& exa, int a0, int MA, const Exchange& exb, int b0, int MB,
                                         const float* sa, const float* sb, int nw, float out_scale, float* __restrict__ gW,
                                         float* __restrict__ gb, int n_out, int n_in, int m0, int n0) {
  dw_phase_rows(exa, a0, MA, exb, b0, MB, sa, sb, nw, out_scale, gW, gb, ContigRows{m0, n_out}, n_in, n0);
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

// ---------------------------------------------------------------------------------------------------------------------
// what the layers' gradient kernels share
// ---------------------------------------------------------------------------------------------------------------------
template <int MT_MAX>
__device__ __forceinline__ uint32_t pack_signs(const Hidden<MT_MAX, 1>& h) {
  uint32_t bits = 0;
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t half = (h.hi[0][m][r >> 1] >> (16 * (r & 1))) & 0xffffu;
      bits |= (half != 0u && (half & 0x8000u) == 0u) ? 1u << (4 * m + r) : 0u;  // LeakyReLU keeps the sign
    }
  return bits;
}

// (main + corr 2^-11) * scale, times the LeakyReLU derivative of the hidden vector whose sign bits are `bits`
template <int MT_MAX>
__device__ __forceinline__ void chain_result(const Acc<MT_MAX, 1>& acc, float scale, uint32_t bits, f32x4 (&dv)[MT_MAX]) {
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m) {
    const f32x4 p = (acc.corr[0][m] * kSplitInvScale + acc.main[0][m]) * scale;
#pragma unroll
    for (int r = 0; r < 4; ++r) dv[m][r] = p[r] * ((bits >> (4 * m + r)) & 1u ? 1.f : kLeakySlope);
  }
}

// a split hidden vector back to fp32: (head + residual 2^-11) * the row's scale
__device__ __forceinline__ float half_of(uint32_t word, int hi) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)(hi ? word >> 16 : word & 0xffffu));
}
template <int MT_MAX>
__device__ __forceinline__ void unsplit(const Hidden<MT_MAX, 1>& h, f32x4 (&v)[MT_MAX]) {
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m) {
    const u32x2 wh = h.hi[0][m], wl = h.lo[0][m];
    v[m][0] = (half_of(wl[0], 0) * kSplitInvScale + half_of(wh[0], 0)) * h.up[0];
    v[m][1] = (half_of(wl[0], 1) * kSplitInvScale + half_of(wh[0], 1)) * h.up[0];
    v[m][2] = (half_of(wl[1], 0) * kSplitInvScale + half_of(wh[1], 0)) * h.up[0];
    v[m][3] = (half_of(wl[1], 1) * kSplitInvScale + half_of(wh[1], 1)) * h.up[0];
  }
}

// fp32 tiles -> split tiles with the row's power-of-two scale (the B operands of the next chain product)
template <int MT_MAX>
__device__ __forceinline__ void split_rows(const f32x4 (&v)[MT_MAX], Hidden<MT_MAX, 1>& h) {
  float fm = 0.f;
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) fm = __builtin_fmaxf(fm, finite_abs(v[m][r]));
  const int e = down_exponent(max_over_q(fm), 13);
  const float down = pow2f(-e);
  h.up[0] = pow2f(e);
  float unused = 0.f;
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m) split_tile(v[m] * down, h.hi[0][m], h.lo[0][m], unused);
}

// first exchange tile of hidden vector H_i (i >= 1) of net nd
__host__ __device__ inline int exH_tile_of(const NetDesc& nd, int i) {
  int t = 0;
  for (int k = 1; k < i; ++k) t += (nd.sizes[k] + 15) >> 4;
  return t;
}


constexpr int kBwdHeadWords = 32 + 8 * (kMaxBwdLayers + 2);  // scratch 16 | sA 8 | sC 8 | sH (layers + 1) x 8 | spare

struct BwdLds {
  float *sA, *sC, *sH;   // power-of-two scales of the wave-tiles' exchange copies: deltas | chunk tiles | hidden vector i
  Exchange exH, exC;  // hidden vectors of one net (a layer's deltas take the place of its dead hidden vector) | a chunk's cotangent / input tiles
  uint32_t* meta_bits;   // this wave's [hidden vector i][lane] sign bits
  int ct_tiles;
  int nw, wave, lane, q;
  f16x4 ident;
};
__device__ __forceinline__ BwdLds bwd_lds(uint32_t* lds0, float* after_weights, int ht_tiles, int dt_tiles, int ct_tiles) {
  BwdLds L;
  float* scratch = reinterpret_cast<float*>(lds0);
  L.sA = scratch + 16;
  L.sC = scratch + 24;
  L.sH = scratch + 32;
  L.nw = blockDim.x >> 6;
  L.wave = threadIdx.x >> 6;
  L.lane = threadIdx.x & 63;
  L.q = L.lane >> 4;
  L.exH = Exchange{reinterpret_cast<uint16_t*>(after_weights), 16 * L.nw};
  L.exC = Exchange{L.exH.base + (size_t)(ht_tiles + dt_tiles) * L.exH.tile_halves(), 16 * L.nw};
  L.meta_bits = reinterpret_cast<uint32_t*>(L.exC.base + (size_t)ct_tiles * L.exH.tile_halves()) + L.wave * ((kMaxBwdLayers + 1) * 64);
  L.ct_tiles = ct_tiles;
  L.ident = identity_operand(L.lane & 15, L.q);
  return L;
}
// bytes of LDS behind the weight stream for a workgroup of nw waves
inline size_t bwd_lds_bytes(int nw, int ht_tiles, int dt_tiles, int ct_tiles) {
  const size_t tile_bytes = (size_t)2 * 16 * (16 * nw + kExPad) * 2;
  return (size_t)(ht_tiles + dt_tiles + ct_tiles) * tile_bytes + (size_t)nw * (kMaxBwdLayers + 1) * 64 * 4;
}

// The forward recompute of one conditioner net that KEEPS every hidden vector: its true values turned into the exchange
// area (tiles from exH tile 0 on, with the wave-tile's scale in sH[i][wave]) and its sign bits (the LeakyReLU derivative).
template <int MT_MAX, typename Src, typename LoadX>
__device__ __forceinline__ void forward_keep(Src& src, const float* __restrict__ flat, const NetDesc& nd, int n_hid,
                                             int no_act_layer, float wup, const BwdLds& L, const LoadX& load_x,
                                             Hidden<MT_MAX, 1>& h) {
  auto use_x = [&](int, int, const f32x4&, const f32x4&) {};
  int tile0 = 0;
  auto hook = [&](int i, const Hidden<MT_MAX, 1>& hh) {
    const int MT = tiles16(nd.sizes[i]);
    f32x4 hv[MT_MAX];  // the vector's true values (head + residual, times the row's scale)
    unsplit<MT_MAX>(hh, hv);
    const float sc = exchange_store<MT_MAX>(hv, MT, L.exH, tile0, 16 * L.wave, L.lane, L.ident);
    if (L.lane == 0) L.sH[i * 8 + L.wave] = sc;
    tile0 += MT;
    L.meta_bits[i * 64 + L.lane] = pack_signs<MT_MAX>(hh);
  };
  net_to_hidden<MT_MAX, 1, false>(src, flat, nd, n_hid, no_act_layer, wup, L.lane, L.q, load_x, use_x, h, hook);
}

// From dv = the cotangent of H_n's pre-activation (n = n_hid; in units of the gradient scale) down to the net's input:
// per hidden layer dW_{i-1}, db_{i-1} through the exchange area and the chain step  delta_{i-1} = (W_{i-1}^T delta_i) *
// act'(H_{i-1});  then the first layer input tile by input tile: add_in(mi, W_0^T delta_1 of the tile's 16 columns, times
// inv_gs) and dW_0 += delta_1 (x) load_in(mi).  n_in0 = the first layer's input width.  The hidden vector whose index is
// no_act_hidden has no activation (derivative 1).
template <int MT_MAX, typename Src, typename LoadIn, typename AddIn>
__device__ __forceinline__ void backward_tail(Src& src, const float* __restrict__ flat, float* gflat, const NetDesc& nd,
                                              int n_hid, int no_act_hidden, f32x4 (&dv)[MT_MAX], const BwdLds& L, float wup,
                                              float inv_gs, int n_in0, const LoadIn& load_in, const AddIn& add_in) {
  const int lane = L.lane, wave = L.wave, q = L.q, nw = L.nw;
#pragma unroll 1
  for (int i = n_hid; i >= 2; --i) {
    const int MTi = tiles16(nd.sizes[i]), MTp = tiles16(nd.sizes[i - 1]);
    if (gflat) {
      // delta_i goes into the exchange tiles of H_i: that vector's last reader was the previous phase (dW_{i+1}, or the
      // caller's output-layer phase for i = n_hid), which every wave has left behind this barrier -- no tiles of their own
      // for the deltas, so that more rows fit a workgroup (the per-row-block cost is what these kernels are bound by)
      lds_barrier();
      const Exchange exD{L.exH.tile(exH_tile_of(nd, i)), L.exH.R};
      const float sc = exchange_store<MT_MAX>(dv, MTi, exD, 0, 16 * wave, lane, L.ident);
      if (lane == 0) L.sA[wave] = sc;
      lds_barrier();
      dw_phase(exD, 0, MTi, L.exH, exH_tile_of(nd, i - 1), MTp, L.sA, L.sH + (i - 1) * 8, nw, inv_gs, gflat + nd.w_off[i - 1],
               gflat + nd.b_off[i - 1], nd.sizes[i], nd.sizes[i - 1], 0, 0);
    }
    // delta_{i-1} = (W_{i-1}^T delta_i) * act'(H_{i-1}): K = the units of H_i, output tiles = those of H_{i-1}
    Hidden<MT_MAX, 1> hd;
    split_rows<MT_MAX>(dv, hd);
    const int KS = steps32(16 * MTi);
    int KC = src.cb / MTp;
    if (KC < 1) KC = 1;
    Acc<MT_MAX, 1> acc;
    acc.zero();
    const uint32_t* bufT = nullptr;
    int next_start = 0, chunk_start = 0;
#pragma unroll
    for (int ks = 0; ks < MT_MAX / 2; ++ks)
      if (ks < KS) {
        if (ks == next_start) {
          const int kc = KS - ks < KC ? KS - ks : KC;
          uint32_t* b = src.cur_blocks();
          stage_blocks(b, kc * MTp, DenseTKMajor{flat + nd.w_off[i - 1], nd.sizes[i - 1], nd.sizes[i], MTp, ks}, src.wdown);
          src.commit();
          bufT = b;
          chunk_start = ks;
          next_start = ks + kc;
        }
        f16x8 bh[1], bl[1];
        hidden_operand<MT_MAX, 1>(hd, ks, bh, bl);
        mac_kstep<MT_MAX, 1>(bufT, (ks - chunk_start) * MTp, MTp, lane, bh, bl, acc.main, acc.corr);
      }
    chain_result<MT_MAX>(acc, wup * hd.up[0], i - 1 == no_act_hidden ? 0xffffffffu : L.meta_bits[(i - 1) * 64 + lane], dv);
  }
  // ---- first layer: dv = delta_1
  const int MT1 = tiles16(nd.sizes[1]), KS1 = steps32(16 * MT1), MI = tiles16(n_in0);
  const Exchange exD1{L.exH.tile(exH_tile_of(nd, 1)), L.exH.R};  // (delta_1 in H_1's tiles, as above)
  if (gflat) {
    lds_barrier();
    const float sc = exchange_store<MT_MAX>(dv, MT1, exD1, 0, 16 * wave, lane, L.ident);
    if (lane == 0) L.sA[wave] = sc;
  }
  Hidden<MT_MAX, 1> hd;
  split_rows<MT_MAX>(dv, hd);
  int CI = src.cb / KS1;
  if (CI > L.ct_tiles) CI = L.ct_tiles;
  if (CI > MT_MAX) CI = MT_MAX;
  if (CI < 1) CI = 1;
  for (int mi0 = 0; mi0 < MI; mi0 += CI) {
    const int ci = MI - mi0 < CI ? MI - mi0 : CI;
    uint32_t* buf = src.cur_blocks();
    float* bbuf = src.cur_bias();
    stage_blocks(buf, ci * KS1, DenseTMMajor{flat + nd.w_off[0], n_in0, nd.sizes[1], KS1, mi0}, src.wdown);
    stage_bias(bbuf, 1, NoBias{});
    src.commit();
    f32x4 xv[MT_MAX];
#pragma unroll
    for (int ml = 0; ml < MT_MAX; ++ml) {
      xv[ml] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ml < ci) {
        f32x4 gx[1];
        out_tile<MT_MAX, 1>(buf, ml * KS1, KS1, bbuf, lane, q, hd, wup, gx);
        add_in(mi0 + ml, gx[0] * inv_gs);
        xv[ml] = load_in(mi0 + ml);
      }
    }
    if (gflat) {
      const float sc = exchange_store<MT_MAX>(xv, ci, L.exC, 0, 16 * wave, lane, L.ident);
      if (lane == 0) L.sC[wave] = sc;
      lds_barrier();
      dw_phase(exD1, 0, MT1, L.exC, 0, ci, L.sA, L.sC, nw, inv_gs, gflat + nd.w_off[0], mi0 == 0 ? gflat + nd.b_off[0] : nullptr,
               nd.sizes[1], n_in0, 0, mi0);
    }
  }
}

}  // namespace rt
}  // namespace mnf
