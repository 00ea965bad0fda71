// fp32 products on the f16 matrix pipe ("split" arithmetic), shared by the kernels that use it.
//
// gfx950 runs v_mfma_f32_16x16x4_f32 at the fp32 VALU rate (32 cycles per 2,048 MACs... i.e. 256
// flop/cycle/CU) and it does not overlap with VALU work, which is what bounds the conditioner nets
// of the coupling layers.  v_mfma_f32_16x16x32_f16 does 8x the MACs in half the cycles and leaves
// the vector issue port free for half of them.  A fp32 value v is carried as two f16 numbers
//
//     hi = f16(v)                       (11 significant bits)
//     lo = f16((v - f32(hi)) * 2^11)    (the next 11 bits, pre-scaled so it stays a normal f16)
//
// and a fp32 product sum  sum_k w_k v_k  is evaluated as three f16 MFMAs with fp32 accumulation
//
//     main = sum w_hi v_hi           corr = sum (w_hi v_lo + w_lo v_hi)         result = main + corr * 2^-11
//
// Every f16 x f16 product is exact in fp32; the dropped w_lo v_lo term is 2^-22 relative, so the
// result carries ~22 significant bits per product (fp32: 24) -- measured against float64 the split
// MLP is 1.7e-7 normwise where the fp32 MLP is 0.7e-7 (tests/test_hip_parity.py, tools/split_error.py);
// the parity bar is 1e-5.  16 x fewer matrix-pipe cycles per MAC x 3 products = 5.3 x fewer cycles.
//
// Range: f16 tops out at 65,504 and lo is scaled by 2^11, so the scheme needs |v| < 2^15 for every
// operand.  The kernels track max|v| per 16-row tile (one v_max3 per two values) and recompute a tile
// that reaches kSplitLimit -- or, for the whole launch, whose weights exceed kSplitWeightLimit, flagged by
// the pack kernel -- on the fp32 MFMA path, so results do not depend on the input range.  An operand
// below 2^-14 is carried with an ABSOLUTE error of at most 2^-36 (hi and lo are f16 subnormals there,
// which the MFMA honours); the two limits keep the other factor of such a product small enough for that
// to stay below fp32's own rounding of the sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mnf_ahf_shape.h"

namespace mnf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr float kSplitScale = 2048.f;          // 2^11
constexpr float kSplitInvScale = 1.f / 2048.f;
constexpr float kSplitLimit = 8192.f;           // operands at or above 2^13 go to the fp32 path (f16 ends at 65,504;
                                               // keeping |v| small also bounds |v| 2^-36, the absolute error a
                                               // sub-2^-14 weight contributes per product)
constexpr float kSplitWeightLimit = 256.f;     // likewise for weights: a larger one sends the launch to fp32
constexpr int kSplitTailWords = 4;             // [max|w| bits, 0, 0, 0] after the plain words
constexpr int32_t kSplitLoBit = 1 << 30;       // index-table entry: take the lo part of the source value

typedef float f32x2 __attribute__((ext_vector_type(2)));

// v - f32(low / high half of hp), exact, in one instruction (the compiler emits cvt + sub for the C form)
__device__ __forceinline__ float residual_lo(uint32_t hp, float v) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}
__device__ __forceinline__ float residual_hi(uint32_t hp, float v) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> packed f16 hi pair and packed scaled-residual pair; mx tracks max|v|.
// Six vector instructions: v_cvt_pk_f16_f32 (round to nearest even, gfx950), 2 x fma_mix, pk_mul,
// v_cvt_pk_f16_f32, max3.  With round-to-nearest |v - hi| <= 2^-12 |v| and the residual is itself rounded
// to 11 bits, so hi + lo 2^-11 carries v to 2^-24 relative -- fp32's own precision -- as long as hi is a
// normal f16; below 2^-14 the ABSOLUTE error is bounded by 2^-36 instead (f16 subnormals are honoured by
// the MFMA, measured), which is why the guard also has a lower-than-overflow weight limit.
__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t& hi, uint32_t& lo, float& mx) {
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v0, v1}, f16x2));
  f32x2 r = f32x2{residual_lo(hi, v0), residual_hi(hi, v1)};
  r = r * kSplitScale;
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
  mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(v0)), __builtin_fabsf(v1));
}

// the four accumulator rows of one 16x16 tile (rows 4q..4q+3) -> two words of hi, two of lo
__device__ __forceinline__ void split_tile(const f32x4& v, u32x2& hi, u32x2& lo, float& mx) {
  uint32_t h0, l0, h1, l1;
  split_pair(v[0], v[1], h0, l0, mx);
  split_pair(v[2], v[3], h1, l1, mx);
  hi = u32x2{h0, h1};
  lo = u32x2{l0, l1};
}

// B operand of one K = 32 step: slots 8q..8q+3 <- tile a rows 4q..4q+3, slots 8q+4..8q+7 <- tile b
__device__ __forceinline__ f16x8 pair_operand(const u32x2& a, const u32x2& b) {
  return __builtin_bit_cast(f16x8, u32x4{a[0], a[1], b[0], b[1]});
}

__device__ __forceinline__ f32x4 mfma_h(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// main += Ah Bh ; corr += Ah Bl + Al Bh
__device__ __forceinline__ void split_mac(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl,
                                          f32x4& main, f32x4& corr) {
  main = mfma_h(ah, bh, main);
  corr = mfma_h(ah, bl, corr);
  corr = mfma_h(al, bh, corr);
}

__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// initial value of a tile's running max|operand|: the weights' verdict (image tail word = max |weight| bits)
__device__ __forceinline__ float split_guard_seed(float wmax) { return wmax <= kSplitWeightLimit ? 0.f : __builtin_inff(); }

// ------------------------------------------------------------------------------------------------
// Tiling of a coupling layer's two conditioner nets (s and t, HID hidden units each, evaluated as
// one concatenated net with block-diagonal hidden layers) for 16x16x32 MFMAs.
//   hidden vector u in [0, 2 HID): u < HID is s-net unit u, else t-net unit u - HID; tile m holds
//   u = 16 m + i in accumulator row i.  A hidden K-step pairs two tiles (the accumulator registers
//   of tiles a, b ARE the B operand after the split -- no data movement between layers).
// ------------------------------------------------------------------------------------------------
template <int H, int HID>
struct SplitShape {
  static_assert(H % 16 == 0 && HID % 4 == 0, "unsupported conditioner shape");
  static constexpr int G = H / 16;                 // 16-dim groups per half row = output tiles per net
  static constexpr int GC = G < 4 ? G : 4;         // output tiles per chunk (operand order: chunk, net, K-step, tile)
  static_assert(G % GC == 0, "output tiles come in whole chunks");
  static constexpr int KS1 = (G + 1) / 2;          // K-steps of the first layer
  static constexpr int NT = (2 * HID + 15) / 16;   // tiles of the concatenated hidden vector
  static constexpr int NKS = (NT + 1) / 2;         // hidden K-steps
  static constexpr int ks_a(int ks) { return (NT > 1 && (NT & 1) && ks == NKS - 1) ? NT - 2 : 2 * ks; }
  static constexpr int ks_b(int ks) { return ks_a(ks) + 1 < NT ? ks_a(ks) + 1 : -1; }
  static constexpr bool ks_has(int ks, int tile) { return ks_a(ks) == tile || ks_b(ks) == tile; }
  // nets: bit 0 = s, bit 1 = t
  static constexpr bool tile_needed(int nets, int tile) {
    for (int i = 0; i < 16; ++i) {
      const int u = 16 * tile + i;
      if (u < 2 * HID && ((nets >> (u / HID)) & 1)) return true;
    }
    return false;
  }
  static constexpr int tile_nets(int tile) {  // nets with a unit in this tile
    int nets = 0;
    for (int i = 0; i < 16; ++i)
      if (16 * tile + i < 2 * HID) nets |= 1 << ((16 * tile + i) / HID);
    return nets;
  }
  static constexpr int single_cover(int nets) {
    for (int ks = 0; ks < NKS; ++ks) {
      bool all = true;
      for (int t = 0; t < NT; ++t)
        if (tile_needed(nets, t) && !ks_has(ks, t)) all = false;
      if (all) return ks;
    }
    return -1;
  }
  // the K-step through which input tile `tile` feeds an output that depends on `nets`
  static constexpr int assigned_ks(int nets, int tile) {
    const int sc = single_cover(nets);
    if (sc >= 0) return sc;
    for (int ks = 0; ks < NKS; ++ks)
      if (ks_has(ks, tile)) return ks;
    return -1;
  }
  static constexpr bool uses(int nets, int ks) {
    for (int t = 0; t < NT; ++t)
      if (tile_needed(nets, t) && assigned_ks(nets, t) == ks) return true;
    return false;
  }
  static constexpr int count_ops() {
    int n = NT * KS1;
    for (int m = 0; m < NT; ++m)
      for (int ks = 0; ks < NKS; ++ks)
        if (uses(tile_nets(m), ks)) n += 2;
    for (int net = 0; net < 2; ++net)
      for (int ks = 0; ks < NKS; ++ks)
        if (uses(1 << net, ks)) n += G;
    return n;
  }
  static constexpr int hidden_ops() {  // operand pairs of one hidden layer
    int n = 0;
    for (int m = 0; m < NT; ++m)
      for (int ks = 0; ks < NKS; ++ks)
        if (uses(tile_nets(m), ks)) ++n;
    return n;
  }
  static constexpr int N_OPS = count_ops();              // (hi, lo) A-operand pairs = MFMA triples
  static constexpr int SPLIT_WORDS = N_OPS * 2 * 256;    // [op][hi|lo][lane][4 words]
  static constexpr int N_BIAS_TILES = 3 * NT + 2 * G;
  static constexpr int PLAIN_WORDS = N_BIAS_TILES * 16;
  static constexpr int IMAGE_WORDS = SPLIT_WORDS + PLAIN_WORDS + kSplitTailWords;
};

// The conditioner on the split path: cnd (G groups of the conditioning half, accumulator layout
// lane (j, q) reg r <-> dim 16 g + 4 q + r) -> raw s and t of the same layout.  `img` points at the
// LDS copy of the split image.  mx accumulates max|operand| (see kSplitLimit).
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
struct NoEmit {
  static constexpr bool enabled = false;
};
// per-chunk consumer of the output layer: abort_fn(mx) -> true = the caller takes the fp32 path instead;
// fn(g0, s_chunk, t_chunk) gets the raw s and t of output tiles g0 .. g0 + GC - 1
template <typename AbortFn, typename Fn>
struct ChunkEmit {
  static constexpr bool enabled = true;
  AbortFn abort_fn;
  Fn fn;
  __device__ __forceinline__ bool abort(float mx) { return abort_fn(mx); }
  template <typename SC, typename TC>
  __device__ __forceinline__ void operator()(int g0, const SC& s, const TC& t) { fn(g0, s, t); }
};
template <typename AbortFn, typename Fn>
__device__ __forceinline__ ChunkEmit<AbortFn, Fn> make_chunk_emit(AbortFn a, Fn f) {
  return ChunkEmit<AbortFn, Fn>{a, f};
}

// NTL row tiles (16 rows each) share every A-operand read: the weights come out of LDS once per NTL
// tiles (LDS bandwidth, not the matrix pipe, is the co-bottleneck of the conditioner: 30 KB of operands
// per tile per layer).  at_stage(0) runs once cnd has been turned into MFMA operands (cnd is dead from
// there on: the single-layer kernel issues its prefetch of the next tile there); at_stage(1..3) after each
// of the three activation blocks (the stack kernel spreads its intermediate-tensor stores over them).
// ABL != 0 only in tools/split_microbench.hip (1 = MFMAs skipped, 3 = operand splitting skipped).
template <int H, int HID, int NTL = 1, typename Hook = NoHook, int ABL = 0, typename Emit = NoEmit>
__device__ __forceinline__ void split_conditioner(const uint32_t* img, int lane, int q,
                                                  const f32x4 (&cnd)[NTL][H / 16], f32x4 (&s4)[NTL][H / 16],
                                                  f32x4 (&t4)[NTL][H / 16], float& mx, Hook at_stage = Hook(),
                                                  Emit emit = Emit()) {
  using S = SplitShape<H, HID>;
  constexpr int G = S::G, NT = S::NT, NKS = S::NKS, KS1 = S::KS1;
  // opaque offsets: keep the (loop-invariant) operand reads inside the tile loop instead of in VGPRs
  int a_off = lane * 4, b_off = S::SPLIT_WORDS + q * 4;
  asm volatile("" : "+v"(a_off), "+v"(b_off));
  const f16x8* A8 = reinterpret_cast<const f16x8*>(img + a_off);  // + 64 * (2 op + part)
  const f32x4* B4 = reinterpret_cast<const f32x4*>(img + b_off);  // + 4 * tile
  int op = 0, bt = 0;
  const u32x2 zero2 = u32x2{0u, 0u};
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  auto split_tile = [&](const f32x4& v, u32x2& hi, u32x2& lo, float& m) {
    if (ABL == 3) {
      hi = u32x2{__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1])};
      lo = u32x2{__builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])};
    } else {
      mnf::split_tile(v, hi, lo, m);
    }
  };
  // the three products of every (output tile, row tile) pair, issued product by product: an accumulator's
  // two dependent MFMAs (corr) are then NM * NTL - 1 independent MFMAs apart instead of back to back
  auto split_mac_phased_impl = [&](auto nm_tag, const auto& ah, const auto& al, const f16x8 (&bh)[NTL],
                                   const f16x8 (&bl)[NTL], auto& mn, auto& cr, auto used, int m0 = 0) {
    constexpr int NM = decltype(nm_tag)::value;
#pragma unroll
    for (int phase = 0; phase < 3; ++phase)
#pragma unroll
      for (int m = 0; m < NM; ++m)
        if (used(m)) {
#pragma unroll
          for (int t = 0; t < NTL; ++t) {
            if (ABL == 1) {
              if (phase == 0) mn[t][m0 + m] += __builtin_bit_cast(f32x4, ah[m]) * __builtin_bit_cast(f32x4, bh[t]);
              if (phase == 1) cr[t][m0 + m] += __builtin_bit_cast(f32x4, al[m]) * __builtin_bit_cast(f32x4, bl[t]);
            } else if (phase == 0) {
              mn[t][m0 + m] = mfma_h(ah[m], bh[t], mn[t][m0 + m]);
            } else if (phase == 1) {
              cr[t][m0 + m] = mfma_h(ah[m], bl[t], cr[t][m0 + m]);
            } else {
              cr[t][m0 + m] = mfma_h(al[m], bh[t], cr[t][m0 + m]);
            }
          }
        }
  };
  // With two or more row tiles per wave there are registers to spare (the kernel runs at 2 waves/SIMD
  // anyway): a layer's operands are then requested from LDS BEFORE the vector work on the previous layer's
  // results and consumed after it, so the LDS latency is covered (left alone, hipcc sinks every ds_read to
  // just in front of its MFMA to save registers and exposes ~100 cycles per operand).  sched_barrier(0)
  // pins the three blocks [reads][vector work][MFMAs] in that order.
  constexpr bool PRE = NTL >= 2;
  auto fence = [] {
    if (PRE) __builtin_amdgcn_sched_barrier(0);
  };

  // ---- layer 1
  f16x8 a1h[KS1][NT], a1l[KS1][NT];
  f32x4 bias1[NT];
  auto read_layer1 = [&]() {
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        a1h[ks][m] = A8[64 * (2 * op)];
        a1l[ks][m] = A8[64 * (2 * op + 1)];
        ++op;
      }
#pragma unroll
    for (int m = 0; m < NT; ++m) bias1[m] = B4[4 * (bt++)];
  };
  if (PRE) read_layer1();
  fence();
  u32x2 xh[NTL][G], xl[NTL][G];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g) split_tile(cnd[t][g], xh[t][g], xl[t][g], mx);
  at_stage(0);
  fence();
  if (!PRE) read_layer1();  // one tile per wave: reads stay next to their MFMAs (registers are what is scarce)
  f32x4 main[NTL][NT], corr[NTL][NT];
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      main[t][m] = bias1[m];
      corr[t][m] = zero4;
    }
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) {
    f16x8 bh[NTL], bl[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      bh[t] = pair_operand(xh[t][2 * ks], 2 * ks + 1 < G ? xh[t][2 * ks + 1 < G ? 2 * ks + 1 : 0] : zero2);
      bl[t] = pair_operand(xl[t][2 * ks], 2 * ks + 1 < G ? xl[t][2 * ks + 1 < G ? 2 * ks + 1 : 0] : zero2);
    }
    split_mac_phased_impl(std::integral_constant<int, NT>{}, a1h[ks], a1l[ks], bh, bl, main, corr,
                          [](int) { return true; });
  }
  u32x2 hh[NTL][NT], hl[NTL][NT];
  auto activate = [&]() {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        // pre-activation = main + corr 2^-11, LeakyReLU = max(p, 0.2 p): vector forms so that the
        // multiply-adds go out as packed fp32 instructions (two values each)
        const f32x4 p = corr[t][m] * kSplitInvScale + main[t][m];
        const f32x4 v = __builtin_elementwise_max(p, p * kLeakySlope);
        split_tile(v, hh[t][m], hl[t][m], mx);
      }
  };
  auto hidden_operand = [&](const u32x2 (&h)[NTL][NT], int ks, f16x8 (&b)[NTL]) {
    const int ta = S::ks_a(ks), tb = S::ks_b(ks);
#pragma unroll
    for (int t = 0; t < NTL; ++t) b[t] = pair_operand(h[t][ta], tb >= 0 ? h[t][tb >= 0 ? tb : 0] : zero2);
  };

  // ---- hidden layers 2 and 3 (block diagonal)
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
    f16x8 ah[NKS][NT], al[NKS][NT];
    f32x4 bias[NT];
    auto read_hidden = [&]() {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int m = 0; m < NT; ++m)
          if (S::uses(S::tile_nets(m), ks)) {
            ah[ks][m] = A8[64 * (2 * op)];
            al[ks][m] = A8[64 * (2 * op + 1)];
            ++op;
          }
#pragma unroll
      for (int m = 0; m < NT; ++m) bias[m] = B4[4 * (bt++)];
    };
    if (PRE) read_hidden();
    fence();
    activate();  // the previous layer's accumulators -> this layer's B operands
    at_stage(1 + layer);
    fence();
    if (!PRE) read_hidden();
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        main[t][m] = bias[m];
        corr[t][m] = zero4;
      }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      f16x8 bh[NTL], bl[NTL];
      hidden_operand(hh, ks, bh);
      hidden_operand(hl, ks, bl);
      split_mac_phased_impl(std::integral_constant<int, NT>{}, ah[ks], al[ks], bh, bl, main, corr,
                            [ks](int m) { return S::uses(S::tile_nets(m), ks); });
    }
  }

  // ---- output layer: s from the s-net units, t from the t-net units, in chunks of GC output tiles (operand
  // order in the image: chunk, net, K-step, tile).  With an Emit consumer (d = 256: the sixteen s/t tiles of a
  // row would take 128 VGPRs) a chunk's s and t are handed over as soon as they are complete and s4 / t4
  // are not written; emit.abort(mx) is asked first, with the final max|operand|, whether the caller wants
  // the fp32 path instead (the range guard has to be settled before anything is consumed).
  constexpr int GC = S::GC;
  fence();
  activate();
  at_stage(3);
  fence();
  if constexpr (Emit::enabled) {
    if (emit.abort(mx)) return;
  }
#pragma unroll
  for (int g0 = 0; g0 < G; g0 += GC) {
    f32x4 chunk[2][NTL][GC];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      f32x4 oc[NTL][GC];
#pragma unroll
      for (int g = 0; g < GC; ++g) {
        const f32x4 bias = B4[4 * (3 * NT + net * G + g0 + g)];
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          chunk[net][t][g] = bias;
          oc[t][g] = zero4;
        }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        if (S::uses(1 << net, ks)) {
          f16x8 bh[NTL], bl[NTL];
          hidden_operand(hh, ks, bh);
          hidden_operand(hl, ks, bl);
          f16x8 ah[GC], al[GC];
#pragma unroll
          for (int g = 0; g < GC; ++g) {
            ah[g] = A8[64 * (2 * op)];
            al[g] = A8[64 * (2 * op + 1)];
            ++op;
          }
          split_mac_phased_impl(std::integral_constant<int, GC>{}, ah, al, bh, bl, chunk[net], oc,
                                [](int) { return true; });
        }
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int g = 0; g < GC; ++g) chunk[net][t][g] = oc[t][g] * kSplitInvScale + chunk[net][t][g];
    }
    if constexpr (Emit::enabled) {
      emit(g0, chunk[0], chunk[1]);
    } else {
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          s4[t][g0 + g] = chunk[0][t][g];
          t4[t][g0 + g] = chunk[1][t][g];
        }
    }
  }
}

}  // namespace mnf
