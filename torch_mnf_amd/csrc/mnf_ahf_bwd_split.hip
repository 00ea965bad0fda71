// AffineHalfFlow gradients on the f16 matrix pipe in split form (the training step's hot kernel).
//
// mnf_ahf_bwd_mfma.hip evaluates the same gradients with v_mfma_f32_16x16x4_f32, which runs at the fp32 vector rate:
// 336 MFMAs x 32 cycles per 16 rows, i.e. 0.33 ms per layer per 2^20 rows with a perfectly busy pipe (measured: 0.88).
// Here every product is a split one (mnf_split.h: v = hi + lo 2^-11 as two f16 numbers, three exact f16 products,
// fp32 accumulation), 225 MFMAs x 16 cycles per 16 rows.  One wave owns 16 rows at a time and
//
//   1. recomputes the two conditioner nets exactly like the forward kernels (same operand image order), keeping
//      the split activations x, h1, h2, h3 (LeakyReLU' is read off their sign);
//   2. forms the output deltas from grad_y / grad_ld and the transform (as the fp32 kernel does);
//   3. back-propagates through the TRANSPOSED weights, a second operand image of the same form: the accumulator
//      layout of one product is the B-operand layout of the next, so the delta chain needs no data movement either;
//   4. accumulates dW_l += delta_l^T a_{l-1} with the 16 rows on the K axis (as K = 32 products whose two K halves carry
//      the head and the residual of delta: acc_outer32).  That product
//      wants both operands with ROWS along a lane's registers, the chain has UNITS there: a tile is transposed by
//      one more MFMA against the identity matrix (exact: every product is x * 1), the hi and the lo part separately;
//      the lo part is multiplied by 2^-11 on the way (identity entries 2^-11), so the three products
//      d_hi a_hi + d_hi a_lo' + d_lo' a_hi of a weight-gradient tile all go into ONE fp32 accumulator.
//
// Gradients of a mean over 2^20 rows are ~1e-6, far below f16's normal range, so the caller passes a power of two
// (`g_scale`, exact) that brings max |grad| near 1; grad_x and the parameter gradients are scaled back on the way out.
// Operands beyond the split range (mnf_split.h kSplitLimit) would overflow f16: such 16-row tiles are not
// accumulated, their indices go to `cold_list` and mnf_affine_half_bwd_mfma recomputes exactly those on the fp32
// pipe afterwards (same stream).
//
// The 28 weight-gradient tiles (112 VGPRs at d = 64) stay in registers across all tiles of a wave; at the end the
// four waves of a workgroup add them up in LDS and issue one atomic add per parameter, through the fp32 kernel's
// flush tables (same tile order).  One wave per SIMD (the whole register file), one workgroup per CU.
#include <hip/hip_runtime.h>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_split.h"
#include "mnf_ahf_bwd_shape.h"
#include "mnf_agpr.h"

#include <utility>

namespace mnf {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kBsWaves = 4;

template <int H, int HID>
struct BwdSplitShape {
  using S = SplitShape<H, HID>;
  using F = BwdShape<H, HID>;
  static constexpr int G = S::G, NT = S::NT, NKS = S::NKS, KS1 = S::KS1;
  static_assert(G == S::GC, "one output chunk (d <= 128)");
  static_assert(G == 1 || (G & 1) == 0, "delta-4 tiles pair up inside a net");
  // delta-4 tiles c in [0, 2 G): [s tiles | t tiles]; K-step p of the transposed output layer pairs tiles 2p, 2p + 1
  static constexpr int NP4 = G;
  static constexpr int c_net(int c) { return c / G; }
  static constexpr bool needs4(int m, int p) {
    const int nets = S::tile_nets(m);
    return ((nets >> c_net(2 * p)) & 1) || ((nets >> c_net(2 * p + 1)) & 1);
  }
  static constexpr int t4_ops() {
    int n = 0;
    for (int p = 0; p < NP4; ++p)
      for (int m = 0; m < NT; ++m) n += needs4(m, p) ? 1 : 0;
    return n;
  }
  static constexpr int t1_ops() {  // grad x0 tile g <- every hidden tile, each through ONE K-step
    int n = 0;
    for (int ks = 0; ks < NKS; ++ks)
      if (S::uses(3, ks)) n += G;
    return n;
  }
  // the last K-step hidden tile m takes part in (hidden layers / the transposed output layer): its vector work can
  // start behind that step while the later steps' MFMAs run
  static constexpr int last_ks(int m) {
    int r = 0;
    for (int ks = 0; ks < NKS; ++ks)
      if (S::uses(S::tile_nets(m), ks)) r = ks;
    return r;
  }
  static constexpr int last_p4(int m) {
    int r = 0;
    for (int p = 0; p < NP4; ++p)
      if (needs4(m, p)) r = p;
    return r;
  }
  static constexpr int T_OPS = t4_ops() + 2 * S::hidden_ops() + t1_ops();
  static constexpr int FWD_WORDS = S::SPLIT_WORDS;
  static constexpr int T_WORDS = T_OPS * 2 * 256;
  static constexpr int SPLIT_WORDS = FWD_WORDS + T_WORDS;
  static constexpr int PLAIN_WORDS = S::PLAIN_WORDS;
  static constexpr int IMAGE_WORDS = SPLIT_WORDS + PLAIN_WORDS + kSplitTailWords;
  static constexpr int INDEX_INTS = 2 * SPLIT_WORDS + PLAIN_WORDS;
  static constexpr int RED_FLOATS = F::DW_TILES * 256 + F::DB_TILES * 16;
  static constexpr int LDS_WORDS = IMAGE_WORDS > RED_FLOATS ? IMAGE_WORDS : RED_FLOATS;
};

template <typename Fn, int... I>
__device__ __forceinline__ void bs_static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename Fn>
__device__ __forceinline__ void bs_static_for(Fn&& f) {
  bs_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Position of a weight-gradient tile in the fp32 kernel's order (mnf_ahf_bwd_shape.h: layer 1, hidden 1, hidden 2,
// output), as compile-time functions: the accumulators are addressed by register number (below).
template <int H, int HID>
struct DwOrder {
  using S = SplitShape<H, HID>;
  using F = BwdShape<H, HID>;
  static constexpr int G = S::G, NT = S::NT;
  static constexpr bool out_used(int mo, int mi) { return ((S::tile_nets(mi) >> (mo / G)) & 1) != 0; }
  static constexpr int out_rank(int mo, int mi) {
    int n = 0;
    for (int a = 0; a < 2 * G; ++a)
      for (int b = 0; b < NT; ++b) {
        if (a == mo && b == mi) return n;
        if (out_used(a, b)) ++n;
      }
    return -1;
  }
  static constexpr int hid_rank(int mo, int mi) {
    int n = 0;
    for (int a = 0; a < NT; ++a)
      for (int b = 0; b < NT; ++b) {
        if (a == mo && b == mi) return n;
        if (F::needs(a, b)) ++n;
      }
    return -1;
  }
};

// The weight- and bias-gradient accumulators live in the accumulator half of the register file under fixed numbers,
// a[92 + 4 t : 92 + 4 t + 3] for tile t (the top of the file; the compiler puts the few values of its own that
// overflow the vector registers at the bottom), touched only by the MFMAs below (as loop-carried C++ values hipcc moved them to a
// new place on every trip: ~800 copy instructions per 16 rows, as many as the arithmetic).  mnf_agpr.h's rules
// apply: the file is built with the AGPR flags and check_agpr.py guards the build.  Back-to-back MFMAs on the SAME
// accumulator registers need no wait states (the hardware interlocks a full overlap of SrcC and vDst).
// A vector instruction's result needs 2 wait states before an MFMA may read it; hipcc inserts them for the MFMAs it
// generates but cannot see into these, and it does place operand-producing instructions (a v_cvt_pk of a transpose,
// the v_mov that materialises the ones) directly in front: every statement therefore starts with its own s_nop 1.
// One weight-gradient tile: acc += dh ah + dl ah + dh al + dl al as TWO K = 32 products (round 4; three K = 16 ones
// before -- v_mfma_f32_16x16x16_f16 occupies the matrix pipe as long as v_mfma_f32_16x16x32_f16 does): a K = 32
// product sums its two K halves, so A = [dh | dl] against B = [ah | ah] gives (dh + dl) ah and against [al | al]
// (dh + dl) al -- the lo x lo term the three-product form drops comes along for free.
template <int T>
__device__ __forceinline__ void acc_outer32(const f16x8& d_hl, const f16x8& a_hh, const f16x8& a_ll) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 a[%0:%1], %2, %3, a[%0:%1]\n\t"
               "v_mfma_f32_16x16x32_f16 a[%0:%1], %2, %4, a[%0:%1]" ::"n"(kTopAgprBase + 4 * T),
               "n"(kTopAgprBase + 4 * T + 3), "v"(d_hl), "v"(a_hh), "v"(a_ll));
}
// One bias-gradient tile: acc += (dh + dl) ones: one K = 32 product against [ones | ones]
template <int T>
__device__ __forceinline__ void acc_bias32(const f16x8& d_hl, const f16x8& ones8) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(kTopAgprBase + 4 * T),
               "n"(kTopAgprBase + 4 * T + 3), "v"(d_hl), "v"(ones8));
}
template <int T>
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\t"
               "v_accvgpr_write_b32 a[%3], 0" ::"n"(kTopAgprBase + 4 * T),
               "n"(kTopAgprBase + 4 * T + 1), "n"(kTopAgprBase + 4 * T + 2), "n"(kTopAgprBase + 4 * T + 3));
}
template <int T>
__device__ __forceinline__ f32x4 acc_read() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
               "v_accvgpr_read_b32 %3, a[%7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
               : "n"(kTopAgprBase + 4 * T), "n"(kTopAgprBase + 4 * T + 1), "n"(kTopAgprBase + 4 * T + 2),
                 "n"(kTopAgprBase + 4 * T + 3));
  return v;
}

// LeakyReLU'(pre-activation) = 1 where the activation v is positive, read off its split form instead of keeping the
// fp32 activations of three layers alive through the whole tile (36 registers).  Element r of the tile = half r & 1
// of word r >> 1.  The 32-bit key [head | tail] has v's sign: the head's, and where the head is +0 (v < 2^-25) the
// tail's, which is >= 0 there (a negative v rounds to the head -0: sign bit set).  One v_perm_b32 per element.
__device__ __forceinline__ bool unit_active(const u32x2& hi, const u32x2& lo, int r) {
  const uint32_t key = __builtin_amdgcn_perm(hi[r >> 1], lo[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
  return (int32_t)key > 0;
}

__device__ __forceinline__ f32x4 mfma_x16(const f16x4& a, const f16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16x4 as_f16x4(const u32x2& v) { return __builtin_bit_cast(f16x4, v); }

// LP: the launch differentiates log p = log_det + log N(y; 0, I) of a density pass whose LAST layer this is: the
// cotangents are not read but formed from the per-row d loss / d log p (`lp_grad`): grad_ld = g, grad_y = -y g, with the
// layer's output y recomputed here (y_cond = x_cond; y_act from s, t) -- no `-z g` elementwise launch, no grad_y read.
// A tile the range verdict hands to the fp32 pass leaves its grad_y rows in `gy_scratch` for that pass.
template <int H, int HID, bool INV, bool LP>
__global__ void __launch_bounds__(kBsWaves * 64, 1)
ahf_bwd_split_kernel(const float* __restrict__ x, const float* __restrict__ grad_y, const float* __restrict__ grad_ld,
                     float* __restrict__ grad_x, float* __restrict__ grad_flat, const uint32_t* __restrict__ image,
                     const int32_t* __restrict__ index, int64_t rows, int parity, const float* __restrict__ scale_dev,
                     int32_t* __restrict__ cold_list, int cold_capacity, float* __restrict__ partials, const float* __restrict__ lp_grad, float* __restrict__ gy_scratch) {
  using B = BwdSplitShape<H, HID>;
  using S = typename B::S;
  using F = typename B::F;
  constexpr int G = B::G, NT = B::NT, NKS = B::NKS, KS1 = B::KS1, dim = 2 * H;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  // weights beyond the split range (max |w|, written behind the image by the pack kernel): the whole launch belongs
  // to the fp32 pass
  if (!(__builtin_bit_cast(float, image[B::SPLIT_WORDS + B::PLAIN_WORDS]) <= kSplitWeightLimit)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) cold_list[0] = -1;
    return;
  }
  {
    const uint4* src = reinterpret_cast<const uint4*>(image);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = threadIdx.x; i < B::IMAGE_WORDS / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int cond_off = parity ? H : 0, act_off = parity ? 0 : H;
  const float g_scale = scale_dev[0], g_unscale = 1.0f / g_scale;  // a power of two: both exact

  // identity operands of the transposing MFMA: B[k = 4 q + e][n = j] = (k == n), and the same times 2^-11
  f16x4 ident, ident_lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ident[e] = (_Float16)((4 * q + e == j) ? 1.0f : 0.0f);
    ident_lo[e] = (_Float16)((4 * q + e == j) ? kSplitInvScale : 0.0f);
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};

  // accumulators: tiles 0 .. DW_TILES - 1 the weight gradients, then DB_TILES bias-gradient tiles (every column of
  // such a tile holds the same sums: delta^T times a matrix of ones)
  using O = DwOrder<H, HID>;
  constexpr int ACC_TILES = F::DW_TILES + F::DB_TILES;
  static_assert(kTopAgprBase + 4 * ACC_TILES <= 256, "accumulator registers");
  reserve_agprs_top();
  bs_static_for<ACC_TILES>([&](auto t) { acc_zero<decltype(t)::value>(); });
  f16x4 ones;
#pragma unroll
  for (int e = 0; e < 4; ++e) ones[e] = (_Float16)1.0f;
  constexpr int DW_L1 = 0, DW_H1 = NT * G, DW_H2 = DW_H1 + F::PAIRS, DW_OUT = DW_H2 + F::PAIRS;
  constexpr int DB_L1 = 0, DB_H1 = NT, DB_H2 = 2 * NT, DB_OUT = 3 * NT;

  const int n_tiles = (int)((rows + 15) >> 4);
  const int tile_step = (int)gridDim.x * kBsWaves;
  // One wave per SIMD: nobody hides this wave's memory latency, so the next tile's rows are requested at the top of
  // the current tile and carried in registers (hipcc keeps the requests there; measured 5.14 -> 4.65 ms per 9-layer
  // training step at 2^20 rows; pulling them through the caches by LDS-DMA instead gave 4.77).
  f32x4 n_cnd[G], n_act[G], n_gc[G], n_ga[G];
  float n_gl;
  const float* const gy_or_x = grad_y ? grad_y : x;
  const float* const gl_or_x = LP ? lp_grad : (grad_ld ? grad_ld : x);
  auto load_rows = [&](int tile) {
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* xr = x + rowc * dim + 4 * q;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      n_cnd[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
      n_act[g] = *reinterpret_cast<const f32x4*>(xr + act_off + 16 * g);
      // (no branch around a load: behind one hipcc's wait counts fall back to vmcnt(0), and the tile's first use of its
      //  rows would also wait for the previous tile's grad_x stores.  A missing cotangent reads x instead and a row past
      //  the end the last row: both are multiplied by zero where they are used.)
      if constexpr (!LP) {
        n_gc[g] = *reinterpret_cast<const f32x4*>(gy_or_x + rowc * dim + 4 * q + cond_off + 16 * g);
        n_ga[g] = *reinterpret_cast<const f32x4*>(gy_or_x + rowc * dim + 4 * q + act_off + 16 * g);
      }
    }
    n_gl = gl_or_x[rowc];
  };
  // ------------------------------------------------------------------ weight gradients: rows on the K axis
  // The split tiles of a row tile that its weight-gradient products read (declared out here for the blocks below).
  // (Issuing a tile's products one tile later, block by block between the next tile's forward stages -- in registers,
  // no LDS round trip -- measured 272 us against 266: profiles/r5/ahf_bwd_split_ablation.txt, commit 3bb0743.)
  using std::integral_constant;
  u32x2 xh[G], xl[G], hh[3][NT], hl[3][NT];
  u32x2 d4h[2 * G], d4l[2 * G], dh[3][NT], dl_[3][NT];  // dh[2] = delta 3 (pre-activation of h3), dh[0] = delta 1
  // T(v): the tile with rows along the registers: lane (unit = j, q) holds rows 4 q .. 4 q + 3, head and residual
  auto transpose = [&](const u32x2& hi, const u32x2& lo, f16x4& th, f16x4& tl) {
    const f32x4 o = mfma_x16(as_f16x4(hi), ident, zero4);
    const f32x4 ol = mfma_x16(as_f16x4(lo), ident_lo, zero4);  // (lo 2^-11: the residual itself)
    th = __builtin_convertvector(o, f16x4);
    tl = __builtin_convertvector(ol, f16x4);
  };
  // a delta tile as the A operand [head | residual] of the K = 32 products, an activation tile as the two B operands
  // [head | head], [residual | residual]
  auto delta_op = [&](const u32x2& hi, const u32x2& lo) -> f16x8 {
    f16x4 th, tl;
    transpose(hi, lo, th, tl);
    return __builtin_shufflevector(th, tl, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto act_ops = [&](const u32x2& hi, const u32x2& lo, f16x8& hh_, f16x8& ll_) {
    f16x4 th, tl;
    transpose(hi, lo, th, tl);
    hh_ = __builtin_shufflevector(th, th, 0, 1, 2, 3, 4, 5, 6, 7);
    ll_ = __builtin_shufflevector(tl, tl, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const f16x8 ones8 = __builtin_shufflevector(ones, ones, 0, 1, 2, 3, 4, 5, 6, 7);
  // MNF_BS_ABL (timing-only builds, tools/lib_variant.sh; results are wrong): 1 = no weight-gradient phase at all,
  // 2 = the transposes without the products
#ifndef MNF_BS_ABL
#define MNF_BS_ABL 0
#endif
  // one weight-gradient tile += delta^T a (all four split products), one bias-gradient tile += delta^T ones
  auto outer = [&](auto t, const f16x8& d_hl, const f16x8& a_hh, const f16x8& a_ll) {
    if (MNF_BS_ABL & 2)
      asm volatile("" ::"v"(d_hl), "v"(a_hh), "v"(a_ll));
    else
      acc_outer32<decltype(t)::value>(d_hl, a_hh, a_ll);
  };
  auto bias = [&](auto t, const f16x8& d_hl) {
    if (MNF_BS_ABL & 2)
      asm volatile("" ::"v"(d_hl));
    else
      acc_bias32<F::DW_TILES + decltype(t)::value>(d_hl, ones8);
  };
  // output layer: delta 4 (2 G tiles) x h3 (NT tiles)
  auto w_out = [&] {
    if (MNF_BS_ABL & 1) return;
    f16x8 a_hh[NT], a_ll[NT], d_hl[2 * G];
#pragma unroll
    for (int m = 0; m < NT; ++m) act_ops(hh[2][m], hl[2][m], a_hh[m], a_ll[m]);
#pragma unroll
    for (int c = 0; c < 2 * G; ++c) d_hl[c] = delta_op(d4h[c], d4l[c]);
    bs_static_for<2 * G>([&](auto mo_c) {
      constexpr int mo = decltype(mo_c)::value;
      bias(integral_constant<int, DB_OUT + mo>{}, d_hl[mo]);
      bs_static_for<NT>([&](auto mi_c) {
        constexpr int mi = decltype(mi_c)::value;
        if constexpr (O::out_used(mo, mi))
          outer(integral_constant<int, DW_OUT + O::out_rank(mo, mi)>{}, d_hl[mo], a_hh[mi], a_ll[mi]);
      });
    });
  };
  // hidden layer W_2 (h2 -> h3, l = 2): delta 3 x h2;  W_1 (h1 -> h2, l = 1): delta 2 x h1
  auto w_hid = [&](auto lc) {
    if (MNF_BS_ABL & 1) return;
    constexpr int l = decltype(lc)::value;
    f16x8 a_hh[NT], a_ll[NT], d_hl[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      d_hl[m] = delta_op(dh[l][m], dl_[l][m]);
      act_ops(hh[l - 1][m], hl[l - 1][m], a_hh[m], a_ll[m]);
    }
    bs_static_for<NT>([&](auto mo_c) {
      constexpr int mo = decltype(mo_c)::value;
      bias(integral_constant<int, (l == 2 ? DB_H2 : DB_H1) + mo>{}, d_hl[mo]);
      bs_static_for<NT>([&](auto mi_c) {
        constexpr int mi = decltype(mi_c)::value;
        if constexpr (F::needs(mo, mi))
          outer(integral_constant<int, (l == 2 ? DW_H2 : DW_H1) + O::hid_rank(mo, mi)>{}, d_hl[mo], a_hh[mi], a_ll[mi]);
      });
    });
  };
  // layer 1: delta 1 x x0
  auto w_l1 = [&](const u32x2* xh_, const u32x2* xl_) {
    if (MNF_BS_ABL & 1) return;
    f16x8 x_hh[G], x_ll[G], d_hl[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) d_hl[m] = delta_op(dh[0][m], dl_[0][m]);
#pragma unroll
    for (int g = 0; g < G; ++g) act_ops(xh_[g], xl_[g], x_hh[g], x_ll[g]);
    bs_static_for<NT>([&](auto mo_c) {
      constexpr int mo = decltype(mo_c)::value;
      bias(integral_constant<int, DB_L1 + mo>{}, d_hl[mo]);
      bs_static_for<G>([&](auto mi_c) {
        constexpr int mi = decltype(mi_c)::value;
        outer(integral_constant<int, DW_L1 + mo * G + mi>{}, d_hl[mo], x_hh[mi], x_ll[mi]);
      });
    });
  };

  const int first_tile = (int)blockIdx.x * kBsWaves + wave;
  load_rows(first_tile < n_tiles ? first_tile : 0);
  for (int tile = first_tile; tile < n_tiles; tile += tile_step) {
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float gy_on = (grad_y && live) ? g_scale : 0.f;
    const float g_raw = n_gl;                          // (LP: d loss / d log p of the row, unscaled)
    const float g_row = live ? n_gl * g_scale : 0.f;   // (LP: the same, scaled; zero past the last row)
    f32x4 cnd[G], act[G], gc[G], ga[G], y_act[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      cnd[g] = n_cnd[g];
      act[g] = n_act[g];
      if constexpr (LP) {
        gc[g] = cnd[g] * -g_row;  // grad_y of the conditioning half: y_cond = x_cond
      } else {
        gc[g] = n_gc[g] * gy_on;
        ga[g] = n_ga[g] * gy_on;
      }
    }
    const float gl = LP ? g_row : n_gl * ((grad_ld && live) ? g_scale : 0.f);
    load_rows(tile + tile_step < n_tiles ? tile + tile_step : tile);  // (past the end: this tile again, unused)

    int a_off = lane * 4, b_off = B::SPLIT_WORDS + q * 4;
    asm volatile("" : "+v"(a_off), "+v"(b_off));  // keep the operand reads inside the tile loop
    const f16x8* A8 = reinterpret_cast<const f16x8*>(lds + a_off);  // + 64 * (2 op + part)
    const f32x4* B4 = reinterpret_cast<const f32x4*>(lds + b_off);  // + 4 * bias tile
    int op = 0;
    float mx = 0.f;
    auto pair_of = [&](const u32x2* v, int a, int b) { return pair_operand(v[a], b >= 0 ? v[b >= 0 ? b : 0] : zero2); };
    // A stage = [its operands requested from LDS] [the vector work that builds its B operands] [its MFMAs], pinned in
    // that order (sched_barrier): with one wave per SIMD nothing else covers the LDS latency, and left alone hipcc
    // sinks every ds_read to just in front of its MFMA (SQ counters: 40 % of the wave's cycles in s_waitcnt).
#ifndef MNF_BS_RFENCE
#define MNF_BS_RFENCE 1
#endif
    auto fence = [] {
      if (MNF_BS_RFENCE) __builtin_amdgcn_sched_barrier(0);
    };
    // the fence between a stage's vector work and the next stage's MFMAs is NOT placed (round 4): without it hipcc moves
    // the first MFMAs of the next stage in between the vector instructions of the last tile, 308 -> 300 us per launch
    // at 2^20 x 64 (same box; without the read fences as well 306-310)
#ifndef MNF_BS_EARLY
#define MNF_BS_EARLY 1
#endif
#ifndef MNF_BS_VFENCE
#define MNF_BS_VFENCE 0
#endif
    auto fence_v = [] {
      if (MNF_BS_VFENCE) __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int N1 = NT * KS1, NH = S::hidden_ops();
    auto read_ops = [&](auto n_tag, f16x8* ah, f16x8* al) {
      constexpr int N = decltype(n_tag)::value;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        ah[i] = A8[64 * (2 * op)];
        al[i] = A8[64 * (2 * op + 1)];
        ++op;
      }
    };
    constexpr int N_OUT = [] {
      int n = 0;
      for (int net = 0; net < 2; ++net)
        for (int ks = 0; ks < NKS; ++ks)
          if (S::uses(1 << net, ks)) n += G;
      return n;
    }();
    constexpr int N_T4 = B::t4_ops(), N_T1 = B::t1_ops();

    // ------------------------------------------------------------------ forward recompute (split_conditioner's order)
    {
      f16x8 ah[N1], al[N1];
      read_ops(integral_constant<int, N1>{}, ah, al);
      fence();
#pragma unroll
      for (int g = 0; g < G; ++g) split_tile(cnd[g], xh[g], xl[g], mx);
      fence_v();
      f32x4 mn[NT], cr[NT];
      int i = 0;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const f16x8 bh = pair_of(xh, 2 * ks, 2 * ks + 1 < G ? 2 * ks + 1 : -1);
        const f16x8 bl = pair_of(xl, 2 * ks, 2 * ks + 1 < G ? 2 * ks + 1 : -1);
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if (ks == 0) {
            mn[m] = B4[4 * m];
            cr[m] = zero4;
          }
          split_mac(ah[i], al[i], bh, bl, mn[m], cr[m]);
          ++i;
        }
      }
      f16x8 ah2[NH], al2[NH];
      read_ops(integral_constant<int, NH>{}, ah2, al2);
      fence();
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const f32x4 p = cr[m] * kSplitInvScale + mn[m];
        split_tile(__builtin_elementwise_max(p, p * kLeakySlope), hh[0][m], hl[0][m], mx);
      }
      fence_v();
      // hidden layers 2 and 3; the operands of the layer after each are requested before its activation
#pragma unroll
      for (int l = 1; l <= 2; ++l) {
        f16x8* const ahl = ah2;
        f16x8* const all_ = al2;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          mn[m] = B4[4 * (l * NT + m)];
          cr[m] = zero4;
        }
        i = 0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const f16x8 bh = pair_of(hh[l - 1], S::ks_a(ks), S::ks_b(ks)), bl = pair_of(hl[l - 1], S::ks_a(ks), S::ks_b(ks));
#pragma unroll
          for (int m = 0; m < NT; ++m)
            if (S::uses(S::tile_nets(m), ks)) {
              split_mac(ahl[i], all_[i], bh, bl, mn[m], cr[m]);
              ++i;
            }
          // (a tile no later K-step adds to: its activation goes in front of the read fence, next to those MFMAs)
#pragma unroll
          for (int m = 0; m < NT; ++m)
            if (MNF_BS_EARLY && ks < NKS - 1 && B::last_ks(m) == ks) {
              const f32x4 p = cr[m] * kSplitInvScale + mn[m];
              split_tile(__builtin_elementwise_max(p, p * kLeakySlope), hh[l][m], hl[l][m], mx);
            }
        }
        if (l == 1) read_ops(integral_constant<int, NH>{}, ah2, al2);  // (hidden layer 3's; the registers are free again)
        fence();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if (MNF_BS_EARLY && B::last_ks(m) < NKS - 1) continue;
          const f32x4 p = cr[m] * kSplitInvScale + mn[m];
          split_tile(__builtin_elementwise_max(p, p * kLeakySlope), hh[l][m], hl[l][m], mx);
        }
        fence_v();
      }
    }
    f32x4 st[2][G];  // raw s (net 0) and t (net 1)
    f16x8 t4h[N_T4], t4l[N_T4];
    {
      f16x8 ah[N_OUT], al[N_OUT];
      read_ops(integral_constant<int, N_OUT>{}, ah, al);
      int i = 0;
#pragma unroll
      for (int net = 0; net < 2; ++net) {
        f32x4 oc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          st[net][g] = B4[4 * (3 * NT + net * G + g)];
          oc[g] = zero4;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          if (S::uses(1 << net, ks)) {
            const f16x8 bh = pair_of(hh[2], S::ks_a(ks), S::ks_b(ks)), bl = pair_of(hl[2], S::ks_a(ks), S::ks_b(ks));
#pragma unroll
            for (int g = 0; g < G; ++g) {
              split_mac(ah[i], al[i], bh, bl, st[net][g], oc[g]);
              ++i;
            }
          }
#pragma unroll
        for (int g = 0; g < G; ++g) st[net][g] = oc[g] * kSplitInvScale + st[net][g];
      }
    }

    // ------------------------------------------------------------------ output deltas, grad of the transformed half
    //   forward: y = e^s v + t          g_v = g e^s      g_s = g e^s v + g_ld      g_t = g
    //   inverse: y = (v - t) e^-s       g_v = g e^-s     g_s = -g y - g_ld         g_t = -g e^-s
    // (grad_x is stored as soon as it exists: if the range verdict below hands the tile to the fp32 pass, that pass
    //  overwrites it -- same stream, later)
    f32x4 d4[2 * G];  // [s tiles | t tiles]
    float* const gr = grad_x + rowc * dim + 4 * q;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      f32x4 gv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = st[0][g][r], t = st[1][g][r], v = act[g][r];
        const float e = exp6(INV ? -s : s);
        float gy;
        if constexpr (LP) {
          y_act[g][r] = INV ? (v - t) * e : __builtin_fmaf(e, v, t);
          gy = y_act[g][r] * -g_row;
        } else {
          gy = ga[g][r];
        }
        gv[r] = gy * e * g_unscale;
        d4[g][r] = live ? (INV ? -gy * ((v - t) * e) - gl : gy * e * v + gl) : 0.f;
        d4[G + g][r] = INV ? -gy * e : gy;
      }
      if (live) *reinterpret_cast<f32x4*>(gr + act_off + 16 * g) = gv;
    }

    // ------------------------------------------------------------------ the delta chain through the transposed weights
    read_ops(integral_constant<int, N_T4>{}, t4h, t4l);  // (not earlier: the delta arithmetic above is where the
    fence();                                               //  register demand peaks)
#pragma unroll
    for (int c = 0; c < 2 * G; ++c) split_tile(d4[c], d4h[c], d4l[c], mx);
    fence_v();
    f16x8 t1h[N_T1], t1l[N_T1];
    {
      f32x4 mn[NT], cr[NT];
#pragma unroll
      for (int m = 0; m < NT; ++m) mn[m] = cr[m] = zero4;
      int i = 0;
#pragma unroll
      for (int p = 0; p < B::NP4; ++p) {
        const f16x8 bh = pair_operand(d4h[2 * p], d4h[2 * p + 1]), bl = pair_operand(d4l[2 * p], d4l[2 * p + 1]);
#pragma unroll
        for (int m = 0; m < NT; ++m)
          if (B::needs4(m, p)) {
            split_mac(t4h[i], t4l[i], bh, bl, mn[m], cr[m]);
            ++i;
          }
#pragma unroll
        for (int m = 0; m < NT; ++m)
          if (MNF_BS_EARLY && p < B::NP4 - 1 && B::last_p4(m) == p) {
            f32x4 d = cr[m] * kSplitInvScale + mn[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = unit_active(hh[2][m], hl[2][m], r) ? d[r] : kLeakySlope * d[r];
            split_tile(d, dh[2][m], dl_[2][m], mx);
          }
      }
      f16x8 ah[NH], al[NH];
      read_ops(integral_constant<int, NH>{}, ah, al);
      fence();
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        if (MNF_BS_EARLY && B::last_p4(m) < B::NP4 - 1) continue;
        f32x4 d = cr[m] * kSplitInvScale + mn[m];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = unit_active(hh[2][m], hl[2][m], r) ? d[r] : kLeakySlope * d[r];
        split_tile(d, dh[2][m], dl_[2][m], mx);
      }
      fence_v();
#pragma unroll
      for (int l = 2; l >= 1; --l) {  // delta_l = W_l^T delta_{l+1} .* LeakyReLU'(h_l)
#pragma unroll
        for (int m = 0; m < NT; ++m) mn[m] = cr[m] = zero4;
        i = 0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const f16x8 bh = pair_of(dh[l], S::ks_a(ks), S::ks_b(ks)), bl = pair_of(dl_[l], S::ks_a(ks), S::ks_b(ks));
#pragma unroll
          for (int m = 0; m < NT; ++m)
            if (S::uses(S::tile_nets(m), ks)) {
              split_mac(ah[i], al[i], bh, bl, mn[m], cr[m]);
              ++i;
            }
#pragma unroll
          for (int m = 0; m < NT; ++m)
            if (MNF_BS_EARLY && ks < NKS - 1 && B::last_ks(m) == ks) {
              f32x4 d = cr[m] * kSplitInvScale + mn[m];
#pragma unroll
              for (int r = 0; r < 4; ++r) d[r] = unit_active(hh[l - 1][m], hl[l - 1][m], r) ? d[r] : kLeakySlope * d[r];
              split_tile(d, dh[l - 1][m], dl_[l - 1][m], mx);
            }
        }
        // the next stage's operands: the other hidden layer's (l = 2), the first layer's (l = 1)
        if (l == 2) read_ops(integral_constant<int, NH>{}, ah, al);
        else read_ops(integral_constant<int, N_T1>{}, t1h, t1l);
        fence();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if (MNF_BS_EARLY && B::last_ks(m) < NKS - 1) continue;
          f32x4 d = cr[m] * kSplitInvScale + mn[m];
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r] = unit_active(hh[l - 1][m], hl[l - 1][m], r) ? d[r] : kLeakySlope * d[r];
          split_tile(d, dh[l - 1][m], dl_[l - 1][m], mx);
        }
        fence_v();
      }
    }
    f32x4 gx0[G];
    {
      f32x4 cr[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        gx0[g] = gc[g];
        cr[g] = zero4;
      }
      int i = 0;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        if (S::uses(3, ks)) {
          const f16x8 bh = pair_of(dh[0], S::ks_a(ks), S::ks_b(ks)), bl = pair_of(dl_[0], S::ks_a(ks), S::ks_b(ks));
#pragma unroll
          for (int g = 0; g < G; ++g) {
            split_mac(t1h[i], t1l[i], bh, bl, gx0[g], cr[g]);
            ++i;
          }
        }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        gx0[g] = (cr[g] * kSplitInvScale + gx0[g]) * g_unscale;
        if (live) *reinterpret_cast<f32x4*>(gr + cond_off + 16 * g) = gx0[g];
      }
    }

    // ------------------------------------------------------------------ range verdict of the whole tile
    if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {
      // an operand left the split range: this tile's gradients come from the fp32 kernel (the caller runs it on the
      // listed tiles next); nothing of it has been accumulated
      if (lane == 0) {
        const int slot = atomicAdd(cold_list, 1);
        if (slot < cold_capacity) cold_list[1 + slot] = tile;
      }
      if constexpr (LP) {  // the fp32 pass reads its cotangents: this tile's grad_y rows, unscaled
        if (live) {
          float* const gs = gy_scratch + rowc * dim + 4 * q;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            *reinterpret_cast<f32x4*>(gs + cond_off + 16 * g) = cnd[g] * -g_raw;
            *reinterpret_cast<f32x4*>(gs + act_off + 16 * g) = y_act[g] * -g_raw;
          }
        }
      }
      continue;
    }
    // ------------------------------------------------------------------ weight gradients: rows on the K axis
    w_out();
    w_hid(integral_constant<int, 2>{});
    w_hid(integral_constant<int, 1>{});
    w_l1(xh, xl);
  }

  // ------------------------------------------------------------------ flush: sum over the waves in LDS, one atomic per parameter
  if (grad_flat == nullptr) return;
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);  // the images are no longer needed
  constexpr int DW_FLOATS = F::DW_TILES * 256;
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the last MFMAs' results, before the accumulators are read)
  for (int w = 0; w < kBsWaves; ++w) {
    if (wave == w) {
      bs_static_for<F::DW_TILES>([&](auto t) {
        constexpr int T = decltype(t)::value;
        f32x4* p = reinterpret_cast<f32x4*>(red + T * 256 + lane * 4);
        const f32x4 v = acc_read<T>();
        *p = w == 0 ? v : *p + v;
      });
      bs_static_for<F::DB_TILES>([&](auto t) {
        constexpr int T = decltype(t)::value;
        const f32x4 v = acc_read<F::DW_TILES + T>();  // lane (column n, q): units 4 q + r; every column the same
        if (j == 0) {
          f32x4* p = reinterpret_cast<f32x4*>(red + DW_FLOATS + T * 16 + 4 * q);
          *p = w == 0 ? v : *p + v;
        }
      });
    }
    __syncthreads();
  }
  if (partials) {  // two-stage flush: this workgroup's sums as one coalesced block, ahf_bwd_reduce_kernel adds them up
    float* dst = partials + (int64_t)blockIdx.x * B::RED_FLOATS;
    for (int i = threadIdx.x; i < B::RED_FLOATS; i += blockDim.x) dst[i] = red[i];
    return;
  }
  const int32_t* flush_w = index + F::IMAGE_FLOATS;
  const int32_t* flush_b = flush_w + DW_FLOATS;
  for (int i = threadIdx.x; i < DW_FLOATS; i += blockDim.x) {
    const int32_t dst = flush_w[i];
    if (dst >= 0) atomicAdd(grad_flat + dst, red[i] * g_unscale);
  }
  for (int i = threadIdx.x; i < F::DB_TILES * 16; i += blockDim.x) {
    const int32_t dst = flush_b[i];
    if (dst >= 0) atomicAdd(grad_flat + dst, red[DW_FLOATS + i] * g_unscale);
  }
}

// Second stage of the flush.  One atomic per parameter per workgroup is 1.9 M device-scope atomics per launch on 7,376
// addresses: ~40 us of the kernel whatever the row count (measured: 51 us for ONE tile per wave, 64 us at 65,536 rows,
// 117 at 262,144).  With a workspace every workgroup stores its sums as one block and this kernel adds the blocks up:
// thread i owns entry i of the flush tables (every parameter appears in exactly one entry: plain add, no atomics).
__global__ void __launch_bounds__(256) ahf_bwd_reduce_kernel(const float* __restrict__ partials, int n_blocks, int red_floats,
                                                             const int32_t* __restrict__ flush, float* __restrict__ grad_flat,
                                                             const float* __restrict__ scale_dev,
                                                             const int32_t* __restrict__ cold_list) {
  if (cold_list[0] < 0) return;  // the whole launch went to the fp32 pass: nothing was stored
  // 32 entries per workgroup x 8 slices of the blocks: the loads of a thread are few and independent (as one thread
  // per entry walking all 256 blocks this kernel took 22 us, all of it load latency)
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  const bool mine = i < red_floats;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (mine) {
    int b = slice;
    for (; b + 24 < n_blocks; b += 32) {
      s0 += partials[(int64_t)(b + 0) * red_floats + i];
      s1 += partials[(int64_t)(b + 8) * red_floats + i];
      s2 += partials[(int64_t)(b + 16) * red_floats + i];
      s3 += partials[(int64_t)(b + 24) * red_floats + i];
    }
    for (; b < n_blocks; b += 8) s0 += partials[(int64_t)b * red_floats + i];
  }
  part[slice][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && mine) {
    const int32_t dst = flush[i];
    if (dst >= 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += part[k][e];
      grad_flat[dst] += t * (1.0f / scale_dev[0]);
    }
  }
}

// The power of two that brings max(|grad_y|, |grad_ld|) over a sample of `sample` rows into [1, 2) (1 for an all-zero or
// non-finite sample): one workgroup, the sample is small.  The sample is spread evenly over the whole batch (row
// s * stride): a sorted, masked or weighted batch whose first rows carry no or atypically small cotangents would
// otherwise set a scale that leaves the rest far outside [1, 2).
__device__ __forceinline__ float scale_of_max(float m) {
  float scale = 1.f;
  if (m > 0.f && m < __builtin_inff()) {
    int e;
    frexpf(m, &e);  // m = f 2^e, f in [0.5, 1)
    e = 1 - e;
    e = e > 120 ? 120 : (e < -120 ? -120 : e);
    scale = ldexpf(1.f, e);
  }
  return scale;
}
__global__ void __launch_bounds__(1024) grad_scale_kernel(const float* __restrict__ grad_y, const float* __restrict__ grad_ld,
                                                          int64_t sample, int64_t stride, int dim,
                                                          float* __restrict__ scale_out) {
  __shared__ float part[16];
  float m = 0.f;
  if (grad_y)
    for (int64_t i = threadIdx.x; i < sample * dim; i += blockDim.x) {
      const int64_t r = i / dim, c = i - r * dim;
      m = fmaxf(m, fabsf(grad_y[r * stride * dim + c]));
    }
  if (grad_ld)
    for (int64_t i = threadIdx.x; i < sample; i += blockDim.x) m = fmaxf(m, fabsf(grad_ld[i * stride]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, part[w]);
    scale_out[0] = scale_of_max(m);
  }
}

// Wide rows (RNVP at d = 800: 512 x 800 sample values): the same maximum over several workgroups -- sample rows
// blockIdx.x, blockIdx.x + gridDim.x, ... each -- collected with one atomic per workgroup into *scale_out (zeroed by the
// caller; non-negative floats order like their bit patterns), then turned into the power of two by one thread.
__global__ void __launch_bounds__(256) grad_max_kernel(const float* __restrict__ grad_y, const float* __restrict__ grad_ld,
                                                       int64_t sample, int64_t stride, int dim,
                                                       uint32_t* __restrict__ max_bits) {
  __shared__ float part[4];
  float m = 0.f;
  for (int64_t r = blockIdx.x; r < sample; r += gridDim.x) {
    if (grad_y)
      for (int c = threadIdx.x; c < dim; c += blockDim.x) m = fmaxf(m, fabsf(grad_y[r * stride * dim + c]));
    if (grad_ld && threadIdx.x == 0) m = fmaxf(m, fabsf(grad_ld[r * stride]));
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, part[w]);
    if (m > 0.f) atomicMax(max_bits, __builtin_bit_cast(uint32_t, m));
  }
}
__global__ void grad_scale_finalize_kernel(float* __restrict__ scale_out) {
  scale_out[0] = scale_of_max(scale_out[0]);
}

// ---------------------------------------------------------------- host: index table of the backward split image
// [forward split entries (mnf_affine_half_split_index) | transposed split entries | plain (bias) entries]
template <int H, int HID>
static int build_bwd_split_index(int32_t* idx) {
  using B = BwdSplitShape<H, HID>;
  using S = typename B::S;
  constexpr int G = B::G, NT = B::NT, NKS = B::NKS;
  int hidden[3] = {HID, HID, HID};
  {
    int32_t* tmp = new int32_t[2 * (int64_t)S::SPLIT_WORDS + S::PLAIN_WORDS];
    const int rc = mnf_affine_half_split_index(2 * H, 3, hidden, 1, 1, tmp);
    if (rc != MNF_OK) {
      delete[] tmp;
      return rc;
    }
    for (int64_t i = 0; i < 2 * (int64_t)S::SPLIT_WORDS; ++i) idx[i] = tmp[i];
    for (int64_t i = 0; i < S::PLAIN_WORDS; ++i) idx[2 * (int64_t)B::SPLIT_WORDS + i] = tmp[2 * (int64_t)S::SPLIT_WORDS + i];
    delete[] tmp;
  }
  int sizes[5] = {H, HID, HID, HID, H};
  NetDesc net[2];
  int64_t off = fill_net(net[0], 5, sizes, 0);
  fill_net(net[1], 5, sizes, off);
  for (int64_t i = 2 * (int64_t)B::FWD_WORDS; i < 2 * (int64_t)B::SPLIT_WORDS; ++i) idx[i] = -1;
  int op = S::N_OPS;  // transposed operands follow the forward ones
  auto put = [&](int lane, int e, int32_t src) {
    for (int part = 0; part < 2; ++part)
      idx[(((int64_t)(2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] = src | (part ? kSplitLoBit : 0);
  };
  auto netof = [&](int u) { return u / HID; };
  auto valid = [&](int u) { return u < 2 * HID; };
  // delta 3 [unit 16 m + i] += W4[dim of K slot][unit]: K-step p pairs delta-4 tiles 2 p, 2 p + 1
  for (int p = 0; p < B::NP4; ++p)
    for (int m = 0; m < NT; ++m) {
      if (!B::needs4(m, p)) continue;
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
        if (!valid(u)) continue;
        for (int e = 0; e < 8; ++e) {
          const int c = 2 * p + (e >> 2), nn = B::c_net(c), d = 16 * (c % G) + 4 * kq + (e & 3);
          if (netof(u) == nn) put(lane, e, net[nn].w_off[3] + d * HID + u % HID);
        }
      }
      ++op;
    }
  // delta_l [in unit 16 m + i] += W_l[out unit of K slot][in unit], l = 2, 1 (forward hidden layers' tiling, roles swapped)
  auto unit_k = [&](int ks, int kq, int e, int& tile) {
    tile = e < 4 ? S::ks_a(ks) : S::ks_b(ks);
    if (tile < 0) return -1;
    const int u = 16 * tile + 4 * kq + (e & 3);
    return u < 2 * HID ? u : -1;
  };
  for (int l = 2; l >= 1; --l)
    for (int ks = 0; ks < NKS; ++ks)
      for (int m = 0; m < NT; ++m) {
        if (!S::uses(S::tile_nets(m), ks)) continue;
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, ui = 16 * m + i;
          if (!valid(ui)) continue;
          for (int e = 0; e < 8; ++e) {
            int tile;
            const int uo = unit_k(ks, kq, e, tile);
            if (uo < 0 || netof(uo) != netof(ui) || S::assigned_ks(S::tile_nets(m), tile) != ks) continue;
            put(lane, e, net[netof(ui)].w_off[l] + (uo % HID) * HID + ui % HID);
          }
        }
        ++op;
      }
  // grad x0 [dim 16 g + i] += W1[unit of K slot][dim]
  for (int ks = 0; ks < NKS; ++ks) {
    if (!S::uses(3, ks)) continue;
    for (int g = 0; g < G; ++g) {
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4;
        for (int e = 0; e < 8; ++e) {
          int tile;
          const int u = unit_k(ks, kq, e, tile);
          if (u < 0 || S::assigned_ks(3, tile) != ks) continue;
          put(lane, e, net[netof(u)].w_off[0] + (u % HID) * H + 16 * g + i);
        }
      }
      ++op;
    }
  }
  return op == S::N_OPS + B::T_OPS ? MNF_OK : MNF_ERR_INVALID_ARG;
}

template <int H, int HID>
static int launch_bwd_split(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                            const uint32_t* image, const int32_t* index, int64_t rows, int parity, int inverse,
                            const float* scale_dev, int32_t* cold_list, int cold_capacity, float* workspace,
                            int64_t workspace_floats, hipStream_t stream, const float* lp_grad = nullptr,
                            float* gy_scratch = nullptr) {
  using B = BwdSplitShape<H, HID>;
  using Kernel = void (*)(const float*, const float*, const float*, float*, float*, const uint32_t*, const int32_t*, int64_t,
                          int, const float*, int32_t*, int, float*, const float*, float*);
  static constexpr size_t lds_bytes = B::LDS_WORDS * sizeof(uint32_t);
  const Kernel all[4] = {ahf_bwd_split_kernel<H, HID, true, false>, ahf_bwd_split_kernel<H, HID, false, false>,
                         ahf_bwd_split_kernel<H, HID, true, true>, ahf_bwd_split_kernel<H, HID, false, true>};
  static DeviceMemo memo;
  const int cus = memo.get([&](int dev) {
    bool ok = true;
    for (int k = 0; k < 4; ++k)
      ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(all[k]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes) == hipSuccess;
    return ok ? device_cus(dev) : -1;
  });
  if (cus <= 0) return MNF_ERR_UNSUPPORTED;
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + kBsWaves - 1) / kBsWaves;
  if (blocks > cus) blocks = cus;  // one persistent workgroup per CU (one wave per SIMD: the whole register file)
  const dim3 grid((unsigned)blocks), block(kBsWaves * 64);
  float* partials = (grad_flat && workspace && workspace_floats >= blocks * B::RED_FLOATS) ? workspace : nullptr;
  tag_kernel("ahf_bwd_split");
  hipLaunchKernelGGL(all[(lp_grad ? 2 : 0) + (inverse ? 0 : 1)], grid, block, lds_bytes, stream, x, grad_y, grad_ld, grad_x,
                     grad_flat, image, index, rows, parity, scale_dev, cold_list, cold_capacity, partials, lp_grad,
                     gy_scratch);
  if (partials)
    hipLaunchKernelGGL(ahf_bwd_reduce_kernel, dim3((B::RED_FLOATS + 31) / 32), dim3(256), 0, stream, partials, (int)blocks,
                       (int)B::RED_FLOATS, index + B::F::IMAGE_FLOATS, grad_flat, scale_dev, cold_list);
  return check_launch();
}

static bool bwd_split_uniform3(int n_hidden, const int* hidden, int& hid) {
  if (n_hidden != 3 || !hidden) return false;
  hid = hidden[0];
  return hidden[1] == hid && hidden[2] == hid;
}

}  // namespace mnf

// shapes: those of the fp32 gradient kernel
#define MNF_AHF_BWD_SPLIT_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16)

extern "C" {

int mnf_affine_half_bwd_split_layout(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                     int64_t* n_split_words, int64_t* n_plain_words) {
  int hid = 0;
  if (!n_split_words || !n_plain_words || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!has_scale || !has_shift || !mnf::bwd_split_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                   \
  if (dim == 2 * HH && hid == HD) {                                 \
    *n_split_words = mnf::BwdSplitShape<HH, HD>::SPLIT_WORDS;       \
    *n_plain_words = mnf::BwdSplitShape<HH, HD>::PLAIN_WORDS;       \
    return MNF_OK;                                                  \
  }
  MNF_AHF_BWD_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_affine_half_bwd_split_index(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                    int32_t* idx_host) {
  int hid = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!has_scale || !has_shift || !mnf::bwd_split_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD) \
  if (dim == 2 * HH && hid == HD) return mnf::build_bwd_split_index<HH, HD>(idx_host);
  MNF_AHF_BWD_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int64_t mnf_affine_half_bwd_split_workspace(int64_t rows, int dim, int n_hidden, const int* hidden) {
  int hid = 0;
  if (rows < 0 || !mnf::hidden_ok(n_hidden, hidden) || !mnf::bwd_split_uniform3(n_hidden, hidden, hid)) return 0;
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + mnf::kBsWaves - 1) / mnf::kBsWaves;
  const int cus = mnf::device_cus(mnf::current_device());
  if (blocks > cus) blocks = cus;
#define X(HH, HD) \
  if (dim == 2 * HH && hid == HD) return blocks * mnf::BwdSplitShape<HH, HD>::RED_FLOATS;
  MNF_AHF_BWD_SPLIT_SHAPES(X)
#undef X
  return 0;
}

int mnf_affine_half_grad_scale(const float* grad_y, const float* grad_ld, int64_t rows, int dim, float* scale_out,
                               void* stream) {
  if (!scale_out || rows < 0 || dim < 1 || (!grad_y && !grad_ld)) return MNF_ERR_INVALID_ARG;
  const int64_t sample = rows < 512 ? rows : 512;  // (one workgroup: a larger sample costs more than it tells)
  const int64_t stride = sample > 0 ? rows / sample : 1;  // rows 0, stride, 2 stride, ...: spread over the whole batch
  if (grad_y && sample * dim > 65536) {  // wide rows: several workgroups (one took 150 us at 512 x 800)
    if (int rc = mnf::zero_word_async(scale_out, (hipStream_t)stream)) return rc;
    hipLaunchKernelGGL(mnf::grad_max_kernel, dim3((unsigned)(sample < 128 ? sample : 128)), dim3(256), 0,
                       (hipStream_t)stream, grad_y, grad_ld, sample, stride, dim, reinterpret_cast<uint32_t*>(scale_out));
    if (int rc = mnf::check_launch()) return rc;
    hipLaunchKernelGGL(mnf::grad_scale_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale_out);
    return mnf::check_launch();
  }
  hipLaunchKernelGGL(mnf::grad_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, grad_y, grad_ld, sample,
                     stride, dim, scale_out);
  return mnf::check_launch();
}

int mnf_affine_half_bwd_split(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                              float* grad_flat, const void* bwd_image, const int32_t* index_dev, int64_t rows, int dim,
                              int parity, int inverse, int n_hidden, const int* hidden, const float* grad_scale_dev,
                              int32_t* cold_list, int cold_capacity, float* workspace, int64_t workspace_floats,
                              void* stream) {
  int hid = 0;
  if (!x || !grad_x || !bwd_image || !index_dev || !cold_list || cold_capacity < 0 || rows < 0 || dim < 2 || (dim & 1) ||
      !mnf::hidden_ok(n_hidden, hidden) || !grad_scale_dev)
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (!mnf::bwd_split_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x) |
       reinterpret_cast<uintptr_t>(bwd_image)) & 15)
    return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                                                                      \
  if (dim == 2 * HH && hid == HD)                                                                                      \
    return mnf::launch_bwd_split<HH, HD>(x, grad_y, grad_ld, grad_x, grad_flat, static_cast<const uint32_t*>(bwd_image), \
                                         index_dev, rows, parity != 0, inverse != 0, grad_scale_dev, cold_list,            \
                                         cold_capacity, workspace, workspace_floats, (hipStream_t)stream);
  MNF_AHF_BWD_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}


int mnf_affine_half_bwd_split_lp(const float* x, const float* lp_grad, float* gy_scratch, float* grad_x,
                              float* grad_flat, const void* bwd_image, const int32_t* index_dev, int64_t rows, int dim,
                              int parity, int inverse, int n_hidden, const int* hidden, const float* grad_scale_dev,
                              int32_t* cold_list, int cold_capacity, float* workspace, int64_t workspace_floats,
                              void* stream) {
  int hid = 0;
  if (!x || !lp_grad || !gy_scratch || !grad_x || !bwd_image || !index_dev || !cold_list || cold_capacity < 0 || rows < 0 || dim < 2 || (dim & 1) ||
      !mnf::hidden_ok(n_hidden, hidden) || !grad_scale_dev)
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (!mnf::bwd_split_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy_scratch) | reinterpret_cast<uintptr_t>(grad_x) |
       reinterpret_cast<uintptr_t>(bwd_image)) & 15)
    return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                                                                      \
  if (dim == 2 * HH && hid == HD)                                                                                      \
    return mnf::launch_bwd_split<HH, HD>(x, nullptr, nullptr, grad_x, grad_flat, static_cast<const uint32_t*>(bwd_image), \
                                         index_dev, rows, parity != 0, inverse != 0, grad_scale_dev, cold_list,            \
                                         cold_capacity, workspace, workspace_floats, (hipStream_t)stream, lp_grad, gy_scratch);
  MNF_AHF_BWD_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
