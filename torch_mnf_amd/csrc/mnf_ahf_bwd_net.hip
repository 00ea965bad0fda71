// AffineHalfFlow gradients in split form, ONE CONDITIONER NET PER WAVE (round 4; the training step's hot kernel).
//
// mnf_ahf_bwd_split.hip gives a wave both nets of a 16-row tile: 164 accumulator registers of weight-gradient sums plus
// ~350 of working values, i.e. the whole register file of a SIMD, ONE wave per SIMD -- and that wave spends half its
// cycles waiting for its own MFMA / LDS results (SQ counters, profiles/r4/c2t_valu_issue.json: matrix pipe 30 % busy,
// vector issue 17 %).  The two nets of a coupling layer only meet at three points (s and t in the transform, the two
// shares of grad x_cond, the range verdict), so here a 16-row tile belongs to a PAIR of waves, wave `slot` (net s) and
// wave `slot + 4` (net t) of an 8-wave workgroup: each keeps its own net's weight-gradient sums (24 tiles = 96
// accumulator registers at d = 64, hidden 24) and runs its own net's chain in ~150 registers: two waves per SIMD, one
// covers the other's latencies.  The pair exchanges through per-wave LDS mailboxes around a workgroup barrier:
//
//   1. (inverse direction only) raw s <-> t after the forward recompute: d_s needs t, d_t needs e^-s;
//   2. net s's share of grad x_cond and both waves' max |operand| after the delta chain: the t wave adds the shares and
//      stores grad x_cond, both waves take the same range verdict.
//
// The s wave stores grad of the transformed half.  Every layer of a net is ONE K = 32 step (half and padded hidden
// width <= 32): the operand image holds, per net, the forward weights and their transposes as [out tile] A operands
// (hi | lo parts, mnf_split.h), the bias tiles behind them.  Split arithmetic, the gradient scale, the range guard
// (cold_list -> mnf_affine_half_bwd_mfma_tiles) and the transposing MFMA of the weight-gradient products are those of
// mnf_ahf_bwd_split.hip (read its header first).  At the end the four waves of a net add their sums in LDS in the
// order of the flat parameter vector: one coalesced block per workgroup, summed by ahf_bwd_net_reduce_kernel.
//
// MEASURED (profiles/r4/README.md): 400-430 us per layer-launch at 2^20 x 64 against 300 us for the joint kernel, so
// this form only runs under MNF_AHF_BWD_SPLIT=net.  Splitting by net pads each net's 24 hidden units to 32 (four hidden
// tiles per pair of waves instead of three) and splits x twice: 31 % more vector instructions per tile (SQ_INSTS_VALU
// 9.9e7 vs 7.6e7 per launch), and the vector-instruction slots, not the waits two waves per SIMD can cover, are what
// the joint kernel is short of.  Barriers or flags, either wave-to-SIMD map: the same time; without the weight-gradient
// phase 354 us, without the meetings 386.
#include <hip/hip_runtime.h>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_split.h"
#include "mnf_agpr.h"

#include <utility>

namespace mnf {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kNetSlots = 4;              // 16-row tiles in flight per workgroup
constexpr int kNetWaves = 2 * kNetSlots;

template <int H, int HID>
struct NetShape {
  static constexpr int G = H / 16, HP = (HID + 15) / 16 * 16, NTN = HP / 16, dim = 2 * H;
  static_assert(H % 16 == 0 && G >= 1 && G <= 2 && NTN >= 1 && NTN <= 2, "every layer is one K = 32 step");
  // layer l of a net: sizes[l] -> sizes[l + 1], sizes = [H, HID, HID, HID, H]; weight [out][in] then bias (fill_net)
  static constexpr int in_size(int l) { return l == 0 ? H : HID; }
  static constexpr int out_size(int l) { return l == 3 ? H : HID; }
  static constexpr int w_off(int l) {
    int off = 0;
    for (int k = 0; k < l; ++k) off += in_size(k) * out_size(k) + out_size(k);
    return off;
  }
  static constexpr int b_off(int l) { return w_off(l) + in_size(l) * out_size(l); }
  static constexpr int NET_FLOATS = w_off(4), FLAT_FLOATS = 2 * NET_FLOATS;
  static constexpr int in_tiles(int l) { return l == 0 ? G : NTN; }
  static constexpr int out_tiles(int l) { return l == 3 ? G : NTN; }
  // operand numbers inside a net: forward layer l's [out tile] A operands, then the transposed ones ([in tile])
  static constexpr int op_fwd(int l) { return l * NTN; }
  static constexpr int op_tr(int l) { return 3 * NTN + G + (l == 3 ? 0 : l == 2 ? NTN : l == 1 ? 2 * NTN : 3 * NTN); }
  static constexpr int OPS_NET = 6 * NTN + 2 * G;
  static constexpr int SPLIT_WORDS = 2 * OPS_NET * 512;
  // bias tiles inside a net: layer l's at l NTN (the output layer's G tiles last)
  static constexpr int BT_NET = 3 * NTN + G;
  static constexpr int PLAIN_WORDS = 2 * BT_NET * 16;
  static constexpr int IMAGE_WORDS = SPLIT_WORDS + PLAIN_WORDS + kSplitTailWords;
  static constexpr int INDEX_INTS = 2 * SPLIT_WORDS + PLAIN_WORDS;
  // accumulator tiles of a wave: weight gradients of layer l at dw(l) + mo in_tiles(l) + mi, then the bias gradients
  static constexpr int dw(int l) {
    int t = 0;
    for (int k = 0; k < l; ++k) t += out_tiles(k) * in_tiles(k);
    return t;
  }
  static constexpr int DW_TILES = dw(4);
  // A hidden width below its padded one leaves structural-zero units in the activations h1 .. h3: the first of them is
  // set to ONE in the weight-gradient products' activation operand, which makes column PAD_COL of the last in-tile's
  // weight-gradient tiles the bias gradient (sum over the rows of delta) -- no tiles, no products of its own.
  static constexpr bool PAD = HP > HID;
  static constexpr int PAD_COL = HID - 16 * (NTN - 1);
  static constexpr bool own_bias_tiles(int l) { return l == 0 || !PAD; }
  static constexpr int db(int l) { return DW_TILES + (PAD ? 0 : l * NTN); }  // (PAD: layer 0's only)
  static constexpr int ACC_TILES = DW_TILES + (PAD ? NTN : BT_NET);
  // the accumulators are the top 4 ACC_TILES vector registers of the wave's 256, the compiler gets the rest
  static constexpr int ACC_BASE = (256 - 4 * ACC_TILES) / 8 * 8;
  static_assert(ACC_BASE >= 160, "accumulator registers");
  // LDS: [image | per wave: box 1 (raw s / t), 2 x { box 2 (grad x_cond share), max |operand| }]
  static constexpr int MAIL_OFF = (IMAGE_WORDS + 63) / 64 * 64;
  static constexpr int BOX_WORDS = G * 256;
  static constexpr int MAIL_WAVE = BOX_WORDS + 2 * (BOX_WORDS + 64);
  static constexpr int FLAG_OFF = MAIL_OFF + kNetWaves * MAIL_WAVE;  // one word per wave (64 apart)
  static constexpr int LDS_WORDS = FLAG_OFF + kNetWaves * 64;
  static_assert(FLAT_FLOATS <= MAIL_OFF, "the flush area (flat parameter order) fits in front of the mailboxes");
  static_assert(LDS_WORDS * 4 <= 160 * 1024, "fits the CU's LDS");
};

template <typename Fn, int... I>
__device__ __forceinline__ void net_static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename Fn>
__device__ __forceinline__ void net_static_for(Fn&& f) {
  net_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- the accumulators: v[BASE + 4 t : BASE + 4 t + 3] for tile t, hand-assigned VECTOR registers (BASE = NetShape::
// ACC_BASE, the top of the file).  With two waves per SIMD a wave has 256 registers; the kernel is declared
// amdgpu_num_vgpr(BASE / 2) -- hipcc doubles the number on gfx90a and later --, which makes v[BASE] .. v255 reserved
// registers for the compiler (it never allocates them), and names no accumulator register in any constraint, so the
// whole file stays in vector registers (hipcc splits a wave's budget evenly between the two halves of the register file as soon
// as an asm statement mentions an accumulator register: 128 + 128, and nothing keeps its own overflow values out of
// hand-assigned accumulator registers BETWEEN the statements -- tried, see profiles/r4/README.md).  The clobber of v255
// makes the kernel's register count 256.  Touched only by the statements below (the MFMAs' C / D operands are vector
// registers: -amdgpu-mfma-vgpr-form); mnf_ahf_bwd_split.hip's notes on wait states apply.
__device__ __forceinline__ void reserve_acc_registers() { asm volatile("" ::: "v255"); }
// acc += (dh + dl) ah + (dh + dl) al: two K = 32 products, A = [dh | dl], B = [ah | ah], [al | al]
template <int BASE, int T>
__device__ __forceinline__ void net_acc_outer(const f16x8& d_hl, const f16x8& a_hh, const f16x8& a_ll) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 v[%0:%1], %2, %3, v[%0:%1]\n\t"
               "v_mfma_f32_16x16x32_f16 v[%0:%1], %2, %4, v[%0:%1]" ::"n"(BASE + 4 * T),
               "n"(BASE + 4 * T + 3), "v"(d_hl), "v"(a_hh), "v"(a_ll));
}
template <int BASE, int T>
__device__ __forceinline__ void net_acc_bias(const f16x8& d_hl, const f16x8& ones8) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 v[%0:%1], %2, %3, v[%0:%1]" ::"n"(BASE + 4 * T),
               "n"(BASE + 4 * T + 3), "v"(d_hl), "v"(ones8));
}
template <int BASE, int T>
__device__ __forceinline__ void net_acc_zero() {
  asm volatile("v_mov_b32 v[%0], 0\n\tv_mov_b32 v[%1], 0\n\tv_mov_b32 v[%2], 0\n\tv_mov_b32 v[%3], 0" ::"n"(BASE + 4 * T),
               "n"(BASE + 4 * T + 1), "n"(BASE + 4 * T + 2), "n"(BASE + 4 * T + 3));
}
template <int BASE, int T>
__device__ __forceinline__ f32x4 net_acc_read() {
  f32x4 v;
  asm volatile("v_mov_b32 %0, v[%4]\n\tv_mov_b32 %1, v[%5]\n\tv_mov_b32 %2, v[%6]\n\tv_mov_b32 %3, v[%7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
               : "n"(BASE + 4 * T), "n"(BASE + 4 * T + 1), "n"(BASE + 4 * T + 2),
                 "n"(BASE + 4 * T + 3));
  return v;
}

// LeakyReLU'(pre-activation) read off the split activation's sign (mnf_ahf_bwd_split.hip: unit_active)
__device__ __forceinline__ bool net_unit_active(const u32x2& hi, const u32x2& lo, int r) {
  const uint32_t key = __builtin_amdgcn_perm(hi[r >> 1], lo[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
  return (int32_t)key > 0;
}

// the pair's meeting point: everything this wave wrote to LDS has landed, nothing is said about global memory (the
// grad_x stores of this tile stay in flight: __syncthreads() would wait for them)
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MNF_NET_SYNC 1: the pair meets through a flag word per wave in LDS instead -- this wave's mailbox writes, then its
// flag = seq (a wave's LDS operations execute in order), then it polls the partner's flag -- so the four pairs of a
// workgroup drift apart and cover each other's waits.  A box is written again two meetings later, after the partner's
// next flag, which it raises after it has consumed the box.
#ifndef MNF_NET_SYNC
#define MNF_NET_SYNC 1
#endif
// MNF_NET_MAP 1: wave = 2 slot + net (a SIMD hosts the same net of two different tiles); 0: wave = slot + 4 net
#ifndef MNF_NET_MAP
#define MNF_NET_MAP 1
#endif
// MNF_NET_ABL (timing experiments, results wrong): 1 no global loads / stores, 2 no weight-gradient phase, 4 no meetings
#ifndef MNF_NET_ABL
#define MNF_NET_ABL 0
#endif
__device__ __forceinline__ void pair_meet(uint32_t* lds, int my_flag, int pr_flag, int seq, int lane) {
#if MNF_NET_ABL & 4
  return;
#endif
#if MNF_NET_SYNC
  asm volatile("" ::: "memory");
  if (lane == 0) *reinterpret_cast<volatile int*>(lds + my_flag) = seq;
  while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile const int*>(lds + pr_flag)) < seq) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
#else
  pair_barrier();
#endif
}

template <int H, int HID, bool INV>
__device__ __forceinline__ void ahf_bwd_net_body(const float* __restrict__ x, const float* __restrict__ grad_y,
                                                 const float* __restrict__ grad_ld, float* __restrict__ grad_x,
                                                 float* __restrict__ grad_flat, const uint32_t* __restrict__ image,
                                                 int64_t rows, int parity, const float* __restrict__ scale_dev,
                                                 int32_t* __restrict__ cold_list, int cold_capacity,
                                                 float* __restrict__ partials) {
  using N = NetShape<H, HID>;
  constexpr int G = N::G, NTN = N::NTN, dim = N::dim;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (!(__builtin_bit_cast(float, image[N::SPLIT_WORDS + N::PLAIN_WORDS]) <= kSplitWeightLimit)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) cold_list[0] = -1;  // weights beyond the split range: the fp32 pass's launch
    return;
  }
  {
    const uint4* src = reinterpret_cast<const uint4*>(image);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = threadIdx.x; i < N::IMAGE_WORDS / 4; i += kNetWaves * 64) dst[i] = src[i];
    lds[N::FLAG_OFF + threadIdx.x] = 0;
  }
  __syncthreads();
  const int lane = __lane_id();
#if MNF_NET_MAP
  const int slot = wave >> 1, net = wave & 1, partner = wave ^ 1;
#else
  const int slot = wave & (kNetSlots - 1), net = wave >> 2, partner = wave ^ kNetSlots;
#endif
  const int my_flag = N::FLAG_OFF + 64 * wave, pr_flag = N::FLAG_OFF + 64 * partner;
  int seq = 0;
  const int j = lane & 15, q = lane >> 4;
  const int cond_off = parity ? H : 0, act_off = parity ? 0 : H;
  const float g_scale = scale_dev[0], g_unscale = 1.0f / g_scale;  // a power of two: both exact

  // identity operands of the transposing MFMA: B[k = 4 q + e][n = j] = (k == n), and the same times 2^-11
  f16x4 ident, ident_lo, ones;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ident[e] = (_Float16)((4 * q + e == j) ? 1.0f : 0.0f);
    ident_lo[e] = (_Float16)((4 * q + e == j) ? kSplitInvScale : 0.0f);
    ones[e] = (_Float16)1.0f;
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};

  reserve_acc_registers();
  net_static_for<N::ACC_TILES>([&](auto t) { net_acc_zero<N::ACC_BASE, decltype(t)::value>(); });

  const int n_tiles = (int)((rows + 15) >> 4);
  int round = 0;
  // every wave of the workgroup makes the same trips (the barriers are the workgroup's): a slot past the last tile
  // computes on the last row with zero cotangents and stores nothing
  for (int base = (int)blockIdx.x * kNetSlots; base < n_tiles; base += (int)gridDim.x * kNetSlots, round ^= 1) {
    const int tile = base + slot;
    const bool valid = tile < n_tiles;
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = (MNF_NET_ABL & 1) ? false : valid && row < rows;
    const int64_t rowc = row < rows ? row : rows - 1;
    const float* xr = x + rowc * dim + 4 * q;
    const float* gyr = grad_y + rowc * dim + 4 * q;
    float* const gr = grad_x + rowc * dim + 4 * q;
    // rows: both waves the conditioning half and the cotangent of the transformed half; the s wave the transformed
    // half and grad_ld, the t wave the cotangent of the conditioning half
    f32x4 cnd[G], ga[G], oth[G];  // oth: act (s wave) / grad_y cond (t wave)
    float gl = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (MNF_NET_ABL & 1) {
        cnd[g] = f32x4{0.1f, 0.2f, -0.3f, 0.4f} * (float)(lane + tile);
        oth[g] = cnd[g];
        ga[g] = cnd[g];
        continue;
      }
      cnd[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
      ga[g] = (grad_y && live) ? *reinterpret_cast<const f32x4*>(gyr + act_off + 16 * g) * g_scale : zero4;
      if (net == 0) oth[g] = *reinterpret_cast<const f32x4*>(xr + act_off + 16 * g);
      else oth[g] = (grad_y && live) ? *reinterpret_cast<const f32x4*>(gyr + cond_off + 16 * g) * g_scale : zero4;
    }
    if (net == 0) gl = (grad_ld && live) ? grad_ld[rowc] * g_scale : 0.f;

    // LDS addresses: one register each, the operand's place as the instruction's offset
    int a_off = net * (N::OPS_NET * 512) + lane * 4, b_off = N::SPLIT_WORDS + net * (N::BT_NET * 16) + q * 4;
    // (mailbox of wave w at MAIL_OFF + w MAIL_WAVE: [box 1][trip parity 0: box 2, 64 maxima][parity 1: likewise];
    //  boxes hold one f32x4 per lane and 16-dim group, the maxima one word per lane)
    constexpr int TRIP = N::BOX_WORDS + 64;
    int m_off = N::MAIL_OFF + wave * N::MAIL_WAVE + lane * 4, p_off = N::MAIL_OFF + partner * N::MAIL_WAVE + lane * 4;
    int mm_off = N::MAIL_OFF + wave * N::MAIL_WAVE + 2 * N::BOX_WORDS + round * TRIP + lane;
    int pm_off = N::MAIL_OFF + partner * N::MAIL_WAVE + 2 * N::BOX_WORDS + round * TRIP + lane;
    asm volatile("" : "+v"(a_off), "+v"(b_off), "+v"(m_off), "+v"(p_off), "+v"(mm_off), "+v"(pm_off));
    const f16x8* A8 = reinterpret_cast<const f16x8*>(lds + a_off);  // + 64 * (2 op + part)
    const f32x4* B4 = reinterpret_cast<const f32x4*>(lds + b_off);  // + 4 * bias tile
    f32x4* const my1 = reinterpret_cast<f32x4*>(lds + m_off);       // box 1: + 64 g
    const f32x4* const pr1 = reinterpret_cast<const f32x4*>(lds + p_off);
    f32x4* const my2 = my1 + 64 * G + round * (TRIP / 4);           // box 2 of this trip: + 64 g
    const f32x4* const pr2 = pr1 + 64 * G + round * (TRIP / 4);
    float* const my_max = reinterpret_cast<float*>(lds + mm_off);
    const float* const pr_max = reinterpret_cast<const float*>(lds + pm_off);
    float mx = 0.f;
    auto pair_of = [&](const u32x2* v, int n) { return pair_operand(v[0], n > 1 ? v[n > 1 ? 1 : 0] : zero2); };
    // out tile m of one layer: main (+ bias) and correction sums of W [tile m] x B
    auto product = [&](int op, const f16x8& bh, const f16x8& bl, f32x4 init) {
      f32x4 mn = init, cr = zero4;
      split_mac(A8[64 * (2 * op)], A8[64 * (2 * op + 1)], bh, bl, mn, cr);
      return cr * kSplitInvScale + mn;
    };

    // ------------------------------------------------------------------ forward recompute of this wave's net
    u32x2 xh[G], xl[G], hh[3][NTN], hl[3][NTN];
    f32x4 d4[G];
#pragma unroll
    for (int g = 0; g < G; ++g) split_tile(cnd[g], xh[g], xl[g], mx);
    {
      f16x8 bh = pair_of(xh, G), bl = pair_of(xl, G);
#pragma unroll
      for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int m = 0; m < NTN; ++m) {
          const f32x4 p = product(N::op_fwd(l) + m, bh, bl, B4[4 * (l * NTN + m)]);
          split_tile(__builtin_elementwise_max(p, p * kLeakySlope), hh[l][m], hl[l][m], mx);
        }
        bh = pair_of(hh[l], NTN);
        bl = pair_of(hl[l], NTN);
      }
      f32x4 st[G], other[G];  // this net's raw output, the other net's
#pragma unroll
      for (int g = 0; g < G; ++g) st[g] = product(N::op_fwd(3) + g, bh, bl, B4[4 * (3 * NTN + g)]);
      if constexpr (INV) {
#pragma unroll
        for (int g = 0; g < G; ++g) my1[64 * g] = st[g];
        pair_meet(lds, my_flag, pr_flag, ++seq, lane);
#pragma unroll
        for (int g = 0; g < G; ++g) other[g] = pr1[64 * g];
      }

      // ---------------------------------------------------------------- output deltas, grad of the transformed half
      //   forward: y = e^s v + t          g_v = g e^s      g_s = g e^s v + g_ld      g_t = g
      //   inverse: y = (v - t) e^-s       g_v = g e^-s     g_s = -g y - g_ld         g_t = -g e^-s
      // (stored as soon as it exists: a tile the range verdict hands to the fp32 pass is overwritten by that pass)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 d;
        if (net == 0) {
          f32x4 gv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s = st[g][r], gy = ga[g][r], v = oth[g][r];
            const float e = exp6(INV ? -s : s);
            gv[r] = gy * e * g_unscale;
            if constexpr (INV) d[r] = live ? -gy * ((v - other[g][r]) * e) - gl : 0.f;
            else d[r] = live ? gy * e * v + gl : 0.f;
          }
          if (live) *reinterpret_cast<f32x4*>(gr + act_off + 16 * g) = gv;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (INV) d[r] = -ga[g][r] * exp6(-other[g][r]);
            else d[r] = ga[g][r];
          }
        }
        d4[g] = d;
      }
    }
    u32x2 d4h[G], d4l[G], dh[3][NTN], dl[3][NTN];  // dh[2] = delta 3 (pre-activation of h3), dh[0] = delta 1
#pragma unroll
    for (int g = 0; g < G; ++g) split_tile(d4[g], d4h[g], d4l[g], mx);

    // ------------------------------------------------------------------ the delta chain through the transposed weights
    f32x4 share[G];
    {
      f16x8 bh = pair_of(d4h, G), bl = pair_of(d4l, G);
#pragma unroll
      for (int l = 3; l >= 1; --l) {  // delta_l = W_l^T delta_{l+1} .* LeakyReLU'(h_l)   (dh[l - 1])
#pragma unroll
        for (int m = 0; m < NTN; ++m) {
          f32x4 d = product(N::op_tr(l) + m, bh, bl, zero4);
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r] = net_unit_active(hh[l - 1][m], hl[l - 1][m], r) ? d[r] : kLeakySlope * d[r];
          split_tile(d, dh[l - 1][m], dl[l - 1][m], mx);
        }
        bh = pair_of(dh[l - 1], NTN);
        bl = pair_of(dl[l - 1], NTN);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) share[g] = product(N::op_tr(0) + g, bh, bl, zero4);
    }
    // ------------------------------------------------------------------ the pair meets: grad x_cond, range verdict
    if (net == 0) {
#pragma unroll
      for (int g = 0; g < G; ++g) my2[64 * g] = share[g];
    }
    *my_max = mx;
    pair_meet(lds, my_flag, pr_flag, ++seq, lane);
    mx = __builtin_fmaxf(mx, *pr_max);
    if (net == 1) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 gx0 = (oth[g] + share[g] + pr2[64 * g]) * g_unscale;
        if (live) *reinterpret_cast<f32x4*>(gr + cond_off + 16 * g) = gx0;
      }
    }
    if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {
      // an operand left the split range: this tile's gradients come from the fp32 kernel (the caller runs it on the
      // listed tiles next); neither wave has accumulated anything of it
      if (net == 0 && lane == 0 && valid) {
        const int at = atomicAdd(cold_list, 1);
        if (at < cold_capacity) cold_list[1 + at] = tile;
      }
      continue;
    }
    if (grad_flat == nullptr || (MNF_NET_ABL & 2)) continue;
    // ------------------------------------------------------------------ weight gradients: rows on the K axis
    // T(v): the tile with rows along the registers: lane (unit = j, q) holds rows 4 q .. 4 q + 3, head and residual
    auto transpose = [&](const u32x2& hi, const u32x2& lo, f16x4& th, f16x4& tl) {
      const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, hi), ident, zero4, 0, 0, 0);
      const f32x4 ol = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, lo), ident_lo, zero4, 0, 0, 0);
      th = __builtin_convertvector(o, f16x4);
      tl = __builtin_convertvector(ol, f16x4);
    };
    auto delta_op = [&](const u32x2& hi, const u32x2& lo) -> f16x8 {  // A operand [head | residual]
      f16x4 th, tl;
      transpose(hi, lo, th, tl);
      return __builtin_shufflevector(th, tl, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // (one: the tile's structural-zero unit PAD_COL reads 1 -- NetShape::PAD)
    auto act_ops = [&](const u32x2& hi, const u32x2& lo, f16x8& a_hh, f16x8& a_ll, bool one) {  // B operands [h | h], [r | r]
      f16x4 th, tl;
      transpose(hi, lo, th, tl);
      if (one) {
        const f16x4 z = f16x4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        th = j == N::PAD_COL ? ones : th;
        tl = j == N::PAD_COL ? z : tl;
      }
      a_hh = __builtin_shufflevector(th, th, 0, 1, 2, 3, 4, 5, 6, 7);
      a_ll = __builtin_shufflevector(tl, tl, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const f16x8 ones8 = __builtin_shufflevector(ones, ones, 0, 1, 2, 3, 4, 5, 6, 7);
    // layer l: delta_{l+1} (out_tiles(l) tiles) x a_l (in_tiles(l) tiles); delta 4 = d4, a_0 = x
    net_static_for<4>([&](auto lc) {
      constexpr int l = 3 - decltype(lc)::value;
      constexpr int NO = N::out_tiles(l), NI = N::in_tiles(l);
      f16x8 a_hh[NI], a_ll[NI], d_hl[NO];
#pragma unroll
      for (int m = 0; m < NI; ++m) {
        if constexpr (l == 0) act_ops(xh[m], xl[m], a_hh[m], a_ll[m], false);
        else act_ops(hh[l - 1][m], hl[l - 1][m], a_hh[m], a_ll[m], N::PAD && m == NI - 1);
      }
#pragma unroll
      for (int m = 0; m < NO; ++m) {
        if constexpr (l == 3) d_hl[m] = delta_op(d4h[m], d4l[m]);
        else d_hl[m] = delta_op(dh[l][m], dl[l][m]);
      }
      net_static_for<NO>([&](auto mo_c) {
        constexpr int mo = decltype(mo_c)::value;
        if constexpr (N::own_bias_tiles(l)) net_acc_bias<N::ACC_BASE, N::db(l) + mo>(d_hl[mo], ones8);
        net_static_for<NI>([&](auto mi_c) {
          constexpr int mi = decltype(mi_c)::value;
          net_acc_outer<N::ACC_BASE, N::dw(l) + mo * NI + mi>(d_hl[mo], a_hh[mi], a_ll[mi]);
        });
      });
    });
  }

  // ------------------------------------------------------------------ flush: the four waves of a net add up in LDS
  if (grad_flat == nullptr) return;
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);  // [FLAT_FLOATS] in the flat vector's order; the image is no longer needed
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the last MFMAs' results, before the accumulators are read)
  const int lane_f = __lane_id(), n_f = lane_f & 15, q_f = lane_f >> 4;
  for (int w = 0; w < kNetSlots; ++w) {
    if (slot == w) {
      float* const rn = red + net * N::NET_FLOATS;
      net_static_for<4>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int NO = N::out_tiles(l), NI = N::in_tiles(l), IN = N::in_size(l), OUT = N::out_size(l);
        net_static_for<NO * NI>([&](auto tc) {
          constexpr int mo = decltype(tc)::value / NI, mi = decltype(tc)::value % NI;
          const f32x4 v = net_acc_read<N::ACC_BASE, N::dw(l) + mo * NI + mi>();  // lane (column n, q): delta units 4 q + r
          const int in = 16 * mi + n_f;
          const bool bias_col = !N::own_bias_tiles(l) && in == IN;  // (the ones column: NetShape::PAD)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int out = 16 * mo + 4 * q_f + r;
            if ((in < IN || bias_col) && out < OUT) {
              float* p = rn + (bias_col ? N::b_off(l) + out : N::w_off(l) + out * IN + in);
              *p = w == 0 ? v[r] : *p + v[r];
            }
          }
        });
        if constexpr (N::own_bias_tiles(l)) net_static_for<NO>([&](auto mc) {
          constexpr int mo = decltype(mc)::value;
          const f32x4 v = net_acc_read<N::ACC_BASE, N::db(l) + mo>();  // every column the same
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int out = 16 * mo + 4 * q_f + r;
            if (n_f == 0 && out < OUT) {
              float* p = rn + N::b_off(l) + out;
              *p = w == 0 ? v[r] : *p + v[r];
            }
          }
        });
      });
    }
    __syncthreads();
  }
  const int tid = wave * 64 + lane_f;
  if (partials) {  // two-stage flush: this workgroup's sums as one coalesced block, ahf_bwd_net_reduce_kernel adds them up
    float* dst = partials + (int64_t)blockIdx.x * N::FLAT_FLOATS;
    for (int i = tid; i < N::FLAT_FLOATS; i += kNetWaves * 64) dst[i] = red[i];
    return;
  }
  for (int i = tid; i < N::FLAT_FLOATS; i += kNetWaves * 64) atomicAdd(grad_flat + i, red[i] * g_unscale);
}

// The kernels proper: amdgpu_num_vgpr wants a literal, so one pair (inverse, forward) per shape, the compiler's share of
// the register file (NetShape::ACC_BASE) spelled out
typedef void (*NetKernel)(const float*, const float*, const float*, float*, float*, const uint32_t*, int64_t, int,
                          const float*, int32_t*, int, float*);
template <int H, int HID, bool INV>
struct NetKernelOf;
#define MNF_NET_KERNEL(HH, HD, INV, NAME, BASE)                                                                         \
  __global__ void __launch_bounds__(kNetWaves * 64, 1) __attribute__((amdgpu_num_vgpr(BASE / 2)))                       \
  NAME(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat, const uint32_t* image, \
       int64_t rows, int parity, const float* scale_dev, int32_t* cold_list, int cold_capacity, float* partials) {      \
    static_assert(BASE == NetShape<HH, HD>::ACC_BASE, "the compiler's registers end where the accumulators begin");     \
    ahf_bwd_net_body<HH, HD, INV>(x, grad_y, grad_ld, grad_x, grad_flat, image, rows, parity, scale_dev, cold_list,     \
                                  cold_capacity, partials);                                                            \
  }                                                                                                                     \
  template <>                                                                                                           \
  struct NetKernelOf<HH, HD, INV> {                                                                                     \
    static NetKernel get() { return NAME; }                                                                             \
  };
MNF_NET_KERNEL(32, 24, true, ahf_bwd_net_kernel_32_24_inv, 184)
MNF_NET_KERNEL(32, 24, false, ahf_bwd_net_kernel_32_24_fwd, 184)
MNF_NET_KERNEL(16, 24, true, ahf_bwd_net_kernel_16_24_inv, 200)
MNF_NET_KERNEL(16, 24, false, ahf_bwd_net_kernel_16_24_fwd, 200)
MNF_NET_KERNEL(32, 16, true, ahf_bwd_net_kernel_32_16_inv, 208)
MNF_NET_KERNEL(32, 16, false, ahf_bwd_net_kernel_32_16_fwd, 208)
MNF_NET_KERNEL(16, 16, true, ahf_bwd_net_kernel_16_16_inv, 224)
MNF_NET_KERNEL(16, 16, false, ahf_bwd_net_kernel_16_16_fwd, 224)
#undef MNF_NET_KERNEL

// Second stage of the flush: entry i of every workgroup's block belongs to parameter i (32 entries per workgroup x 8
// slices of the blocks, as ahf_bwd_reduce_kernel)
__global__ void __launch_bounds__(256) ahf_bwd_net_reduce_kernel(const float* __restrict__ partials, int n_blocks, int n_flat,
                                                                 float* __restrict__ grad_flat,
                                                                 const float* __restrict__ scale_dev,
                                                                 const int32_t* __restrict__ cold_list) {
  if (cold_list[0] < 0) return;  // the whole launch went to the fp32 pass: nothing was stored
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  const bool mine = i < n_flat;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (mine) {
    int b = slice;
    for (; b + 24 < n_blocks; b += 32) {
      s0 += partials[(int64_t)(b + 0) * n_flat + i];
      s1 += partials[(int64_t)(b + 8) * n_flat + i];
      s2 += partials[(int64_t)(b + 16) * n_flat + i];
      s3 += partials[(int64_t)(b + 24) * n_flat + i];
    }
    for (; b < n_blocks; b += 8) s0 += partials[(int64_t)b * n_flat + i];
  }
  part[slice][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && mine) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][e];
    grad_flat[i] += t * (1.0f / scale_dev[0]);
  }
}

// ---------------------------------------------------------------- host: index table of the per-net operand image
// split entries of operand (net, op), part, lane, element e at idx[(((2 (net OPS_NET + op) + part) 64 + lane) 4 +
// (e >> 1)) 2 + (e & 1)]: A[i = lane & 15][K slot 8 (lane >> 4) + e], K slot 8 kq + e = unit 16 (e >> 2) + 4 kq + (e & 3)
template <int H, int HID>
static int build_net_index(int32_t* idx) {
  using N = NetShape<H, HID>;
  int sizes[5] = {H, HID, HID, HID, H};
  NetDesc nd[2];
  const int64_t n0 = fill_net(nd[0], 5, sizes, 0);
  fill_net(nd[1], 5, sizes, n0);
  if (n0 != N::NET_FLOATS) return MNF_ERR_INVALID_ARG;
  for (int64_t i = 0; i < N::INDEX_INTS; ++i) idx[i] = -1;
  auto put = [&](int op, int lane, int e, int32_t src) {
    for (int part = 0; part < 2; ++part)
      idx[(((int64_t)(2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] = src | (part ? kSplitLoBit : 0);
  };
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < 4; ++l) {
      const int in_size = sizes[l], out_size = sizes[l + 1];
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4;
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * (e >> 2) + 4 * kq + (e & 3);
          // forward: A[out unit 16 m + i][in unit k]
          for (int m = 0; m < N::out_tiles(l); ++m) {
            const int out = 16 * m + i;
            if (out < out_size && k < in_size)
              put(net * N::OPS_NET + N::op_fwd(l) + m, lane, e, nd[net].w_off[l] + out * in_size + k);
          }
          // transposed: A[in unit 16 m + i][out unit k]
          for (int m = 0; m < N::in_tiles(l); ++m) {
            const int in = 16 * m + i;
            if (in < in_size && k < out_size)
              put(net * N::OPS_NET + N::op_tr(l) + m, lane, e, nd[net].w_off[l] + k * in_size + in);
          }
        }
      }
      for (int m = 0; m < N::out_tiles(l); ++m)
        for (int u = 0; u < 16; ++u)
          if (16 * m + u < out_size)
            idx[2 * (int64_t)N::SPLIT_WORDS + (net * N::BT_NET + l * N::NTN + m) * 16 + u] = nd[net].b_off[l] + 16 * m + u;
    }
  return MNF_OK;
}

static int64_t net_blocks(int64_t rows) {
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + kNetSlots - 1) / kNetSlots;
  const int cus = device_cus(current_device());
  return blocks > cus ? cus : blocks;  // one persistent workgroup per CU
}

template <int H, int HID>
static int launch_bwd_net(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                          const uint32_t* image, int64_t rows, int parity, int inverse, const float* scale_dev,
                          int32_t* cold_list, int cold_capacity, float* workspace, int64_t workspace_floats,
                          hipStream_t stream) {
  using N = NetShape<H, HID>;
  static constexpr size_t lds_bytes = N::LDS_WORDS * sizeof(uint32_t);
  const NetKernel inv_kernel = NetKernelOf<H, HID, true>::get(), fwd_kernel = NetKernelOf<H, HID, false>::get();
  static DeviceMemo memo;
  const int ok = memo.get([&](int) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes) == hipSuccess
               ? 1
               : -1;
  });
  if (ok <= 0) return MNF_ERR_UNSUPPORTED;
  const int64_t blocks = net_blocks(rows);
  float* partials = (grad_flat && workspace && workspace_floats >= blocks * N::FLAT_FLOATS) ? workspace : nullptr;
  hipLaunchKernelGGL(inverse ? inv_kernel : fwd_kernel, dim3((unsigned)blocks), dim3(kNetWaves * 64), lds_bytes, stream, x,
                     grad_y, grad_ld, grad_x, grad_flat, image, rows, parity, scale_dev, cold_list, cold_capacity, partials);
  if (partials)
    hipLaunchKernelGGL(ahf_bwd_net_reduce_kernel, dim3((N::FLAT_FLOATS + 31) / 32), dim3(256), 0, stream, partials,
                       (int)blocks, (int)N::FLAT_FLOATS, grad_flat, scale_dev, cold_list);
  return check_launch();
}

// shapes: those of mnf_ahf_bwd_split.hip
#define MNF_AHF_BWD_NET_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16)

bool bwd_net_mode() {
  static const bool v = [] {
    const char* e = getenv("MNF_AHF_BWD_SPLIT");  // "net": this file's kernels instead of mnf_ahf_bwd_split.hip's own
    return e && e[0] == 'n';
  }();
  return v;
}

int bwd_net_layout(int dim, int hid, int64_t* n_split_words, int64_t* n_plain_words) {
#define X(HH, HD)                                             \
  if (dim == 2 * HH && hid == HD) {                           \
    *n_split_words = NetShape<HH, HD>::SPLIT_WORDS;           \
    *n_plain_words = NetShape<HH, HD>::PLAIN_WORDS;           \
    return MNF_OK;                                            \
  }
  MNF_AHF_BWD_NET_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int bwd_net_index(int dim, int hid, int32_t* idx_host) {
#define X(HH, HD) \
  if (dim == 2 * HH && hid == HD) return build_net_index<HH, HD>(idx_host);
  MNF_AHF_BWD_NET_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int64_t bwd_net_workspace(int64_t rows, int dim, int hid) {
#define X(HH, HD) \
  if (dim == 2 * HH && hid == HD) return net_blocks(rows) * NetShape<HH, HD>::FLAT_FLOATS;
  MNF_AHF_BWD_NET_SHAPES(X)
#undef X
  return 0;
}

int bwd_net_launch(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                   const void* image, int64_t rows, int dim, int hid, int parity, int inverse, const float* scale_dev,
                   int32_t* cold_list, int cold_capacity, float* workspace, int64_t workspace_floats, hipStream_t stream) {
#define X(HH, HD)                                                                                                   \
  if (dim == 2 * HH && hid == HD)                                                                                   \
    return launch_bwd_net<HH, HD>(x, grad_y, grad_ld, grad_x, grad_flat, static_cast<const uint32_t*>(image), rows, \
                                  parity, inverse, scale_dev, cold_list, cold_capacity, workspace, workspace_floats, stream);
  MNF_AHF_BWD_NET_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf
