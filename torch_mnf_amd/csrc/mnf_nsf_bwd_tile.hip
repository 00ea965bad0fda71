// NSF_CL gradients with the WHOLE conditioner on the f16 matrix pipe (split arithmetic, mnf_split.h), 16 rows per wave.
//
// mnf_nsf_bwd_rows.hip gives a lane one (row, element) and evaluates the 8 -> 16 (3K-1) output layer of the conditioner
// on the vector units in both directions (the parameters: 8 fma x (3K-1) per element, W4^T g_p the same again), which
// is a third of that kernel's vector instructions, and its DPP hidden layers tie it to d = 32, hidden width <= 8.
// Here a wave owns a 16-row tile in the forward kernel's layout (mnf_nsf_mfma.hip): lane (j = lane & 15, q = lane >> 4)
// holds, as float4s, the elements {16 g + 4 q + r} of both halves of row j, and
//
//   * the hidden layers and the output layer run as split MFMAs exactly like the forward kernel: the tile for "slot"
//     s = 4 g + r and parameter block kb puts parameter 4 kb + r' of element 16 g + 4 q + r of row j into register r'
//     of lane (j, q) -- after ceil((3K-1)/4) tiles a lane owns all 3K-1 raw parameters of one of ITS OWN elements and
//     differentiates the spline on it in registers (mnf_nsf_spline_grad.h, the code of the row kernel);
//   * W4^T g_p is the same product with the operand image transposed: the lane's g_p registers ARE the B operand
//     (K = the 16 (element, parameter) entries of the tile, N = rows), 3 MFMAs per tile into ONE accumulator pair;
//   * weight gradients are sums over rows: dW[m][n] += sum_rows delta[m][row] a[n][row] wants rows along a lane's
//     registers where the chain has them along the lanes, so a tile is transposed by one MFMA against the identity
//     (exact: every product is x * 1; head and residual separately, the residual scaled back by 2^-11 on the way) and
//     a weight-gradient tile is two K = 32 products [d_hi | d_lo] x [a_hi | a_hi], [a_lo | a_lo] into ONE fp32
//     accumulator (the scheme of mnf_ahf_bwd_split.hip).  The accumulators live in hand-assigned accumulator
//     registers (mnf_agpr.h's rules; check_agpr.py guards the build): the slot loop is a run-time loop and the tile an
//     MFMA adds to is selected by a scalar branch, which compiler-managed loop-carried values would not survive
//     without copies.
//
// Two launches per layer, one per half-step, each over all tiles with every wave on its own: stage 0 differentiates the
// SECOND half-step (its net's sums, the cotangent of the half it transforms, and the first half-step's output half's
// cotangent, which it parks in grad_x), stage 1 the FIRST (reading both cotangents back from grad_x).  The second
// net's conditioner input -- the half the first half-step produced -- is a column block of the layer's OUTPUT, which the
// forward pass has written and autograd keeps: the caller passes it (`y`) and nothing of the first half-step is
// recomputed to get it (the row kernel spent a fifth of its instructions there).  A wave therefore holds ONE net's
// sums and LDS one net's operand image (d = 64 fits), and there is no hand-over between waves: round 5's first form
// of this kernel gave a tile to a pair of waves (one net each, a mailbox and a workgroup barrier per tile) and the
// two waves of a SIMD then ran in lockstep -- both in their MFMA bursts, both in the spline -- with the vector units
// 68 % busy; independent waves drift apart and cover each other's bursts.
//
// Cotangents of a mean over 2^20 rows are ~1e-6, below f16's normal range: the caller passes a power of two
// (`scale_dev`) that brings max |cotangent| near 1; grad_x and the parameter gradients are scaled back on the way out.
// Range: a tile whose forward operands (x, h1, h2, h3 of either net) reach kSplitLimit is not computed at all -- stage 0
// runs both nets' hidden layers at the top of a tile, before anything is accumulated, and flags the tile for stage 1
// -- and goes to `cold`; the caller
// runs mnf_nsf_cl_bwd_tile_fixup (the generic kernel over the listed tiles) next.  Gradient operands (g_p, the deltas)
// are only known while the sums are being formed: one beyond f16's range poisons the launch (cold[1] = 1), the
// reduction kernel then adds nothing and the fix-up pass recomputes every row.  The sums leave the kernel as one block
// per workgroup in parameter order and nsf_tile_reduce_kernel adds the blocks up in a fixed order: no float atomics,
// results repeat bit for bit.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "mnf_agpr.h"
#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_nsf_spline_grad.h"
#include "mnf_split.h"

namespace mnf {
namespace {

using namespace nsfgrad;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// H: padded half width (16 or 32; the real half may be narrower in whole float4 groups), NH: hidden width the nets run
// at (8 or 16; narrower layers get structural-zero units), K: bins
template <int H, int NH, int K>
struct NtShape {
  static_assert(H % 16 == 0 && (NH == 8 || NH == 16) && K >= 2, "unsupported NSF_CL shape");
  static constexpr int G = H / 16;        // float4 groups per half row
  static constexpr int S = H / 4;         // slots = elements per lane per half
  static constexpr int P = 3 * K - 1;
  static constexpr int NB = (P + 3) / 4;  // parameter blocks (tiles) per slot
  static constexpr bool BIASCOL = NH <= 8;  // the activation tiles have a spare column: a column of ones sums the bias
  // operands per net: [lane][hi0 hi1 lo0 lo1] -- one ds_read_b128 per lane brings head and residual
  static constexpr int OP_F1 = 0;                  // G: layer 1, cond group g
  static constexpr int OP_F2 = G, OP_F3 = G + 1;   // hidden layers 2, 3
  static constexpr int OP_F4 = G + 2;              // S * NB: output layer, slot s, block kb
  static constexpr int OP_T4 = OP_F4 + S * NB;     // S * NB: the same transposed
  static constexpr int OP_T3 = OP_T4 + S * NB, OP_T2 = OP_T3 + 1;
  static constexpr int OP_T1 = OP_T2 + 1;          // G: layer 1 transposed, cond group g
  static constexpr int N_OPS = OP_T1 + G;
  static constexpr int SPLIT_WORDS_NET = N_OPS * 256;
  static constexpr int BIAS_TILES = 3 + S * NB;    // b1, b2, b3, b4[s][kb]
  static constexpr int PLAIN_WORDS_NET = BIAS_TILES * 16;
  // image: [f1 operands][f2 operands][f1 biases][f2 biases][tail]
  static constexpr int SPLIT_WORDS = 2 * SPLIT_WORDS_NET;
  static constexpr int PLAIN_WORDS = 2 * PLAIN_WORDS_NET;
  static constexpr int IMAGE_WORDS = SPLIT_WORDS + PLAIN_WORDS + kSplitTailWords;
  // accumulator tiles per net
  static constexpr int T_W4 = 0;                   // S * NB: [entry (q, r) of tile (s, kb)][unit | ones]
  static constexpr int T_W3 = S * NB, T_W2 = T_W3 + 1;
  static constexpr int T_W1 = T_W2 + 1;            // G: [unit][cond feature of group g]
  // bias tiles: [A operand's feature][one-hot column]: columns 0, 1, 2 = b1, b2, b3 (rows = units), and without the
  // ones column, column 3 + t = b4 of output tile t (rows = its 16 entries)
  static constexpr int T_B = T_W1 + G;
  static constexpr int N_BCOLS = 3 + (BIASCOL ? 0 : S * NB);
  static constexpr int N_BT = (N_BCOLS + 15) / 16;
  static constexpr int TILES = T_B + N_BT;
  static constexpr int RED_FLOATS = TILES * 256;
  // one wave per SIMD when the sums take more than 112 of the wave's 256 registers at two.  (At 116 -- hidden 16, K = 8,
  // the reference's test shape -- two waves spilled 88-95 registers per tile: 2.27 ms forward + backward per layer at
  // 2^20 x 32 against 2.01 with one wave and none; at 112 -- hidden 8, K = 8, 26 spilled -- two waves win, 996 us per
  // gradient pass against 1,165.)
  static constexpr int WAVES_PER_SIMD = (4 * TILES <= 112) ? 2 : 1;
  static constexpr int WAVES = 4 * WAVES_PER_SIMD;               // per workgroup (one workgroup per CU)
  static constexpr bool ACC_AG = WAVES_PER_SIMD == 1;            // the sums in accumulator registers / at the top of the vector file
  static constexpr int ACC_BASE = 256 - 4 * TILES;  // (of the accumulator file / of the vector file)
  // LDS: [the stage's net: operands | biases][stage 0: the other net's hidden layers: G + 2 operands | 3 bias tiles]
  static constexpr int OTHER_OPS = G + 2;
  static constexpr int LDS_NET = SPLIT_WORDS_NET + PLAIN_WORDS_NET;
  static constexpr int LDS_IMG = LDS_NET + OTHER_OPS * 256 + 3 * 16;
  static constexpr int LDS_WORDS = LDS_IMG > RED_FLOATS ? LDS_IMG : RED_FLOATS;
  static_assert(4 * TILES <= 256, "accumulator registers");
  static_assert(LDS_WORDS * 4 <= 160 * 1024, "LDS");
};

template <bool AG, int R>
struct AccReg;
template <typename Sh, int T>
using AccT = AccReg<Sh::ACC_AG, Sh::ACC_BASE + 4 * T>;  // accumulator tile T of a shape

template <int I, int N, typename F>
__device__ __forceinline__ void nt_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    nt_static_for<I + 1, N>(f);
  }
}

// ---- accumulators: tile t = four hand-assigned registers, touched only by the statements below (see
// mnf_ahf_bwd_split.hip: back to back MFMAs on the same accumulator need no wait states; a vector instruction's result
// needs 2 before an MFMA may read it and hipcc cannot see into these, so every statement starts with its own s_nop 1).
// Which registers depends on the occupancy the shape runs at (NtShape::WAVES_PER_SIMD):
//   two waves per SIMD (256 registers per wave): VECTOR registers v[BASE + 4 t ..] at the top of the file, BASE = 256 -
//     4 TILES; the kernel is declared amdgpu_num_vgpr(BASE / 2) (hipcc doubles the number), which keeps the compiler
//     below BASE, and the clobber of v255 makes the register count 256 (the scheme of round 4's one-net-per-wave
//     AffineHalfFlow kernel: as soon as an asm statement names an ACCUMULATOR register hipcc splits the wave's budget
//     128 + 128 -- too few vector registers for the spline's derivative); check_vgpr_top.py guards the build;
//   one wave per SIMD (512): the TOP of the accumulator half, a[BASE + 4 t ..], BASE = 256 - 4 TILES (mnf_agpr.h's
//     rules; the compiler parks values of its own that overflow the vector registers in accumulator registers from a0
//     upwards -- a clobber list does not keep it from doing so -- and check_agpr.py is told how far up it may go).
#define MNF_NT_ACC(AGV, RF, ZERO, READ)                                                                                 \
  template <int R>                                                                                                      \
  struct AccReg<AGV, R> {                                                                                               \
    static __device__ __forceinline__ void outer32(const f16x8& d_hl, const f16x8& a_hh, const f16x8& a_ll) {            \
      asm volatile("s_nop 1\n\t"                                                                                        \
                   "v_mfma_f32_16x16x32_f16 " RF "[%0:%1], %2, %3, " RF "[%0:%1]\n\t"                                    \
                   "v_mfma_f32_16x16x32_f16 " RF "[%0:%1], %2, %4, " RF "[%0:%1]" ::"n"(R),                              \
                   "n"(R + 3), "v"(d_hl), "v"(a_hh), "v"(a_ll));                                                        \
    }                                                                                                                   \
    static __device__ __forceinline__ void one32(const f16x8& d_hl, const f16x8& b) {                                    \
      asm volatile("s_nop 1\n\t"                                                                                        \
                   "v_mfma_f32_16x16x32_f16 " RF "[%0:%1], %2, %3, " RF "[%0:%1]" ::"n"(R),                              \
                   "n"(R + 3), "v"(d_hl), "v"(b));                                                                      \
    }                                                                                                                   \
    static __device__ __forceinline__ void zero() {                                                                     \
      asm volatile(ZERO " " RF "[%0], 0\n\t" ZERO " " RF "[%1], 0\n\t" ZERO " " RF "[%2], 0\n\t" ZERO " " RF "[%3], 0" ::"n"(R), \
                   "n"(R + 1), "n"(R + 2), "n"(R + 3));                                                                 \
    }                                                                                                                   \
    static __device__ __forceinline__ f32x4 read() {                                                                    \
      f32x4 v;                                                                                                          \
      asm volatile(READ " %0, " RF "[%4]\n\t" READ " %1, " RF "[%5]\n\t" READ " %2, " RF "[%6]\n\t" READ " %3, " RF "[%7]" \
                   : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])                                                     \
                   : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));                                                       \
      return v;                                                                                                         \
    }                                                                                                                   \
  };
MNF_NT_ACC(true, "a", "v_accvgpr_write_b32", "v_accvgpr_read_b32")
MNF_NT_ACC(false, "v", "v_mov_b32", "v_mov_b32")
#undef MNF_NT_ACC
// a0 .. a255 are part of the kernel's register allocation (the one-wave-per-SIMD shapes)
#define MNF_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
template <bool AG>
__device__ __forceinline__ void reserve_acc() {
  if constexpr (AG)
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MNF_A10(1), MNF_A10(2), MNF_A10(3),
                 MNF_A10(4), MNF_A10(5), MNF_A10(6), MNF_A10(7), MNF_A10(8), MNF_A10(9), MNF_A10(10), MNF_A10(11),
                 MNF_A10(12), MNF_A10(13), MNF_A10(14), MNF_A10(15), MNF_A10(16), MNF_A10(17), MNF_A10(18), MNF_A10(19),
                 MNF_A10(20), MNF_A10(21), MNF_A10(22), MNF_A10(23), MNF_A10(24), "a250", "a251", "a252", "a253", "a254",
                 "a255");
  else
    asm volatile("" ::: "v255");
}
#undef MNF_A10

__device__ __forceinline__ f32x4 mfma_x16(const f16x4& a, const f16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16x4 as_f16x4(const u32x2& v) { return __builtin_bit_cast(f16x4, v); }

// LeakyReLU'(pre-activation) read off the split form of the activation (mnf_ahf_bwd_split.hip)
__device__ __forceinline__ bool unit_active(const u32x2& hi, const u32x2& lo, int r) {
  const uint32_t key = __builtin_amdgcn_perm(hi[r >> 1], lo[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
  return (int32_t)key > 0;
}

struct NtArgs {
  const float* x;
  const float* y;  // the layer's output (the forward pass's)
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  const uint32_t* image;
  float* grad_flat;
  const int32_t* flush;  // per flat parameter: its element in the workgroup's reduced sums
  float* partials;       // [workgroup][n_params]
  const float* scale_dev;
  int32_t* cold;         // [0] count (-1: weights out of range), [1] poison, [2 ..] tiles, [2 + capacity ..] per-tile flags
  int cold_capacity;
  int64_t rows;
  float T;
  int hr;        // real half width (a multiple of 4, <= H)
  int n_params;  // both nets
};

// everything a wave needs to run one net
template <int H, int NH, int K>
struct NetView {
  using Sh = NtShape<H, NH, K>;
  const u32x4* ops;   // + op * 64 (already offset by the lane)
  const f32x4* bias;  // + 4 * tile (already offset by 4 q floats)
  __device__ __forceinline__ void op(int o, u32x2& ah, u32x2& al) const {
    const u32x4 w = ops[o * 64];
    ah = u32x2{w[0], w[1]};
    al = u32x2{w[2], w[3]};
  }
};

// main += Ah Bh ; corr += Ah Bl + Al Bh  (K = 16 products)
__device__ __forceinline__ void mac16(const u32x2& ah, const u32x2& al, const u32x2& bh, const u32x2& bl, f32x4& mn,
                                      f32x4& cr) {
  mn = mfma_x16(as_f16x4(ah), as_f16x4(bh), mn);
  cr = mfma_x16(as_f16x4(ah), as_f16x4(bl), cr);
  cr = mfma_x16(as_f16x4(al), as_f16x4(bh), cr);
}

// the three hidden layers on `cond`: split activations of every layer (hh[l], hl[l]) and of the input (xh, xl)
template <int H, int NH, int K>
__device__ __forceinline__ void hidden_forward(const NetView<H, NH, K>& nv, const f32x4 (&cond)[H / 16],
                                               u32x2 (&xh)[H / 16], u32x2 (&xl)[H / 16], u32x2 (&hh)[3], u32x2 (&hl)[3],
                                               float& mx) {
  using Sh = NtShape<H, NH, K>;
  constexpr int G = Sh::G;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G; ++g) split_tile(cond[g], xh[g], xl[g], mx);
  f32x4 mn = nv.bias[0], cr = zero4;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    u32x2 ah, al;
    nv.op(Sh::OP_F1 + g, ah, al);
    mac16(ah, al, xh[g], xl[g], mn, cr);
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const f32x4 p = cr * kSplitInvScale + mn;
    split_tile(__builtin_elementwise_max(p, p * kLeakySlope), hh[l], hl[l], mx);
    if (l < 2) {
      u32x2 ah, al;
      nv.op(Sh::OP_F2 + l, ah, al);
      mn = nv.bias[4 * (1 + l)];
      cr = zero4;
      mac16(ah, al, hh[l], hl[l], mn, cr);
    }
  }
}

// the 3K-1 raw spline parameters of this lane's element of slot s
template <int H, int NH, int K>
__device__ __forceinline__ void slot_params(const NetView<H, NH, K>& nv, int s, const u32x2& h3h, const u32x2& h3l,
                                            float (&p)[4 * NtShape<H, NH, K>::NB]) {
  using Sh = NtShape<H, NH, K>;
  constexpr int NB = Sh::NB;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // the slot's operands and bias tiles are requested together and waited for ONCE (left alone hipcc puts each read in
  // front of its MFMAs: six exposed LDS latencies per slot with only one other wave on the SIMD to cover them)
  f32x4 prm[NB], prc[NB];
  u32x2 ah[NB], al[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    nv.op(Sh::OP_F4 + s * NB + kb, ah[kb], al[kb]);
    prm[kb] = nv.bias[4 * (3 + s * NB + kb)];
    prc[kb] = zero4;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) mac16(ah[kb], al[kb], h3h, h3l, prm[kb], prc[kb]);
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const f32x4 v = prc[kb] * kSplitInvScale + prm[kb];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[4 * kb + r] = v[r];
  }
}

// One half-step backwards.  cond: the conditioning half; val: the transformed half BEFORE the spline;
// g_val: cotangent of that half after the spline (becomes the cotangent of val); g_cond += the net's share.
// Returns false -- before anything has been accumulated -- when a forward operand is out of the split range.
template <int H, int NH, int K, bool INV>
__device__ __forceinline__ bool half_backward(const NetView<H, NH, K>& nv, int lane, const f32x4 (&cond)[H / 16],
                                              const f32x4 (&val)[H / 16], float g_ld, f32x4 (&g_val)[H / 16],
                                              f32x4 (&g_cond)[H / 16], float T, float wmax_seed, float& mx_grad) {
  using Sh = NtShape<H, NH, K>;
  constexpr int G = Sh::G, P = Sh::P, NB = Sh::NB;
  const int j = lane & 15, q = lane >> 4;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x2 xh[G], xl[G], hh[3], hl[3];
  float mx = wmax_seed;

  hidden_forward<H, NH, K>(nv, cond, xh, xl, hh, hl, mx);
  if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) return false;

  // identity operands of the transposing MFMA: B[k = 4 q + e][n = j] = (k == n), and the same times 2^-11
  f16x4 ident, ident_lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ident[e] = (_Float16)((4 * q + e == j) ? 1.0f : 0.0f);
    ident_lo[e] = (_Float16)((4 * q + e == j) ? kSplitInvScale : 0.0f);
  }
  // T(v): the tile with rows along the registers: lane (feature = j, q) holds rows 4 q .. 4 q + 3, head and residual
  auto transpose = [&](const u32x2& hi, const u32x2& lo, f16x4& th, f16x4& tl) {
    const f32x4 o = mfma_x16(as_f16x4(hi), ident, zero4);
    const f32x4 ol = mfma_x16(as_f16x4(lo), ident_lo, zero4);  // (lo 2^-11: the residual itself)
    th = __builtin_convertvector(o, f16x4);
    tl = __builtin_convertvector(ol, f16x4);
  };
  auto delta_op = [&](const u32x2& hi, const u32x2& lo) -> f16x8 {  // A operand [head | residual]
    f16x4 th, tl;
    transpose(hi, lo, th, tl);
    return __builtin_shufflevector(th, tl, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // an activation tile as the two B operands [head | head], [residual | residual]; ONES: column 8 is a column of ones
  // (its sums are the bias gradients)
  const f16x4 ones4 = {(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
  const f16x4 zeros4 = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
  auto act_ops = [&](const u32x2& hi, const u32x2& lo, bool ones_col, f16x8& a_hh, f16x8& a_ll) {
    f16x4 th, tl;
    transpose(hi, lo, th, tl);
    if (ones_col) {
      th = j == 8 ? ones4 : th;
      tl = j == 8 ? zeros4 : tl;
    }
    a_hh = __builtin_shufflevector(th, th, 0, 1, 2, 3, 4, 5, 6, 7);
    a_ll = __builtin_shufflevector(tl, tl, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // B operand whose column c is all ones (the sums of a delta tile over the rows land in column c)
  auto onehot = [&](int c) -> f16x8 {
    const f16x4 v = j == c ? ones4 : zeros4;
    return __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  f16x8 h3_hh, h3_ll;
  act_ops(hh[2], hl[2], Sh::BIASCOL, h3_hh, h3_ll);
  f32x4 y_mn = zero4, y_cr = zero4;

  // ---- the slots: parameters, the spline's derivative, W4^T g_p, dW4
#pragma unroll
  for (int g = 0; g < G; ++g) {
    f32x4 v4 = val[g], go4 = g_val[g];
#pragma nounroll
    for (int r = 0; r < 4; ++r) {
      const int s = 4 * g + r;
      float p[4 * NB];
      slot_params<H, NH, K>(nv, s, hh[2], hl[2], p);
      float pp[P], g_p[P];
#pragma unroll
      for (int i = 0; i < P; ++i) pp[i] = p[i];
      float g_v;
      rqs_grad<K, INV>(v4[0], T, pp, go4[0], g_ld, g_v, g_p);
      v4 = f32x4{v4[1], v4[2], v4[3], v4[0]};
      go4 = f32x4{go4[1], go4[2], go4[3], g_v};
      f16x8 d_hl[NB];
      u32x2 t4h[NB], t4l[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) nv.op(Sh::OP_T4 + s * NB + kb, t4h[kb], t4l[kb]);  // (requested together: see slot_params)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        f32x4 gt;
#pragma unroll
        for (int i = 0; i < 4; ++i) gt[i] = 4 * kb + i < P ? g_p[4 * kb + i < P ? 4 * kb + i : 0] : 0.f;
        u32x2 gh, gl;
        split_tile(gt, gh, gl, mx_grad);
        mac16(t4h[kb], t4l[kb], gh, gl, y_mn, y_cr);
        d_hl[kb] = delta_op(gh, gl);
      }
      // the slot's accumulators: a scalar branch on r (g is unrolled)
      auto outer_slot = [&](auto rc) {
        constexpr int sc = decltype(rc)::value;
        nt_static_for<0, NB>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          constexpr int t = sc * NB + kb;
          AccT<Sh, Sh::T_W4 + t>::outer32(d_hl[kb], h3_hh, h3_ll);
          if constexpr (!Sh::BIASCOL) AccT<Sh, Sh::T_B + (3 + t) / 16>::one32(d_hl[kb], onehot((3 + t) & 15));
        });
      };
      nt_static_for<0, G>([&](auto gc) {
        constexpr int gg = decltype(gc)::value;
        if (gg == g) {
          if (r == 0) outer_slot(std::integral_constant<int, 4 * gg + 0>{});
          else if (r == 1) outer_slot(std::integral_constant<int, 4 * gg + 1>{});
          else if (r == 2) outer_slot(std::integral_constant<int, 4 * gg + 2>{});
          else outer_slot(std::integral_constant<int, 4 * gg + 3>{});
        }
      });
    }
    g_val[g] = go4;
  }

  // ---- the hidden layers backwards
  auto masked_split = [&](const f32x4& mn, const f32x4& cr, const u32x2& ah, const u32x2& al, u32x2& dh, u32x2& dl) {
    f32x4 d = cr * kSplitInvScale + mn;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = unit_active(ah, al, r) ? d[r] : kLeakySlope * d[r];
    split_tile(d, dh, dl, mx_grad);
  };
  u32x2 dh[3], dl[3];  // dh[2] = delta 3 (pre-activation of h3), dh[0] = delta 1
  masked_split(y_mn, y_cr, hh[2], hl[2], dh[2], dl[2]);
  {
    f16x8 a_hh, a_ll;
    // dW3 (h2 -> h3): delta 3 x h2
    f16x8 d = delta_op(dh[2], dl[2]);
    act_ops(hh[1], hl[1], Sh::BIASCOL, a_hh, a_ll);
    AccT<Sh, Sh::T_W3>::outer32(d, a_hh, a_ll);
    if constexpr (!Sh::BIASCOL) AccT<Sh, Sh::T_B>::one32(d, onehot(2));
    // delta 2 = W3^T delta 3 .* LeakyReLU'(h2)
    u32x2 ah, al;
    f32x4 mn = zero4, cr = zero4;
    nv.op(Sh::OP_T3, ah, al);
    mac16(ah, al, dh[2], dl[2], mn, cr);
    masked_split(mn, cr, hh[1], hl[1], dh[1], dl[1]);
    // dW2 (h1 -> h2): delta 2 x h1
    d = delta_op(dh[1], dl[1]);
    act_ops(hh[0], hl[0], Sh::BIASCOL, a_hh, a_ll);
    AccT<Sh, Sh::T_W2>::outer32(d, a_hh, a_ll);
    if constexpr (!Sh::BIASCOL) AccT<Sh, Sh::T_B>::one32(d, onehot(1));
    // delta 1 = W2^T delta 2 .* LeakyReLU'(h1)
    mn = cr = zero4;
    nv.op(Sh::OP_T2, ah, al);
    mac16(ah, al, dh[1], dl[1], mn, cr);
    masked_split(mn, cr, hh[0], hl[0], dh[0], dl[0]);
    // dW1: delta 1 x cond, db1
    d = delta_op(dh[0], dl[0]);
    AccT<Sh, Sh::T_B>::one32(d, onehot(0));
    nt_static_for<0, G>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      act_ops(xh[g], xl[g], false, a_hh, a_ll);
      AccT<Sh, Sh::T_W1 + g>::outer32(d, a_hh, a_ll);
    });
    // the conditioning half's cotangent += W1^T delta 1
#pragma unroll
    for (int g = 0; g < G; ++g) {
      mn = cr = zero4;
      nv.op(Sh::OP_T1 + g, ah, al);
      mac16(ah, al, dh[0], dl[0], mn, cr);
      g_cond[g] += cr * kSplitInvScale + mn;
    }
  }
  return true;
}

// ST = 0: the second half-step (forward: f2, inverse: f1), cotangents from grad_y; ST = 1: the first, cotangents from
// where stage 0 left them in grad_x
template <int H, int NH, int K, bool INV, int ST>
__device__ __forceinline__ void nsf_bwd_tile_body(const NtArgs& a) {
  using Sh = NtShape<H, NH, K>;
  constexpr int G = Sh::G, WAVES = Sh::WAVES;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  // weights beyond the split range (max |w|, written behind the image by the pack kernel): the fix-up pass does it all
  const float wmax = __builtin_bit_cast(float, a.image[Sh::SPLIT_WORDS + Sh::PLAIN_WORDS]);
  if (!(wmax <= kSplitWeightLimit)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.cold[0] = -1;
    return;
  }
  // forward:  up1 = S(up0; f1(lo0)),  lo1 = S(lo0; f2(up1))       -> reverse: f2's step, then f1's
  // inverse:  lo1 = S^-1(lo0; f2(up0)), up1 = S^-1(up0; f1(lo1))  -> reverse: f1's step, then f2's
  constexpr int net_first = INV ? 1 : 0;                     // the first half-step's net (f1 = 0, f2 = 1)
  constexpr int my_net = ST ? net_first : 1 - net_first, other_net = 1 - my_net;
  {
    auto copy = [&](int dst_word, int src_word, int n_words) {
      const uint4* src = reinterpret_cast<const uint4*>(a.image + src_word);
      uint4* dst = reinterpret_cast<uint4*>(lds + dst_word);
      for (int i = threadIdx.x; i < n_words / 4; i += blockDim.x) dst[i] = src[i];
    };
    copy(0, my_net * Sh::SPLIT_WORDS_NET, Sh::SPLIT_WORDS_NET);
    copy(Sh::SPLIT_WORDS_NET, Sh::SPLIT_WORDS + my_net * Sh::PLAIN_WORDS_NET, Sh::PLAIN_WORDS_NET);
    if (ST == 0) {
      copy(Sh::LDS_NET, other_net * Sh::SPLIT_WORDS_NET, Sh::OTHER_OPS * 256);
      copy(Sh::LDS_NET + Sh::OTHER_OPS * 256, Sh::SPLIT_WORDS + other_net * Sh::PLAIN_WORDS_NET, 3 * 16);
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float g_scale = a.scale_dev[0], g_unscale = 1.0f / g_scale;  // a power of two: both exact

  reserve_acc<Sh::ACC_AG>();
  nt_static_for<0, Sh::TILES>([&](auto t) { AccT<Sh, decltype(t)::value>::zero(); });

  const int hr = a.hr, dim = 2 * hr;
  // columns of the half the first half-step transforms (col_first) and of the other one
  const int col_first = INV ? 0 : hr, col_second = hr - col_first;
  const int col_val = ST ? col_first : col_second, col_cond = hr - col_val;  // this stage's value half / conditioner half
  NetView<H, NH, K> nv, nv_other;
  {
    int o_off = lane * 4, b_off = Sh::SPLIT_WORDS_NET + q * 4;
    asm volatile("" : "+v"(o_off), "+v"(b_off));  // keep the operand reads inside the tile loop
    nv.ops = reinterpret_cast<const u32x4*>(lds + o_off);
    nv.bias = reinterpret_cast<const f32x4*>(lds + b_off);
    nv_other.ops = reinterpret_cast<const u32x4*>(lds + (o_off + Sh::LDS_NET));
    nv_other.bias = reinterpret_cast<const f32x4*>(lds + (b_off - Sh::SPLIT_WORDS_NET + Sh::LDS_NET + Sh::OTHER_OPS * 256));
  }
  const float wseed = split_guard_seed(wmax);

  const int n_rows = (int)a.rows;
  const int n_tiles = (n_rows + 15) >> 4, stride = (int)gridDim.x * WAVES;
  int32_t* const cold_flag = a.cold + 2 + a.cold_capacity;

  // this lane's float4 groups that exist (the real half may be narrower than H): a dead group reads group 0, as zeros
  bool g_live[G];
  int g_off[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    g_live[g] = 16 * g + 4 * q < hr;
    g_off[g] = g_live[g] ? 16 * g + 4 * q : 0;
  }
  // The NEXT tile's rows are requested at the top of a tile, no branch around the loads (behind one hipcc's wait counts
  // fall back to vmcnt(0)); a missing cotangent reads x, a trip past the end the last tile: both are not used.
  //   0 the conditioner half (stage 0: of y, stage 1: of x), 1 the value half of x, 2 / 3 their cotangents, grad_ld
  const float* const g_src = ST ? a.grad_x : (a.grad_y ? a.grad_y : a.x);
  const bool have_g = ST || a.grad_y != nullptr;
  const float* const gl_or_x = a.grad_ld ? a.grad_ld : a.x;
  f32x4 nx[4][G];
  float nx_gl;
  auto request_rows = [&](int t) {
    t = t < n_tiles ? t : n_tiles - 1;
    const int rw = t * 16 + j;
    const uint32_t r = (uint32_t)(rw < n_rows ? rw : n_rows - 1);
    const float* p0 = (ST ? a.x : a.y) + r * dim + col_cond;
    const float* p1 = a.x + r * dim + col_val;
    const float* p2 = g_src + r * dim + col_cond;
    const float* p3 = g_src + r * dim + col_val;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      nx[0][g] = *reinterpret_cast<const f32x4*>(p0 + g_off[g]);
      nx[1][g] = *reinterpret_cast<const f32x4*>(p1 + g_off[g]);
      nx[2][g] = *reinterpret_cast<const f32x4*>(p2 + g_off[g]);
      nx[3][g] = *reinterpret_cast<const f32x4*>(p3 + g_off[g]);
    }
    nx_gl = gl_or_x[r];
  };
  float mx_grad = 0.f;
  const int tile0 = (int)blockIdx.x * WAVES + wave;
  // the second wave of a SIMD starts half a slot late: both run the same code, and from the same start they would sit in
  // their MFMA bursts together and in the spline together
  if (Sh::WAVES_PER_SIMD == 2 && wave >= 4) __builtin_amdgcn_s_sleep(31);
  if (tile0 < n_tiles) request_rows(tile0);
  for (int tile = tile0; tile < n_tiles; tile += stride) {
    f32x4 cur[4][G];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < G; ++g) cur[i][g] = g_live[g] ? nx[i][g] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float cur_gl = nx_gl;
    const int row = tile * 16 + j;
    const bool live = row < n_rows;
    const float gl = (a.grad_ld && live) ? cur_gl * g_scale : 0.f;
    const float g_on = (have_g && live) ? g_scale : 0.f;
    bool cold;
    if (ST == 0) {
      // the range verdict of the WHOLE tile first: the other net will run its hidden layers on this stage's value half
      // (stage 1 could only find out after this stage's sums have taken the tile in)
      u32x2 xh[G], xl[G], hh[3], hl[3];
      float mx = wseed;
      hidden_forward<H, NH, K>(nv_other, cur[1], xh, xl, hh, hl, mx);
      cold = wave_any(!(mx <= kSplitLimit));
    } else {
      cold = cold_flag[tile] != 0;
    }
    if (!cold) {
      f32x4 g_cond[G], g_val[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        g_cond[g] = cur[2][g] * g_on;
        g_val[g] = cur[3][g] * g_on;
      }
      cold = !half_backward<H, NH, K, INV>(nv, lane, cur[0], cur[1], gl, g_val, g_cond, a.T, wseed, mx_grad);
      // (the next tile's rows are requested HERE, behind the tile's arithmetic: at the top of the tile the 17 registers
      //  they land in were live across the whole slot loop and the kernel spilled ~40 registers per tile around it --
      //  scratch traffic of 230 MB per launch; the other wave of the SIMD covers the latency)
      request_rows(tile + stride);
      if (!cold && live) {
        float* gr = a.grad_x + (uint32_t)row * dim;
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (g_live[g]) {
            *reinterpret_cast<f32x4*>(gr + col_cond + g_off[g]) = g_cond[g] * g_unscale;
            *reinterpret_cast<f32x4*>(gr + col_val + g_off[g]) = g_val[g] * g_unscale;
          }
      }
    }
    else {
      request_rows(tile + stride);
    }
    if (ST == 0 && cold && lane == 0) {
      cold_flag[tile] = 1;
      const int slot = atomicAdd(a.cold, 1);
      if (slot < a.cold_capacity) a.cold[2 + slot] = tile;
    }
  }
  if (wave_any(!(mx_grad <= 32768.f)) && lane == 0) a.cold[1] = 1;  // a gradient operand beyond f16: see the header

  // ------------------------------------------------------------------ flush: the waves' sums added up in LDS, then one block per workgroup
  __syncthreads();  // (every wave is done with the image)
  float* red = reinterpret_cast<float*>(lds);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the last MFMAs' results, before the accumulators are read)
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) {
      nt_static_for<0, Sh::TILES>([&](auto t) {
        constexpr int TT = decltype(t)::value;
        f32x4* p = reinterpret_cast<f32x4*>(red + TT * 256 + lane * 4);
        const f32x4 v = AccT<Sh, TT>::read();
        *p = w == 0 ? v : *p + v;
      });
    }
    __syncthreads();
  }
  const int per_net = a.n_params / 2, base = my_net * per_net;
  float* dst = a.partials + (int64_t)blockIdx.x * a.n_params + base;
  for (int i = threadIdx.x; i < per_net; i += blockDim.x) dst[i] = red[a.flush[base + i]];
}

// The kernels proper.  amdgpu_num_vgpr wants a literal: one kernel per (shape, direction, stage), the compiler's share of
// a two-waves-per-SIMD shape's register file (NtShape::ACC_BASE) spelled out; the one-wave-per-SIMD shapes keep their sums
// at the top of the accumulator file and carry no limit (BASE: the first accumulator register of the sums).
typedef void (*NtKernel)(NtArgs);
template <int H, int NH, int K, bool INV, int ST>
struct NtKernelOf;
#define MNF_NT_KERNEL_DECL(HH, NHH, KK, INVV, STV, NAME, ATTR, BASE)                                   \
  __global__ void __launch_bounds__((NtShape<HH, NHH, KK>::WAVES * 64), 1) ATTR NAME(NtArgs a) {        \
    static_assert(BASE == NtShape<HH, NHH, KK>::ACC_BASE, "the compiler's registers end where the accumulators begin"); \
    nsf_bwd_tile_body<HH, NHH, KK, INVV, STV>(a);                                                      \
  }                                                                                                    \
  template <>                                                                                          \
  struct NtKernelOf<HH, NHH, KK, INVV, STV> {                                                          \
    static NtKernel get() { return NAME; }                                                             \
  };
#define MNF_NT_KERNEL4(HH, NHH, KK, ATTR, BASE)                                                          \
  MNF_NT_KERNEL_DECL(HH, NHH, KK, true, 0, nsf_bwd_tile_kernel_##HH##_##NHH##_##KK##_inv_s0, ATTR, BASE)  \
  MNF_NT_KERNEL_DECL(HH, NHH, KK, true, 1, nsf_bwd_tile_kernel_##HH##_##NHH##_##KK##_inv_s1, ATTR, BASE)  \
  MNF_NT_KERNEL_DECL(HH, NHH, KK, false, 0, nsf_bwd_tile_kernel_##HH##_##NHH##_##KK##_fwd_s0, ATTR, BASE) \
  MNF_NT_KERNEL_DECL(HH, NHH, KK, false, 1, nsf_bwd_tile_kernel_##HH##_##NHH##_##KK##_fwd_s1, ATTR, BASE)
#define MNF_NT_TOP(BASE) __attribute__((amdgpu_num_vgpr(BASE / 2)))
MNF_NT_KERNEL4(16, 8, 8, MNF_NT_TOP(144), 144)
MNF_NT_KERNEL4(16, 8, 5, MNF_NT_TOP(176), 176)
MNF_NT_KERNEL4(16, 16, 8, , 140)
MNF_NT_KERNEL4(16, 16, 5, MNF_NT_TOP(172), 172)
MNF_NT_KERNEL4(32, 8, 8, , 44)
MNF_NT_KERNEL4(32, 8, 5, , 108)
MNF_NT_KERNEL4(32, 16, 8, , 32)
MNF_NT_KERNEL4(32, 16, 5, , 100)
MNF_NT_KERNEL4(16, 8, 10, , 112)
MNF_NT_KERNEL4(16, 16, 10, , 104)

// grad_flat[p] += (sum over the workgroups' blocks, in a fixed order) / scale; nothing when the launch went cold
__global__ void __launch_bounds__(256) nsf_tile_reduce_kernel(const float* __restrict__ partials, int n_blocks, int n_params,
                                                              float* __restrict__ grad_flat,
                                                              const float* __restrict__ scale_dev,
                                                              const int32_t* __restrict__ cold) {
  if (cold[0] < 0 || cold[1] != 0) return;
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  const bool mine = i < n_params;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (mine) {
    int b = slice;
    for (; b + 24 < n_blocks; b += 32) {
      s0 += partials[(int64_t)(b + 0) * n_params + i];
      s1 += partials[(int64_t)(b + 8) * n_params + i];
      s2 += partials[(int64_t)(b + 16) * n_params + i];
      s3 += partials[(int64_t)(b + 24) * n_params + i];
    }
    for (; b < n_blocks; b += 8) s0 += partials[(int64_t)b * n_params + i];
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

// ---------------------------------------------------------------- host: index tables
// real shape behind the padded one: half width hr <= H (whole float4 groups), hidden widths w[] <= NH
template <int H, int NH, int K>
static void build_tables(int hr, const int* w, int32_t* idx, int32_t* flush) {
  using Sh = NtShape<H, NH, K>;
  constexpr int G = Sh::G, S = Sh::S, P = Sh::P, NB = Sh::NB;
  int sizes[5] = {hr, w[0], w[1], w[2], P * hr};
  NetDesc net[2];
  int64_t off = fill_net(net[0], 5, sizes, 0);
  const int64_t n_params = fill_net(net[1], 5, sizes, off);
  if (idx) {
    const int64_t n_entries = 2 * (int64_t)Sh::SPLIT_WORDS + Sh::PLAIN_WORDS;
    for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  }
  if (flush)
    for (int64_t i = 0; i < n_params; ++i) flush[i] = 0;
  auto elem_of = [](int s, int qq) { return 16 * (s >> 2) + 4 * qq + (s & 3); };
  for (int nn = 0; nn < 2; ++nn) {
    const NetDesc& nd = net[nn];
    // A operand `op`: weight(m = lane & 15, k = 4 (lane >> 4) + e) -> flat offset or -1
    auto fill_op = [&](int op, auto weight) {
      if (!idx) return;
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
          const int32_t src = weight(lane & 15, 4 * (lane >> 4) + e);
          if (src < 0) continue;
          for (int part = 0; part < 2; ++part) {
            const int64_t word = (int64_t)nn * Sh::SPLIT_WORDS_NET + ((int64_t)op * 64 + lane) * 4 + 2 * part + (e >> 1);
            idx[2 * word + (e & 1)] = src | (part ? kSplitLoBit : 0);
          }
        }
    };
    auto fill_bias = [&](int tile, auto bias) {
      if (!idx) return;
      for (int i = 0; i < 16; ++i) {
        const int32_t src = bias(i);
        if (src >= 0) idx[2 * (int64_t)Sh::SPLIT_WORDS + (int64_t)nn * Sh::PLAIN_WORDS_NET + tile * 16 + i] = src;
      }
    };
    for (int g = 0; g < G; ++g) {
      fill_op(Sh::OP_F1 + g, [&](int m, int k) { return m < w[0] && 16 * g + k < hr ? nd.w_off[0] + m * hr + 16 * g + k : -1; });
      fill_op(Sh::OP_T1 + g, [&](int m, int k) { return k < w[0] && 16 * g + m < hr ? nd.w_off[0] + k * hr + 16 * g + m : -1; });
    }
    for (int l = 1; l <= 2; ++l) {
      fill_op(Sh::OP_F2 + l - 1, [&](int m, int k) { return m < w[l] && k < w[l - 1] ? nd.w_off[l] + m * w[l - 1] + k : -1; });
      fill_op(l == 1 ? Sh::OP_T2 : Sh::OP_T3,
              [&](int m, int k) { return k < w[l] && m < w[l - 1] ? nd.w_off[l] + k * w[l - 1] + m : -1; });
    }
    for (int l = 0; l < 3; ++l) fill_bias(l, [&](int i) { return i < w[l] ? nd.b_off[l] + i : -1; });
    // entry i = 4 q' + r' of tile (s, kb): element elem_of(s, q'), parameter 4 kb + r'
    auto out_of = [&](int s, int kb, int i) {
      const int elem = elem_of(s, i >> 2), prm = 4 * kb + (i & 3);
      return prm < P && elem < hr ? elem * P + prm : -1;
    };
    for (int s = 0; s < S; ++s)
      for (int kb = 0; kb < NB; ++kb) {
        fill_op(Sh::OP_F4 + s * NB + kb, [&](int m, int k) {
          const int o = out_of(s, kb, m);
          return o >= 0 && k < w[2] ? nd.w_off[3] + o * w[2] + k : -1;
        });
        fill_op(Sh::OP_T4 + s * NB + kb, [&](int m, int k) {
          const int o = out_of(s, kb, k);
          return o >= 0 && m < w[2] ? nd.w_off[3] + o * w[2] + m : -1;
        });
        fill_bias(3 + s * NB + kb, [&](int i) {
          const int o = out_of(s, kb, i);
          return o >= 0 ? nd.b_off[3] + o : -1;
        });
      }
    if (!flush) continue;
    // accumulator element (tile t, lane (n, q), reg r) = D[m = 4 q + r][n]
    auto at = [&](int t, int m, int n) { return t * 256 + (16 * (m >> 2) + n) * 4 + (m & 3); };
    for (int s = 0; s < S; ++s)
      for (int kb = 0; kb < NB; ++kb)
        for (int m = 0; m < 16; ++m) {
          const int o = out_of(s, kb, m), t = s * NB + kb;
          if (o < 0) continue;
          for (int u = 0; u < w[2]; ++u) flush[nd.w_off[3] + o * w[2] + u] = at(Sh::T_W4 + t, m, u);
          flush[nd.b_off[3] + o] = Sh::BIASCOL ? at(Sh::T_W4 + t, m, 8) : at(Sh::T_B + (3 + t) / 16, m, (3 + t) & 15);
        }
    for (int l = 1; l <= 2; ++l)
      for (int m = 0; m < w[l]; ++m) {
        for (int n = 0; n < w[l - 1]; ++n) flush[nd.w_off[l] + m * w[l - 1] + n] = at(l == 2 ? Sh::T_W3 : Sh::T_W2, m, n);
        flush[nd.b_off[l] + m] = Sh::BIASCOL ? at(l == 2 ? Sh::T_W3 : Sh::T_W2, m, 8) : at(Sh::T_B, m, l);
      }
    for (int m = 0; m < w[0]; ++m) {
      for (int f = 0; f < hr; ++f) flush[nd.w_off[0] + m * hr + f] = at(Sh::T_W1 + (f >> 4), m, f & 15);
      flush[nd.b_off[0] + m] = at(Sh::T_B, m, 0);
    }
  }
}

template <int H, int NH, int K>
static int launch_tile(const NtArgs& a, int inverse, hipStream_t stream) {
  using Sh = NtShape<H, NH, K>;
  const int64_t n_tiles = (a.rows + 15) / 16;
  int64_t blocks = (n_tiles + Sh::WAVES - 1) / Sh::WAVES;
  const int cus = device_cus(current_device());
  if (blocks > cus) blocks = cus;  // one persistent workgroup per CU: the sums take the register file
  const size_t lds_bytes = (size_t)Sh::LDS_WORDS * 4;
  static DeviceMemo attr[4];
  tag_kernel("nsf_bwd_tile");
  for (int st = 0; st < 2; ++st) {
    const NtKernel kernel = inverse ? (st ? NtKernelOf<H, NH, K, true, 1>::get() : NtKernelOf<H, NH, K, true, 0>::get())
                                    : (st ? NtKernelOf<H, NH, K, false, 1>::get() : NtKernelOf<H, NH, K, false, 0>::get());
    const int ok = attr[2 * (inverse ? 1 : 0) + st].get([&](int) {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_bytes) == hipSuccess ? 1 : -1;
    });
    if (ok < 0) return MNF_ERR_LAUNCH;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(Sh::WAVES * 64), lds_bytes, stream, a);
    if (int rc = check_launch()) return rc;
  }
  hipLaunchKernelGGL(nsf_tile_reduce_kernel, dim3((unsigned)((a.n_params + 31) / 32)), dim3(256), 0, stream, a.partials,
                     (int)blocks, a.n_params, a.grad_flat, a.scale_dev, a.cold);
  return check_launch();
}

// (H, NH, K) triples with an instantiated kernel
#define MNF_NT_SHAPES(X) X(16, 8, 8) X(16, 8, 5) X(16, 16, 8) X(16, 16, 5) X(32, 8, 8) X(32, 8, 5) X(32, 16, 8) X(32, 16, 5) X(16, 8, 10) X(16, 16, 10)

struct TileShape {
  int H, NH, K, hr;
  int w[3];
};
static bool tile_shape(int dim, int K, int n_hidden, const int* hidden, TileShape& ts) {
  if (n_hidden != 3 || !hidden || dim < 8 || (dim & 7) || dim > 64) return false;
  int mx = 0;
  for (int i = 0; i < 3; ++i) {
    if (hidden[i] < 1) return false;
    ts.w[i] = hidden[i];
    mx = hidden[i] > mx ? hidden[i] : mx;
  }
  ts.hr = dim / 2;
  ts.H = ts.hr <= 16 ? 16 : 32;
  ts.NH = mx <= 8 ? 8 : (mx <= 16 ? 16 : 0);
  ts.K = K;
#define X(HH, NHH, KK) \
  if (ts.H == HH && ts.NH == NHH && K == KK) return true;
  MNF_NT_SHAPES(X)
#undef X
  return false;
}

}  // namespace
}  // namespace mnf

extern "C" {

int mnf_nsf_cl_bwd_tile_supported(int dim, int K, int n_hidden, const int* hidden) {
  mnf::TileShape ts;
  return mnf::tile_shape(dim, K, n_hidden, hidden, ts) ? 1 : 0;
}

int mnf_nsf_cl_bwd_tile_layout(int dim, int K, int n_hidden, const int* hidden, int64_t* n_split_words,
                               int64_t* n_plain_words, int64_t* n_params) {
  mnf::TileShape ts;
  if (!n_split_words || !n_plain_words || !n_params || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!mnf::tile_shape(dim, K, n_hidden, hidden, ts)) return MNF_ERR_UNSUPPORTED;
  int sizes[5] = {ts.hr, ts.w[0], ts.w[1], ts.w[2], (3 * K - 1) * ts.hr};
  mnf::NetDesc nd;
  *n_params = 2 * mnf::fill_net(nd, 5, sizes, 0);
#define X(HH, NHH, KK)                                             \
  if (ts.H == HH && ts.NH == NHH && K == KK) {                     \
    *n_split_words = mnf::NtShape<HH, NHH, KK>::SPLIT_WORDS;       \
    *n_plain_words = mnf::NtShape<HH, NHH, KK>::PLAIN_WORDS;       \
    return MNF_OK;                                                 \
  }
  MNF_NT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_nsf_cl_bwd_tile_index(int dim, int K, int n_hidden, const int* hidden, int32_t* idx_host, int32_t* flush_host) {
  mnf::TileShape ts;
  if (!idx_host || !flush_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!mnf::tile_shape(dim, K, n_hidden, hidden, ts)) return MNF_ERR_UNSUPPORTED;
#define X(HH, NHH, KK)                                                     \
  if (ts.H == HH && ts.NH == NHH && K == KK) {                             \
    mnf::build_tables<HH, NHH, KK>(ts.hr, ts.w, idx_host, flush_host);     \
    return MNF_OK;                                                         \
  }
  MNF_NT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int64_t mnf_nsf_cl_bwd_tile_workspace(int64_t rows, int dim, int K, int n_hidden, const int* hidden) {
  int64_t sw = 0, pw = 0, np = 0;
  if (rows < 0 || mnf_nsf_cl_bwd_tile_layout(dim, K, n_hidden, hidden, &sw, &pw, &np) != MNF_OK) return 0;
  return (int64_t)mnf::device_cus(mnf::current_device()) * np;
}

int mnf_nsf_cl_bwd_tile(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                        const void* image, const int32_t* flush, int64_t rows, int dim, int K, float tail_bound,
                        int inverse, int n_hidden, const int* hidden, const float* scale_dev, int32_t* cold,
                        int cold_capacity, float* workspace, int64_t workspace_floats, void* stream) {
  if (!x || !y || !grad_x || !grad_flat || !image || !flush || !scale_dev || !cold || !workspace || rows < 0 || dim < 2 ||
      (dim & 1) || K < 2 || !(tail_bound > 0.f) || !mnf::hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  mnf::TileShape ts;
  if (!mnf::tile_shape(dim, K, n_hidden, hidden, ts)) return MNF_ERR_UNSUPPORTED;
  if (rows * dim >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;  // (32-bit element offsets: the caller's generic kernel)
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(grad_y) |
       reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(image)) & 15)
    return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  if (cold_capacity < (rows + 15) / 16 || workspace_floats < mnf_nsf_cl_bwd_tile_workspace(rows, dim, K, n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  mnf::NtArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat;
  a.image = static_cast<const uint32_t*>(image); a.flush = flush; a.partials = workspace; a.scale_dev = scale_dev;
  a.cold = cold; a.cold_capacity = cold_capacity; a.rows = rows; a.T = tail_bound; a.hr = ts.hr;
  int sizes[5] = {ts.hr, ts.w[0], ts.w[1], ts.w[2], (3 * K - 1) * ts.hr};
  mnf::NetDesc nd;
  a.n_params = (int)(2 * mnf::fill_net(nd, 5, sizes, 0));
#define X(HH, NHH, KK) \
  if (ts.H == HH && ts.NH == NHH && K == KK) return mnf::launch_tile<HH, NHH, KK>(a, inverse != 0, (hipStream_t)stream);
  MNF_NT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
