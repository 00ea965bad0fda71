// NSF_CL (neural-spline coupling layer) with the conditioner on the fp32 matrix cores and the
// rational-quadratic spline evaluated out of accumulator registers, gfx950.
//
// One wave owns 16 rows; lane (j = lane & 15, q = lane >> 4) holds, as float4s, the elements
// {16 gg + 4 q + g'} of both halves of row j.  The conditioner MLP(H, NH, NH, NH, (3K-1) H) runs
// transposed on v_mfma_f32_16x16x4_f32 exactly like the AffineHalfFlow kernel (accumulator
// register r of a tile is the next layer's K-step operand; bias = initial accumulator).  Its
// last layer is arranged so that the tile for "slot" s = (gg, g') and parameter block kb puts
// parameter 4 kb + r of element 16 gg + 4 q + g' of row j into register r of lane (j, q):
// after ceil((3K-1)/4) tiles a lane owns all 3K-1 spline parameters of one of ITS OWN elements,
// and evaluates the spline on it with every index a compile-time constant -- no LDS traffic for
// parameters, no cross-lane movement, all 64 lanes busy in the transcendental part.
//
// The spline reproduces the reference's double normalisation and knot construction
// (spline_flow.py:254-256, :95-113); see rqs_regs below.  This kernel is VALU/transcendental
// bound by construction (~36 v_exp per element at K = 8): SURVEY.md 8d.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_nsf_spline.h"
#include "mnf_split.h"

#ifndef MNF_NSF_NT
#define MNF_NSF_NT 1  // the block's intermediate tensors are written once and not read by this or the next launch:
                      // non-temporal stores (C3: 436-438 -> 431 us per block launch; 0 = A/B switch)
#endif

namespace mnf {

template <int H, int NH, int K>
struct NsfShape {
  static_assert(H % 16 == 0 && NH % 4 == 0 && K >= 2, "unsupported NSF_CL shape");
  static constexpr int G = H / 16;            // float4 groups per half row
  static constexpr int QH = NH / 4;           // K-steps (quads) of a hidden layer
  static constexpr int NTH = (QH + 3) / 4;    // 16-row tiles of a hidden layer
  static constexpr int P = 3 * K - 1;         // spline parameters per element
  static constexpr int NB = (P + 3) / 4;      // parameter blocks (tiles) per slot
  static constexpr int S = H / 4;             // slots = elements per lane per half
  static constexpr int N_L1 = (H / 4) * NTH;
  static constexpr int N_L23 = QH * NTH;
  static constexpr int N_L4 = S * NB * QH;
  static constexpr int N_MFMA = N_L1 + 2 * N_L23 + N_L4;
  static constexpr int A_FLOATS = ((N_MFMA + 3) / 4) * 256;
  static constexpr int BIAS_TILES = 3 * NTH + S * NB;
  static constexpr int NET_FLOATS = A_FLOATS + BIAS_TILES * 16;
  static constexpr int IMAGE_FLOATS = 2 * NET_FLOATS;  // f1 then f2
};

#ifndef MNF_NSF_WAVES
#define MNF_NSF_WAVES 4  // waves per workgroup (experiment switch, with MNF_NSF_WPE = waves per SIMD for the register cap)
#endif
constexpr int kNsfWaves = MNF_NSF_WAVES;

// One half-step: params = net(cond); act <- spline(act; params); returns this lane's sum of
// log-derivatives over its S elements.
// live: bit g set = this lane's float4 group g exists (a half narrower than H: the dead groups' elements run through the
// spline like the others -- zero weights, value 0 -- but their log-derivatives must not be counted)
template <int H, int NH, int K, bool INV>
__device__ __forceinline__ float nsf_half_step(const float* lds_net, int lane, int q, const f32x4 (&cond)[H / 16],
                                               f32x4 (&act)[H / 16], float T, unsigned live = ~0u) {
  using S_ = NsfShape<H, NH, K>;
  constexpr int QH = S_::QH, NTH = S_::NTH, NB = S_::NB, SL = S_::S;
  int a_off = lane * 4, b_off = S_::A_FLOATS + q * 4;
  asm volatile("" : "+v"(a_off), "+v"(b_off));  // keep the operand reads inside the tile loop (see mnf_ahf_mfma.hip)
  const f32x4* A4 = reinterpret_cast<const f32x4*>(lds_net + a_off);
  const f32x4* B4 = reinterpret_cast<const f32x4*>(lds_net + b_off);
  int n = 0, bt = 0;
  f32x4 a4;

  f32x4 h1[NTH];
#pragma unroll
  for (int m = 0; m < NTH; ++m) h1[m] = B4[4 * (bt++)];
#pragma unroll
  for (int c1 = 0; c1 < H / 4; ++c1)
#pragma unroll
    for (int m = 0; m < NTH; ++m) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      h1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], cond[c1 >> 2][c1 & 3], h1[m], 0, 0, 0);
      ++n;
    }
  f32x4 h2[NTH], h3[NTH];
#pragma unroll
  for (int m = 0; m < NTH; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h1[m][r] = __builtin_fmaxf(h1[m][r], kLeakySlope * h1[m][r]);
    h2[m] = B4[4 * (bt++)];
  }
#pragma unroll
  for (int c = 0; c < QH; ++c)
#pragma unroll
    for (int m = 0; m < NTH; ++m) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      h2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h1[c >> 2][c & 3], h2[m], 0, 0, 0);
      ++n;
    }
#pragma unroll
  for (int m = 0; m < NTH; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[m][r] = __builtin_fmaxf(h2[m][r], kLeakySlope * h2[m][r]);
    h3[m] = B4[4 * (bt++)];
  }
#pragma unroll
  for (int c = 0; c < QH; ++c)
#pragma unroll
    for (int m = 0; m < NTH; ++m) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      h3[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h2[c >> 2][c & 3], h3[m], 0, 0, 0);
      ++n;
    }
#pragma unroll
  for (int m = 0; m < NTH; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) h3[m][r] = __builtin_fmaxf(h3[m][r], kLeakySlope * h3[m][r]);

  float lad_sum = 0.f;
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    f32x4 prm[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) prm[kb] = B4[4 * (bt++)];
#pragma unroll
    for (int c = 0; c < QH; ++c)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
        prm[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], h3[c >> 2][c & 3], prm[kb], 0, 0, 0);
        ++n;
      }
    float p[4 * NB];
#pragma unroll
    for (int i = 0; i < 4 * NB; ++i) p[i] = prm[i >> 2][i & 3];
    float o, l;
    rqs_regs<K, INV, 4 * NB>(act[s >> 2][s & 3], T, p, o, l);
    act[s >> 2][s & 3] = o;
    lad_sum += ((live >> (s >> 2)) & 1) ? l : 0.f;
  }
  return lad_sum;
}

// ------------------------------------------------------------------------------------------------
// The conditioner in split (hi + lo) fp32 arithmetic on v_mfma_f32_16x16x16_f16 (mnf_split.h; K = 16
// covers the hidden widths 8 and 16 without the 4x zero padding a K = 32 step would carry in LDS):
// 81 f16 MFMAs per half step at (16, 8, 8) instead of 64 fp32 MFMAs that cost 32 non-overlapping cycles
// each.  Same tiling as the fp32 version (slot s / parameter block kb output tiles), hidden unit u of
// tile m in accumulator row u - 16 m.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int H, int NH, int K>
struct NsfSplitShape {
  using F = NsfShape<H, NH, K>;
  static constexpr int G = H / 16;
  static constexpr int NTH = (NH + 15) / 16;
  static constexpr int NB = F::NB, S = F::S, P = F::P;
  static constexpr int N_OPS = G * NTH + 2 * NTH * NTH + S * NTH * NB;
  static constexpr int OP_WORDS = 256;                       // [hi: 64 lanes x 2 words][lo: 64 lanes x 2 words]
  static constexpr int SPLIT_WORDS_NET = N_OPS * OP_WORDS;
  static constexpr int BIAS_TILES = 3 * NTH + S * NB;
  static constexpr int PLAIN_WORDS_NET = BIAS_TILES * 16;
  // image: [f1 operands][f2 operands][f1 biases][f2 biases][tail]
  static constexpr int SPLIT_WORDS = 2 * SPLIT_WORDS_NET;
  static constexpr int PLAIN_WORDS = 2 * PLAIN_WORDS_NET;
  static constexpr int IMAGE_WORDS = SPLIT_WORDS + PLAIN_WORDS + kSplitTailWords;
};

__device__ __forceinline__ f32x4 mfma_h16(const u32x2& a, const u32x2& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}

// ops: this net's operand words (in LDS), bias: this net's bias tiles (in LDS); mx: max |operand| so far
template <int H, int NH, int K, bool INV>
__device__ __forceinline__ float nsf_half_step_split(const uint32_t* ops, const float* bias, int lane, int q,
                                                     const f32x4 (&cond)[H / 16], f32x4 (&act)[H / 16], float T,
                                                     float& mx, unsigned live = ~0u) {
  using S_ = NsfSplitShape<H, NH, K>;
  constexpr int G = S_::G, NTH = S_::NTH, NB = S_::NB;
  int a_off = lane * 2, b_off = q * 4;
  asm volatile("" : "+v"(a_off), "+v"(b_off));  // keep the operand reads inside the tile loop
  const u32x2* A2 = reinterpret_cast<const u32x2*>(ops + a_off);   // + 64 * (2 op + part)
  const f32x4* B4 = reinterpret_cast<const f32x4*>(bias + b_off);  // + 4 * tile
  int op = 0, bt = 0;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mac = [&](const u32x2& bh, const u32x2& bl, f32x4& main, f32x4& corr) {
    const u32x2 ah = A2[64 * (2 * op)], al = A2[64 * (2 * op + 1)];
    main = mfma_h16(ah, bh, main);
    corr = mfma_h16(ah, bl, corr);
    corr = mfma_h16(al, bh, corr);
    ++op;
  };
  u32x2 xh[G], xl[G];
#pragma unroll
  for (int g = 0; g < G; ++g) split_tile(cond[g], xh[g], xl[g], mx);
  f32x4 main[NTH], corr[NTH];
  u32x2 hh[NTH], hl[NTH];
  auto activate = [&]() {
#pragma unroll
    for (int m = 0; m < NTH; ++m) {
      const f32x4 p = corr[m] * kSplitInvScale + main[m];
      split_tile(__builtin_elementwise_max(p, p * kLeakySlope), hh[m], hl[m], mx);
    }
  };
#pragma unroll
  for (int m = 0; m < NTH; ++m) {
    main[m] = B4[4 * (bt++)];
    corr[m] = zero4;
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int m = 0; m < NTH; ++m) mac(xh[g], xl[g], main[m], corr[m]);
  activate();
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
    u32x2 ph[NTH], pl[NTH];
#pragma unroll
    for (int m = 0; m < NTH; ++m) {
      ph[m] = hh[m];
      pl[m] = hl[m];
      main[m] = B4[4 * (bt++)];
      corr[m] = zero4;
    }
#pragma unroll
    for (int ks = 0; ks < NTH; ++ks)
#pragma unroll
      for (int m = 0; m < NTH; ++m) mac(ph[ks], pl[ks], main[m], corr[m]);
    activate();
  }
  // The slots as a RUN-TIME loop (round 5): unrolled, hipcc hoisted the operand reads of all H / 4 slots to the top --
  // 243-254 registers at d = 32 (two waves per SIMD), 512 with 66-82 spilled at d = 64.  The loop always works on
  // component 0 of the group's float4, which is rotated by one element per trip (back in place after four).
  float lad_sum = 0.f;
  const int op0 = op, bt0 = bt;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    f32x4 a4 = act[g];
#pragma nounroll
    for (int r = 0; r < 4; ++r) {
      const int s = 4 * g + r;
      const u32x2* As = A2 + 128 * (op0 + s * (NTH * NB));  // + 64 * (2 o + part), o = ks * NB + kb
      const f32x4* Bs = B4 + 4 * (bt0 + s * NB);
      f32x4 prm[NB], prc[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        prm[kb] = Bs[4 * kb];
        prc[kb] = zero4;
      }
#pragma unroll
      for (int ks = 0; ks < NTH; ++ks)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          const u32x2 ah = As[64 * (2 * (ks * NB + kb))], al = As[64 * (2 * (ks * NB + kb) + 1)];
          prm[kb] = mfma_h16(ah, hh[ks], prm[kb]);
          prc[kb] = mfma_h16(ah, hl[ks], prc[kb]);
          prc[kb] = mfma_h16(al, hh[ks], prc[kb]);
        }
      float p[4 * NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const f32x4 v = prc[kb] * kSplitInvScale + prm[kb];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[4 * kb + i] = v[i];
      }
      float o, l;
      rqs_regs<K, INV, 4 * NB>(a4[0], T, p, o, l);
      a4 = f32x4{a4[1], a4[2], a4[3], o};
      lad_sum += ((live >> g) & 1) ? l : 0.f;
    }
    act[g] = a4;
  }
  return lad_sum;
}

// fp32 half step from the fp32 image in global memory, out of line: the range-guard path of the split kernel
template <int G>
struct HalfStepIO {
  f32x4 act[G];
  float lad;
  unsigned live;
};
template <int H, int NH, int K, bool INV>
__device__ __attribute__((noinline)) HalfStepIO<H / 16> nsf_half_step_cold(const float* net_f32, int lane, int q,
                                                                          HalfStepIO<H / 16> cond_in,
                                                                          HalfStepIO<H / 16> act_in, float T) {
  HalfStepIO<H / 16> out = act_in;
  out.lad = nsf_half_step<H, NH, K, INV>(net_f32, lane, q, cond_in.act, out.act, T, act_in.live);
  return out;
}

// one guarded half step: split path on a copy, redone in fp32 if an operand left the f16 range
template <int H, int NH, int K, bool INV>
__device__ __forceinline__ float nsf_half_step_guarded(const uint32_t* ops, const float* bias, const float* net_f32,
                                                       float wmax, int lane, int q, const f32x4 (&cond)[H / 16],
                                                       f32x4 (&act)[H / 16], float T, unsigned live = ~0u) {
  constexpr int G = H / 16;
  f32x4 trial[G];
#pragma unroll
  for (int g = 0; g < G; ++g) trial[g] = act[g];
  float mx = split_guard_seed(wmax);
  float lad = nsf_half_step_split<H, NH, K, INV>(ops, bias, lane, q, cond, trial, T, mx, live);
  if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {
    HalfStepIO<G> c, a;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      c.act[g] = cond[g];
      a.act[g] = act[g];
    }
    c.lad = a.lad = 0.f;
    c.live = a.live = live;
    const HalfStepIO<G> r = nsf_half_step_cold<H, NH, K, INV>(net_f32, lane, q, c, a, T);
#pragma unroll
    for (int g = 0; g < G; ++g) trial[g] = r.act[g];
    lad = r.lad;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) act[g] = trial[g];
  return lad;
}

// row <- row @ A + b for the whole (lower | upper) row held as float4s, on the same MFMA scheme as
// mnf_linear_mfma.hip: `aff` = [D*D operand image][D bias], D = 2H.
template <int H>
__device__ __forceinline__ void affine_rows(const float* aff, int lane, int q, f32x4 (&lo)[H / 16],
                                            f32x4 (&up)[H / 16]) {
  constexpr int G = H / 16, D = 2 * H, NK = D / 4;
  int a_off = lane * 4, b_off = D * D + q * 4;
  asm volatile("" : "+v"(a_off), "+v"(b_off));
  const f32x4* A4 = reinterpret_cast<const f32x4*>(aff + a_off);
  const f32x4* B4 = reinterpret_cast<const f32x4*>(aff + b_off);
  f32x4 acc[2 * G];
#pragma unroll
  for (int m = 0; m < 2 * G; ++m) acc[m] = B4[4 * m];  // bias of dims 16 m + 4 q .. + 3
  int n = 0;
  f32x4 a4;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int m = 0; m < 2 * G; ++m) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      const float b = (kk >> 2) < G ? lo[kk >> 2][kk & 3] : up[(kk >> 2) - G][kk & 3];
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], b, acc[m], 0, 0, 0);
      ++n;
    }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    lo[g] = acc[g];
    up[g] = acc[G + g];
  }
}

// AFF: 0 = plain NSF_CL.  Opt-in fusion of the reference's [ActNorm, Glow, NSF_CL] block (SURVEY.md
// 8f rank 3): 1 = the affine map runs before the spline steps (forward: z e^s + t, then @ W),
// 2 = after them (inverse: @ W^-1, then (. - t) e^-s); both collapse to one  row @ A + b  and a
// row-independent log-det constant.  The block's two intermediate tensors are never written.
// SPLIT: the conditioner on f16 MFMAs in split arithmetic (`simage`), with `image` (fp32, read from global
// memory) behind it for tiles whose operands leave the f16 range.
// RAG (AFF = 0 only): the real half `hr` is narrower than H, in whole float4 groups (dim = 2 hr, a multiple of 8): a lane's
// groups at or beyond hr are dead -- loaded as zeros from group 0's address, not stored, their log-derivatives not
// counted; the index tables leave their weights out.
template <int H, int NH, int K, bool INV, int AFF = 0, bool SPLIT = false, bool RAG = false>
#ifdef MNF_NSF_WPE
__global__ void __launch_bounds__(kNsfWaves * 64, MNF_NSF_WPE)
#else
__global__ void __launch_bounds__(kNsfWaves * 64)
#endif
nsf_mfma_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ log_det,
                const float* __restrict__ image, const uint32_t* __restrict__ simage, int64_t rows, float T,
                int accumulate,
                const float* __restrict__ aff_image, float ld_const, const float* __restrict__ scale_shift,
                float* __restrict__ mid1, float* __restrict__ mid2, float* __restrict__ log_prob,
                double* __restrict__ log_prob_sum, int hr) {
  using S_ = NsfShape<H, NH, K>;
  constexpr int G = S_::G;
  static_assert(G >= 1 && !(RAG && AFF), "");
  const int dim = RAG ? 2 * hr : 2 * H, up_off = RAG ? hr : H;  // (compile-time constants without RAG)
  constexpr int AFF_FLOATS = AFF ? 2 * H * 2 * H + 2 * H : 0;
  constexpr int SS_FLOATS = AFF ? 2 * 2 * H : 0;  // ActNorm's exp(s) and t, for the block's intermediate tensors
  using SS_ = NsfSplitShape<H, NH, K>;
  constexpr int NET_IMAGE = SPLIT ? SS_::IMAGE_WORDS : S_::IMAGE_FLOATS;  // words of LDS for the conditioner nets
  __shared__ __attribute__((aligned(16))) float lds[NET_IMAGE + AFF_FLOATS + SS_FLOATS];
  {
    const float4* src = SPLIT ? reinterpret_cast<const float4*>(simage) : reinterpret_cast<const float4*>(image);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < NET_IMAGE / 4; i += blockDim.x) dst[i] = src[i];
    if (AFF) {
      const float4* asrc = reinterpret_cast<const float4*>(aff_image);
      float4* adst = reinterpret_cast<float4*>(lds + NET_IMAGE);
      for (int i = threadIdx.x; i < AFF_FLOATS / 4; i += blockDim.x) adst[i] = asrc[i];
      if (scale_shift)
        for (int i = threadIdx.x; i < SS_FLOATS; i += blockDim.x) lds[NET_IMAGE + AFF_FLOATS + i] = scale_shift[i];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float* f1 = lds;
  const float* f2 = lds + S_::NET_FLOATS;
  // split image: [f1 operands][f2 operands][f1 biases][f2 biases][tail: max |weight|]
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(lds);
  const float wmax = SPLIT ? lds[SS_::SPLIT_WORDS + SS_::PLAIN_WORDS] : 0.f;
  // this lane's float4 groups that exist
  unsigned live_mask = ~0u;
  int g_off[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const bool lv = !RAG || 16 * g + 4 * q < hr;
    if (!lv) live_mask &= ~(1u << g);
    g_off[g] = lv ? 16 * g + 4 * q : 0;
  }
  auto half_step = [&](auto inv_tag, int net, const f32x4 (&cond)[G], f32x4 (&act)[G]) -> float {
    constexpr bool kInv = decltype(inv_tag)::value;
    float lad;
    if constexpr (SPLIT)
      lad = nsf_half_step_guarded<H, NH, K, kInv>(sw + net * SS_::SPLIT_WORDS_NET,
                                                  lds + SS_::SPLIT_WORDS + net * SS_::PLAIN_WORDS_NET,
                                                  image + net * S_::NET_FLOATS, wmax, lane, q, cond, act, T, live_mask);
    else
      lad = nsf_half_step<H, NH, K, kInv>(net ? f2 : f1, lane, q, cond, act, T, live_mask);
    if constexpr (RAG) {  // (a dead element's spline output is f(0) of an all-zero parameter set: keep it at exactly 0)
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (!((live_mask >> g) & 1)) act[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return lad;
  };

  double lp_acc = 0.0;
  const int n_tiles = (int)((rows + 15) >> 4);
  // The NEXT tile's row (and its log_det entry when this launch adds to it) is requested at the top of a tile (round 4):
  // loaded where it was used, the row waited for its own latency AND -- vector-memory operations complete in order --
  // for the previous tile's stores, and the log_det read-modify-write at the tile's end did so again.  No branch around
  // the requests (behind one hipcc's wait counts fall back to vmcnt(0)): past the end the same tile is read again.
  const int tile_step = (int)gridDim.x * kNsfWaves, tile0 = (int)blockIdx.x * kNsfWaves + wave;
  const float* const ld_or_x = (log_det && accumulate) ? log_det : x;
  f32x4 n_lo[G], n_up[G];
  float n_ld;
  auto request_tile = [&](int t) {
    const int64_t rw = (int64_t)(t < n_tiles ? t : n_tiles - 1) * 16 + j;
    const int64_t rc = rw < rows ? rw : rows - 1;
    const float* xr = x + rc * dim;
#pragma unroll
    for (int g = 0; g < G; ++g) n_lo[g] = *reinterpret_cast<const f32x4*>(xr + g_off[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) n_up[g] = *reinterpret_cast<const f32x4*>(xr + up_off + g_off[g]);
    n_ld = ld_or_x[rc];
  };
  if (n_tiles > 0) request_tile(tile0);
  for (int tile = tile0; tile < n_tiles; tile += tile_step) {
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    float* yr = y + rowc * dim;
    f32x4 lo[G], up[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool lv = (live_mask >> g) & 1;
      lo[g] = lv ? n_lo[g] : f32x4{0.f, 0.f, 0.f, 0.f};
      up[g] = lv ? n_up[g] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float ld_before = n_ld;
    request_tile(tile + tile_step);
    float ld;
    // the block's two intermediate tensors (mid1, mid2 in application order), written once from registers:
    //   forward: ActNorm(x) = x e^s + t elementwise, then the affine result = Glow(ActNorm(x))
    //   inverse: NSF^-1(x), then Glow^-1 of it = (final) e^s + t elementwise from the affine result
    const float* ss = lds + NET_IMAGE + AFF_FLOATS + 4 * q;
    // (every lane stores: a lane past the last row holds the last row's values -- clamped loads, the same arithmetic -- and
    //  writes them to the last row again.  A store under a branch would make the next tile's first use of its prefetched
    //  row wait for vmcnt(0), i.e. for these stores to reach memory.)
    auto store_row = [&](float* base, const f32x4 (&a)[G], const f32x4 (&b)[G]) {
      float* mr = base + rowc * dim + 4 * q;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (MNF_NSF_NT) __builtin_nontemporal_store(a[g], reinterpret_cast<f32x4*>(mr + 16 * g));
        else *reinterpret_cast<f32x4*>(mr + 16 * g) = a[g];
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (MNF_NSF_NT) __builtin_nontemporal_store(b[g], reinterpret_cast<f32x4*>(mr + H + 16 * g));
        else *reinterpret_cast<f32x4*>(mr + H + 16 * g) = b[g];
      }
    };
    auto store_actnorm_of = [&](float* base) {  // rows e^s + t
      f32x4 a[G], b[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        a[g] = lo[g] * *reinterpret_cast<const f32x4*>(ss + 16 * g) + *reinterpret_cast<const f32x4*>(ss + dim + 16 * g);
        b[g] = up[g] * *reinterpret_cast<const f32x4*>(ss + H + 16 * g) +
               *reinterpret_cast<const f32x4*>(ss + dim + H + 16 * g);
      }
      store_row(base, a, b);
    };
    if (AFF == 1) {
      if (mid1) store_actnorm_of(mid1);
      affine_rows<H>(lds + NET_IMAGE, lane, q, lo, up);
      if (mid2) store_row(mid2, lo, up);
    }
    if (!INV) {  // f1(lower) moves upper, then f2(upper') moves lower (spline_flow.py:249-266)
      ld = half_step(std::false_type{}, 0, lo, up);
      ld += half_step(std::false_type{}, 1, up, lo);
    } else {     // (:268-285)
      ld = half_step(std::true_type{}, 1, up, lo);
      ld += half_step(std::true_type{}, 0, lo, up);
    }
    if (AFF == 2) {
      if (mid1) store_row(mid1, lo, up);
      affine_rows<H>(lds + NET_IMAGE, lane, q, lo, up);
      if (mid2) store_actnorm_of(mid2);
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (!RAG || ((live_mask >> g) & 1)) *reinterpret_cast<f32x4*>(yr + g_off[g]) = lo[g];
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (!RAG || ((live_mask >> g) & 1)) *reinterpret_cast<f32x4*>(yr + up_off + g_off[g]) = up[g];
    if (log_det) {
      ld = sum_over_q(ld) + ld_const;
      if (accumulate) ld += ld_before;
      log_det[rowc] = ld;  // (the row's four lanes and any lane past the end: the same value)
      if (log_prob) {  // last launch of a density pass: log p = log_det - |y|^2 / 2 - d/2 log(2 pi)  (core.py:46-49)
        float sq = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) sq = fmaf(lo[g][r], lo[g][r], fmaf(up[g][r], up[g][r], sq));
        sq = sum_over_q(sq);
        const float lp = ld + (-0.5f * sq - (float)dim * kHalfLog2Pi);
        if (live && q == 0) {
          log_prob[row] = lp;
          lp_acc += (double)lp;
        }
      }
    }
  }
  if (log_prob_sum) {  // fp64 sum over the rows: wave shuffle, one native fp64 atomic per wave
    for (int off = 32; off > 0; off >>= 1) lp_acc += __shfl_down(lp_acc, off, 64);
    if (lane == 0) atomicAdd(log_prob_sum, lp_acc);
  }
}

// ---------------------------------------------------------------- host: image index table
// hr: the real half width (<= H, whole float4 groups): features and elements at or beyond it have no weights
template <int H, int NH, int K>
static void build_index(int32_t* idx, const int* widths = nullptr, int hr = H) {
  using S_ = NsfShape<H, NH, K>;
  constexpr int QH = S_::QH, NTH = S_::NTH, NB = S_::NB, SL = S_::S, P = S_::P;
  // real widths of the three hidden layers (<= NH; the other units are structural zeros: LeakyReLU(0) = 0)
  const int w[3] = {widths ? widths[0] : NH, widths ? widths[1] : NH, widths ? widths[2] : NH};
  int sizes[5] = {hr, w[0], w[1], w[2], P * hr};
  NetDesc net[2];
  int64_t off = fill_net(net[0], 5, sizes, 0);
  fill_net(net[1], 5, sizes, off);
  for (int64_t i = 0; i < S_::IMAGE_FLOATS; ++i) idx[i] = -1;
  auto unit_of = [&](int m, int i) { return 16 * m + 4 * (i & 3) + (i >> 2); };  // hidden unit of acc row i
  for (int nn = 0; nn < 2; ++nn) {
    int32_t* A = idx + (int64_t)nn * S_::NET_FLOATS;
    int32_t* B = A + S_::A_FLOATS;
    int n = 0, bt = 0;
    auto put = [&](int lane, int32_t src) { A[(n >> 2) * 256 + lane * 4 + (n & 3)] = src; };
    for (int m = 0; m < NTH; ++m, ++bt)
      for (int i = 0; i < 16; ++i)
        if (unit_of(m, i) < w[0]) B[bt * 16 + i] = net[nn].b_off[0] + unit_of(m, i);
    for (int c1 = 0; c1 < H / 4; ++c1) {
      const int g = c1 >> 2, e = c1 & 3;
      for (int m = 0; m < NTH; ++m) {
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = unit_of(m, i);
          if (u < w[0] && 16 * g + 4 * kq + e < hr) put(lane, net[nn].w_off[0] + u * hr + 16 * g + 4 * kq + e);
        }
        ++n;
      }
    }
    for (int l = 1; l <= 2; ++l) {
      for (int m = 0; m < NTH; ++m, ++bt)
        for (int i = 0; i < 16; ++i)
          if (unit_of(m, i) < w[l]) B[bt * 16 + i] = net[nn].b_off[l] + unit_of(m, i);
      for (int c = 0; c < QH; ++c)
        for (int m = 0; m < NTH; ++m) {
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, u = unit_of(m, i);
            if (u < w[l] && 4 * c + kq < w[l - 1]) put(lane, net[nn].w_off[l] + u * w[l - 1] + 4 * c + kq);
          }
          ++n;
        }
    }
    for (int s = 0; s < SL; ++s) {
      // accumulator row i = 4 q' + r of tile (s, kb): element 16 (s/4) + 4 q' + (s%4), parameter 4 kb + r
      auto out_of = [&](int kb, int i) {
        const int elem = 16 * (s >> 2) + 4 * (i >> 2) + (s & 3), prm = 4 * kb + (i & 3);
        return prm < P && elem < hr ? elem * P + prm : -1;
      };
      for (int kb = 0; kb < NB; ++kb, ++bt)
        for (int i = 0; i < 16; ++i)
          if (out_of(kb, i) >= 0) B[bt * 16 + i] = net[nn].b_off[3] + out_of(kb, i);
      for (int c = 0; c < QH; ++c)
        for (int kb = 0; kb < NB; ++kb) {
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, o = out_of(kb, i);
            if (o >= 0 && 4 * c + kq < w[2]) put(lane, net[nn].w_off[3] + o * w[2] + 4 * c + kq);
          }
          ++n;
        }
    }
  }
}

// 2 entries per split word (low half, high half), then 1 entry per plain (bias) word -- mnf_pack_gather_split
template <int H, int NH, int K>
static void build_split_index(int32_t* idx, const int* widths = nullptr, int hr = H) {
  using S_ = NsfSplitShape<H, NH, K>;
  constexpr int G = S_::G, NTH = S_::NTH, NB = S_::NB, SL = S_::S, P = S_::P;
  const int w[3] = {widths ? widths[0] : NH, widths ? widths[1] : NH, widths ? widths[2] : NH};  // real widths <= NH
  int sizes[5] = {hr, w[0], w[1], w[2], P * hr};
  NetDesc net[2];
  int64_t off = fill_net(net[0], 5, sizes, 0);
  fill_net(net[1], 5, sizes, off);
  const int64_t n_entries = 2 * (int64_t)S_::SPLIT_WORDS + S_::PLAIN_WORDS;
  for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  for (int nn = 0; nn < 2; ++nn) {
    int op = 0, bt = 0;
    int32_t* B = idx + 2 * (int64_t)S_::SPLIT_WORDS + (int64_t)nn * S_::PLAIN_WORDS_NET;
    // element e (0..3) of lane (i, kq) of operand op: weight(row i of the output tile, input slot 4 kq + e)
    auto put = [&](int lane, int e, int32_t src) {
      for (int part = 0; part < 2; ++part)
        idx[2 * ((int64_t)nn * S_::SPLIT_WORDS_NET) +
            (((int64_t)(2 * op + part) * 64 + lane) * 2 + (e >> 1)) * 2 + (e & 1)] = src | (part ? kSplitLoBit : 0);
    };
    for (int m = 0; m < NTH; ++m, ++bt)
      for (int i = 0; i < 16; ++i)
        if (16 * m + i < w[0]) B[bt * 16 + i] = net[nn].b_off[0] + 16 * m + i;
    for (int g = 0; g < G; ++g)
      for (int m = 0; m < NTH; ++m, ++op)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
          if (u >= w[0]) continue;
          for (int e = 0; e < 4; ++e)
            if (16 * g + 4 * kq + e < hr) put(lane, e, net[nn].w_off[0] + u * hr + 16 * g + 4 * kq + e);
        }
    for (int l = 1; l <= 2; ++l) {
      for (int m = 0; m < NTH; ++m, ++bt)
        for (int i = 0; i < 16; ++i)
          if (16 * m + i < w[l]) B[bt * 16 + i] = net[nn].b_off[l] + 16 * m + i;
      for (int ks = 0; ks < NTH; ++ks)
        for (int m = 0; m < NTH; ++m, ++op)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
            if (u >= w[l]) continue;
            for (int e = 0; e < 4; ++e) {
              const int ui = 16 * ks + 4 * kq + e;
              if (ui < w[l - 1]) put(lane, e, net[nn].w_off[l] + u * w[l - 1] + ui);
            }
          }
    }
    for (int s = 0; s < SL; ++s) {
      // accumulator row i = 4 q' + r of tile (s, kb): element 16 (s/4) + 4 q' + (s%4), parameter 4 kb + r
      auto out_of = [&](int kb, int i) {
        const int elem = 16 * (s >> 2) + 4 * (i >> 2) + (s & 3), prm = 4 * kb + (i & 3);
        return prm < P && elem < hr ? elem * P + prm : -1;
      };
      for (int kb = 0; kb < NB; ++kb, ++bt)
        for (int i = 0; i < 16; ++i)
          if (out_of(kb, i) >= 0) B[bt * 16 + i] = net[nn].b_off[3] + out_of(kb, i);
      for (int ks = 0; ks < NTH; ++ks)
        for (int kb = 0; kb < NB; ++kb, ++op)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, o = out_of(kb, i);
            if (o < 0) continue;
            for (int e = 0; e < 4; ++e) {
              const int ui = 16 * ks + 4 * kq + e;
              if (ui < w[2]) put(lane, e, net[nn].w_off[3] + o * w[2] + ui);
            }
          }
    }
  }
}

template <int H, int NH, int K>
static int launch(const float* x, float* y, float* log_det, int accumulate, const float* image,
                  const uint32_t* simage, int64_t rows, float T, int inverse, hipStream_t stream,
                  const float* aff = nullptr, float ld_const = 0.f, const float* scale_shift = nullptr,
                  float* mid1 = nullptr, float* mid2 = nullptr, float* log_prob = nullptr,
                  double* log_prob_sum = nullptr, int hr = H) {
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + kNsfWaves - 1) / kNsfWaves;
  auto resident_of = [](auto kernel, int dev) {
    return resident_by_occupancy(kernel, kNsfWaves * 64, dev, 2);
  };
  // The fp32-MFMA variants (no split image: force_fp32_mfma / MNF_FP32_MFMA=1) exist for the K = 8 shapes of full width
  // only; an fp32 request at another shape is MNF_ERR_UNSUPPORTED here and runs the VALU kernel (fp32 too) -- the
  // run-time-shaped kernels are split arithmetic and do not take it (mnf_generic.hip).
  // (nor for the fused [ActNorm, Glow, NSF_CL] block: the three layers then run one after the other)
  constexpr bool kHasFp32 = K == 8 && NH <= 16;
  if (!simage && (!kHasFp32 || hr != H || aff)) return MNF_ERR_UNSUPPORTED;
  static DeviceMemo memo_f32, memo_split;
  int resident = 0;
  if (simage) resident = memo_split.get([&](int dev) { return resident_of(nsf_mfma_kernel<H, NH, K, true, 2, true>, dev); });
  else if constexpr (kHasFp32)
    resident = memo_f32.get([&](int dev) { return resident_of(nsf_mfma_kernel<H, NH, K, true, 2, false>, dev); });
  if (blocks > resident) blocks = resident;
  const dim3 grid((unsigned)blocks), block(kNsfWaves * 64);
#define MNF_NSF_LAUNCH_R(INVV, AFFV, SPL, RAGV)                                                                        \
  hipLaunchKernelGGL((nsf_mfma_kernel<H, NH, K, INVV, AFFV, SPL, RAGV>), grid, block, 0, stream, x, y, log_det, image, \
                     simage, rows, T, accumulate, aff, ld_const, scale_shift, mid1, mid2, log_prob, log_prob_sum, hr)
#define MNF_NSF_LAUNCH(INVV, AFFV, SPL) MNF_NSF_LAUNCH_R(INVV, AFFV, SPL, false)
  tag_kernel(simage ? (aff ? "nsf_block_split" : "nsf_mfma_split") : "nsf_mfma_fp32");
  if (hr != H) {  // a half narrower than the tile (plain layer only)
    if (aff) return MNF_ERR_UNSUPPORTED;
    if (inverse) MNF_NSF_LAUNCH_R(true, 0, true, true); else MNF_NSF_LAUNCH_R(false, 0, true, true);
  } else if (simage) {
    if (aff) {
      if (inverse) MNF_NSF_LAUNCH(true, 2, true); else MNF_NSF_LAUNCH(false, 1, true);
    } else {
      if (inverse) MNF_NSF_LAUNCH(true, 0, true); else MNF_NSF_LAUNCH(false, 0, true);
    }
  } else if constexpr (kHasFp32) {
    if (inverse) MNF_NSF_LAUNCH(true, 0, false); else MNF_NSF_LAUNCH(false, 0, false);
  }
#undef MNF_NSF_LAUNCH
#undef MNF_NSF_LAUNCH_R
  return check_launch();
}

// (H, NH, K) triples with an instantiated kernel
// (three hidden layers of 32 units at dim <= 32 had kernels here until round 6: the run-time-shaped kernel of mnf_nsf_rt.hip
// runs those within 1.1-1.2x of them, profiles/r6/coverage_map.txt)
#define MNF_NSF_SHAPES(X) X(16, 8, 8) X(16, 16, 8) X(16, 8, 5) X(32, 8, 8) X(32, 8, 5) X(32, 16, 8) X(16, 16, 5) X(32, 16, 5) X(16, 8, 10) X(16, 16, 10)
// ... and those that also have the fused [ActNorm, Glow, NSF_CL] variants (the affine image must be one the Glow
// MFMA kernel supports: dim 32 and 64 are)
#define MNF_NSF_FUSED_SHAPES(X) X(16, 8, 8) X(16, 16, 8) X(16, 8, 5) X(16, 16, 5) X(32, 8, 8) X(32, 8, 5) X(32, 16, 8)

// three hidden layers of at most 32 units: nh = the width the kernels run them at (8, 16 or -- dim <= 32 -- 32; narrower layers get
// structural-zero units)
static bool uniform_hidden3(int n_hidden, const int* hidden, int& nh) {
  if (n_hidden != 3 || !hidden) return false;
  int mx = 0;
  for (int i = 0; i < 3; ++i) {
    if (hidden[i] < 1) return false;
    mx = hidden[i] > mx ? hidden[i] : mx;
  }
  nh = mx <= 8 ? 8 : mx <= 16 ? 16 : mx <= 32 ? 32 : 0;
  return nh != 0;
}

// half width the plain-layer kernels run dim at: 16 or 32, the real half narrower in whole float4 groups (0: none)
static int nsf_padded_half(int dim) {
  if (dim < 8 || (dim & 7) || dim > 64) return 0;
  return dim / 2 <= 16 ? 16 : 32;
}

int nsf_mfma_launch(const float* x, float* y, float* log_det, int accumulate, const float* image,
                    const void* split_image, int64_t rows, int dim, int K, float tail_bound, int inverse,
                    int n_hidden, const int* hidden, hipStream_t stream) {
  int nh = 0;
  if (!uniform_hidden3(n_hidden, hidden, nh)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(image) |
       reinterpret_cast<uintptr_t>(split_image)) & 15)
    return MNF_ERR_UNSUPPORTED;
  const int hp = nsf_padded_half(dim);
#define X(HH, NHH, KK) \
  if (hp == HH && nh == NHH && K == KK) \
    return launch<HH, NHH, KK>(x, y, log_det, accumulate, image, static_cast<const uint32_t*>(split_image), rows, \
                               tail_bound, inverse != 0, stream, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr,   \
                               nullptr, dim / 2);
  MNF_NSF_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int nsf_fused_launch(const float* x, float* y, float* log_det, int accumulate, const float* image,
                     const void* split_image, const float* aff, float ld_const, const float* scale_shift,
                     float* mid1, float* mid2, float* log_prob, double* log_prob_sum,
                     int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden, const int* hidden,
                     hipStream_t stream) {
  int nh = 0;
  if (!uniform_hidden3(n_hidden, hidden, nh)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(image) |
       reinterpret_cast<uintptr_t>(aff) | reinterpret_cast<uintptr_t>(mid1) | reinterpret_cast<uintptr_t>(mid2) |
       reinterpret_cast<uintptr_t>(split_image)) & 15)
    return MNF_ERR_UNSUPPORTED;
#define X(HH, NHH, KK) \
  if (dim == 2 * HH && nh == NHH && K == KK) \
    return launch<HH, NHH, KK>(x, y, log_det, accumulate, image, static_cast<const uint32_t*>(split_image), rows, \
                               tail_bound, inverse != 0, stream, aff, ld_const, scale_shift, mid1, mid2, log_prob,    \
                               log_prob_sum);
  MNF_NSF_FUSED_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf

extern "C" {

// Opt-in fused [ActNorm, Glow, NSF_CL] block: forward y = NSF(x @ A + b), inverse y = NSF^-1(x) @ A + b,
// log_det = spline terms + ld_const.  aff = [dim*dim operand image of A (mnf_linear_rows_image_index)][dim bias].
int mnf_nsf_cl_fused(const float* x, float* y, float* log_det, int accumulate, const float* image,
                     const void* split_image, const float* aff, float ld_const, const float* scale_shift,
                     float* mid1, float* mid2, float* log_prob, double* log_prob_sum,
                     int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden, const int* hidden,
                     void* stream) {
  if (!x || !y || x == y || !image || !aff || rows < 0 || dim < 2 || (dim & 1) || K < 2 || !(tail_bound > 0.f) ||
      !mnf::hidden_ok(n_hidden, hidden) || ((mid1 || mid2) && !scale_shift) || (mid1 && (mid1 == y || mid1 == x)) ||
      (mid2 && (mid2 == y || mid2 == x || mid2 == mid1)) || ((log_prob || log_prob_sum) && !log_det) ||
      (log_prob_sum && !log_prob))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  return mnf::nsf_fused_launch(x, y, log_det, accumulate, image, split_image, aff, ld_const, scale_shift, mid1, mid2,
                               log_prob, log_prob_sum, rows, dim, K,
                               tail_bound, inverse, n_hidden, hidden, (hipStream_t)stream);
}

int mnf_nsf_cl_split_layout(int dim, int K, int n_hidden, const int* hidden, int64_t* n_split_words,
                            int64_t* n_plain_words) {
  int nh = 0;
  if (!n_split_words || !n_plain_words || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!mnf::uniform_hidden3(n_hidden, hidden, nh)) return MNF_ERR_UNSUPPORTED;
  const int hp = mnf::nsf_padded_half(dim);
#define X(HH, NHH, KK)                                                   \
  if (hp == HH && nh == NHH && K == KK) {                                \
    *n_split_words = mnf::NsfSplitShape<HH, NHH, KK>::SPLIT_WORDS;       \
    *n_plain_words = mnf::NsfSplitShape<HH, NHH, KK>::PLAIN_WORDS;       \
    return MNF_OK;                                                       \
  }
  MNF_NSF_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_nsf_cl_split_index(int dim, int K, int n_hidden, const int* hidden, int32_t* idx_host) {
  int nh = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!mnf::uniform_hidden3(n_hidden, hidden, nh)) return MNF_ERR_UNSUPPORTED;
  const int hp = mnf::nsf_padded_half(dim);
#define X(HH, NHH, KK)                                              \
  if (hp == HH && nh == NHH && K == KK) {                           \
    mnf::build_split_index<HH, NHH, KK>(idx_host, hidden, dim / 2); \
    return MNF_OK;                                                  \
  }
  MNF_NSF_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int64_t mnf_nsf_cl_image_floats(int dim, int K, int n_hidden, const int* hidden) {
  int nh = 0;
  if (!mnf::hidden_ok(n_hidden, hidden) || !mnf::uniform_hidden3(n_hidden, hidden, nh)) return 0;
  const int hp = mnf::nsf_padded_half(dim);
#define X(HH, NHH, KK) \
  if (hp == HH && nh == NHH && K == KK) return mnf::NsfShape<HH, NHH, KK>::IMAGE_FLOATS;
  MNF_NSF_SHAPES(X)
#undef X
  return 0;
}

int mnf_nsf_cl_image_index(int dim, int K, int n_hidden, const int* hidden, int32_t* idx_host) {
  int nh = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!mnf::uniform_hidden3(n_hidden, hidden, nh)) return MNF_ERR_UNSUPPORTED;
  const int hp = mnf::nsf_padded_half(dim);
#define X(HH, NHH, KK)                                        \
  if (hp == HH && nh == NHH && K == KK) {                     \
    mnf::build_index<HH, NHH, KK>(idx_host, hidden, dim / 2); \
    return MNF_OK;                                            \
  }
  MNF_NSF_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
