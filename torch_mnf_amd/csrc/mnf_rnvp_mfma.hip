// Masked / gated RNVP coupling (flows/rnvp.py:25-39) as two chained fp32-MFMA GEMMs, gfx950.
//
//   y      = Wn (m * z) + bn                 (h  <- d)     GEMM 1, K = d
//   shift  = Wt y + bt ; scale = Ws y + bs   (d  <- h)     GEMM 2, K = h
//   gate   = sigmoid(scale)
//   x      = (1-m) z gate + (1-gate) shift + m z ;  log_det = sum_j (1-m_j) log gate_j
//
// One wave owns 16 rows, a 512-thread workgroup 128 rows.  Both GEMMs run transposed on
// v_mfma_f32_16x16x4_f32 with the batch on the N axis, as in the AffineHalfFlow kernel: the
// 16 accumulator registers of y (h padded to 64 = 4 tiles) are directly the K-step operands of
// GEMM 2, and a GEMM-2 output tile (16 dims x 16 rows) has the lane layout of a float4 of the
// row, so gate / transform / store / log-det run from registers.  At d = 800 the operand image
// (538 KB) does not fit LDS: it is streamed from L2 through a double-buffered 2 x 16 KiB LDS
// window that the eight waves of a workgroup share; row values and operands of chunk c+1 are
// requested before chunk c is computed and handed over at one barrier per chunk.
// z and the mask are read twice (once as the GEMM-1 operand, once in the epilogue; the second
// read mostly hits L2 / Infinity Cache) and x is written once.
//
// Supported here: one hidden layer (net is a bare Linear, as MNFLinear uses it) of width 50 or 30, d >= 49
// (d % 16 != 0: the ragged variants, see row_load4); everything else runs the generic kernel.
#include <hip/hip_runtime.h>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_split.h"
#include "mnf_rnvp_common.h"

namespace mnf {

template <int HN, bool SEEDED, bool RAG>
__global__ void __launch_bounds__(kRnvpWaves * 64, 4)  // two 8-wave workgroups per CU: <= 128 VGPRs
rnvp_mfma_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x,
                 float* __restrict__ log_det, const float* __restrict__ image, int64_t rows, int d,
                 int accumulate, uint64_t seed, int dm, int vec) {
  __shared__ __attribute__((aligned(16))) float lds[2][RnvpShape<HN>::CHUNK_FLOATS];
  const int n_groups = (int)((rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves));
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x)
    rnvp_group_f32<HN, SEEDED, RAG>(lds, grp, z, mask, x, log_det, image, rows, d, accumulate, seed, nullptr, dm,
                                    vec != 0);
}

// ================================================================================================
// The same layer with both GEMMs on the f16 matrix pipe in split (hi + lo) fp32 arithmetic
// (mnf_split.h): per 16 rows at d = 800, h = 50: 900 v_mfma_f32_16x16x32_f16 of 16 cycles instead of
// 2,100 v_mfma_f32_16x16x4_f32 of 32 cycles.  Same streaming scheme (double-buffered LDS window,
// one barrier per chunk).  A 128-row group whose operands leave the f16 range -- detected after
// GEMM 1, before anything is stored -- is recomputed by rnvp_group_f32 inside the same launch.
// ================================================================================================
#ifndef MNF_RNVP_NT
#define MNF_RNVP_NT 0
#endif
constexpr bool kRnvpNtStore = (MNF_RNVP_NT & 1) != 0, kRnvpNtLoad2 = (MNF_RNVP_NT & 2) != 0;
#ifndef MNF_RNVP_ABL
#define MNF_RNVP_ABL 0  // experiments only: 1 no x stores, 2 no phase-2 row loads, 3 no phase-1 row loads, 4 no MFMAs
#endif
constexpr int kRnvpAbl = MNF_RNVP_ABL;

constexpr int kRnvpMaxPrologueDim = 1024;  // fused sample_z prologue: mean and std of up to this many dims in LDS


// returns false (block-uniform) when the group has to be recomputed on the fp32 path
// RES (round 5): the WHOLE operand image is resident in LDS (lds0; narrow layers: d <= 128 is <= 96 KB) -- no operand
// staging, no per-chunk barriers.  At d = 50 the streamed form moved 48 KB of operands from L2 through eight barriers
// for every 25 KB of rows (128 rows per workgroup trip): 123 us per launch at 256,000 rows with nothing saturated.
template <int HN, bool SEEDED, bool RAG, bool RES = false>
__device__ __forceinline__ bool rnvp_group_split(uint32_t* lds0, uint32_t* lds1, int grp, const float* __restrict__ z,
                                                 const float* __restrict__ mask, float* __restrict__ x,
                                                 float* __restrict__ log_det, const uint32_t* __restrict__ simage,
                                                 int64_t rows, int d, int accumulate, uint64_t seed,
                                                 const float* zprm, int dm_ragged, bool vec, float* __restrict__ y_out) {
  using S = RnvpSplitShape<HN>;
  const int dm = RAG ? dm_ragged : d;  // row width in memory (d: rounded up to whole 16-dim groups)
  constexpr int YT = S::YT, NKS2 = S::NKS2, KC = S::KC, MC = S::MC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int G = d / 16;                  // 16-dim groups of a row = GEMM-2 output tiles
  const int n_ks1 = (G + 1) / 2;         // GEMM-1 K-steps (two groups each)
  const int nc1 = (n_ks1 + KC - 1) / KC, nc2 = (G + MC - 1) / MC, nc = nc1 + nc2;
  const uint32_t* img1 = simage;
  const uint32_t* img2 = simage + S::part1_words(d);
  const float* bias2 = reinterpret_cast<const float*>(simage + S::split_words(d));
  const float* bias_y = bias2 + (int64_t)G * 32;

  auto chunk_src = [&](int c, int& n4) -> const uint4* {
    if (c < nc1) {
      n4 = min(KC, n_ks1 - c * KC) * (S::KS1_WORDS / 4);
      return reinterpret_cast<const uint4*>(img1 + (int64_t)c * KC * S::KS1_WORDS);
    }
    const int m0 = (c - nc1) * MC;
    n4 = min(MC, G - m0) * (S::TILE2_WORDS / 4);
    return reinterpret_cast<const uint4*>(img2 + (int64_t)m0 * S::TILE2_WORDS);
  };
  // the i-th 16-dim group chunk c works on, or -1 past the end of the row
  auto row_group = [&](int c, int i) -> int {
    const int g = c < nc1 ? 2 * (c * KC) + i : (c - nc1) * MC + i;
    return g < G ? g : -1;
  };

  const int64_t row = (int64_t)grp * (16 * kRnvpWaves) + wave * 16 + j;
  const bool live = row < rows;
  const int64_t rowc = live ? row : rows - 1;
  const float* zr = z + rowc * dm + 4 * q;
  const float* mr = SEEDED ? nullptr : mask + rowc * dm + 4 * q;
  float* xr = x + rowc * dm + 4 * q;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mask4 = [&](int g) -> f32x4 {  // mask of dims 16 g + 4 q .. + 3; groups past the row end read as 0
    if (g < 0) return zero4;
    if (!SEEDED) return row_load4<RAG>(mr, 16 * g, 4 * q, dm, vec);
    const int dd = 16 * g + 4 * q;
    const uint32_t w = rnvp_mask_word(seed, rowc, dd >> 5) >> (dd & 31);
    return f32x4{(float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u)};
  };
  // in-kernel mask as all-ones / all-zeros words (element r of the lane's float4): one v_bfe_i32 each, so that
  // m z is an AND and the  m ? z : gated  choice a v_bfi -- the float form costs a convert and a multiply more
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  auto mask_bits = [&](int g) -> i32x4 {
    if (g < 0) return i32x4{0, 0, 0, 0};
    const int dd = 16 * g + 4 * q;
    const int32_t w = (int32_t)(rnvp_mask_word(seed, rowc, dd >> 5) >> (dd & 31));
    return i32x4{(int32_t)__builtin_amdgcn_sbfe(w, 0, 1), (int32_t)__builtin_amdgcn_sbfe(w, 1, 1),
                 (int32_t)__builtin_amdgcn_sbfe(w, 2, 1), (int32_t)__builtin_amdgcn_sbfe(w, 3, 1)};
  };
  auto and_bits = [](const f32x4& v, const i32x4& m) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, v) & m);
  };
  // zprm != nullptr: the sample_z prologue fused into the loads, z = q0_mean + q0_std * eps
  auto z_of = [&](const f32x4& v, int g) -> f32x4 {
    if (zprm == nullptr) return v;
    const int dd = 16 * (g < 0 ? 0 : g) + 4 * q;
    return v * *reinterpret_cast<const f32x4*>(zprm + d + dd) + *reinterpret_cast<const f32x4*>(zprm + dd);
  };
  auto z4 = [&](int g) -> f32x4 { return z_of(row_load4<RAG>(zr, 16 * (g < 0 ? 0 : g), 4 * q, dm, vec), g); };
  const f32x4 fake4 = f32x4{0.25f, -0.5f, 0.125f, 1.f};

  // Row data (z, and the mask when it is an input) comes from HBM with ~2 us of latency under load, while
  // a chunk of split MFMAs takes well under 1 us: rows are therefore requested D chunks ahead into a ring
  // of register sets (the chunk loops are unrolled by D so that the ring index is static).  A set is
  // re-requested as soon as it has been turned into MFMA operands.  With an explicit float mask the ring
  // would need twice the registers, so that variant keeps a depth of one.
  constexpr int D1 = SEEDED ? 2 : 1;  // GEMM 1: 2 KC groups per set
  constexpr int D2 = SEEDED ? 3 : 1;  // GEMM 2: MC groups per set
  uint4 st[S::STAGE_U4];
  auto request_operands = [&](int c, int& n4) {
    if constexpr (RES) {
      n4 = 0;
    } else {
      const uint4* src = chunk_src(c < nc ? c : nc - 1, n4);
#pragma unroll
      for (int i = 0; i < S::STAGE_U4; ++i) {
        const int k = threadIdx.x + i * (kRnvpWaves * 64);
        st[i] = src[k < n4 ? k : 0];
      }
    }
  };
  auto hand_over = [&](uint32_t* buf, int n4) {
    if constexpr (!RES) {
      uint4* dst = reinterpret_cast<uint4*>(buf);
#pragma unroll
      for (int i = 0; i < S::STAGE_U4; ++i) {
        const int k = threadIdx.x + i * (kRnvpWaves * 64);
        if (k < n4) dst[k] = st[i];
      }
    }
  };
  // the LDS words chunk c's operands start at
  auto chunk_buf = [&](int c) -> const uint32_t* {
    if constexpr (RES)
      return c < nc1 ? lds0 + (int64_t)c * KC * S::KS1_WORDS
                     : lds0 + S::part1_words(d) + (int64_t)(c - nc1) * MC * S::TILE2_WORDS;
    else
      return (c & 1) ? lds1 : lds0;
  };
  f32x4 z1[D1][2 * KC], m1[SEEDED ? 1 : D1][2 * KC];
  auto request_rows1 = [&](int c, int u) {
#pragma unroll
    for (int i = 0; i < 2 * KC; ++i) {
      z1[u][i] = (kRnvpAbl == 3 || kRnvpAbl >= 5) ? fake4 : z4(row_group(c < nc1 ? c : nc1 - 1, i));
      if (!SEEDED) m1[u][i] = mask4(row_group(c < nc1 ? c : nc1 - 1, i));
    }
  };
  if constexpr (!RES) __syncthreads();  // the previous group's last chunk is fully consumed
  {
    int n4;
#pragma unroll
    for (int u = 0; u < D1; ++u) request_rows1(u, u);
    request_operands(0, n4);
    hand_over(lds0, n4);
  }
  if constexpr (!RES) __syncthreads();

  auto split_mac = [&](const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x4& mn, f32x4& cr) {
    if (kRnvpAbl == 4 || kRnvpAbl == 6) {
      mn += __builtin_bit_cast(f32x4, ah) * __builtin_bit_cast(f32x4, bh);
      cr += __builtin_bit_cast(f32x4, al) * __builtin_bit_cast(f32x4, bl);
    } else {
      mnf::split_mac(ah, al, bh, bl, mn, cr);
    }
  };
  f32x4 ym[YT], yc[YT];
#pragma unroll
  for (int m = 0; m < YT; ++m) {
    ym[m] = *reinterpret_cast<const f32x4*>(bias_y + m * 16 + 4 * q);
    yc[m] = zero4;
  }
  float mx = 0.f;
  // ---- GEMM 1: y^T += Wn[:, 32 dims] (m z)^T, two 16-dim groups per K-step
  for (int c0 = 0; c0 < nc1; c0 += D1) {
#pragma unroll
    for (int u = 0; u < D1; ++u) {
      const int c = c0 + u;
      if (c < nc1) {
        f16x8 bh[KC], bl[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          u32x2 h0, l0, h1, l1;
          if (SEEDED) {
            split_tile(and_bits(z1[u][2 * kk], mask_bits(row_group(c, 2 * kk))), h0, l0, mx);
            split_tile(and_bits(z1[u][2 * kk + 1], mask_bits(row_group(c, 2 * kk + 1))), h1, l1, mx);
          } else {
            split_tile(m1[SEEDED ? 0 : u][2 * kk] * z1[u][2 * kk], h0, l0, mx);
            split_tile(m1[SEEDED ? 0 : u][2 * kk + 1] * z1[u][2 * kk + 1], h1, l1, mx);
          }
          bh[kk] = pair_operand(h0, h1);
          bl[kk] = pair_operand(l0, l1);
        }
        int n4_next = 0;
        request_rows1(c + D1, u);
        request_operands(c + 1, n4_next);  // c + 1 == nc1: the first GEMM-2 chunk
        const uint32_t* buf = chunk_buf(c);
        const f16x8* A8 = reinterpret_cast<const f16x8*>(buf) + lane;  // + 64 * operand
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          if (c * KC + kk < n_ks1) {
#pragma unroll
            for (int m = 0; m < YT; ++m)
              split_mac(A8[64 * (2 * (kk * YT + m))], A8[64 * (2 * (kk * YT + m) + 1)], bh[kk], bl[kk], ym[m], yc[m]);
          }
        }
        hand_over((c & 1) ? lds0 : lds1, n4_next);
        // (LDS-only barrier: __syncthreads() would also wait -- vmcnt(0) -- for the row loads requested chunks ahead)
        if (!RES && c < nc1 - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
  // rows of the first GEMM-2 chunks: their latency overlaps the y split and the group-wide verdict
  f32x4 z2[D2][MC], m2[SEEDED ? 1 : D2][MC];
  auto request_rows2 = [&](int c, int u) {
#pragma unroll
    for (int i = 0; i < MC; ++i) {
      const int g2 = row_group(c < nc ? c : nc - 1, i);
      z2[u][i] = (kRnvpAbl == 2 || kRnvpAbl >= 5) ? fake4
                 : (kRnvpNtLoad2 && !RAG)
                     ? z_of(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zr + 16 * (g2 < 0 ? 0 : g2))), g2)
                     : z4(g2);
      if (!SEEDED) m2[u][i] = mask4(row_group(c < nc ? c : nc - 1, i));
    }
  };
#pragma unroll
  for (int u = 0; u < D2; ++u) request_rows2(nc1 + u, u);
  // y complete: operands of GEMM 2, and the range verdict for the whole 128-row group
  u32x2 yh[YT], yl[YT];
  const u32x2 zero2 = u32x2{0u, 0u};
#pragma unroll
  for (int m = 0; m < YT; ++m) split_tile(yc[m] * kSplitInvScale + ym[m], yh[m], yl[m], mx);
  if (__syncthreads_or(!(mx <= kSplitLimit) ? 1 : 0)) return false;  // nothing has been stored yet
  if (y_out && live) {  // training: y kept for the gradient pass (mnf_rnvp_seeded_train)
#pragma unroll
    for (int m = 0; m < YT; ++m)
      *reinterpret_cast<f32x4*>(y_out + row * (16 * YT) + 16 * m + 4 * q) = yc[m] * kSplitInvScale + ym[m];
  }

  // ---- GEMM 2 + gate, 16 output dims per tile
  float ld = 0.f, ld2 = 0.f;  // ld2: sum of log2(1 + e^-s) over the gated elements (seeded path)
  for (int c0 = nc1; c0 < nc; c0 += D2) {
#pragma unroll
    for (int u = 0; u < D2; ++u) {
      const int c = c0 + u;
      if (c < nc) {
        int n4_next = 0;
        request_operands(c + 1, n4_next);  // past the last chunk: the same chunk again (branch-free)
        const int m0 = (c - nc1) * MC;
        f32x4 bt[MC], bs[MC];
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
          const int m = m0 + mi < G ? m0 + mi : G - 1;
          bt[mi] = *reinterpret_cast<const f32x4*>(bias2 + (int64_t)m * 32 + 4 * q);
          bs[mi] = *reinterpret_cast<const f32x4*>(bias2 + (int64_t)m * 32 + 16 + 4 * q);
        }
        const uint32_t* buf = chunk_buf(c);
        const f16x8* A8 = reinterpret_cast<const f16x8*>(buf) + lane;
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
          const int m = m0 + mi;
          if (m < G) {
            const f16x8* T8 = A8 + 64 * (mi * (S::TILE2_WORDS / 256));
            f32x4 tm = zero4, tc = zero4, sm = zero4, sc = zero4;
#pragma unroll
            for (int ks = 0; ks < NKS2; ++ks) {
              const f16x8 bh = pair_operand(yh[2 * ks], 2 * ks + 1 < YT ? yh[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
              const f16x8 bl = pair_operand(yl[2 * ks], 2 * ks + 1 < YT ? yl[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
              split_mac(T8[64 * (2 * ks)], T8[64 * (2 * ks + 1)], bh, bl, tm, tc);
              split_mac(T8[64 * (2 * (NKS2 + ks))], T8[64 * (2 * (NKS2 + ks) + 1)], bh, bl, sm, sc);
            }
            const f32x4 t4 = tc * kSplitInvScale + tm + bt[mi];
            const f32x4 s4 = sc * kSplitInvScale + sm + bs[mi];
            f32x4 o;
            if (SEEDED) {
              // binary mask: x = (1 - gate) t + (m ? z : gate z);  log_det -= (1 - m) ln(1 + e^-s)   (rnvp.py:36-37;
              // the shift term reaches the kept elements too)
              const i32x4 mb = mask_bits(m);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float zz = z2[u][mi][r];
                const float den = 1.f + __builtin_amdgcn_exp2f(s4[r] * -1.44269504088896341f);
                const float gate = __builtin_amdgcn_rcpf(den);
                const int32_t mr_ = mb[r];
                const float zsel = __builtin_bit_cast(float, (mr_ & __builtin_bit_cast(int32_t, zz)) |
                                                                 (~mr_ & __builtin_bit_cast(int32_t, zz * gate)));
                o[r] = __builtin_fmaf(-gate, t4[r], t4[r]) + zsel;
                ld2 += __builtin_bit_cast(float, ~mr_ & __builtin_bit_cast(int32_t, __builtin_amdgcn_logf(den)));
              }
            } else {
              const f32x4 mk = m2[SEEDED ? 0 : u][mi];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float zz = z2[u][mi][r], mm = mk[r];
                const float gate = __builtin_amdgcn_rcpf(1.f + exp6r(-s4[r]));
                const float keep = mm * zz;                            // z2 = m z
                const float gated = (1.f - mm) * zz;                   // z1 = (1-m) z
                o[r] = (gated * gate + (1.f - gate) * t4[r]) + keep;   // rnvp.py:37
                ld += (1.f - mm) * (__builtin_amdgcn_logf(gate) * 0.693147180559945309f);  // :36
              }
            }
            if (live && ((kRnvpAbl != 1 && kRnvpAbl < 5) || o[0] == 1.2345e30f)) {
              if (kRnvpNtStore && !RAG) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(xr + 16 * m));
              else row_store4<RAG>(xr, 16 * m, 4 * q, dm, vec, o);
            }
          }
        }
        request_rows2(c + D2, u);
        hand_over((c & 1) ? lds0 : lds1, n4_next);
        if (!RES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS-only, as above: the row stores stay in flight)
      }
    }
  }
  if (log_det) {
    ld = sum_over_q(ld - 0.693147180559945309f * ld2);
    if (live && q == 0) log_det[row] = accumulate ? log_det[row] + ld : ld;
  }
  return true;
}

// out-of-line on purpose: inlined next to the split path the two bodies compete for the 128 VGPRs of
// a 4-waves/SIMD kernel and the hot path spills
template <int HN, bool SEEDED, bool RAG>
__device__ __attribute__((noinline)) void rnvp_group_f32_cold(float* lds, int grp, const float* z, const float* mask,
                                                             float* x, float* log_det, const float* image,
                                                             int64_t rows, int d, int accumulate, uint64_t seed,
                                                             const float* zprm, int dm, bool vec) {
  rnvp_group_f32<HN, SEEDED, RAG>(*reinterpret_cast<float(*)[2][RnvpShape<HN>::CHUNK_FLOATS]>(lds), grp, z, mask, x,
                                  log_det, image, rows, d, accumulate, seed, zprm, dm, vec);
}

// ================================================================================================
// Narrow layers (d <= 64 after padding: MNFLinear(50, 10)'s flows, BASELINE configs[4]'s second layer): a LATENCY kernel.
// The streamed kernel above gives a 128-row group to a workgroup and walks it chunk by chunk behind barriers -- built
// for d = 800, where the operand image does not fit LDS.  At d = 50 the image is 48 KB and a row 200 bytes: there the
// launch (123-131 us at 256,000 rows, nothing saturated: vector units 0.24, matrix pipe 0.06) was a chain of exposed
// memory round trips per group with eight waves per CU to hide them.  Here the image is resident in LDS, SIXTEEN waves
// per CU each own 16-row tiles outright -- no barrier in the tile loop, <= 128 registers: the row is 16 registers, no
// prefetch rings --, and the latency of one wave's loads is covered by the other fifteen.  A tile whose operands leave
// the split range is noted in LDS and redone on the fp32 body (rnvp_group_f32, stores enabled for that tile's wave
// only) after the loop, with the image's LDS as its staging windows.
// ================================================================================================
constexpr int kNarrowWaves = 16;
constexpr int kNarrowMaxTrips = 1024;  // tile-loop trips of a workgroup (one 16-bit cold mask each); more rows: the streamed kernel

template <int HN, bool SEEDED>
__global__ void __launch_bounds__(kNarrowWaves * 64, 1)
rnvp_narrow_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x,
                   float* __restrict__ log_det, const uint32_t* __restrict__ simage, const float* __restrict__ image,
                   int64_t rows, int d, int accumulate, uint64_t seed, const float* __restrict__ q0_mean,
                   const float* __restrict__ q0_log_var, int dm, int vec_ok) {
  using S = RnvpSplitShape<HN>;
  constexpr int YT = S::YT, NKS2 = S::NKS2;
  extern __shared__ __attribute__((aligned(16))) uint32_t img_lds[];  // the split image; later the fp32 body's windows
  __shared__ __attribute__((aligned(16))) float zprm_lds[2 * 64];
  // trip t of the tile loop = the 256-row group blockIdx.x + t gridDim.x, wave w its tile w: bit w of cold[t] = that tile
  // left the split range
  __shared__ uint32_t cold[kNarrowMaxTrips];
  const bool vec = vec_ok != 0;
  const int G = d / 16;  // <= 4
  const float* zprm = nullptr;
  if (q0_mean != nullptr) {  // fused sample_z prologue: z = q0_mean + sqrt(exp(q0_log_var)) eps (mnf_linear.py:59-62)
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      zprm_lds[i] = i < dm ? q0_mean[i] : 0.f;
      zprm_lds[d + i] = i < dm ? sqrtf(expf(q0_log_var[i])) : 0.f;
    }
    zprm = zprm_lds;
  }
  for (int i = threadIdx.x; i < kNarrowMaxTrips; i += blockDim.x) cold[i] = 0u;
  {
    const uint4* src = reinterpret_cast<const uint4*>(simage);
    uint4* dst = reinterpret_cast<uint4*>(img_lds);
    for (int i = threadIdx.x; i < (int)(S::split_words(d) / 4); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const float wmax = __builtin_bit_cast(float, simage[S::split_words(d) + S::plain_words(d)]);
  const bool split_ok = wmax <= kSplitWeightLimit;  // (false: every tile goes to the fp32 body below)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n_ks1 = (G + 1) / 2;
  const float* bias2 = reinterpret_cast<const float*>(simage + S::split_words(d));
  const float* bias_y = bias2 + (int64_t)G * 32;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  int a_off = lane * 4;
  asm volatile("" : "+v"(a_off));  // keep the operand reads inside the tile loop
  const f16x8* A1 = reinterpret_cast<const f16x8*>(img_lds + a_off);                        // + 64 * (2 (ks YT + m) + part)
  const f16x8* A2 = reinterpret_cast<const f16x8*>(img_lds + a_off + S::part1_words(d));    // + 64 * (m TILE2 / 256 + ...)

  const int64_t n_tiles = (rows + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * kNarrowWaves;
  int trip = 0;
  for (int64_t tile = (int64_t)blockIdx.x * kNarrowWaves + wave; split_ok && tile < n_tiles; tile += stride, ++trip) {
    const int64_t row = tile * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* zr = z + rowc * dm + 4 * q;
    float* xr = x + rowc * dm + 4 * q;
    // the row (and the mask when it is an input): G float4 groups per lane, dims 16 g + 4 q .. + 3
    f32x4 zz[4], mk[4];
    i32x4 mb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < G) {
        f32x4 v = row_load4<true>(zr, 16 * g, 4 * q, dm, vec);
        if (zprm) {
          const int dd = 16 * g + 4 * q;
          v = v * *reinterpret_cast<const f32x4*>(zprm + d + dd) + *reinterpret_cast<const f32x4*>(zprm + dd);
        }
        zz[g] = v;
        if (SEEDED) {
          const int dd = 16 * g + 4 * q;
          const int32_t w = (int32_t)(rnvp_mask_word(seed, rowc, dd >> 5) >> (dd & 31));
          mb[g] = i32x4{(int32_t)__builtin_amdgcn_sbfe(w, 0, 1), (int32_t)__builtin_amdgcn_sbfe(w, 1, 1),
                        (int32_t)__builtin_amdgcn_sbfe(w, 2, 1), (int32_t)__builtin_amdgcn_sbfe(w, 3, 1)};
        } else {
          mk[g] = row_load4<true>(mask + rowc * dm + 4 * q, 16 * g, 4 * q, dm, vec);
        }
      } else {
        zz[g] = zero4;
        mk[g] = zero4;
        mb[g] = i32x4{0, 0, 0, 0};
      }
    }
    // ---- GEMM 1: y^T = Wn (m z)^T + bn, two 16-dim groups per K = 32 step
    float mx = 0.f;
    u32x2 kh[4], kl[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 kept = SEEDED ? __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, zz[g]) & mb[g]) : mk[g] * zz[g];
      split_tile(kept, kh[g], kl[g], mx);
    }
    f32x4 ym[YT], yc[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) {
      ym[m] = *reinterpret_cast<const f32x4*>(bias_y + m * 16 + 4 * q);
      yc[m] = zero4;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < n_ks1) {
        const f16x8 bh = pair_operand(kh[2 * ks], kh[2 * ks + 1]), bl = pair_operand(kl[2 * ks], kl[2 * ks + 1]);
#pragma unroll
        for (int m = 0; m < YT; ++m)
          split_mac(A1[64 * (2 * (ks * YT + m))], A1[64 * (2 * (ks * YT + m) + 1)], bh, bl, ym[m], yc[m]);
      }
    }
    u32x2 yh[YT], yl[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) split_tile(yc[m] * kSplitInvScale + ym[m], yh[m], yl[m], mx);
    if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {  // nothing has been stored yet: the fp32 body redoes the tile
      if (lane == 0) atomicOr(&cold[trip], 1u << wave);
      continue;
    }
    // ---- GEMM 2 + gate, 16 output dims per tile
    float ld = 0.f, ld2 = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < G) {
        const f16x8* T8 = A2 + 64 * (m * (S::TILE2_WORDS / 256));
        f32x4 tm = zero4, tc = zero4, sm = zero4, sc = zero4;
#pragma unroll
        for (int ks = 0; ks < NKS2; ++ks) {
          const f16x8 bh = pair_operand(yh[2 * ks], 2 * ks + 1 < YT ? yh[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
          const f16x8 bl = pair_operand(yl[2 * ks], 2 * ks + 1 < YT ? yl[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
          split_mac(T8[64 * (2 * ks)], T8[64 * (2 * ks + 1)], bh, bl, tm, tc);
          split_mac(T8[64 * (2 * (NKS2 + ks))], T8[64 * (2 * (NKS2 + ks) + 1)], bh, bl, sm, sc);
        }
        const f32x4 t4 = tc * kSplitInvScale + tm + *reinterpret_cast<const f32x4*>(bias2 + m * 32 + 4 * q);
        const f32x4 s4 = sc * kSplitInvScale + sm + *reinterpret_cast<const f32x4*>(bias2 + m * 32 + 16 + 4 * q);
        f32x4 o;
        if (SEEDED) {
          // binary mask: x = (1 - gate) t + (m ? z : gate z);  log_det -= (1 - m) ln(1 + e^-s)   (rnvp.py:36-37)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = zz[m][r];
            const float den = 1.f + __builtin_amdgcn_exp2f(s4[r] * -1.44269504088896341f);
            const float gate = __builtin_amdgcn_rcpf(den);
            const int32_t mr_ = mb[m][r];
            const float zsel = __builtin_bit_cast(float, (mr_ & __builtin_bit_cast(int32_t, v)) |
                                                             (~mr_ & __builtin_bit_cast(int32_t, v * gate)));
            o[r] = __builtin_fmaf(-gate, t4[r], t4[r]) + zsel;
            ld2 += __builtin_bit_cast(float, ~mr_ & __builtin_bit_cast(int32_t, __builtin_amdgcn_logf(den)));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = zz[m][r], mm = mk[m][r];
            const float gate = __builtin_amdgcn_rcpf(1.f + exp6r(-s4[r]));
            o[r] = (((1.f - mm) * v) * gate + (1.f - gate) * t4[r]) + mm * v;                 // rnvp.py:37
            ld += (1.f - mm) * (__builtin_amdgcn_logf(gate) * 0.693147180559945309f);          // :36
          }
        }
        if (live) row_store4<true>(xr, 16 * m, 4 * q, dm, vec, o);
      }
    }
    if (log_det) {
      ld = sum_over_q(ld - 0.693147180559945309f * ld2);
      if (live && q == 0) log_det[row] = accumulate ? log_det[row] + ld : ld;
    }
  }
  // ---- tiles that left the split range (or every tile when the weights did): the fp32 body on the trip's 256-row group
  // with this workgroup's 16 waves, stores enabled for the noted tiles' waves only
  __syncthreads();
  float(&win)[2][RnvpShape<HN>::CHUNK_FLOATS] = *reinterpret_cast<float(*)[2][RnvpShape<HN>::CHUNK_FLOATS]>(img_lds);
  const int n_groups = (int)((rows + 16 * kNarrowWaves - 1) / (16 * kNarrowWaves));
  int t = 0;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x, ++t) {
    const uint32_t bits = split_ok ? cold[t] : 0xffffu;  // (block-uniform)
    if (bits)
      rnvp_group_f32<HN, SEEDED, true, kNarrowWaves>(win, grp, z, mask, x, log_det, image, rows, d, accumulate, seed, zprm,
                                                     dm, vec, ((bits >> wave) & 1u) != 0);
  }
}

// OCC = waves per SIMD the register allocation aims at.  4 (<= 128 registers, two 8-wave workgroups per CU, 50-140
// registers spilled) or 2 (no spills, one workgroup per CU), chosen per launch -- measured at 256,000 rows, seeded mask,
// us per launch at OCC 4 / 2: ragged rows d = 50: 146 / 123, 96: 118 / 114, 200: 274 / 269, 400: 387 / 397, 799: 1,518 /
// 1,250; whole 16-dim groups d = 64: 89 / 80, 128: 151 / 140, 256: 254 / 272, 512: 474 / 481, 784: 754 / 734, 800: 714 /
// 777, 1,024: 919 / 1,471.  (The spills are scratch traffic: at d = 50 the launch moved 241 MB of writes for a 51 MB
// output, `rocprofv3 --pmc WRITE_SIZE`.)  So: 2 for ragged rows and for d <= 128, else 4.
template <int HN, bool SEEDED, bool RAG, int OCC, bool RES = false>
__global__ void __launch_bounds__(kRnvpWaves * 64, OCC)
rnvp_split_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x,
                  float* __restrict__ log_det, const uint32_t* __restrict__ simage, const float* __restrict__ image,
                  int64_t rows, int d, int accumulate, uint64_t seed, const float* __restrict__ q0_mean,
                  const float* __restrict__ q0_log_var, int dm_ragged, int vec_ok, float* __restrict__ y_out) {
  const int dm = RAG ? dm_ragged : d;
  const bool vec = vec_ok != 0;
  using S = RnvpSplitShape<HN>;
  using F = RnvpShape<HN>;
  constexpr int WORDS = S::CHUNK_WORDS > F::CHUNK_FLOATS ? S::CHUNK_WORDS : F::CHUNK_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[2][WORDS];
  // fused sample_z prologue (q0_mean != nullptr: `z` holds eps): mean and std = sqrt(exp(log_var)) per dim
  __shared__ __attribute__((aligned(16))) float zprm_lds[2 * kRnvpMaxPrologueDim];
  const float* zprm = nullptr;
  if (q0_mean != nullptr) {
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      zprm_lds[i] = i < dm ? q0_mean[i] : 0.f;
      zprm_lds[d + i] = i < dm ? sqrtf(expf(q0_log_var[i])) : 0.f;  // mnf_linear.py:60
    }
    zprm = zprm_lds;
    __syncthreads();
  }
  // weights outside the f16 range (flagged by the pack kernel): every group on the fp32 path
  const float wmax = __builtin_bit_cast(float, simage[S::split_words(d) + S::plain_words(d)]);
  const bool split_ok = wmax <= kSplitWeightLimit;
  extern __shared__ __attribute__((aligned(16))) uint32_t res_lds[];  // RES: the whole split operand image
  if constexpr (RES) {
    const uint4* src = reinterpret_cast<const uint4*>(simage);
    uint4* dst = reinterpret_cast<uint4*>(res_lds);
    for (int i = threadIdx.x; i < (int)(S::split_words(d) / 4); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  const int n_groups = (int)((rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves));
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    if (split_ok && rnvp_group_split<HN, SEEDED, RAG, RES>(RES ? res_lds : reinterpret_cast<uint32_t*>(lds[0]),
                                                           reinterpret_cast<uint32_t*>(lds[1]), grp, z, mask, x, log_det,
                                                           simage, rows, d, accumulate, seed, zprm, dm, vec, y_out))
      continue;
    if constexpr (RES) __syncthreads();  // (the fp32 body opens with a barrier of its own; kept explicit)
    rnvp_group_f32_cold<HN, SEEDED, RAG>(&lds[0][0], grp, z, mask, x, log_det, image, rows, d, accumulate, seed, zprm,
                                         dm, vec);
    if (y_out) {  // no y from the fp32 body: NaN rows make the gradient pass's launch A flag the group for its fix-up
      constexpr int W = 16 * S::YT, GR = 16 * kRnvpWaves;
      for (int i = threadIdx.x; i < GR * W; i += blockDim.x) {
        const int64_t r = (int64_t)grp * GR + i / W;
        if (r < rows) y_out[r * W + i % W] = __builtin_nanf("");
      }
    }
  }
}

// narrow layers (padded d <= 64, no y to keep): the latency kernel
template <int HN>
static int launch_rnvp_narrow(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                              const uint32_t* simage, const float* image, int64_t rows, int dim, uint64_t seed,
                              const float* q0_mean, const float* q0_log_var, int dm, int vec, hipStream_t stream) {
  using S = RnvpSplitShape<HN>;
  const int64_t n_groups = (rows + 16 * kNarrowWaves - 1) / (16 * kNarrowWaves);
  const int cus = device_cus(current_device());
  const int64_t blocks = n_groups < cus ? n_groups : cus;
  if ((n_groups + blocks - 1) / blocks > kNarrowMaxTrips) return MNF_ERR_UNSUPPORTED;
  const int64_t img_bytes = S::split_words(dim) * 4, win_bytes = 2 * (int64_t)RnvpShape<HN>::CHUNK_FLOATS * 4;
  const int lds_bytes = (int)(img_bytes > win_bytes ? img_bytes : win_bytes);
  typedef void (*Kern)(const float*, const float*, float*, float*, const uint32_t*, const float*, int64_t, int, int,
                       uint64_t, const float*, const float*, int, int);
  const Kern kern = mask ? static_cast<Kern>(rnvp_narrow_kernel<HN, false>) : static_cast<Kern>(rnvp_narrow_kernel<HN, true>);
  static DeviceMemo memo[2];
  const int ok = memo[mask ? 0 : 1].get([&](int) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               64 * 1024) == hipSuccess ? 1 : -1;
  });
  if (ok < 0) return MNF_ERR_UNSUPPORTED;
  tag_kernel("rnvp_narrow");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kNarrowWaves * 64), lds_bytes, stream, z, mask, x, log_det, simage,
                     image, rows, dim, accumulate, seed, q0_mean, q0_log_var, dm, vec);
  return check_launch();
}

// narrow layers (padded d <= 128): the operand image resident in LDS, one workgroup per CU
template <int HN, bool RAG>
static int launch_rnvp_split_res(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                                 const uint32_t* simage, const float* image, int64_t rows, int dim, uint64_t seed,
                                 const float* q0_mean, const float* q0_log_var, int dm, int vec, hipStream_t stream,
                                 float* y_out) {
  using S = RnvpSplitShape<HN>;
  const int64_t n_groups = (rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves);
  const int lds_bytes = (int)S::split_words(dim) * 4;
  typedef void (*Kern)(const float*, const float*, float*, float*, const uint32_t*, const float*, int64_t, int, int,
                       uint64_t, const float*, const float*, int, int, float*);
  const Kern kern = mask ? static_cast<Kern>(rnvp_split_kernel<HN, false, RAG, 2, true>)
                         : static_cast<Kern>(rnvp_split_kernel<HN, true, RAG, 2, true>);
  static DeviceMemo memo[2];
  const int ok = memo[mask ? 0 : 1].get([&](int) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               96 * 1024) == hipSuccess ? 1 : -1;
  });
  if (ok < 0) return MNF_ERR_UNSUPPORTED;
  const int cus = device_cus(current_device());
  const int64_t blocks = n_groups < cus ? n_groups : cus;
  tag_kernel("rnvp_split_resident_operands");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kRnvpWaves * 64), lds_bytes, stream, z, mask, x, log_det, simage,
                     image, rows, dim, accumulate, seed, q0_mean, q0_log_var, dm, vec, y_out);
  return check_launch();
}

template <int HN, bool RAG, int OCC>
static int launch_rnvp_split_occ(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                                 const uint32_t* simage, const float* image, int64_t rows, int dim, uint64_t seed,
                                 const float* q0_mean, const float* q0_log_var, int dm, int vec, hipStream_t stream,
                                 float* y_out) {
  const int64_t n_groups = (rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves);
  static DeviceMemo memo_mask, memo_seed;
  const int resident_mask = memo_mask.get(
      [](int dev) { return resident_by_occupancy(rnvp_split_kernel<HN, false, RAG, OCC>, kRnvpWaves * 64, dev, 1); });
  const int resident_seed = memo_seed.get(
      [](int dev) { return resident_by_occupancy(rnvp_split_kernel<HN, true, RAG, OCC>, kRnvpWaves * 64, dev, 1); });
  const int resident = mask ? resident_mask : resident_seed;
  const int64_t blocks = n_groups < resident ? n_groups : resident;
  tag_kernel("rnvp_split");
  if (mask)
    hipLaunchKernelGGL((rnvp_split_kernel<HN, false, RAG, OCC>), dim3((unsigned)blocks), dim3(kRnvpWaves * 64), 0, stream,
                       z, mask, x, log_det, simage, image, rows, dim, accumulate, seed, q0_mean, q0_log_var, dm, vec, y_out);
  else
    hipLaunchKernelGGL((rnvp_split_kernel<HN, true, RAG, OCC>), dim3((unsigned)blocks), dim3(kRnvpWaves * 64), 0, stream,
                       z, mask, x, log_det, simage, image, rows, dim, accumulate, seed, q0_mean, q0_log_var, dm, vec, y_out);
  return check_launch();
}

template <int HN, bool RAG>
static int launch_rnvp_split(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                             const uint32_t* simage, const float* image, int64_t rows, int dim, uint64_t seed,
                             const float* q0_mean, const float* q0_log_var, int dm, int vec, hipStream_t stream,
                             float* y_out = nullptr) {
  constexpr int forced = 0;  // (2 | 4: one register target for rows of whole 16-dim groups, for A/B measurements)
  if (dim <= 64 && !forced && !y_out) {
    const int rc = launch_rnvp_narrow<HN>(z, mask, x, log_det, accumulate, simage, image, rows, dim, seed, q0_mean,
                                          q0_log_var, dm, vec, stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if (dim <= 128 && !forced) {
    const int rc = launch_rnvp_split_res<HN, RAG>(z, mask, x, log_det, accumulate, simage, image, rows, dim, seed, q0_mean,
                                                  q0_log_var, dm, vec, stream, y_out);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  const bool two = forced == 2 || (forced != 4 && (RAG || dim <= 128));
  if constexpr (RAG) {
    return launch_rnvp_split_occ<HN, true, 2>(z, mask, x, log_det, accumulate, simage, image, rows, dim, seed, q0_mean,
                                              q0_log_var, dm, vec, stream, y_out);
  } else {
    if (two)
      return launch_rnvp_split_occ<HN, false, 2>(z, mask, x, log_det, accumulate, simage, image, rows, dim, seed, q0_mean,
                                                 q0_log_var, dm, vec, stream, y_out);
    return launch_rnvp_split_occ<HN, false, 4>(z, mask, x, log_det, accumulate, simage, image, rows, dim, seed, q0_mean,
                                               q0_log_var, dm, vec, stream, y_out);
  }
}

// ---------------------------------------------------------------- host: image index table
template <int HN>
static void build_index(int dm, int d, int32_t* idx, int hn = HN) {
  using S = RnvpShape<HN>;
  constexpr int KQ = S::KQ, YT = S::YT;
  // flat layout: net.0.weight (hn, dm), net.0.bias (hn), t.weight (dm, hn), t.bias (dm), s.weight (dm, hn), s.bias (dm)
  const int64_t wn = 0, bn = wn + (int64_t)hn * dm, wt = bn + hn, bt = wt + (int64_t)dm * hn, ws = bt + dm,
                bs = ws + (int64_t)dm * hn;
  const int64_t total = S::image_floats(d);
  for (int64_t i = 0; i < total; ++i) idx[i] = -1;
  auto unit_of = [&](int m, int i) { return 16 * m + 4 * (i & 3) + (i >> 2); };
  // part 1: group per K-step kk; component = y tile m
  for (int kk = 0; kk < d / 4; ++kk) {
    const int g = kk >> 2, e = kk & 3;
    for (int m = 0; m < YT; ++m)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, u = unit_of(m, i);
        const int col = 16 * g + 4 * kq + e;
        if (u < hn && col < dm) idx[(int64_t)kk * 256 + lane * 4 + m] = (int32_t)(wn + (int64_t)u * dm + col);
      }
  }
  // part 2: per output tile m: sequence n = 2 c + which
  int32_t* p2 = idx + S::part1_floats(d);
  for (int m = 0; m < d / 16; ++m)
    for (int c = 0; c < KQ; ++c)
      for (int which = 0; which < 2; ++which) {
        const int n = 2 * c + which;
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, unit = 4 * c + kq;
          if (unit < hn && 16 * m + i < dm)
            p2[(int64_t)m * S::TILE2_FLOATS + (n >> 2) * 256 + lane * 4 + (n & 3)] =
                (int32_t)((which ? ws : wt) + (int64_t)(16 * m + i) * hn + unit);
        }
      }
  for (int m = 0; m < d / 16; ++m)
    for (int i = 0; i < 16; ++i) {
      const bool real = 16 * m + i < dm;
      p2[(int64_t)m * S::TILE2_FLOATS + S::G2 * 256 + i] = real ? (int32_t)(bt + 16 * m + i) : -1;
      p2[(int64_t)m * S::TILE2_FLOATS + S::G2 * 256 + 16 + i] = real ? (int32_t)(bs + 16 * m + i) : kPackBigBias;
    }
  int32_t* pb = p2 + S::part2_floats(d);
  for (int m = 0; m < YT; ++m)
    for (int i = 0; i < 16; ++i)
      if (unit_of(m, i) < hn) pb[m * 16 + i] = (int32_t)(bn + unit_of(m, i));
}

template <int HN, bool RAG>
static int launch_rnvp(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                       const float* image, int64_t rows, int dim, uint64_t seed, int dm, int vec, hipStream_t stream) {
  const int64_t n_groups = (rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves);
  static DeviceMemo memo_mask, memo_seed;
  const int resident_mask = memo_mask.get(
      [](int dev) { return resident_by_occupancy(rnvp_mfma_kernel<HN, false, RAG>, kRnvpWaves * 64, dev, 1); });
  const int resident_seed = memo_seed.get(
      [](int dev) { return resident_by_occupancy(rnvp_mfma_kernel<HN, true, RAG>, kRnvpWaves * 64, dev, 1); });
  const int resident = mask ? resident_mask : resident_seed;
  const int64_t blocks = n_groups < resident ? n_groups : resident;
  tag_kernel("rnvp_mfma_fp32");
  if (mask)
    hipLaunchKernelGGL((rnvp_mfma_kernel<HN, false, RAG>), dim3((unsigned)blocks), dim3(kRnvpWaves * 64), 0, stream, z,
                       mask, x, log_det, image, rows, dim, accumulate, seed, dm, vec);
  else
    hipLaunchKernelGGL((rnvp_mfma_kernel<HN, true, RAG>), dim3((unsigned)blocks), dim3(kRnvpWaves * 64), 0, stream, z,
                       mask, x, log_det, image, rows, dim, accumulate, seed, dm, vec);
  return check_launch();
}

int rnvp_mfma_launch(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                     const float* image, const void* split_image, int64_t rows, int dim, int n_hidden,
                     const int* hidden, uint64_t seed, hipStream_t stream, const float* q0_mean,
                     const float* q0_log_var, float* y_out, int* y_written) {
  if (y_written) *y_written = 0;
  if (!rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
  const int d = rnvp_padded_dim(dim);
  if (q0_mean && (!split_image || !q0_log_var || d > kRnvpMaxPrologueDim)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(split_image)) & 15) return MNF_ERR_UNSUPPORTED;
  const bool rows_aligned =
      ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  const bool ragged = d != dim;
  const int hn_pad = rnvp_padded_hidden(n_hidden, hidden);
  if (!ragged && !rows_aligned) return MNF_ERR_UNSUPPORTED;
  const int vec = rows_aligned && (dim & 3) == 0;
  if (split_image && !mask && rows_aligned && y_out) {  // training: the kernel that can keep y for the gradient pass
    const int rc = rnvp_resident_launch(z, x, log_det, accumulate, split_image, image, rows, dim, hn_pad, seed, q0_mean,
                                        q0_log_var, vec, stream, y_out);
    if (rc != MNF_ERR_UNSUPPORTED) {
      if (rc == MNF_OK && y_written) *y_written = 1;
      return rc;
    }
  }
  if (split_image && !mask && rows_aligned && !y_out) {  // in-kernel mask: the register-resident kernels where they exist
    const int rc = rnvp_resident_launch(z, x, log_det, accumulate, split_image, image, rows, dim, hn_pad, seed, q0_mean, q0_log_var,
                              vec, stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if (split_image) {
    const uint32_t* simage = static_cast<const uint32_t*>(split_image);
    if (y_out && y_written) *y_written = 1;  // (every kernel below keeps y when asked)
#define X(HN)                                                                                                        \
  if (hn_pad == HN)                                                                                                  \
    return ragged ? launch_rnvp_split<HN, true>(z, mask, x, log_det, accumulate, simage, image, rows, d, seed,       \
                                                q0_mean, q0_log_var, dim, vec, stream, y_out)                        \
                  : launch_rnvp_split<HN, false>(z, mask, x, log_det, accumulate, simage, image, rows, d, seed,      \
                                                 q0_mean, q0_log_var, dim, vec, stream, y_out);
    MNF_RNVP_HIDDEN(X)
#undef X
    if (y_written) *y_written = 0;
  }
#define X(HN)                                                                                                        \
  if (hn_pad == HN)                                                                                                  \
    return ragged ? launch_rnvp<HN, true>(z, mask, x, log_det, accumulate, image, rows, d, seed, dim, vec, stream)   \
                  : launch_rnvp<HN, false>(z, mask, x, log_det, accumulate, image, rows, d, seed, dim, vec, stream);
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf

extern "C" {

int mnf_rnvp_y_floats_per_row(int n_hidden, const int* hidden) {
  if (n_hidden < 1 || !mnf::hidden_ok(n_hidden, hidden)) return 0;
  const int hn = mnf::rnvp_padded_hidden(n_hidden, hidden);
  return hn > 0 ? 16 * ((hn + 15) / 16) : 0;
}

int64_t mnf_rnvp_image_floats(int dim, int n_hidden, const int* hidden) {
  if (!mnf::rnvp_shape_ok(dim, n_hidden, hidden)) return 0;
#define X(HN) if (mnf::rnvp_padded_hidden(n_hidden, hidden) == HN) return mnf::RnvpShape<HN>::image_floats(mnf::rnvp_padded_dim(dim));
  MNF_RNVP_HIDDEN(X)
#undef X
  return 0;
}

int mnf_rnvp_split_layout(int dim, int n_hidden, const int* hidden, int64_t* n_split_words, int64_t* n_plain_words) {
  if (!n_split_words || !n_plain_words) return MNF_ERR_INVALID_ARG;
  if (!mnf::rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
#define X(HN)                                                        \
  if (mnf::rnvp_padded_hidden(n_hidden, hidden) == HN) {             \
    *n_split_words = mnf::RnvpSplitShape<HN>::split_words(mnf::rnvp_padded_dim(dim));  \
    *n_plain_words = mnf::RnvpSplitShape<HN>::plain_words(mnf::rnvp_padded_dim(dim));  \
    return MNF_OK;                                                   \
  }
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_rnvp_split_index(int dim, int n_hidden, const int* hidden, int32_t* idx_host) {
  if (!idx_host) return MNF_ERR_INVALID_ARG;
  if (!mnf::rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
#define X(HN)                                      \
  if (mnf::rnvp_padded_hidden(n_hidden, hidden) == HN) {  \
    mnf::build_split_index<HN>(dim, mnf::rnvp_padded_dim(dim), idx_host, hidden[0]);  \
    return MNF_OK;                                 \
  }
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_rnvp_image_index(int dim, int n_hidden, const int* hidden, int32_t* idx_host) {
  if (!idx_host) return MNF_ERR_INVALID_ARG;
  if (!mnf::rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
#define X(HN)                               \
  if (mnf::rnvp_padded_hidden(n_hidden, hidden) == HN) {  \
    mnf::build_index<HN>(dim, mnf::rnvp_padded_dim(dim), idx_host, hidden[0]);  \
    return MNF_OK;                          \
  }
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
