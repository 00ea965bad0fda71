// Device code shared by the RNVP kernels (mnf_rnvp_mfma.hip: streaming kernels; mnf_rnvp_resident.hip: the
// register-resident kernel): operand-image shapes, ragged row accesses, and the fp32-MFMA group body that also
// serves as the range-guard path of the split kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "mnf_device.h"
#include "mnf_split.h"

namespace mnf {

constexpr int kRnvpWaves = 8;    // 512-thread workgroups: 128 rows share every staged operand chunk
constexpr int kRnvpChunkK = 16;  // GEMM-1 K-steps per chunk (64 dims): 16 x 4 tiles x 256 B = 16 KiB
constexpr int kRnvpChunkM = 2;   // GEMM-2 output tiles per chunk

template <int HN, int WAVES = 8>
struct RnvpShape {
  static constexpr int KQ = (HN + 3) / 4;       // K-steps of GEMM 2 (quads of y units)
  static constexpr int YT = (KQ + 3) / 4;       // 16-row tiles of y
  static_assert(YT >= 1 && YT <= 4, "GEMM-1 operand groups hold up to four y tiles (hidden width <= 64)");
  static constexpr int G2 = (2 * KQ + 3) / 4;   // operand groups (of 4 MFMAs) per GEMM-2 output tile
  static constexpr int TILE2_FLOATS = G2 * 256 + 32;  // operands, then the tile's t and s biases (16 + 16)
  static constexpr int64_t part1_floats(int d) { return (int64_t)(d / 4) * 256; }         // one group per K-step
  static constexpr int64_t part2_floats(int d) { return (int64_t)(d / 16) * TILE2_FLOATS; }
  static constexpr int64_t bias_floats(int) { return YT * 16; }
  static constexpr int64_t image_floats(int d) { return part1_floats(d) + part2_floats(d) + bias_floats(d); }
  static constexpr int CHUNK_FLOATS =
      (kRnvpChunkK * 256 > kRnvpChunkM * TILE2_FLOATS ? kRnvpChunkK * 256 : kRnvpChunkM * TILE2_FLOATS);
  static constexpr int STAGE_F4 = (CHUNK_FLOATS / 4 + WAVES * 64 - 1) / (WAVES * 64);  // float4 per thread
};

// Ragged rows (RAG): the row in memory is `dm` floats wide, the kernels work on d = dm rounded up to a multiple of
// 16 -- the operand images are zero in the padded columns (and carry kPackBigBias as the padded scale bias, so
// gate = 1 and log gate = 0 there: nothing reaches log_det).  Loads return 0 past the row end, stores skip it;
// `vec`: dm % 4 == 0 and 16-byte aligned bases keep the 16-byte access, else element by element.
// (`rowq` = row start + 4 q, the lane's own float4 column inside a 16-dim group; `col` = 16 g; `q4` = 4 q)
// Ragged rows (RAG: dm is not a multiple of 16): the loads are UNCONDITIONAL -- a piece past the row's end reads the
// row's first piece instead and is zeroed by a select (round 5: as `if (in range) load` every piece was a load under a
// branch, behind which hipcc's wait counts fall back to vmcnt(0): at d = 50 a wave spent 55 % of its time waiting).
// vec: rows and dm multiples of 4 floats (one 16-byte piece); else, dm even (rows 8-byte aligned): two 8-byte pieces;
// else four dwords.
typedef float f32x2_row __attribute__((ext_vector_type(2)));
template <bool RAG>
__device__ __forceinline__ f32x4 row_load4(const float* rowq, int col, int q4, int dm, bool vec) {
  if (!RAG) return *reinterpret_cast<const f32x4*>(rowq + col);
  const float* row0 = rowq - q4;  // the row's first element: always readable
  if (vec) {
    const bool ok = col + q4 < dm;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? rowq + col : row0);
    return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if ((dm & 1) == 0 && (reinterpret_cast<uintptr_t>(row0) & 7) == 0) {
    const bool ok0 = col + q4 < dm, ok1 = col + q4 + 2 < dm;
    const f32x2_row a = *reinterpret_cast<const f32x2_row*>(ok0 ? rowq + col : row0);
    const f32x2_row b = *reinterpret_cast<const f32x2_row*>(ok1 ? rowq + col + 2 : row0);
    return f32x4{ok0 ? a[0] : 0.f, ok0 ? a[1] : 0.f, ok1 ? b[0] : 0.f, ok1 ? b[1] : 0.f};
  }
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = col + q4 + r < dm;
    const float e = *(ok ? rowq + col + r : row0);
    v[r] = ok ? e : 0.f;
  }
  return v;
}
template <bool RAG>
__device__ __forceinline__ void row_store4(float* rowq, int col, int q4, int dm, bool vec, const f32x4& v) {
  if (!RAG) {
    *reinterpret_cast<f32x4*>(rowq + col) = v;
  } else if (vec) {
    if (col + q4 < dm) *reinterpret_cast<f32x4*>(rowq + col) = v;
  } else if ((dm & 1) == 0 && (reinterpret_cast<uintptr_t>(rowq - q4) & 7) == 0) {
    if (col + q4 < dm) *reinterpret_cast<f32x2_row*>(rowq + col) = f32x2_row{v[0], v[1]};
    if (col + q4 + 2 < dm) *reinterpret_cast<f32x2_row*>(rowq + col + 2) = f32x2_row{v[2], v[3]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col + q4 + r < dm) rowq[col + r] = v[r];
  }
}

// 1/(1+exp(-v)) with the 6-instruction exp of the AffineHalfFlow kernel
__device__ __forceinline__ float exp6r(float x) {
  const float c_hi = 1.44269502162933349609375f, c_lo = 1.925963033500011e-08f, ln2 = 0.693147182464599609375f;
  const float t = x * c_hi;
  const float err = __builtin_fmaf(x, c_hi, -t);
  const float tl = __builtin_fmaf(x, c_lo, err);
  return __builtin_amdgcn_exp2f(t) * __builtin_fmaf(tl, ln2, 1.0f);
}

// The operand image is streamed through a double-buffered LDS window.  Per chunk c every thread
//   1. requests the rows' z (and mask) values chunk c+1 will need and its share of chunk c+1's
//      operands into registers,
//   2. computes chunk c out of LDS buffer c&1 with the z values requested one chunk earlier,
//   3. writes the staged operands into buffer (c+1)&1 and meets the others at ONE barrier.
// So neither HBM/L2 latency (rows, operands) nor the LDS fill is on the MFMA chain's critical path.
// SEEDED: the mask is regenerated from (seed, row, dim) wherever it is needed instead of being
// read -- 8d fewer bytes per row (a float mask is otherwise read twice).
template <int HN, bool SEEDED, bool RAG = false, int WAVES = kRnvpWaves>
__device__ __forceinline__ void rnvp_group_f32(float (&lds)[2][RnvpShape<HN>::CHUNK_FLOATS], int grp,
                                               const float* __restrict__ z, const float* __restrict__ mask,
                                               float* __restrict__ x, float* __restrict__ log_det,
                                               const float* __restrict__ image, int64_t rows, int d, int accumulate,
                                               uint64_t seed, const float* zprm = nullptr, int dm_ragged = 0,
                                               bool vec = true, bool enable = true) {
  // enable (wave-uniform): false = this wave's 16 rows take part in the group's staging but store nothing (the narrow
  // kernel's fix-up: one tile of the group is redone)
  const int dm = RAG ? dm_ragged : d;  // row width in memory
  using S = RnvpShape<HN, WAVES>;
  constexpr int KQ = S::KQ, YT = S::YT, KC = kRnvpChunkK, MC = kRnvpChunkM;
  constexpr int NROW = (KC / 4 > MC ? KC / 4 : MC);  // float4 row loads per chunk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n_k = d / 4;    // GEMM-1 K-steps
  const int n_m = d / 16;   // GEMM-2 output tiles
  const int nc1 = (n_k + KC - 1) / KC, nc2 = (n_m + MC - 1) / MC, nc = nc1 + nc2;
  const float* img1 = image;
  const float* img2 = image + S::part1_floats(d);
  const float* bias_y = img2 + S::part2_floats(d);

  // operands of chunk c: source and number of float4s
  auto chunk_src = [&](int c, int& n4) -> const float4* {
    if (c < nc1) {
      n4 = min(KC, n_k - c * KC) * 64;
      return reinterpret_cast<const float4*>(img1 + (int64_t)c * KC * 256);
    }
    const int m0 = (c - nc1) * MC;
    n4 = min(MC, n_m - m0) * (S::TILE2_FLOATS / 4);
    return reinterpret_cast<const float4*>(img2 + (int64_t)m0 * S::TILE2_FLOATS);
  };
  // first dim of the i-th float4 of row data chunk c needs (clamped inside the row)
  auto row_dim = [&](int c, int i) -> int {
    const int dim0 = (c < nc1) ? (c * KC + 4 * i) * 4 : 16 * ((c - nc1) * MC + i);
    return dim0 < d ? dim0 : 0;
  };

  {
    const int64_t row = (int64_t)grp * (16 * WAVES) + wave * 16 + j;
    const bool live = row < rows && enable;
    const int64_t rowc = row < rows ? row : rows - 1;
    const float* zr = z + rowc * dm + 4 * q;
    const float* mr = SEEDED ? nullptr : mask + rowc * dm + 4 * q;
    float* xr = x + rowc * dm + 4 * q;
    // zprm: the sample_z prologue fused into the loads, z = q0_mean + q0_std * eps (mnf_linear.py:59-62)
    auto load_z = [&](int dim0) -> f32x4 {
      const f32x4 v = row_load4<RAG>(zr, dim0, 4 * q, dm, vec);
      if (zprm == nullptr) return v;
      return v * *reinterpret_cast<const f32x4*>(zprm + d + dim0 + 4 * q) +
             *reinterpret_cast<const f32x4*>(zprm + dim0 + 4 * q);
    };
    auto mask4 = [&](int dim0) -> f32x4 {  // four consecutive dims share one 32-bit mask word
      if (!SEEDED) return row_load4<RAG>(mr, dim0, 4 * q, dm, vec);
      const int dd = dim0 + 4 * q;
      const uint32_t w = rnvp_mask_word(seed, rowc, dd >> 5) >> (dd & 31);
      return f32x4{(float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u)};
    };

    f32x4 zc[NROW], mc[NROW], zn[NROW], mn[NROW];
    float4 st[S::STAGE_F4];
    // prologue: rows + operands of chunk 0 (the only exposed latency of the group)
    __syncthreads();  // the previous group's last chunk is fully consumed
    {
      int n4;
      const float4* src = chunk_src(0, n4);
#pragma unroll
      for (int i = 0; i < NROW; ++i) {
        zc[i] = load_z(row_dim(0, i));
        mc[i] = mask4(row_dim(0, i));
      }
#pragma unroll
      for (int i = 0; i < S::STAGE_F4; ++i) {
        const int k = threadIdx.x + i * (WAVES * 64);
        if (k < n4) reinterpret_cast<float4*>(lds[0])[k] = src[k];
      }
    }
    __syncthreads();

    f32x4 yacc[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) yacc[m] = *reinterpret_cast<const f32x4*>(bias_y + m * 16 + 4 * q);
    float ld = 0.f;

    for (int c = 0; c < nc; ++c) {
      // 1. request what chunk c+1 needs.  Branch-free on purpose: after the last chunk the same
      //    chunk is simply requested again (a conditional here makes hipcc park st[] in scratch
      //    behind a vmcnt wait, which serialises the prefetch).
      const int cn = c + 1 < nc ? c + 1 : c;
      int n4_next = 0;
      const float4* src_next = chunk_src(cn, n4_next);
#pragma unroll
      for (int i = 0; i < NROW; ++i) {
        zn[i] = load_z(row_dim(cn, i));
        mn[i] = mask4(row_dim(cn, i));
      }
#pragma unroll
      for (int i = 0; i < S::STAGE_F4; ++i) {
        const int k = threadIdx.x + i * (WAVES * 64);
        st[i] = src_next[k < n4_next ? k : 0];
      }
      // 2. compute chunk c
      const float* buf = lds[c & 1];
      if (c < nc1) {  // GEMM 1: y^T (64 x 16) += Wn[:, chunk] . (m*z)^T[chunk]
        const int nk = min(KC, n_k - c * KC);
        const f32x4* A4 = reinterpret_cast<const f32x4*>(buf) + lane;
#pragma unroll
        for (int g = 0; g < KC / 4; ++g) {
          if (4 * g < nk) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float b = mc[g][e] * zc[g][e];
              const f32x4 a4 = A4[64 * (4 * g + e)];
#pragma unroll
              for (int m = 0; m < YT; ++m)
                yacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m], b, yacc[m], 0, 0, 0);
            }
          }
        }
      } else {  // GEMM 2 + gate, 16 output dims per tile
        const int m0 = (c - nc1) * MC;
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
          const int m = m0 + mi;
          if (m < n_m) {
            const float* tile = buf + mi * S::TILE2_FLOATS;
            const f32x4* A4 = reinterpret_cast<const f32x4*>(tile) + lane;
            f32x4 t4 = *reinterpret_cast<const f32x4*>(tile + S::G2 * 256 + 4 * q);
            f32x4 s4 = *reinterpret_cast<const f32x4*>(tile + S::G2 * 256 + 16 + 4 * q);
            f32x4 a4;
#pragma unroll
            for (int cc = 0; cc < KQ; ++cc) {
              if (((2 * cc) & 3) == 0) a4 = A4[64 * ((2 * cc) >> 2)];
              t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(2 * cc) & 3], yacc[cc >> 2][cc & 3], t4, 0, 0, 0);
              s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(2 * cc + 1) & 3], yacc[cc >> 2][cc & 3], s4, 0, 0, 0);
            }
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float zz = zc[mi][r], mm = mc[mi][r];
              const float gate = __builtin_amdgcn_rcpf(1.f + exp6r(-s4[r]));
              const float keep = mm * zz;                            // z2 = m z
              const float gated = (1.f - mm) * zz;                   // z1 = (1-m) z
              o[r] = (gated * gate + (1.f - gate) * t4[r]) + keep;   // rnvp.py:37
              ld += (1.f - mm) * (__builtin_amdgcn_logf(gate) * 0.693147180559945309f);  // :36
            }
            if (live) row_store4<RAG>(xr, 16 * m, 4 * q, dm, vec, o);
          }
        }
      }
      // 3. hand chunk c+1 over
      {
        float4* dst = reinterpret_cast<float4*>(lds[(c + 1) & 1]);
#pragma unroll
        for (int i = 0; i < S::STAGE_F4; ++i) {
          const int k = threadIdx.x + i * (WAVES * 64);
          if (k < n4_next) dst[k] = st[i];
        }
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
          zc[i] = zn[i];
          mc[i] = mn[i];
        }
      }
      __syncthreads();
    }
    if (log_det) {
      ld = sum_over_q(ld);
      if (live && q == 0) log_det[row] = accumulate ? log_det[row] + ld : ld;
    }
  }
}

// Operand image of the split (f16 hi + lo) kernels: GEMM-1 K-steps (two 16-dim groups each), then GEMM-2 output
// tiles, then fp32 bias words ((bt, bs) per tile, bn) -- see build_split_index in mnf_rnvp_mfma.hip.
template <int HN>
struct RnvpSplitShape {
  static constexpr int YT = (HN + 15) / 16;            // 16-unit tiles of y (unit u = 16 m + i)
  static constexpr int NKS2 = (YT + 1) / 2;            // K = 32 steps of GEMM 2
  static constexpr int KS1_WORDS = YT * 512;           // one GEMM-1 K-step: YT x (hi, lo) operands
  static constexpr int TILE2_WORDS = 2 * NKS2 * 512;   // one GEMM-2 output tile: (t, s) x NKS2 x (hi, lo)
  static constexpr int KC = 2, MC = 2;                 // K-steps / output tiles per chunk
  static constexpr int CHUNK_WORDS = KC * KS1_WORDS > MC * TILE2_WORDS ? KC * KS1_WORDS : MC * TILE2_WORDS;
  static constexpr int NROW = 2 * KC > MC ? 2 * KC : MC;  // 16-dim row groups per chunk
  static constexpr int STAGE_U4 = (CHUNK_WORDS / 4 + kRnvpWaves * 64 - 1) / (kRnvpWaves * 64);
  static constexpr int64_t n_ks1(int d) { return (d / 16 + 1) / 2; }
  static constexpr int64_t part1_words(int d) { return n_ks1(d) * KS1_WORDS; }
  static constexpr int64_t part2_words(int d) { return (int64_t)(d / 16) * TILE2_WORDS; }
  static constexpr int64_t split_words(int d) { return part1_words(d) + part2_words(d); }
  static constexpr int64_t plain_words(int d) { return (int64_t)(d / 16) * 32 + YT * 16; }  // (bt, bs) per tile, then bn
};

// ---------------------------------------------------------------- hand-over between a row-parallel and a dims-slab launch
// (the RNVP gradient kernels, mnf_rnvp_bwd.hip, and MNFLinear's, mnf_mnf_linear_bwd.hip)
// Per 16-row tile, two small per-row vectors "a" and "b" of 16 YT entries each (RNVP: y and g_y over the hidden units;
// MNFLinear: the cotangents of mean and var over the outputs), written by the row-parallel launch as ready-made split
// MFMA operands in BOTH orientations: entries along the lane's registers ([ks][hi|lo][lane][4 words], the A operand of a
// K = entries product with the rows on M) and rows along them ([entry tile][hi|lo'][lane][2 words], the A operand of a
// sum over rows; lo' = the UNSCALED residual, so that such a sum needs one accumulator).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int YT>
struct HandoverShape {
  static constexpr int NKS = (YT + 1) / 2;
  static constexpr int OP_WORDS = NKS * 2 * 256;
  static constexpr int TR_WORDS = YT * 2 * 128;
  static constexpr int TILE_WORDS = 2 * OP_WORDS + 2 * TR_WORDS;
  static constexpr int A_OP = 0, B_OP = OP_WORDS, A_TR = 2 * OP_WORDS, B_TR = 2 * OP_WORDS + TR_WORDS;
};
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
}
// four fp32 values -> f16 heads and UNSCALED f16 residuals (v - head): the operands of a product whose three partial
// products go into one accumulator.  Below 2^-13 the residual is an f16 subnormal or zero: an absolute error of at most
// 2^-25 on data scaled to O(1), fp32's own rounding of a sum whose largest terms are O(1).
__device__ __forceinline__ void split_plain(const f32x4& v, u32x2& hi, u32x2& lo) {
  const uint32_t h0 = cvt_pk(v[0], v[1]), h1 = cvt_pk(v[2], v[3]);
  hi = u32x2{h0, h1};
  lo = u32x2{cvt_pk(residual_lo(h0, v[0]), residual_hi(h0, v[1])), cvt_pk(residual_lo(h1, v[2]), residual_hi(h1, v[3]))};
}
// lane (row j, q) holds entries 16 m + 4 q .. + 3 of its row as split tiles (ah/al, bh/bl: hi and 2^11-scaled lo)
template <int YT>
__device__ __forceinline__ void store_handover(uint32_t* __restrict__ out, const u32x2 (&ah)[YT], const u32x2 (&al)[YT],
                                               const u32x2 (&bh)[YT], const u32x2 (&bl)[YT], int lane, int j, int q) {
  using H = HandoverShape<YT>;
  const u32x2 zero2 = u32x2{0u, 0u};
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // entries on K: operand ks = (tiles 2 ks, 2 ks + 1)
#pragma unroll
  for (int ks = 0; ks < H::NKS; ++ks) {
    const bool two = 2 * ks + 1 < YT;
    const u32x2 a1 = two ? ah[two ? 2 * ks + 1 : 0] : zero2, a1l = two ? al[two ? 2 * ks + 1 : 0] : zero2;
    const u32x2 b1 = two ? bh[two ? 2 * ks + 1 : 0] : zero2, b1l = two ? bl[two ? 2 * ks + 1 : 0] : zero2;
    *reinterpret_cast<u32x4*>(out + H::A_OP + ((2 * ks) * 64 + lane) * 4) = u32x4{ah[2 * ks][0], ah[2 * ks][1], a1[0], a1[1]};
    *reinterpret_cast<u32x4*>(out + H::A_OP + ((2 * ks + 1) * 64 + lane) * 4) = u32x4{al[2 * ks][0], al[2 * ks][1], a1l[0], a1l[1]};
    *reinterpret_cast<u32x4*>(out + H::B_OP + ((2 * ks) * 64 + lane) * 4) = u32x4{bh[2 * ks][0], bh[2 * ks][1], b1[0], b1[1]};
    *reinterpret_cast<u32x4*>(out + H::B_OP + ((2 * ks + 1) * 64 + lane) * 4) = u32x4{bl[2 * ks][0], bl[2 * ks][1], b1l[0], b1l[1]};
  }
  // rows on K: one MFMA against the identity per tile and part turns "lane = row, registers = entries" into
  // "lane = entry, registers = rows" (D[row][e'] = sum_k A[row][k] I[k][e'], every product x 1: exact); the tail goes
  // against 2^-11 I and comes out unscaled
  u32x2 id, ids;
  {
    const _Float16 one = (_Float16)1.f, tiny = (_Float16)kSplitInvScale, zero = (_Float16)0.f;
    f16x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (4 * q + e == j) ? one : zero;
      b[e] = (4 * q + e == j) ? tiny : zero;
    }
    id = __builtin_bit_cast(u32x2, a);
    ids = __builtin_bit_cast(u32x2, b);
  }
#pragma unroll
  for (int m = 0; m < YT; ++m) {
    const f32x4 a = mfma16(ah[m], id, zero4), b = mfma16(al[m], ids, zero4);
    const f32x4 c = mfma16(bh[m], id, zero4), e = mfma16(bl[m], ids, zero4);
    *reinterpret_cast<u32x2*>(out + H::A_TR + ((2 * m) * 64 + lane) * 2) = u32x2{cvt_pk(a[0], a[1]), cvt_pk(a[2], a[3])};
    *reinterpret_cast<u32x2*>(out + H::A_TR + ((2 * m + 1) * 64 + lane) * 2) = u32x2{cvt_pk(b[0], b[1]), cvt_pk(b[2], b[3])};
    *reinterpret_cast<u32x2*>(out + H::B_TR + ((2 * m) * 64 + lane) * 2) = u32x2{cvt_pk(c[0], c[1]), cvt_pk(c[2], c[3])};
    *reinterpret_cast<u32x2*>(out + H::B_TR + ((2 * m + 1) * 64 + lane) * 2) = u32x2{cvt_pk(e[0], e[1]), cvt_pk(e[2], e[3])};
  }
}

// Work items = (row part, slab) of a dims-slab launch over a persistent grid.  With >= 8 row parts, part p belongs to
// XCD p % 8 (workgroups go to the XCDs round robin: block b runs on XCD b % 8) and that XCD's workgroups take its items
// in (part, slab) order: the n_slabs workgroups on one row part then run on ONE XCD at about the same time and walk the
// same rows, so the per-row hand-over (read by every slab) is fetched into that XCD's L2 once.
struct SlabItems {
  int n_items, first, step, n_slabs, xcd;
  bool by_xcd;
  __device__ __forceinline__ SlabItems(int n_slabs_, int row_parts) : n_slabs(n_slabs_) {
    by_xcd = row_parts >= 8;
    xcd = blockIdx.x & 7;
    const int local_parts = by_xcd ? (row_parts - xcd + 7) / 8 : 0;
    n_items = by_xcd ? local_parts * n_slabs : row_parts * n_slabs;
    first = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  }
  __device__ __forceinline__ int slab(int item) const { return item % n_slabs; }
  __device__ __forceinline__ int part(int item) const { return by_xcd ? (item / n_slabs) * 8 + xcd : item / n_slabs; }
};
// host: row parts and grid of such a launch.  Row parts come in multiples of 8 (one XCD each); their number per XCD
// is chosen so that the XCD's items fill whole rounds of its resident workgroups.
inline void plan_slab_launch(int64_t n_pairs, int waves, int n_slabs, int resident, int& row_parts, int& grid,
                             int max_l = 32) {
  const int64_t max_parts = (n_pairs + waves - 1) / waves;  // at least one pair per wave
  if (max_parts < 8) {
    row_parts = (int)(max_parts < 1 ? 1 : max_parts);
    grid = row_parts * n_slabs;
    return;
  }
  const int wgs_xcd = resident / 8 > 0 ? resident / 8 : 1;
  int best_l = 1;
  double best_fill = 0.0;
  for (int l = 1; l <= max_l && (int64_t)l * 8 <= max_parts; ++l) {
    const int items = l * n_slabs, rounds = (items + wgs_xcd - 1) / wgs_xcd;
    const double fill = (double)items / ((double)rounds * wgs_xcd);
    if (fill > best_fill + 0.02) {  // (prefer fewer parts -- fewer flushes -- unless the fill improves by > 2 %)
      best_fill = fill;
      best_l = l;
    }
  }
  row_parts = best_l * 8;
  const int items = best_l * n_slabs;
  grid = 8 * (items < wgs_xcd ? items : wgs_xcd);
}

// ---------------------------------------------------------------- host: shapes and the split image's index table
// hidden widths with an instantiated kernel: 50 (MNFLinear's h_sizes), 30 (RNVP's default) and 64 (the widest the
// four-tile layout holds); any other width up to 64 runs at the next one up with structural-zero units
// (rnvp_padded_hidden).  50 and 64 share the split kernels' shape (four 16-unit tiles); the fp32 kernels differ.
#define MNF_RNVP_HIDDEN(X) X(50) X(30) X(64)
inline int rnvp_padded_hidden(int n_hidden, const int* hidden) {
  if (n_hidden != 1 || !hidden || hidden[0] < 1) return 0;
  return hidden[0] <= 30 ? 30 : hidden[0] <= 50 ? 50 : hidden[0] <= 64 ? 64 : 0;
}


// dim rounded up to whole 16-dim groups (the kernels' d); a layer with dim % 16 != 0 runs the ragged variants
inline int rnvp_padded_dim(int dim) { return (dim + 15) & ~15; }

inline bool rnvp_shape_ok(int dim, int n_hidden, const int* hidden) {
  const int d = rnvp_padded_dim(dim);
  if (n_hidden != 1 || !hidden || dim < 1 || d < 64 || (int64_t)d * 64 * 3 >= (1ll << 30)) return false;
#define X(HN) if (rnvp_padded_hidden(n_hidden, hidden) == HN) return true;
  MNF_RNVP_HIDDEN(X)
#undef X
  return false;
}


// 2 entries per split word (low half, high half), then 1 entry per plain word -- see mnf_pack_gather_split
// dm: the layer's real width (flat-parameter offsets, valid columns / rows); d: dm rounded up to 16
// hn <= HN: the layer's real hidden width (units hn .. HN-1 are structural zeros: y = 0 there)
template <int HN>
inline void build_split_index(int dm, int d, int32_t* idx, int hn = HN) {
  using S = RnvpSplitShape<HN>;
  constexpr int YT = S::YT, NKS2 = S::NKS2;
  const int G = d / 16;
  const int64_t wn = 0, bn = wn + (int64_t)hn * dm, wt = bn + hn, bt = wt + (int64_t)dm * hn, ws = bt + dm,
                bs = ws + (int64_t)dm * hn;
  const int64_t n_entries = 2 * S::split_words(d) + S::plain_words(d);
  for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  // element e of lane (i, kq) of operand `op` (hi at 2 op, lo at 2 op + 1), base = first word of the region
  auto put = [&](int64_t base_words, int op, int lane, int e, int64_t src) {
    for (int part = 0; part < 2; ++part)
      idx[2 * base_words + (((int64_t)(2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] =
          (int32_t)src | (part ? kSplitLoBit : 0);
  };
  // part 1: K-step ks covers groups 2 ks (slots 8 kq + 0..3) and 2 ks + 1 (slots 8 kq + 4..7)
  for (int ks = 0; ks < S::n_ks1(d); ++ks)
    for (int m = 0; m < YT; ++m)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
        if (u >= hn) continue;
        for (int e = 0; e < 8; ++e) {
          const int g = 2 * ks + (e >> 2), col = 16 * g + 4 * kq + (e & 3);
          if (g < G && col < dm) put(0, ks * YT + m, lane, e, wn + (int64_t)u * dm + col);
        }
      }
  // part 2: per output tile m: t operands for K-steps 0..NKS2-1, then s operands
  for (int m = 0; m < G; ++m)
    for (int which = 0; which < 2; ++which)
      for (int ks = 0; ks < NKS2; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4;
          for (int e = 0; e < 8; ++e) {
            const int tile = 2 * ks + (e >> 2), unit = 16 * tile + 4 * kq + (e & 3);
            if (tile < YT && unit < hn && 16 * m + i < dm)
              put(S::part1_words(d) + (int64_t)m * S::TILE2_WORDS, which * NKS2 + ks, lane, e,
                  (which ? ws : wt) + (int64_t)(16 * m + i) * hn + unit);
          }
        }
  int32_t* pl = idx + 2 * S::split_words(d);
  for (int m = 0; m < G; ++m)
    for (int i = 0; i < 16; ++i) {
      const bool real = 16 * m + i < dm;  // padded output dims: shift 0, scale bias "big" (gate 1, log gate 0)
      pl[(int64_t)m * 32 + i] = real ? (int32_t)(bt + 16 * m + i) : -1;
      pl[(int64_t)m * 32 + 16 + i] = real ? (int32_t)(bs + 16 * m + i) : kPackBigBias;
    }
  for (int m = 0; m < YT; ++m)
    for (int i = 0; i < 16; ++i)
      if (16 * m + i < hn) pl[(int64_t)G * 32 + m * 16 + i] = (int32_t)(bn + 16 * m + i);
}


}  // namespace mnf
