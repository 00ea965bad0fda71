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
// Supported here: one hidden layer (net is a bare Linear, as MNFLinear uses it), d % 16 == 0;
// everything else runs the generic kernel.
#include <hip/hip_runtime.h>

#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRnvpWaves = 8;    // 512-thread workgroups: 128 rows share every staged operand chunk
constexpr int kRnvpChunkK = 16;  // GEMM-1 K-steps per chunk (64 dims): 16 x 4 tiles x 256 B = 16 KiB
constexpr int kRnvpChunkM = 2;   // GEMM-2 output tiles per chunk

template <int HN>
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
  static constexpr int STAGE_F4 = (CHUNK_FLOATS / 4 + kRnvpWaves * 64 - 1) / (kRnvpWaves * 64);  // float4 per thread
};

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
template <int HN, bool SEEDED>
__global__ void __launch_bounds__(kRnvpWaves * 64, 4)  // two 8-wave workgroups per CU: <= 128 VGPRs
rnvp_mfma_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x,
                 float* __restrict__ log_det, const float* __restrict__ image, int64_t rows, int d,
                 int accumulate, uint64_t seed) {
  using S = RnvpShape<HN>;
  constexpr int KQ = S::KQ, YT = S::YT, KC = kRnvpChunkK, MC = kRnvpChunkM;
  constexpr int NROW = (KC / 4 > MC ? KC / 4 : MC);  // float4 row loads per chunk
  __shared__ __attribute__((aligned(16))) float lds[2][S::CHUNK_FLOATS];
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

  const int n_groups = (int)((rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves));
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t row = (int64_t)grp * (16 * kRnvpWaves) + wave * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* zr = z + rowc * d + 4 * q;
    const float* mr = SEEDED ? nullptr : mask + rowc * d + 4 * q;
    float* xr = x + rowc * d + 4 * q;
    auto mask4 = [&](int dim0) -> f32x4 {  // four consecutive dims share one 32-bit mask word
      if (!SEEDED) return *reinterpret_cast<const f32x4*>(mr + dim0);
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
        zc[i] = *reinterpret_cast<const f32x4*>(zr + row_dim(0, i));
        mc[i] = mask4(row_dim(0, i));
      }
#pragma unroll
      for (int i = 0; i < S::STAGE_F4; ++i) {
        const int k = threadIdx.x + i * (kRnvpWaves * 64);
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
        zn[i] = *reinterpret_cast<const f32x4*>(zr + row_dim(cn, i));
        mn[i] = mask4(row_dim(cn, i));
      }
#pragma unroll
      for (int i = 0; i < S::STAGE_F4; ++i) {
        const int k = threadIdx.x + i * (kRnvpWaves * 64);
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
            if (live) *reinterpret_cast<f32x4*>(xr + 16 * m) = o;
          }
        }
      }
      // 3. hand chunk c+1 over
      {
        float4* dst = reinterpret_cast<float4*>(lds[(c + 1) & 1]);
#pragma unroll
        for (int i = 0; i < S::STAGE_F4; ++i) {
          const int k = threadIdx.x + i * (kRnvpWaves * 64);
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

// ---------------------------------------------------------------- host: image index table
template <int HN>
static void build_index(int d, int32_t* idx) {
  using S = RnvpShape<HN>;
  constexpr int KQ = S::KQ, YT = S::YT;
  // flat layout: net.0.weight (HN, d), net.0.bias (HN), t.weight (d, HN), t.bias (d), s.weight (d, HN), s.bias (d)
  const int64_t wn = 0, bn = wn + (int64_t)HN * d, wt = bn + HN, bt = wt + (int64_t)d * HN, ws = bt + d,
                bs = ws + (int64_t)d * HN;
  const int64_t total = S::image_floats(d);
  for (int64_t i = 0; i < total; ++i) idx[i] = -1;
  auto unit_of = [&](int m, int i) { return 16 * m + 4 * (i & 3) + (i >> 2); };
  // part 1: group per K-step kk; component = y tile m
  for (int kk = 0; kk < d / 4; ++kk) {
    const int g = kk >> 2, e = kk & 3;
    for (int m = 0; m < YT; ++m)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, u = unit_of(m, i);
        if (u < HN) idx[(int64_t)kk * 256 + lane * 4 + m] = (int32_t)(wn + (int64_t)u * d + 16 * g + 4 * kq + e);
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
          if (unit < HN)
            p2[(int64_t)m * S::TILE2_FLOATS + (n >> 2) * 256 + lane * 4 + (n & 3)] =
                (int32_t)((which ? ws : wt) + (int64_t)(16 * m + i) * HN + unit);
        }
      }
  for (int m = 0; m < d / 16; ++m)
    for (int i = 0; i < 16; ++i) {
      p2[(int64_t)m * S::TILE2_FLOATS + S::G2 * 256 + i] = (int32_t)(bt + 16 * m + i);
      p2[(int64_t)m * S::TILE2_FLOATS + S::G2 * 256 + 16 + i] = (int32_t)(bs + 16 * m + i);
    }
  int32_t* pb = p2 + S::part2_floats(d);
  for (int m = 0; m < YT; ++m)
    for (int i = 0; i < 16; ++i)
      if (unit_of(m, i) < HN) pb[m * 16 + i] = (int32_t)(bn + unit_of(m, i));
}

// hidden widths with an instantiated kernel: 50 (MNFLinear's h_sizes) and 30 (RNVP's default)
#define MNF_RNVP_HIDDEN(X) X(50) X(30)

template <int HN>
static int launch_rnvp(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                       const float* image, int64_t rows, int dim, uint64_t seed, hipStream_t stream) {
  const int64_t n_groups = (rows + 16 * kRnvpWaves - 1) / (16 * kRnvpWaves);
  auto resident_of = [](auto kernel) {
    int per_cu = 0, cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kRnvpWaves * 64, 0) != hipSuccess || per_cu < 1)
      per_cu = 1;
    return per_cu * cus;
  };
  static const int resident_mask = resident_of(rnvp_mfma_kernel<HN, false>);
  static const int resident_seed = resident_of(rnvp_mfma_kernel<HN, true>);
  const int resident = mask ? resident_mask : resident_seed;
  const int64_t blocks = n_groups < resident ? n_groups : resident;
  if (mask)
    hipLaunchKernelGGL((rnvp_mfma_kernel<HN, false>), dim3((unsigned)blocks), dim3(kRnvpWaves * 64), 0, stream, z,
                       mask, x, log_det, image, rows, dim, accumulate, seed);
  else
    hipLaunchKernelGGL((rnvp_mfma_kernel<HN, true>), dim3((unsigned)blocks), dim3(kRnvpWaves * 64), 0, stream, z,
                       mask, x, log_det, image, rows, dim, accumulate, seed);
  return check_launch();
}

static bool rnvp_shape_ok(int dim, int n_hidden, const int* hidden) {
  if (n_hidden != 1 || !hidden || dim < 64 || dim % 16 != 0 || (int64_t)dim * 64 * 3 >= (1ll << 30)) return false;
#define X(HN) if (hidden[0] == HN) return true;
  MNF_RNVP_HIDDEN(X)
#undef X
  return false;
}

int rnvp_mfma_launch(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                     const float* image, int64_t rows, int dim, int n_hidden, const int* hidden,
                     uint64_t seed, hipStream_t stream) {
  if (!rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(x) |
       reinterpret_cast<uintptr_t>(image)) & 15)
    return MNF_ERR_UNSUPPORTED;
#define X(HN) \
  if (hidden[0] == HN) return launch_rnvp<HN>(z, mask, x, log_det, accumulate, image, rows, dim, seed, stream);
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf

extern "C" {

int64_t mnf_rnvp_image_floats(int dim, int n_hidden, const int* hidden) {
  if (!mnf::rnvp_shape_ok(dim, n_hidden, hidden)) return 0;
#define X(HN) if (hidden[0] == HN) return mnf::RnvpShape<HN>::image_floats(dim);
  MNF_RNVP_HIDDEN(X)
#undef X
  return 0;
}

int mnf_rnvp_image_index(int dim, int n_hidden, const int* hidden, int32_t* idx_host) {
  if (!idx_host) return MNF_ERR_INVALID_ARG;
  if (!mnf::rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
#define X(HN)                               \
  if (hidden[0] == HN) {                    \
    mnf::build_index<HN>(dim, idx_host);    \
    return MNF_OK;                          \
  }
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
