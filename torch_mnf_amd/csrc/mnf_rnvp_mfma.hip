// Masked / gated RNVP coupling (flows/rnvp.py:25-39) as two chained fp32-MFMA GEMMs, gfx950.
//
//   y      = Wn (m * z) + bn                 (h  <- d)     GEMM 1, K = d
//   shift  = Wt y + bt ; scale = Ws y + bs   (d  <- h)     GEMM 2, K = h
//   gate   = sigmoid(scale)
//   x      = (1-m) z gate + (1-gate) shift + m z ;  log_det = sum_j (1-m_j) log gate_j
//
// One wave owns 16 rows, a 256-thread workgroup 64 rows.  Both GEMMs run transposed on
// v_mfma_f32_16x16x4_f32 with the batch on the N axis, as in the AffineHalfFlow kernel: the
// 16 accumulator registers of y (h padded to 64 = 4 tiles) are directly the K-step operands of
// GEMM 2, and a GEMM-2 output tile (16 dims x 16 rows) has the lane layout of a float4 of the
// row, so gate / transform / store / log-det run from registers.  At d = 800 the operand image
// (538 KB) does not fit LDS: it is streamed from L2 in chunks that the four waves of a
// workgroup share (stage -> barrier -> compute -> barrier), 40 KiB of LDS per workgroup.
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

constexpr int kRnvpWaves = 4;
constexpr int kRnvpChunkK = 32;  // GEMM-1 K-steps staged per chunk (128 dims): 32 x 4 tiles x 256 B = 32 KiB
constexpr int kRnvpChunkM = 5;   // GEMM-2 output tiles staged per chunk

template <int HN>
struct RnvpShape {
  static constexpr int KQ = (HN + 3) / 4;       // K-steps of GEMM 2 (quads of y units)
  static constexpr int YT = (KQ + 3) / 4;       // 16-row tiles of y
  static_assert(YT >= 1 && YT <= 4, "GEMM-1 operand groups hold up to four y tiles (hidden width <= 64)");
  static constexpr int G2 = (2 * KQ + 3) / 4;   // operand groups (of 4 MFMAs) per GEMM-2 output tile
  static constexpr int TILE2_FLOATS = G2 * 256;
  static constexpr int64_t part1_floats(int d) { return (int64_t)(d / 4) * 256; }         // one group per K-step
  static constexpr int64_t part2_floats(int d) { return (int64_t)(d / 16) * TILE2_FLOATS; }
  static constexpr int64_t bias_floats(int d) { return YT * 16 + 2 * d; }
  static constexpr int64_t image_floats(int d) { return part1_floats(d) + part2_floats(d) + bias_floats(d); }
  static constexpr int LDS_FLOATS =
      (kRnvpChunkK * 256 > kRnvpChunkM * TILE2_FLOATS ? kRnvpChunkK * 256 : kRnvpChunkM * TILE2_FLOATS);
};

// 1/(1+exp(-v)) with the 6-instruction exp of the AffineHalfFlow kernel
__device__ __forceinline__ float exp6r(float x) {
  const float c_hi = 1.44269502162933349609375f, c_lo = 1.925963033500011e-08f, ln2 = 0.693147182464599609375f;
  const float t = x * c_hi;
  const float err = __builtin_fmaf(x, c_hi, -t);
  const float tl = __builtin_fmaf(x, c_lo, err);
  return __builtin_amdgcn_exp2f(t) * __builtin_fmaf(tl, ln2, 1.0f);
}

// SEEDED: the mask is not read from memory but regenerated from (seed, row, dim) wherever it
// is needed -- 8d fewer bytes per row (the mask is otherwise read twice).
template <int HN, bool SEEDED>
__global__ void __launch_bounds__(kRnvpWaves * 64)
rnvp_mfma_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x,
                 float* __restrict__ log_det, const float* __restrict__ image, int64_t rows, int d,
                 int accumulate, uint64_t seed) {
  using S = RnvpShape<HN>;
  constexpr int KQ = S::KQ, YT = S::YT;
  __shared__ __attribute__((aligned(16))) float lds[S::LDS_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n_k = d / 4;    // GEMM-1 K-steps
  const int n_m = d / 16;   // GEMM-2 output tiles
  const float* img1 = image;
  const float* img2 = image + S::part1_floats(d);
  const float* bias_y = img2 + S::part2_floats(d);
  const float* bias_ts = bias_y + YT * 16;  // [tile m][t: 16 floats | s: 16 floats]

  const int n_groups = (int)((rows + 63) >> 6);  // 64 rows per workgroup iteration
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t row = (int64_t)grp * 64 + wave * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* zr = z + rowc * d + 4 * q;
    const float* mr = SEEDED ? nullptr : mask + rowc * d + 4 * q;
    // four consecutive dims 16 g + 4 q .. + 3 share one 32-bit mask word
    auto mask4 = [&](int dim0) -> f32x4 {
      if (!SEEDED) return *reinterpret_cast<const f32x4*>(mr + dim0);
      const int dd = dim0 + 4 * q;
      const uint32_t w = rnvp_mask_word(seed, rowc, dd >> 5) >> (dd & 31);
      return f32x4{(float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u)};
    };
    float* xr = x + rowc * d + 4 * q;

    // ---------------- GEMM 1: y^T (64 x 16) = Wn . (m*z)^T, K = d
    f32x4 yacc[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) yacc[m] = *reinterpret_cast<const f32x4*>(bias_y + m * 16 + 4 * q);
    for (int k0 = 0; k0 < n_k; k0 += kRnvpChunkK) {
      const int nk = min(kRnvpChunkK, n_k - k0);
      __syncthreads();  // previous chunk fully consumed
      {
        const float4* src = reinterpret_cast<const float4*>(img1 + (int64_t)k0 * 256);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < nk * 64; i += blockDim.x) dst[i] = src[i];
      }
      __syncthreads();
      const f32x4* A4 = reinterpret_cast<const f32x4*>(lds) + lane;
      for (int g = 0; g < nk / 4; ++g) {  // 16 dims = 4 K-steps per float4
        const int dim0 = (k0 + 4 * g) * 4;
        f32x4 zz = *reinterpret_cast<const f32x4*>(zr + dim0);
        const f32x4 mm = mask4(dim0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float b = mm[e] * zz[e];
          const f32x4 a4 = A4[64 * (4 * g + e)];
#pragma unroll
          for (int m = 0; m < YT; ++m)
            yacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m], b, yacc[m], 0, 0, 0);
        }
      }
    }

    // ---------------- GEMM 2 + gate, 16 output dims per tile
    float ld = 0.f;
    for (int m0 = 0; m0 < n_m; m0 += kRnvpChunkM) {
      const int nm = min(kRnvpChunkM, n_m - m0);
      __syncthreads();
      {
        const float4* src = reinterpret_cast<const float4*>(img2 + (int64_t)m0 * S::TILE2_FLOATS);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < nm * (S::TILE2_FLOATS / 4); i += blockDim.x) dst[i] = src[i];
      }
      __syncthreads();
      for (int mi = 0; mi < nm; ++mi) {
        const int m = m0 + mi;
        const f32x4* A4 = reinterpret_cast<const f32x4*>(lds + mi * S::TILE2_FLOATS) + lane;
        f32x4 t4 = *reinterpret_cast<const f32x4*>(bias_ts + m * 32 + 4 * q);
        f32x4 s4 = *reinterpret_cast<const f32x4*>(bias_ts + m * 32 + 16 + 4 * q);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 16 * m);
        const f32x4 mm = mask4(16 * m);
        f32x4 a4;
#pragma unroll
        for (int c = 0; c < KQ; ++c) {
          if (((2 * c) & 3) == 0) a4 = A4[64 * ((2 * c) >> 2)];
          t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(2 * c) & 3], yacc[c >> 2][c & 3], t4, 0, 0, 0);
          s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(2 * c + 1) & 3], yacc[c >> 2][c & 3], s4, 0, 0, 0);
        }
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gate = __builtin_amdgcn_rcpf(1.f + exp6r(-s4[r]));
          const float keep = mm[r] * zz[r];                       // z2 = m z
          const float gated = (1.f - mm[r]) * zz[r];              // z1 = (1-m) z
          o[r] = (gated * gate + (1.f - gate) * t4[r]) + keep;    // rnvp.py:37
          ld += (1.f - mm[r]) * (__builtin_amdgcn_logf(gate) * 0.693147180559945309f);  // :36
        }
        if (live) *reinterpret_cast<f32x4*>(xr + 16 * m) = o;
      }
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
  int32_t* pb = p2 + S::part2_floats(d);
  for (int m = 0; m < YT; ++m)
    for (int i = 0; i < 16; ++i)
      if (unit_of(m, i) < HN) pb[m * 16 + i] = (int32_t)(bn + unit_of(m, i));
  pb += YT * 16;
  for (int m = 0; m < d / 16; ++m)
    for (int i = 0; i < 16; ++i) {
      pb[m * 32 + i] = (int32_t)(bt + 16 * m + i);
      pb[m * 32 + 16 + i] = (int32_t)(bs + 16 * m + i);
    }
}

// hidden widths with an instantiated kernel: 50 (MNFLinear's h_sizes) and 30 (RNVP's default)
#define MNF_RNVP_HIDDEN(X) X(50) X(30)

template <int HN>
static int launch_rnvp(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                       const float* image, int64_t rows, int dim, uint64_t seed, hipStream_t stream) {
  const int64_t n_groups = (rows + 63) / 64;
  static const int resident = [] {
    int per_cu = 0, cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rnvp_mfma_kernel<HN, false>, kRnvpWaves * 64, 0) !=
            hipSuccess || per_cu < 1)
      per_cu = 2;
    return per_cu * cus;
  }();
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
