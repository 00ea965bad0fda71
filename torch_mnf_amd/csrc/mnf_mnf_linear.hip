// MNFLinear.forward (layers/mnf_linear.py:46-56) behind the flow path, gfx950: the two GEMMs of the local
// reparametrisation and its noise epilogue in one pass over the rows,
//
//   mean = (x * z) W_mean^T + b_mean          var = x^2 exp(W_log_var)^T + exp(b_log_var)
//   out  = mean + sqrt(var) * eps             eps ~ N(0, 1), injected (tests, fixtures) or generated in-kernel
//
// where z (rows, n_in) is what MNFLinear.sample_z's last RNVP kernel wrote.  The stock composition reads x three
// times and z twice and writes / re-reads the (rows, n_in) product; here every x and z is read once (8 n_in + 4 n_out
// bytes per row, HBM bound) and nothing of size n_in is written.
//
// Same machinery as the RNVP kernels (mnf_rnvp_mfma.hip): one wave owns 16 rows, the batch sits on the MFMA N axis,
// both products run in split fp32 arithmetic (mnf_split.h) on v_mfma_f32_16x16x32_f16 with the weights streamed from
// an operand image through a double-buffered LDS window shared by the eight waves of a workgroup (128 rows per pass).
// exp(W_log_var) is ~1e-4 and smaller after training -- near or below the f16 normal range -- so the image carries it
// times a power of two chosen at pack time (`var_unscale` undoes it, exactly).  A 128-row group whose operands leave
// the f16 range (|x z| or x^2 at 8,192 or above) is flagged in `workspace` and recomputed by the fp32 fix-up kernel
// that every call launches behind the main one (it returns at once when nothing is flagged).
#include <hip/hip_runtime.h>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_split.h"

#include <utility>

namespace mnf {

constexpr int kMlWaves = 8;  // 128 rows share every staged operand chunk
constexpr int kMlKC = 1;     // K-steps (32 input dims each) per chunk
#ifndef MNF_ML_OPS_FIRST
#define MNF_ML_OPS_FIRST 1
#endif
// chunks of row data in flight per wave: 4, but 2 at four output tiles (n_out > 48), where 64 accumulator registers plus a
// 4-deep ring came to 256 VGPRs with 6 spilled: 582 -> 524 us at 256,000 x (800 -> 50)
template <typename Fn, int... I>
__device__ __forceinline__ void mnf_static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename Fn>
__device__ __forceinline__ void mnf_static_for(Fn&& f) {
  mnf_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int YT>
constexpr int kMlRing = YT >= 4 ? 2 : 4;

template <int YT>
struct MlShape {
  static constexpr int OPS = 2 * YT;                 // A operands per K-step: mean tiles, then var tiles
  static constexpr int KS_WORDS = OPS * 512;         // (hi, lo) x 64 lanes x 4 words each
  static constexpr int CHUNK_WORDS = kMlKC * KS_WORDS;
  static constexpr int STAGE_U4 = (CHUNK_WORDS / 4 + kMlWaves * 64 - 1) / (kMlWaves * 64);
  static constexpr int64_t n_ks(int n_in) { return (n_in + 31) / 32; }
  static constexpr int64_t split_words(int n_in) { return n_ks(n_in) * KS_WORDS; }
  static constexpr int64_t plain_words(int) { return 2 * YT * 16; }  // b_mean tiles, then exp(b_log_var) tiles
};

// flat parameter vector the image is gathered from (built by the caller):
//   W_mean (n_out, n_in) | exp(W_log_var) * var_scale (n_out, n_in) | b_mean (n_out) | exp(b_log_var) (n_out)
template <int YT>
static void build_ml_index(int n_in, int n_out, int32_t* idx) {
  using S = MlShape<YT>;
  const int64_t wm = 0, wv = (int64_t)n_out * n_in, bm = 2 * wv, bv = bm + n_out;
  const int64_t n_entries = 2 * S::split_words(n_in) + S::plain_words(n_in);
  for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  auto put = [&](int64_t op, int lane, int e, int64_t src) {  // element e of lane (i, kq) of operand op (hi, lo)
    for (int part = 0; part < 2; ++part)
      idx[(((2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] = (int32_t)src | (part ? kSplitLoBit : 0);
  };
  // K-step ks covers 16-dim groups 2 ks (slots 8 kq + 0..3) and 2 ks + 1 (slots 8 kq + 4..7), as in the RNVP image
  for (int ks = 0; ks < S::n_ks(n_in); ++ks)
    for (int which = 0; which < 2; ++which)
      for (int m = 0; m < YT; ++m)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
          if (u >= n_out) continue;
          for (int e = 0; e < 8; ++e) {
            const int col = 16 * (2 * ks + (e >> 2)) + 4 * kq + (e & 3);
            if (col < n_in) put((int64_t)ks * S::OPS + which * YT + m, lane, e, (which ? wv : wm) + (int64_t)u * n_in + col);
          }
        }
  int32_t* pl = idx + 2 * S::split_words(n_in);
  for (int which = 0; which < 2; ++which)
    for (int m = 0; m < YT; ++m)
      for (int i = 0; i < 16; ++i)
        if (16 * m + i < n_out) pl[(which * YT + m) * 16 + i] = (int32_t)((which ? bv : bm) + 16 * m + i);
}

template <int YT, bool RAG>
__global__ void __launch_bounds__(kMlWaves * 64, 2)
mnf_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ eps,
                      float* __restrict__ out, float* __restrict__ sd_out, const uint32_t* __restrict__ simage,
                      int32_t* __restrict__ flags, int64_t rows, int n_in, int n_out, float var_unscale, uint64_t seed,
                      int vec_ok) {
  using S = MlShape<YT>;
  constexpr int KC = kMlKC, OPS = S::OPS;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][S::CHUNK_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n_ks = (int)S::n_ks(n_in), nc = (n_ks + KC - 1) / KC;
  const int n_groups16 = (n_in + 15) / 16;
  const bool vec = vec_ok != 0;
  const float* bias = reinterpret_cast<const float*>(simage + S::split_words(n_in));
  const float wmax = __builtin_bit_cast(float, simage[S::split_words(n_in) + S::plain_words(n_in)]);
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_row_groups = (int)((rows + 16 * kMlWaves - 1) / (16 * kMlWaves));

  for (int grp = blockIdx.x; grp < n_row_groups; grp += gridDim.x) {
    const int64_t row = (int64_t)grp * (16 * kMlWaves) + wave * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* xr = x + rowc * n_in + 4 * q;
    const float* zr = z + rowc * n_in + 4 * q;
    // the four 16-dim groups chunk c works on (two per K-step); groups past the row end read as zeros
    // (whole rows: a group past the row end -- the odd last group of a K-step, the ring's requests past the last chunk --
    //  reads the row's last group instead, UNCONDITIONALLY: its weights in the image are zeros, and a load under a branch
    //  makes hipcc's wait counts fall back to vmcnt(0) for every load in flight, which empties the ring at every chunk)
    auto load4 = [&](const float* p, int g) -> f32x4 {
      if constexpr (RAG) {
        if (g >= n_groups16) return zero4;
        return row_load4<true>(p, 16 * g, 4 * q, n_in, vec);
      } else {
        return row_load4<false>(p, 16 * (g < n_groups16 ? g : n_groups16 - 1), 4 * q, n_in, vec);
      }
    };
    u32x4 st[S::STAGE_U4];
    auto request_operands = [&](int c, int& n4) {
      const int cc = c < nc ? c : nc - 1;
      const int left = n_ks - cc * KC;
      n4 = (left < KC ? left : KC) * (S::KS_WORDS / 4);
      const u32x4* src = reinterpret_cast<const u32x4*>(simage + (int64_t)cc * KC * S::KS_WORDS);
#pragma unroll
      for (int i = 0; i < S::STAGE_U4; ++i) {
        const int k = threadIdx.x + i * (kMlWaves * 64);
        st[i] = src[k < n4 ? k : 0];
      }
    };
    auto hand_over = [&](uint32_t* buf, int n4) {
      u32x4* dst = reinterpret_cast<u32x4*>(buf);
#pragma unroll
      for (int i = 0; i < S::STAGE_U4; ++i) {
        const int k = threadIdx.x + i * (kMlWaves * 64);
        if (k < n4) dst[k] = st[i];
      }
    };
    // Row data comes from HBM with a couple of microseconds of latency under load while a chunk of split MFMAs takes
    // well under one: rows are requested D chunks ahead into a ring of register sets (the chunk loop is unrolled by D
    // so that the ring index is static); a set is re-requested as soon as it has been turned into MFMA operands.
    constexpr int D = kMlRing<YT>;
    f32x4 xs[D][2 * KC], zs[D][2 * KC];
    __syncthreads();  // the previous group's last chunk is fully consumed
    {
      int n4;
#pragma unroll
      for (int u = 0; u < D; ++u)
#pragma unroll
        for (int i = 0; i < 2 * KC; ++i) {
          xs[u][i] = load4(xr, 2 * KC * u + i);
          zs[u][i] = load4(zr, 2 * KC * u + i);
        }
      request_operands(0, n4);
      hand_over(lds[0], n4);
    }
    __syncthreads();
    f32x4 mm[YT], mc[YT], vm[YT], vc[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) {
      mm[m] = *reinterpret_cast<const f32x4*>(bias + m * 16 + 4 * q);
      vm[m] = zero4;  // (exp(b_log_var) is added after the unscaling)
      mc[m] = zero4;
      vc[m] = zero4;
    }
    float mx = wmax <= kSplitWeightLimit ? 0.f : __builtin_inff();
    // one chunk: ring set u (static), chunk number c.  Whole groups of D chunks run WITHOUT a guard around the loads (see
    // load4); the last nc % D chunks follow one by one.
    auto chunk = [&](int c, auto u_c) {
      constexpr int u = decltype(u_c)::value;
      f16x8 bh[KC], bl[KC], ch[KC], cl[KC];
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        u32x2 ph0, pl0, ph1, pl1, sh0, sl0, sh1, sl1;
        split_tile(xs[u][2 * kk] * zs[u][2 * kk], ph0, pl0, mx);          // x * z        (mnf_linear.py:48)
        split_tile(xs[u][2 * kk + 1] * zs[u][2 * kk + 1], ph1, pl1, mx);
        split_tile(xs[u][2 * kk] * xs[u][2 * kk], sh0, sl0, mx);          // x ** 2       (:53)
        split_tile(xs[u][2 * kk + 1] * xs[u][2 * kk + 1], sh1, sl1, mx);
        bh[kk] = pair_operand(ph0, ph1), bl[kk] = pair_operand(pl0, pl1);
        ch[kk] = pair_operand(sh0, sh1), cl[kk] = pair_operand(sl0, sl1);
      }
      int n4_next = 0;
#if MNF_ML_OPS_FIRST
      // the next chunk's operands are requested BEFORE the rows of chunk c + D: vector-memory loads return in order, so
      // the wait for the operands (hand_over below) must not have the row requests in front of it
      request_operands(c + 1, n4_next);
#endif
#pragma unroll
      for (int i = 0; i < 2 * KC; ++i) {  // the set is free: rows of chunk c + D (past the end: zeros, no load)
        xs[u][i] = load4(xr, 2 * KC * (c + D) + i);
        zs[u][i] = load4(zr, 2 * KC * (c + D) + i);
      }
#if !MNF_ML_OPS_FIRST
      request_operands(c + 1, n4_next);
#endif
      const f16x8* A8 = reinterpret_cast<const f16x8*>(lds[c & 1]) + lane;  // + 64 * operand
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        if (c * KC + kk < n_ks) {
#pragma unroll
          for (int m = 0; m < YT; ++m) {
            split_mac(A8[64 * (2 * (kk * OPS + m))], A8[64 * (2 * (kk * OPS + m) + 1)], bh[kk], bl[kk], mm[m], mc[m]);
            split_mac(A8[64 * (2 * (kk * OPS + YT + m))], A8[64 * (2 * (kk * OPS + YT + m) + 1)], ch[kk], cl[kk],
                      vm[m], vc[m]);
          }
        }
      }
      hand_over(lds[(c + 1) & 1], n4_next);
      // (not __syncthreads(): its release fence is `s_waitcnt vmcnt(0)`, which would make every chunk wait for the
      // row loads of the D chunks ahead -- the whole point of the ring.  What the barrier has to order is LDS only:
      // this wave's reads of the current buffer and its writes to the next one.)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    const int nc_full = nc / D * D;
    for (int c0 = 0; c0 < nc_full; c0 += D) mnf_static_for<D>([&](auto u_c) { chunk(c0 + decltype(u_c)::value, u_c); });
    mnf_static_for<D>([&](auto u_c) {
      if (nc_full + decltype(u_c)::value < nc) chunk(nc_full + decltype(u_c)::value, u_c);
    });
    // range verdict for the whole 128-row group: flagged groups are redone by the fix-up kernel
    const int bad = __syncthreads_or(!(mx <= kSplitLimit) ? 1 : 0);
    if (threadIdx.x == 0) flags[grp] = bad;
    // out = mean + sqrt(var) * eps; lane (j, q) holds outputs 16 m + 4 q .. + 3 of row j
    if (live) {
#pragma unroll
      for (int m = 0; m < YT; ++m) {
        const f32x4 mean = mc[m] * kSplitInvScale + mm[m];
        const f32x4 bvar = *reinterpret_cast<const f32x4*>(bias + (YT + m) * 16 + 4 * q);
        const f32x4 var = (vc[m] * kSplitInvScale + vm[m]) * var_unscale + bvar;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * m + 4 * q + r;
          if (o < n_out) {
            const float e = eps ? eps[row * n_out + o] : ml_normal(seed, row, o);
            const float sd = sqrtf(var[r]);
            out[row * n_out + o] = mean[r] + sd * e;  // :56
            if (sd_out) sd_out[row * n_out + o] = sd;  // (training: the backward pass needs d out / d var = eps / (2 sd))
          }
        }
      }
    }
  }
}

// fp32 recomputation of the flagged 128-row groups, from the flat parameters (k order, fmaf chain)
__global__ void __launch_bounds__(256)
mnf_linear_fixup_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ eps,
                        float* __restrict__ out, float* __restrict__ sd_out, const float* __restrict__ flat,
                        const int32_t* __restrict__ flags, int64_t rows, int n_in, int n_out, float var_unscale,
                        uint64_t seed) {
  const int grp = blockIdx.x;
  if (!flags[grp]) return;
  const float* wm = flat;
  const float* wv = flat + (int64_t)n_out * n_in;
  const float* bm = wv + (int64_t)n_out * n_in;
  const float* bv = bm + n_out;
  const int64_t row0 = (int64_t)grp * (16 * kMlWaves);
  for (int t = threadIdx.x; t < 16 * kMlWaves * n_out; t += blockDim.x) {
    const int64_t row = row0 + t / n_out;
    const int o = t % n_out;
    if (row >= rows) continue;
    float mean = bm[o], var = 0.f;
    for (int k = 0; k < n_in; ++k) {
      const float xv = x[row * n_in + k];
      mean = fmaf(xv * z[row * n_in + k], wm[(int64_t)o * n_in + k], mean);
      var = fmaf(xv * xv, wv[(int64_t)o * n_in + k], var);
    }
    var = var * var_unscale + bv[o];
    const float e = eps ? eps[row * n_out + o] : ml_normal(seed, row, o);
    out[row * n_out + o] = mean + sqrtf(var) * e;
    if (sd_out) sd_out[row * n_out + o] = sqrtf(var);
  }
}

__global__ void mnf_linear_noise_kernel(uint64_t seed, float* __restrict__ eps, int64_t rows, int n_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * n_out) eps[i] = ml_normal(seed, i / n_out, (int)(i % n_out));
}

static int ml_tiles(int n_out) { return n_out < 1 || n_out > 64 ? 0 : (n_out + 15) / 16; }

template <int YT, bool RAG>
static int launch_ml(const float* x, const float* z, const float* eps, float* out, float* sd_out,
                     const uint32_t* simage, int32_t* flags, int64_t rows, int n_in, int n_out, float var_unscale,
                     uint64_t seed, int vec, hipStream_t stream) {
  static DeviceMemo memo;
  const int resident = memo.get(
      [](int dev) { return resident_by_occupancy(mnf_linear_fwd_kernel<YT, RAG>, kMlWaves * 64, dev, 1); });
  const int64_t n_groups = (rows + 16 * kMlWaves - 1) / (16 * kMlWaves);
  const int64_t blocks = n_groups < resident ? n_groups : resident;
  tag_kernel("mnf_linear_fwd");
  hipLaunchKernelGGL((mnf_linear_fwd_kernel<YT, RAG>), dim3((unsigned)blocks), dim3(kMlWaves * 64), 0, stream, x, z, eps,
                     out, sd_out, simage, flags, rows, n_in, n_out, var_unscale, seed, vec);
  return check_launch();
}

}  // namespace mnf

extern "C" {

int mnf_mnf_linear_split_layout(int n_in, int n_out, int64_t* n_split_words, int64_t* n_plain_words) {
  if (!n_split_words || !n_plain_words || n_in < 1) return MNF_ERR_INVALID_ARG;
  const int yt = mnf::ml_tiles(n_out);
  if (yt == 0 || (int64_t)n_in * 64 * 8 >= (1ll << 30)) return MNF_ERR_UNSUPPORTED;
#define X(YT)                                              \
  if (yt == YT) {                                          \
    *n_split_words = mnf::MlShape<YT>::split_words(n_in);  \
    *n_plain_words = mnf::MlShape<YT>::plain_words(n_in);  \
    return MNF_OK;                                         \
  }
  X(1) X(2) X(3) X(4)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_mnf_linear_split_index(int n_in, int n_out, int32_t* idx_host) {
  if (!idx_host || n_in < 1) return MNF_ERR_INVALID_ARG;
  const int yt = mnf::ml_tiles(n_out);
#define X(YT)                                         \
  if (yt == YT) {                                     \
    mnf::build_ml_index<YT>(n_in, n_out, idx_host);   \
    return MNF_OK;                                    \
  }
  X(1) X(2) X(3) X(4)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_mnf_linear_fwd_train(const float* x, const float* z, const float* eps, uint64_t seed, float* out, float* sd_out,
                             const float* flat, const void* split_image, float var_unscale, int32_t* workspace,
                             int64_t rows, int n_in, int n_out, void* stream) {
  if (!x || !z || !out || !flat || !split_image || !workspace || rows < 0 || n_in < 1 || n_out < 1 ||
      !(var_unscale > 0.f))
    return MNF_ERR_INVALID_ARG;
  const int yt = mnf::ml_tiles(n_out);
  if (yt == 0) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  if (reinterpret_cast<uintptr_t>(split_image) & 15) return MNF_ERR_UNSUPPORTED;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(z)) & 15) == 0;
  const bool ragged = (n_in & 15) != 0 || !aligned;
  const int vec = aligned && (n_in & 3) == 0;
  const uint32_t* simage = static_cast<const uint32_t*>(split_image);
  hipStream_t s = (hipStream_t)stream;
  int rc = MNF_ERR_UNSUPPORTED;
#define X(YT)                                                                                                         \
  if (yt == YT)                                                                                                       \
    rc = ragged ? mnf::launch_ml<YT, true>(x, z, eps, out, sd_out, simage, workspace, rows, n_in, n_out, var_unscale, \
                                           seed, vec, s)                                                              \
                : mnf::launch_ml<YT, false>(x, z, eps, out, sd_out, simage, workspace, rows, n_in, n_out,             \
                                            var_unscale, seed, vec, s);
  X(1) X(2) X(3) X(4)
#undef X
  if (rc != MNF_OK) return rc;
  const int64_t n_groups = (rows + 16 * mnf::kMlWaves - 1) / (16 * mnf::kMlWaves);
  hipLaunchKernelGGL(mnf::mnf_linear_fixup_kernel, dim3((unsigned)n_groups), dim3(256), 0, s, x, z, eps, out, sd_out, flat,
                     workspace, rows, n_in, n_out, var_unscale, seed);
  return mnf::check_launch();
}

int mnf_mnf_linear_fwd(const float* x, const float* z, const float* eps, uint64_t seed, float* out, const float* flat,
                       const void* split_image, float var_unscale, int32_t* workspace, int64_t rows, int n_in, int n_out,
                       void* stream) {
  return mnf_mnf_linear_fwd_train(x, z, eps, seed, out, nullptr, flat, split_image, var_unscale, workspace, rows, n_in,
                                  n_out, stream);
}

int mnf_mnf_linear_noise(uint64_t seed, float* eps, int64_t rows, int n_out, void* stream) {
  if (!eps || rows < 0 || n_out < 1) return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  const int64_t n = rows * n_out;
  hipLaunchKernelGGL(mnf::mnf_linear_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     seed, eps, rows, n_out);
  return mnf::check_launch();
}

}  // extern "C"
