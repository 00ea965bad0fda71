// AffineHalfFlow.forward / .inverse (torch_mnf/flows/affine_half_flow.py:44-66) for ANY conditioner shape on the f16
// matrix pipe: run-time layer count and widths (mnf_rt.h), weights read from the plain `flat` parameter vector.  Takes
// every call the per-shape kernels (mnf_ahf_split.hip, mnf_ahf_mfma.hip) have no instantiation for -- h_sizes of any
// length >= 1, hidden widths 4 .. 256, any even dim -- and the VALU kernel of mnf_generic.hip keeps the rest (few rows,
// hidden layers narrower than 4 units, no hidden layer at all).
//
// A wave owns NTL 16-row tiles: the s-net and the t-net run one after the other up to their last hidden vectors (the
// conditioning half streamed from memory K-step by K-step, copied to y on the way), then the output layer is walked
// tile by tile: s and t of 16 columns, the affine transform of those columns, the row's log|det J| in registers.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_host.h"
#include "mnf_rt.h"

namespace mnf {

struct AhfRtArgs {
  const float* x;
  float* y;
  float* log_det;
  float* ysq;
  const float* flat;
  int64_t rows;
  int dim, parity, inverse, accumulate, has_scale, has_shift;
  int n_params;
  int vec;                   // rows and halves are 16-byte aligned: dwordx4 row accesses
  int cb, bt;                // LDS plan (mnf_rt.h Source)
  int block_words, bias_words;
  NetDesc s_net, t_net;
};

constexpr float kLog2e = 1.4426950408889634f;

template <int MT_MAX, int NTL, int VEC, bool PREFILL, typename Src>  // VEC: 0 / 1, or 2 = a.vec
__device__ __forceinline__ void ahf_rt_block(const AhfRtArgs& a, Src& src, float wup, int64_t row0) {
  using namespace rt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const int H = a.dim / 2;
  const int cond_off = a.parity ? H : 0, act_off = a.parity ? 0 : H;
  const bool vec = VEC == 2 ? a.vec != 0 : VEC == 1;
  const float* xrow[NTL];
  float* yrow[NTL];
  bool live[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const int64_t r = row0 + (int64_t)(wave * NTL + t) * 16 + j;
    live[t] = !PREFILL && r < a.rows;
    const int64_t rc = r < a.rows ? r : a.rows - 1;
    xrow[t] = a.x + rc * a.dim;
    yrow[t] = a.y + rc * a.dim;
  }
  float sq[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) sq[t] = 0.f;

  const NetDesc& any_net = a.has_scale ? a.s_net : a.t_net;
  const int n_hid = any_net.n_lin - 1;  // hidden vectors per net
  Hidden<MT_MAX, NTL> hs, ht;
  // the conditioning half: B operands of the first layer; the first net's pass also copies it to y (:50, :60-61)
  const int n_nets = (a.has_scale ? 1 : 0) + (a.has_shift ? 1 : 0);
  auto load_x = [&](int t, int ks, f32x4& xa, f32x4& xb) {
    const int c0 = 32 * ks + 4 * q;
    xa = load4(xrow[t] + cond_off, c0, H, vec);
    xb = load4(xrow[t] + cond_off, c0 + 16, H, vec);
  };
  if (MT_MAX == 4 && n_nets == 2) {
    // both nets' first layers in ONE pass over the conditioning half (it is loaded, copied to y and split once; at wide
    // dims the input side is most of the layer), then each net's hidden layers
    auto use_x = [&](int t, int ks, const f32x4& xa, const f32x4& xb) {
      const int c0 = 32 * ks + 4 * q;
      store4(yrow[t] + cond_off, c0, H, vec, live[t], xa);
      store4(yrow[t] + cond_off, c0 + 16, H, vec, live[t], xb);
      sq[t] += xa[0] * xa[0] + xa[1] * xa[1] + xa[2] * xa[2] + xa[3] * xa[3] + xb[0] * xb[0] + xb[1] * xb[1] + xb[2] * xb[2] +
               xb[3] * xb[3];
    };
    const NetDesc* const nds[2] = {&a.s_net, &a.t_net};
    Hidden<MT_MAX, NTL> h2[2];
    first_layer<MT_MAX, NTL, PREFILL, 2>(src, a.flat, nds, n_hid != 0, wup, lane, q, load_x, use_x, h2);
    hs = h2[0];
    ht = h2[1];
    hidden_layers<MT_MAX, NTL, PREFILL>(src, a.flat, a.s_net, n_hid, -1, wup, lane, q, hs);
    hidden_layers<MT_MAX, NTL, PREFILL>(src, a.flat, a.t_net, n_hid, -1, wup, lane, q, ht);
  } else {
#pragma unroll 1
    for (int net = 0; net < n_nets; ++net) {
      const bool copy = net == 0;
      auto use_x = [&](int t, int ks, const f32x4& xa, const f32x4& xb) {
        const int c0 = 32 * ks + 4 * q;
        store4(yrow[t] + cond_off, c0, H, vec, live[t] && copy, xa);
        store4(yrow[t] + cond_off, c0 + 16, H, vec, live[t] && copy, xb);
        const float ss = xa[0] * xa[0] + xa[1] * xa[1] + xa[2] * xa[2] + xa[3] * xa[3] + xb[0] * xb[0] + xb[1] * xb[1] +
                         xb[2] * xb[2] + xb[3] * xb[3];
        sq[t] += copy ? ss : 0.f;
      };
      // (s_net, t_net are both filled: an absent net is a copy of the other one; net 0 = the first PRESENT net)
      if (net == 1) hs = ht;  // (both nets present: the s-net's vector moves over, the t-net's takes its place)
      net_to_hidden<MT_MAX, NTL, PREFILL>(src, a.flat, net == 0 && a.has_scale ? a.s_net : a.t_net, n_hid, -1, wup, lane, q,
                                          load_x, use_x, ht);
    }
  }
  const bool both = n_nets == 2;  // else the one present net's vector is in ht

  // ---- output layer, tile by tile: blocks [tile][head][K-step]
  const int L = any_net.n_lin - 1;
  const int KS = steps32(16 * tiles16(any_net.sizes[L])), M = tiles16(H);
  const int heads = (a.has_scale ? 1 : 0) + (a.has_shift ? 1 : 0);
  int MO = Src::resident ? M : src.cb / (heads * KS);
  if (MO > src.bt / heads && !Src::resident) MO = src.bt / heads;
  if (MO < 1) MO = 1;
  const float* W0 = a.flat + any_net.w_off[L];
  const float* B0 = a.flat + any_net.b_off[L];
  const int64_t w_stride = (int64_t)a.t_net.w_off[L] - a.s_net.w_off[L], b_stride = (int64_t)a.t_net.b_off[L] - a.s_net.b_off[L];
  float ld[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) ld[t] = 0.f;
  // the transformed half runs kRing output tiles ahead in a register ring (see net_to_hidden)
  f32x4 r1[kRing][NTL];
  if (!PREFILL) {
#pragma unroll
    for (int u = 0; u < kRing; ++u)
#pragma unroll
      for (int t = 0; t < NTL; ++t) r1[u][t] = load4(xrow[t] + act_off, 16 * (u < M ? u : M - 1) + 4 * q, H, vec);
  }
  Chunk c{nullptr, nullptr};
  int next_start = 0, chunk_start = 0;
  for (int m_base = 0; m_base < M; m_base += kRing) {
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
      const int m = m_base + u;
      if (m >= M) continue;
      if (m == next_start) {  // (uniform) a new chunk of output tiles starts here
        const int mo = M - m < MO ? M - m : MO;
        c = src.template chunk<PREFILL>(mo * heads * KS, DenseMMajor{W0, any_net.sizes[L], H, KS, m, heads, w_stride}, mo * heads,
                                        DenseBiasHeads{B0, H, m, heads, b_stride});
        chunk_start = m;
        next_start = m + mo;
      }
      if (PREFILL) continue;
      const int ml = m - chunk_start;
      const int col = 16 * m + 4 * q;
      f32x4 x1[NTL], s[NTL], tt[NTL];
      const int m_ahead = m + kRing < M ? m + kRing : M - 1;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        x1[t] = r1[u][t];
        r1[u][t] = load4(xrow[t] + act_off, 16 * m_ahead + 4 * q, H, vec);
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        tt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (both) {
        out_tile<MT_MAX, NTL>(c.A, (ml * 2) * KS, KS, c.bias + (ml * 2) * 16, lane, q, hs, wup, s);
        out_tile<MT_MAX, NTL>(c.A, (ml * 2 + 1) * KS, KS, c.bias + (ml * 2 + 1) * 16, lane, q, ht, wup, tt);
      } else {
        f32x4 o[NTL];
        out_tile<MT_MAX, NTL>(c.A, ml * KS, KS, c.bias + ml * 16, lane, q, ht, wup, o);
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          if (a.has_scale) s[t] = o[t];
          else tt[t] = o[t];
        }
      }
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        f32x4 y1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // forward: exp(s) z1 + t (:57); inverse: (z1 - t) / exp(s) (:53)
          y1[r] = a.inverse ? (x1[t][r] - tt[t][r]) * __builtin_amdgcn_exp2f(-s[t][r] * kLog2e)
                            : __builtin_amdgcn_exp2f(s[t][r] * kLog2e) * x1[t][r] + tt[t][r];
          ld[t] += s[t][r];  // (padded columns: zero weights and bias, s = 0)
          sq[t] += col + r < H ? y1[r] * y1[r] : 0.f;
        }
        store4(yrow[t] + act_off, col, H, vec, live[t], y1);
      }
    }
  }
  if (PREFILL) return;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const int64_t r = row0 + (int64_t)(wave * NTL + t) * 16 + j;
    const float total = sum_over_q(a.inverse ? -ld[t] : ld[t]);  // log_det = s.sum(1), sign flipped on the way back (:55, :62)
    const float sqt = sum_over_q(sq[t]);
    if (q == 0 && live[t]) {
      if (a.log_det) a.log_det[r] = a.accumulate ? a.log_det[r] + total : total;
      if (a.ysq) a.ysq[r] = sqt;
    }
  }
}

template <int MT_MAX, int NTL, int NW, bool RESIDENT, int VEC>
__global__ void __launch_bounds__(NW * 64) ahf_rt_kernel(AhfRtArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t rt_lds[];
  float* scratch = reinterpret_cast<float*>(rt_lds);
  uint32_t* blocks = rt_lds + 16;
  float* bias = reinterpret_cast<float*>(blocks + a.block_words);
  const float wmax = rt::block_weight_max(a.flat, a.n_params, scratch);
  const int e = rt::weight_exponent(wmax);  // weights are staged as w 2^-e: the largest one just below 2^15
  const float wup = rt::pow2f(e);
  rt::Source<RESIDENT> src{blocks, bias, a.cb, a.bt, 0, 0, 0, rt::pow2f(-e), 0};
  if (RESIDENT) {
    ahf_rt_block<MT_MAX, NTL, VEC, true>(a, src, wup, 0);
    __syncthreads();
  }
  const int64_t rows_per_block = (int64_t)(blockDim.x >> 6) * NTL * 16;
  const int64_t n_blocks = (a.rows + rows_per_block - 1) / rows_per_block;
  for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    src.slot = 0;
    src.btile = 0;
    ahf_rt_block<MT_MAX, NTL, VEC, false>(a, src, wup, b * rows_per_block);
  }
}

// blocks and bias tiles of the whole conditioner (the resident image), and the largest hidden width
static void ahf_rt_plan(const NetDesc& nd, int H, int heads, int64_t& n_blocks, int64_t& n_bias, int& max_hidden) {
  n_blocks = 0;
  n_bias = 0;
  max_hidden = 0;
  const int L = nd.n_lin - 1;
  for (int l = 0; l < L; ++l) {
    const int in_cols = l == 0 ? nd.sizes[0] : 16 * ((nd.sizes[l] + 15) / 16);
    const int KS = (in_cols + 31) / 32, MT = (nd.sizes[l + 1] + 15) / 16;
    n_blocks += (int64_t)heads * KS * MT;
    n_bias += (int64_t)heads * MT;
    if (nd.sizes[l + 1] > max_hidden) max_hidden = nd.sizes[l + 1];
  }
  const int KS = (16 * ((nd.sizes[L] + 15) / 16) + 31) / 32, M = (H + 15) / 16;
  n_blocks += (int64_t)heads * KS * M;
  n_bias += (int64_t)heads * M;
}

template <typename K>
static void rt_allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int MT_MAX, int NTL, int NW>
static int ahf_rt_launch_class(AhfRtArgs& a, int64_t n_blocks, int64_t n_bias, hipStream_t stream) {
  constexpr int kResidentBytes = 158 * 1024, kStreamBlocks = 16, kStreamBias = 16;
  const int64_t resident_bytes = n_blocks * 2048 + n_bias * 64;
  // Rows that are not 16-byte aligned (dim not a multiple of 8, a view at an odd offset) have the resident variant only,
  // except in the widest class, whose streaming kernel takes the alignment at run time (a branch around every row access:
  // 15-20 % on the memory-bound shapes) and serves every width: ahf_rt_launch sends such a call there.
  const bool resident = resident_bytes <= kResidentBytes;
  constexpr int kStreamVec = MT_MAX == 16 ? 2 : 1;
  if (!resident && !a.vec && kStreamVec != 2) return MNF_ERR_UNSUPPORTED;
  if (resident) {
    a.cb = (int)n_blocks;
    a.bt = (int)n_bias;
    a.block_words = (int)n_blocks * rt::kBlockWords;
    a.bias_words = (int)n_bias * 16;
  } else {
    a.cb = kStreamBlocks;
    a.bt = kStreamBias;
    a.block_words = 2 * kStreamBlocks * rt::kBlockWords;
    a.bias_words = 2 * kStreamBias * 16;
  }
  const size_t lds = 64 + (size_t)a.block_words * 4 + (size_t)a.bias_words * 4;
  static DeviceMemo attr;
  attr.get([&](int) {
    rt_allow_big_lds(ahf_rt_kernel<MT_MAX, NTL, NW, true, 1>);
    rt_allow_big_lds(ahf_rt_kernel<MT_MAX, NTL, NW, false, kStreamVec>);
    rt_allow_big_lds(ahf_rt_kernel<MT_MAX, NTL, NW, true, 0>);
    return 1;
  });
  // workgroups of NW waves -- of 4 when the LDS footprint lets a CU hold two or more of them (they overlap each other's
  // barriers, staging and memory waits); persistent grid = what the occupancy query says is resident
  auto kernel = !resident ? ahf_rt_kernel<MT_MAX, NTL, NW, false, kStreamVec>
                          : a.vec ? ahf_rt_kernel<MT_MAX, NTL, NW, true, 1> : ahf_rt_kernel<MT_MAX, NTL, NW, true, 0>;
  const int nw = NW == 8 && resident && lds <= 79 * 1024 ? 4 : NW;  // (streaming: every wave of the CU shares one conversion of the weights)
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nw * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const int64_t rows_per_block = (int64_t)nw * NTL * 16;
  const int64_t need = (a.rows + rows_per_block - 1) / rows_per_block;
  int64_t grid = (int64_t)per_cu * device_cus(current_device());
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nw * 64), lds, stream, a);
  return check_launch();
}

// MNF_ERR_UNSUPPORTED: the shape is outside the run-time-shaped kernel too (the caller runs the VALU kernel)
int ahf_rt_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate, const float* flat, int64_t rows,
                  int dim, int parity, int inverse, int n_hidden, const int* hidden, int has_scale, int has_shift,
                  hipStream_t stream) {
  if (!flat || n_hidden < 1 || (!has_scale && !has_shift) || rows * dim >= (1ll << 40)) return MNF_ERR_UNSUPPORTED;
  AhfRtArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.log_det = log_det; a.ysq = ysq; a.flat = flat; a.rows = rows; a.dim = dim;
  a.parity = parity != 0; a.inverse = inverse != 0; a.accumulate = accumulate != 0;
  a.has_scale = has_scale != 0; a.has_shift = has_shift != 0;
  const int H = dim / 2;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = H;
  int mn = 1 << 30;
  for (int i = 0; i < n_hidden; ++i) {
    sizes[1 + i] = hidden[i];
    mn = hidden[i] < mn ? hidden[i] : mn;
  }
  sizes[n_hidden + 1] = H;
  if (mn < 4) return MNF_ERR_UNSUPPORTED;  // a sum of one or two split products is not a 1e-5 sum (flows.py _MIN_SPLIT_HIDDEN)
  int64_t off = 0;
  if (has_scale) off += fill_net(a.s_net, n_hidden + 2, sizes, off);
  if (has_shift) off += fill_net(a.t_net, n_hidden + 2, sizes, off);
  if (!has_scale) a.s_net = a.t_net;
  if (!has_shift) a.t_net = a.s_net;
  if (off >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;
  a.n_params = (int)off;
  a.vec = dim % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  int64_t n_blocks, n_bias;
  int max_hidden;
  const int heads = (has_scale ? 1 : 0) + (has_shift ? 1 : 0);
  ahf_rt_plan(a.s_net, H, heads, n_blocks, n_bias, max_hidden);
  tag_kernel("ahf_rt");
  if (max_hidden > 256) return MNF_ERR_UNSUPPORTED;
  int rc = MNF_ERR_UNSUPPORTED;
  if (max_hidden <= 64) rc = ahf_rt_launch_class<4, 1, 8>(a, n_blocks, n_bias, stream);
  else if (max_hidden <= 128) rc = ahf_rt_launch_class<8, 1, 8>(a, n_blocks, n_bias, stream);
  if (rc == MNF_ERR_UNSUPPORTED) rc = ahf_rt_launch_class<16, 1, 4>(a, n_blocks, n_bias, stream);
  return rc;
}

}  // namespace mnf
