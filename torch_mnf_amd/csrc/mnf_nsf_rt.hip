// NSF_CL.forward / .inverse (torch_mnf/flows/spline_flow.py:249-285) for ANY dim, K <= 16 and hidden widths 4 .. 64 on the
// f16 matrix pipe: run-time layer count and widths (mnf_rt.h), weights read from the plain `flat` parameter vector.  Takes
// the calls the per-shape kernels of mnf_nsf_mfma.hip have no instantiation for (dim > 64, n_h = 32 at dim >= 48, K other
// than 5 / 8 / 10, ...); the VALU kernel of mnf_generic.hip keeps few rows and hidden layers narrower than 4 units.
//
// A wave owns one 16-row tile and runs the layer's two half-steps one after the other (:251-266, :270-284).  A half-step
// = the conditioner up to its last hidden vector (mnf_rt.h), then the output layer walked SLOT by slot: a slot is one
// element per lane -- lane (row j, q) owns the elements 16 g + 4 q + r of its row, slot = (g, r) -- and its output tiles
// are arranged (at staging time: any row of the weight matrix can go to any row of a block) so that after them the lane
// holds all 3K-1 raw spline parameters of ITS element in registers, at fixed positions 16 c + k (c = widths / heights /
// derivatives); the spline then runs on 64 lanes with compile-time indices (mnf_nsf_spline.h, chosen by a uniform
// switch over K).  The second half-step's conditioner reads the first one's result back from y (same wave, same
// addresses: program order).
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_host.h"
#include "mnf_nsf_spline.h"
#include "mnf_rt.h"

namespace mnf {

struct NsfRtArgs {
  const float* x;
  float* y;
  float* log_det;
  const float* flat;
  int64_t rows;
  int dim, K, inverse, accumulate;
  float T;
  int n_params, vec;  // vec: rows and halves are 16-byte aligned (dwordx4 row accesses)
  int cb, bt;  // LDS plan (mnf_rt.h Source)
  int block_words, bias_words;
  NetDesc f1, f2;
};

// valid parameter tiles of one slot: tile t' holds positions 4 t' .. 4 t' + 3; c = t' / 4, k0 = 4 (t' % 4)
__host__ __device__ inline bool nsf_tile_valid(int tp, int K) {
  const int c = tp >> 2, k0 = 4 * (tp & 3);
  return k0 < (c < 2 ? K : K - 1);
}
__host__ __device__ inline int nsf_tiles_valid(int K) { return 2 * ((K + 3) / 4) + (K - 1 + 3) / 4; }
// the tv-th valid tile
__device__ __forceinline__ int nsf_tile_of(int tv, int K) {
  const int nw = (K + 3) >> 2;
  return tv < nw ? tv : tv < 2 * nw ? 4 + (tv - nw) : 8 + (tv - 2 * nw);
}

// Output layer of one conditioner net, walked [slot][valid tile][K-step] from slot s0: block row i = 4 q' + r' is the
// weight row of (element 16 g + 4 q' + r, position 4 t' + r').
struct NsfOutFetch {  // digits (K-step, valid tile, slot - s0)
  const float* W;  // (P H) x n_in
  int n_in, H, K, R0, R1, s0;  // R0 = KS, R1 = TV
  __device__ __forceinline__ void load(int ks, int tv, int sl, int i, int q, f32x4& va, f32x4& vb) const {
    const int slot = s0 + sl, g = slot >> 2, r = slot & 3, tp = nsf_tile_of(tv, K);
    const int e = 16 * g + 4 * (i >> 2) + r, pos = 4 * tp + (i & 3), c = pos >> 4, kk = pos & 15;
    const bool ok = e < H && kk < (c < 2 ? K : K - 1);
    const int c0 = 32 * ks;
    const bool aligned = (n_in & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    rt::load_row8(W + (int64_t)(ok ? e * (3 * K - 1) + c * K + kk : 0) * n_in, ok, c0, n_in, aligned, q, va, vb);
  }
};
struct NsfOutBias {  // tile t = (slot - s0) * TV + tv
  const float* b;
  int H, K, TV, s0;
  __device__ __forceinline__ float operator()(int t, int u) const {
    const int tv = t % TV, slot = s0 + t / TV;
    const int g = slot >> 2, r = slot & 3, tp = nsf_tile_of(tv, K);
    const int e = 16 * g + 4 * (u >> 2) + r, pos = 4 * tp + (u & 3), c = pos >> 4, kk = pos & 15;
    const bool ok = e < H && kk < (c < 2 ? K : K - 1);
    const float v = b[ok ? e * (3 * K - 1) + c * K + kk : 0];
    return ok ? v : 0.f;
  }
};

// The output layer and the spline of one half-step for a compile-time K (the tiles of a slot and the positions of its
// parameters are then static: the slot's LDS reads and products are scheduled together); `inv` is run-time.
template <int MT_MAX, int K, bool PREFILL, typename Src>
__device__ __forceinline__ float nsf_rt_slots(const NsfRtArgs& a, Src& src, const NetDesc& nd, float wup,
                                              const rt::Hidden<MT_MAX, 1>& h, const float* xrow, float* yrow, int act_off,
                                              bool live, bool inv) {
  using namespace rt;
  const bool VEC = a.vec != 0;  // (uniform)
  constexpr int NW_ = (K + 3) / 4, ND_ = (K - 1 + 3) / 4, TV = 2 * NW_ + ND_;  // tiles of widths / heights, derivatives
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const int H = a.dim / 2;
  const int L = nd.n_lin - 1;
  const int KS = steps32(16 * tiles16(nd.sizes[L]));
  const int S = 4 * tiles16(H);  // slots (g, r); elements >= H are padding
  int SO = Src::resident ? S : src.cb / (TV * KS);  // slots per chunk
  if (!Src::resident && SO > src.bt / TV) SO = src.bt / TV;
  if (SO < 1) SO = 1;
  const float* W = a.flat + nd.w_off[L];
  const float* B = a.flat + nd.b_off[L];
  float lad_sum = 0.f;
  f32x4 vin = f32x4{0.f, 0.f, 0.f, 0.f}, vout = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < S; s0 += SO) {
    const int so = S - s0 < SO ? S - s0 : SO;
    const Chunk c = src.template chunk<PREFILL>(so * TV * KS, NsfOutFetch{W, nd.sizes[L], H, K, KS, TV, s0}, so * TV,
                                                NsfOutBias{B, H, K, TV, s0});
    if (PREFILL) continue;
#pragma unroll 1
    for (int sl = 0; sl < so; ++sl) {
      const int slot = s0 + sl, g = slot >> 2, r = slot & 3;
      if (r == 0) vin = load4(xrow + act_off, 16 * g + 4 * q, H, VEC);  // (uniform) a new float4 group of the row
      // the slot's parameter tiles -> the lane's element's 3K-1 raw parameters
      float p[3 * K - 1];
#pragma unroll
      for (int tv = 0; tv < TV; ++tv) {
        f32x4 o[1];
        out_tile<MT_MAX, 1>(c.A, (sl * TV + tv) * KS, KS, c.bias + (sl * TV + tv) * 16, lane, q, h, wup, o);
        const int cgrp = tv < NW_ ? 0 : tv < 2 * NW_ ? 1 : 2, k0 = 4 * (tv - cgrp * NW_);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0 + e < (cgrp < 2 ? K : K - 1)) p[cgrp * K + k0 + e] = o[0][e];
      }
      float out, lad;
      if (inv) rqs_regs<K, true, 3 * K - 1>(vin[0], a.T, p, out, lad);
      else rqs_regs<K, false, 3 * K - 1>(vin[0], a.T, p, out, lad);
      const bool real = 16 * g + 4 * q + r < H;
      lad_sum += real ? lad : 0.f;
      // the group's float4s rotate by one element per slot: component 0 is always the current one
      vin = f32x4{vin[1], vin[2], vin[3], vin[0]};
      vout = f32x4{vout[1], vout[2], vout[3], out};
      if (r == 3) store4(yrow + act_off, 16 * g + 4 * q, H, VEC, live, vout);
    }
  }
  return lad_sum;
}

template <int MT_MAX, bool PREFILL, typename Src>
__device__ __forceinline__ void nsf_rt_block(const NsfRtArgs& a, Src& src, float wup, int64_t row0) {
  using namespace rt;
  const bool VEC = a.vec != 0;  // (uniform)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const int H = a.dim / 2;
  const int64_t r = row0 + (int64_t)wave * 16 + j;
  const bool live = !PREFILL && r < a.rows;
  const int64_t rc = r < a.rows ? r : a.rows - 1;
  const float* xrow = a.x + rc * a.dim;
  float* yrow = a.y + rc * a.dim;
  float lad = 0.f;
  // forward: f1(lower) moves upper, then f2(upper') moves lower (:251-266); inverse: f2(upper) moves lower back, then
  // f1(lower') moves upper back (:270-284).  The second step's conditioner input is the first step's result, read from y.
#pragma unroll 1
  for (int step = 0; step < 2; ++step) {
    const bool first_net = (step == 0) != (a.inverse != 0);  // f1 in step 0 forward / step 1 inverse
    const NetDesc& nd = first_net ? a.f1 : a.f2;
    const int cond_off = first_net ? 0 : H, act_off = first_net ? H : 0;
    const float* cond_row = step == 0 ? xrow : yrow;
    Hidden<MT_MAX, 1> h;
    auto load_x = [&](int, int ks, f32x4& xa, f32x4& xb) {
      const int c0 = 32 * ks + 4 * q;
      xa = load4(cond_row + cond_off, c0, H, VEC);
      xb = load4(cond_row + cond_off, c0 + 16, H, VEC);
    };
    auto use_x = [&](int, int, const f32x4&, const f32x4&) {};
    net_to_hidden<MT_MAX, 1, PREFILL>(src, a.flat, nd, nd.n_lin - 1, -1, wup, lane, q, load_x, use_x, h);
    switch (a.K) {  // (uniform)
#define MNF_NSF_RT_CASE(KK) \
  case KK: lad += nsf_rt_slots<MT_MAX, KK, PREFILL>(a, src, nd, wup, h, xrow, yrow, act_off, live, a.inverse != 0); break;
      MNF_NSF_RT_CASE(2) MNF_NSF_RT_CASE(3) MNF_NSF_RT_CASE(4) MNF_NSF_RT_CASE(5) MNF_NSF_RT_CASE(6) MNF_NSF_RT_CASE(7)
      MNF_NSF_RT_CASE(8) MNF_NSF_RT_CASE(9) MNF_NSF_RT_CASE(10) MNF_NSF_RT_CASE(11) MNF_NSF_RT_CASE(12) MNF_NSF_RT_CASE(13)
      MNF_NSF_RT_CASE(14) MNF_NSF_RT_CASE(15) MNF_NSF_RT_CASE(16)
#undef MNF_NSF_RT_CASE
      default: break;
    }
  }
  if (PREFILL) return;
  const float total = sum_over_q(lad);
  if (q == 0 && live && a.log_det) a.log_det[r] = a.accumulate ? a.log_det[r] + total : total;
}

template <int MT_MAX, int NW, bool RESIDENT>
__global__ void __launch_bounds__(NW * 64) nsf_rt_kernel(NsfRtArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t rt_lds[];
  float* scratch = reinterpret_cast<float*>(rt_lds);
  uint32_t* blocks = rt_lds + 16;
  float* bias = reinterpret_cast<float*>(blocks + a.block_words);
  const float wmax = rt::block_weight_max(a.flat, a.n_params, scratch);
  const int e = rt::weight_exponent(wmax);
  const float wup = rt::pow2f(e);
  rt::Source<RESIDENT> src{blocks, bias, a.cb, a.bt, 0, 0, 0, rt::pow2f(-e), 0};
  if (RESIDENT) {
    nsf_rt_block<MT_MAX, true>(a, src, wup, 0);
    __syncthreads();
  }
  const int64_t rows_per_block = (int64_t)(blockDim.x >> 6) * 16;
  const int64_t n_blocks = (a.rows + rows_per_block - 1) / rows_per_block;
  for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    src.slot = 0;
    src.btile = 0;
    nsf_rt_block<MT_MAX, false>(a, src, wup, b * rows_per_block);
  }
}

template <typename K>
static void nsf_rt_allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// MNF_ERR_UNSUPPORTED: the shape is outside the run-time-shaped kernel too (the caller runs the VALU kernel)
int nsf_rt_launch(const float* x, float* y, float* log_det, int accumulate, const float* flat, int64_t rows, int dim, int K,
                  float tail_bound, int inverse, int n_hidden, const int* hidden, hipStream_t stream) {
  if (!flat || n_hidden < 1 || K < 2 || K > 16 || rows * dim >= (1ll << 40)) return MNF_ERR_UNSUPPORTED;
  NsfRtArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.log_det = log_det; a.flat = flat; a.rows = rows; a.dim = dim; a.K = K; a.T = tail_bound;
  a.inverse = inverse != 0; a.accumulate = accumulate != 0;
  const int H = dim / 2, P = 3 * K - 1;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = H;
  int mn = 1 << 30, mxh = 0;
  for (int i = 0; i < n_hidden; ++i) {
    sizes[1 + i] = hidden[i];
    mn = hidden[i] < mn ? hidden[i] : mn;
    mxh = hidden[i] > mxh ? hidden[i] : mxh;
  }
  sizes[n_hidden + 1] = P * H;  // spline_flow.py:246
  if (mn < 4 || mxh > 64 || (int64_t)P * H * mxh >= (1ll << 30)) return MNF_ERR_UNSUPPORTED;
  int64_t off = fill_net(a.f1, n_hidden + 2, sizes, 0);
  off += fill_net(a.f2, n_hidden + 2, sizes, off);
  a.n_params = (int)off;
  a.vec = dim % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  // the resident image: both nets, blocks and bias tiles
  int64_t n_blocks = 0, n_bias = 0;
  for (int l = 0; l < n_hidden; ++l) {
    const int in_cols = l == 0 ? H : 16 * ((sizes[l] + 15) / 16);
    n_blocks += (int64_t)((in_cols + 31) / 32) * ((sizes[l + 1] + 15) / 16);
    n_bias += (sizes[l + 1] + 15) / 16;
  }
  const int KS = (16 * ((sizes[n_hidden] + 15) / 16) + 31) / 32, TV = nsf_tiles_valid(K), S = 4 * ((H + 15) / 16);
  n_blocks += (int64_t)S * TV * KS;
  n_bias += (int64_t)S * TV;
  n_blocks *= 2;
  n_bias *= 2;
  constexpr int kResidentBytes = 150 * 1024, kStreamBlocks = 24, kStreamBias = 24;
  const bool resident = n_blocks * 2048 + n_bias * 64 <= kResidentBytes;
  if (resident) {
    a.cb = (int)n_blocks;
    a.bt = (int)n_bias;
    a.block_words = (int)n_blocks * rt::kBlockWords;
    a.bias_words = (int)n_bias * 16;
  } else {
    if (TV * KS > kStreamBlocks) return MNF_ERR_UNSUPPORTED;
    a.cb = kStreamBlocks;
    a.bt = kStreamBias;
    a.block_words = 2 * kStreamBlocks * rt::kBlockWords;
    a.bias_words = 2 * kStreamBias * 16;
  }
  const size_t lds = 64 + (size_t)a.block_words * 4 + (size_t)a.bias_words * 4;
  constexpr int NW = 8;
  static DeviceMemo attr;
  attr.get([&](int) {
    nsf_rt_allow_big_lds(nsf_rt_kernel<4, NW, true>);
    nsf_rt_allow_big_lds(nsf_rt_kernel<4, NW, false>);
    return 1;
  });
  auto kernel = resident ? nsf_rt_kernel<4, NW, true> : nsf_rt_kernel<4, NW, false>;
  const int nw = resident && lds <= 79 * 1024 ? 4 : NW;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nw * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const int64_t rows_per_block = (int64_t)nw * 16;
  const int64_t need = (rows + rows_per_block - 1) / rows_per_block;
  int64_t grid = (int64_t)per_cu * device_cus(current_device());
  if (grid > need) grid = need;
  tag_kernel("nsf_rt");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nw * 64), lds, stream, a);
  return check_launch();
}

}  // namespace mnf
