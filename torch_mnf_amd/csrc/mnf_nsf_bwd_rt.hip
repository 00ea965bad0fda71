// Gradients of NSF_CL.forward / .inverse (torch_mnf/flows/spline_flow.py:249-285 under loss.backward(); the reference
// trains through it: tests/test_flows.py:89-99) for ANY dim, K <= 16 and three (1 .. 4) hidden layers of widths 4 .. 64 on
// the f16 matrix pipe: run-time shapes (mnf_rt.h, mnf_rt_bwd.h), weights read from the plain `flat` parameter vector.
// Takes the calls the per-shape tile kernel (mnf_nsf_bwd_tile.hip: dim a multiple of 8 up to 64, hidden width <= 16,
// K = 5 / 8 / 10) has no instantiation for.
//
// The layer's two half-steps are differentiated in reverse order.  Per half-step (a wave = one 16-row tile): forward
// recompute of its conditioner keeping every hidden vector; then slot by slot as in the forward kernel (mnf_nsf_rt.hip:
// a slot = one element per lane, its 3K-1 raw parameters in the lane's registers) the spline's reverse-mode derivative
// (mnf_nsf_spline_grad.h) gives the element's input cotangent and the 3K-1 parameter cotangents, which go back into the
// slot's tile layout: the first step of the delta chain  W_out^T g  (turned blocks of the slot's rows) and the slot's
// dW_out products through the LDS exchange area; then the hidden layers backwards and the first layer
// (mnf_rt_bwd.h backward_tail).  The second half-step's conditioner input is a column block of the layer's OUTPUT y (not
// recomputed); the cotangent that reaches it through the conditioner is parked in grad_x, where the other half-step
// picks it up as its output cotangent.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_host.h"
#include "mnf_nsf_spline_grad.h"
#include "mnf_rt_bwd.h"

namespace mnf {

struct NsfBwdRtArgs {
  const float* x;
  const float* y;
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  const float* gscale_dev;
  int64_t rows;
  int dim, K, inverse;
  float T;
  int n_params, vec;
  int cb, bt, block_words, bias_words;
  int ht_tiles, dt_tiles, ct_tiles;
  NetDesc f1, f2;
};

// the tv-th parameter tile of a slot (widths | heights | derivatives, four positions per tile) -> first position 16 c + k0
__host__ __device__ inline int nsfb_tiles(int K) { return 2 * ((K + 3) / 4) + (K - 1 + 3) / 4; }
__device__ __forceinline__ void nsfb_tile_pos(int tv, int K, int& c, int& k0) {
  const int nw = (K + 3) >> 2;
  c = tv < nw ? 0 : tv < 2 * nw ? 1 : 2;
  k0 = 4 * (tv - c * nw);
}
// weight row of (slot, tile tv, unit u of the tile): element 16 g + 4 (u >> 2) + r, parameter c K + k0 + (u & 3)
__device__ __forceinline__ int nsfb_row(int slot, int tv, int u, int K, int H) {
  int c, k0;
  nsfb_tile_pos(tv, K, c, k0);
  const int e = 16 * (slot >> 2) + 4 * (u >> 2) + (slot & 3), kk = k0 + (u & 3);
  return e < H && kk < (c < 2 ? K : K - 1) ? e * (3 * K - 1) + c * K + kk : -1;
}
struct NsfbRows {
  int slot, K, H;
  __device__ __forceinline__ int operator()(int m, int u) const { return nsfb_row(slot, m, u, K, H); }
};
// forward blocks of ONE slot's output tiles: digits (K-step, tile)
struct NsfbOutFetch {
  const float* W;
  int n_in, H, K, R0, slot;  // R0 = KS
  static constexpr int R1 = 1 << 30;
  __device__ __forceinline__ void load(int ks, int tv, int, int i, int q, f32x4& va, f32x4& vb) const {
    const int o = nsfb_row(slot, tv, i, K, H);
    const bool aligned = (n_in & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    rt::load_row8(W + (int64_t)(o >= 0 ? o : 0) * n_in, o >= 0, 32 * ks, n_in, aligned, q, va, vb);
  }
};
struct NsfbOutBias {
  const float* b;
  int H, K, slot;
  __device__ __forceinline__ float operator()(int t, int u) const {
    const int o = nsfb_row(slot, t, u, K, H);
    const float v = b[o >= 0 ? o : 0];
    return o >= 0 ? v : 0.f;
  }
};
// turned blocks of one slot: block row i = hidden unit 16 mh + i, K index = (tile 2 kp + (k >> 4), unit k & 15): digits (mh, kp)
struct NsfbOutTFetch {
  const float* W;
  int n_in, H, K, R0, slot, TV;  // R0 = hidden tiles
  static constexpr int R1 = 1 << 30;
  __device__ __forceinline__ void load(int mh, int kp, int, int i, int q, f32x4& va, f32x4& vb) const {
    const int u = 16 * mh + i;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int oa = 2 * kp < TV ? nsfb_row(slot, 2 * kp, 4 * q + e, K, H) : -1;
      const int ob = 2 * kp + 1 < TV ? nsfb_row(slot, 2 * kp + 1, 4 * q + e, K, H) : -1;
      const bool oka = oa >= 0 && u < n_in, okb = ob >= 0 && u < n_in;
      const float xa = W[oka ? (int64_t)oa * n_in + u : 0], xb = W[okb ? (int64_t)ob * n_in + u : 0];
      va[e] = oka ? xa : 0.f;
      vb[e] = okb ? xb : 0.f;
    }
  }
};

// The slots of one half-step for a compile-time K: parameters, spline derivative, chain start, dW_out.
template <int MT_MAX, int K, typename Src>
__device__ __forceinline__ void nsf_bwd_slots(const NsfBwdRtArgs& a, Src& src, const NetDesc& nd, const rt::BwdLds& lds, float wup,
                                              float gs, float inv_gs, const rt::Hidden<MT_MAX, 1>& h, const float* xrow,
                                              const float* gorow, float* gxrow, int act_off, bool live, float gl,
                                              rt::Acc<MT_MAX, 1>& accd, float& downd) {
  using namespace rt;
  const bool VEC = a.vec != 0;  // (uniform)
  constexpr int NW_ = (K + 3) / 4, ND_ = (K - 1 + 3) / 4, TV = 2 * NW_ + ND_, KSO = (TV + 1) / 2;
  const int lane = lds.lane, q = lds.q, wave = lds.wave, nw = lds.nw;
  const int H = a.dim / 2, L = nd.n_lin - 1;
  const int MTh = tiles16(nd.sizes[L]), KS = steps32(16 * MTh), S = 4 * tiles16(H);
  const int ht_last = exH_tile_of(nd, L);
  const float* W = a.flat + nd.w_off[L];
  const float* B = a.flat + nd.b_off[L];
  float* gflat = a.grad_flat;
  const float rowmask = live ? 1.f : 0.f;
  f32x4 vin = f32x4{0.f, 0.f, 0.f, 0.f}, gin = vin, vout = vin;
#pragma unroll 1
  for (int slot = 0; slot < S; ++slot) {
    const int g = slot >> 2, r = slot & 3;
    uint32_t* buf = src.cur_blocks();
    float* bbuf = src.cur_bias();
    stage_blocks(buf, TV * KS, NsfbOutFetch{W, nd.sizes[L], H, K, KS, slot}, src.wdown);
    stage_bias(bbuf, TV, NsfbOutBias{B, H, K, slot});
    uint32_t* bufT = buf + TV * KS * kBlockWords;
    stage_blocks(bufT, KSO * MTh, NsfbOutTFetch{W, nd.sizes[L], H, K, MTh, slot, TV}, src.wdown);
    src.commit();
    if (r == 0) {  // (uniform) a new float4 group of the row: the elements and their output cotangents
      vin = load4(xrow + act_off, 16 * g + 4 * q, H, VEC);
      gin = gorow ? load4(gorow + act_off, 16 * g + 4 * q, H, VEC) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float p[3 * K - 1], g_p[3 * K - 1];
#pragma unroll
    for (int tv = 0; tv < TV; ++tv) {
      f32x4 o[1];
      out_tile<MT_MAX, 1>(buf, tv * KS, KS, bbuf + tv * 16, lane, q, h, wup, o);
      const int cgrp = tv < NW_ ? 0 : tv < 2 * NW_ ? 1 : 2, k0 = 4 * (tv - cgrp * NW_);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e < (cgrp < 2 ? K : K - 1)) p[cgrp * K + k0 + e] = o[0][e];
    }
    float g_v;
    if (a.inverse) nsfgrad::rqs_grad<K, true>(vin[0], a.T, p, gin[0], gl, g_v, g_p);
    else nsfgrad::rqs_grad<K, false>(vin[0], a.T, p, gin[0], gl, g_v, g_p);
    const bool real = 16 * g + 4 * q + r < H;
    const float keep = real ? gs * rowmask : 0.f;
    // the parameter cotangents back in the slot's tile layout (times the gradient scale; padding and dead rows: zero)
    f32x4 gt[TV];
#pragma unroll
    for (int tv = 0; tv < TV; ++tv) {
      const int cgrp = tv < NW_ ? 0 : tv < 2 * NW_ ? 1 : 2, k0 = 4 * (tv - cgrp * NW_);
#pragma unroll
      for (int e = 0; e < 4; ++e) gt[tv][e] = k0 + e < (cgrp < 2 ? K : K - 1) ? g_p[cgrp * K + k0 + e] * keep : 0.f;
    }
    // chain start: accd += (turned blocks of the slot) x [tile 2 kp | tile 2 kp + 1]
#pragma unroll
    for (int kp = 0; kp < KSO; ++kp) {
      const f32x4 g0 = gt[2 * kp], g1 = 2 * kp + 1 < TV ? gt[2 * kp + 1 < TV ? 2 * kp + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
      f16x8 bh[1], bl[1];
      float mx = 0.f;
      split_kstep(g0, g1, downd, bh[0], bl[0], mx);
      if (__builtin_expect(wave_any(!(mx < kSplitLimit)), 0)) {
        float fm = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) fm = __builtin_fmaxf(fm, __builtin_fmaxf(finite_abs(g0[e]), finite_abs(g1[e])));
        const float want = pow2f(-down_exponent(max_over_q(fm), 13));
        if (want < downd) {
          const float f = want / downd;
#pragma unroll
          for (int m = 0; m < MT_MAX; ++m) {
            accd.main[0][m] *= f;
            accd.corr[0][m] *= f;
          }
          downd = want;
        }
        float unused = 0.f;
        split_kstep(g0, g1, downd, bh[0], bl[0], unused);
      }
      mac_kstep<MT_MAX, 1>(bufT, kp * MTh, MTh, lane, bh, bl, accd.main, accd.corr);
    }
    // dW_out, db_out of the slot's rows
    if (gflat) {
      const float sc = exchange_store<TV>(gt, TV, lds.exC, 0, 16 * wave, lane, lds.ident);
      if (lane == 0) lds.sC[wave] = sc;
      lds_barrier();
      dw_phase_rows(lds.exC, 0, TV, lds.exH, ht_last, MTh, lds.sC, lds.sH + L * 8, nw, inv_gs, gflat + nd.w_off[L],
                    gflat + nd.b_off[L], NsfbRows{slot, K, H}, nd.sizes[L], 0);
    }
    // the group's float4s rotate by one element per slot: component 0 is always the current one
    vin = f32x4{vin[1], vin[2], vin[3], vin[0]};
    gin = f32x4{gin[1], gin[2], gin[3], gin[0]};
    vout = f32x4{vout[1], vout[2], vout[3], real ? g_v : 0.f};
    if (r == 3) store4(gxrow + act_off, 16 * g + 4 * q, H, VEC, live, vout);
  }
}

template <int MT_MAX>
__global__ void __launch_bounds__(512) nsf_bwd_rt_kernel(NsfBwdRtArgs a) {
  using namespace rt;
  const bool VEC = a.vec != 0;  // (uniform) rows and halves are 16-byte aligned: dwordx4 row accesses
  extern __shared__ __attribute__((aligned(16))) uint32_t rt_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
  float* scratch = reinterpret_cast<float*>(rt_lds);
  uint32_t* blocks = rt_lds + kBwdHeadWords;
  float* bias = reinterpret_cast<float*>(blocks + a.block_words);
  const BwdLds lds = bwd_lds(rt_lds, bias + a.bias_words, a.ht_tiles, a.dt_tiles, a.ct_tiles);
  const float wmax = block_weight_max(a.flat, a.n_params, scratch);
  const int we = weight_exponent(wmax);
  const float wup = pow2f(we);
  Source<false> src{blocks, bias, a.cb, a.bt, 0, 0, 0, pow2f(-we), 0};
  const float gs = *a.gscale_dev, inv_gs = 1.f / gs;
  const int H = a.dim / 2;
  const int64_t n_blocks = (a.rows + 16 * nw - 1) / (16 * nw);
  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t r = blk * (16 * nw) + 16 * wave + j;
    const bool live = r < a.rows;
    const int64_t rc = live ? r : a.rows - 1;
    const float* xrow = a.x + rc * a.dim;
    const float* yrow = a.y + rc * a.dim;
    const float* gyrow = a.grad_y ? a.grad_y + rc * a.dim : nullptr;
    float* gxrow = a.grad_x + rc * a.dim;
    const float gl = a.grad_ld && live ? a.grad_ld[rc] : 0.f;
#pragma unroll 1
    for (int bstep = 0; bstep < 2; ++bstep) {
      // backward step 0 undoes the forward pass's SECOND half-step (its conditioner saw a half of the output y)
      const bool use_f1 = (bstep == 1) != (a.inverse != 0);
      const NetDesc& nd = use_f1 ? a.f1 : a.f2;
      const int cond_off = use_f1 ? 0 : H, act_off = use_f1 ? H : 0;
      const float* cond_row = bstep == 0 ? yrow : xrow;
      const float* gorow = bstep == 0 ? gyrow : gxrow;  // (step 1: what step 0 parked in grad_x)
      const int n_hid = nd.n_lin - 1;
      Hidden<MT_MAX, 1> h;
      {
        auto load_x = [&](int, int ks, f32x4& xa, f32x4& xb) {
          const int c0 = 32 * ks + 4 * q;
          xa = load4(cond_row + cond_off, c0, H, VEC);
          xb = load4(cond_row + cond_off, c0 + 16, H, VEC);
        };
        forward_keep<MT_MAX>(src, a.flat, nd, n_hid, -1, wup, lds, load_x, h);
      }
      Acc<MT_MAX, 1> accd;
      accd.zero();
      float downd = 1.f;
      switch (a.K) {  // (uniform)
#define MNF_NSF_BWD_CASE(KK) \
  case KK: nsf_bwd_slots<MT_MAX, KK>(a, src, nd, lds, wup, gs, inv_gs, h, xrow, gorow, gxrow, act_off, live, gl, accd, downd); break;
        MNF_NSF_BWD_CASE(2) MNF_NSF_BWD_CASE(3) MNF_NSF_BWD_CASE(4) MNF_NSF_BWD_CASE(5) MNF_NSF_BWD_CASE(6) MNF_NSF_BWD_CASE(7)
        MNF_NSF_BWD_CASE(8) MNF_NSF_BWD_CASE(9) MNF_NSF_BWD_CASE(10) MNF_NSF_BWD_CASE(11) MNF_NSF_BWD_CASE(12)
        MNF_NSF_BWD_CASE(13) MNF_NSF_BWD_CASE(14) MNF_NSF_BWD_CASE(15) MNF_NSF_BWD_CASE(16)
#undef MNF_NSF_BWD_CASE
        default: break;
      }
      f32x4 dv[MT_MAX];
      chain_result<MT_MAX>(accd, wup / downd, lds.meta_bits[n_hid * 64 + lane], dv);
      auto load_in = [&](int mi) { return load4(cond_row + cond_off, 16 * mi + 4 * q, H, VEC); };
      auto add_in = [&](int mi, const f32x4& g) {
        const int col = 16 * mi + 4 * q;
        const f32x4 base = bstep == 0 ? (gyrow ? load4(gyrow + cond_off, col, H, VEC) : f32x4{0.f, 0.f, 0.f, 0.f})
                                      : load4(gxrow + cond_off, col, H, VEC);
        store4(gxrow + cond_off, col, H, VEC, live, base + g);
      };
      backward_tail<MT_MAX>(src, a.flat, a.grad_flat, nd, n_hid, -1, dv, lds, wup, inv_gs, H, load_in, add_in);
    }
  }
}

}  // namespace mnf

using namespace mnf;

extern "C" int mnf_nsf_cl_bwd_rt(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x,
                                 float* grad_flat, const float* flat, const float* grad_scale_dev, int64_t rows, int dim, int K,
                                 float tail_bound, int inverse, int n_hidden, const int* hidden, void* stream) {
  if (!x || !y || !grad_x || !flat || !grad_scale_dev || rows < 0 || dim < 2 || (dim & 1) || K < 1 || !(tail_bound > 0.f) ||
      !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (1e-3 * K > 1.0) return MNF_ERR_DOMAIN;
  if (rows == 0) return MNF_OK;
  if (n_hidden < 1 || n_hidden > rt::kMaxBwdLayers || K < 2 || K > 16 || deterministic() || rows * dim >= (1ll << 40))
    return MNF_ERR_UNSUPPORTED;
  NsfBwdRtArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.gscale_dev = grad_scale_dev; a.rows = rows; a.dim = dim; a.K = K; a.T = tail_bound; a.inverse = inverse != 0;
  const int H = dim / 2, P = 3 * K - 1;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = H;
  int mn = 1 << 30, mxh = 0, ht = 0, dt = 0;
  for (int i = 0; i < n_hidden; ++i) {
    sizes[1 + i] = hidden[i];
    mn = hidden[i] < mn ? hidden[i] : mn;
    mxh = hidden[i] > mxh ? hidden[i] : mxh;
    ht += (hidden[i] + 15) / 16;
    dt = (hidden[i] + 15) / 16 > dt ? (hidden[i] + 15) / 16 : dt;
  }
  sizes[n_hidden + 1] = P * H;
  if (mn < 4 || mxh > 64 || (int64_t)P * H * mxh >= (1ll << 30)) return MNF_ERR_UNSUPPORTED;
  int64_t off = fill_net(a.f1, n_hidden + 2, sizes, 0);
  off += fill_net(a.f2, n_hidden + 2, sizes, off);
  a.n_params = (int)off;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec = dim % 8 == 0 && aligned(x) && aligned(y) && aligned(grad_x) && (!grad_y || aligned(grad_y));
  constexpr int MT_MAX = 4;
  const int TV = nsfb_tiles(K), MTh = (hidden[n_hidden - 1] + 15) / 16, KS = (16 * MTh + 31) / 32, KSO = (TV + 1) / 2;
  a.cb = TV * KS + KSO * MTh;  // a slot: its output tiles' blocks and their turned counterparts
  if (a.cb < 8) a.cb = 8;
  if (a.cb > 40) return MNF_ERR_UNSUPPORTED;
  a.bt = TV > MT_MAX ? TV : MT_MAX;
  a.block_words = 2 * a.cb * rt::kBlockWords;
  a.bias_words = 2 * a.bt * 16;
  a.ht_tiles = ht;
  a.dt_tiles = 0;  // (the deltas reuse the hidden vectors' tiles: mnf_rt_bwd.h backward_tail)
  (void)dt;
  a.ct_tiles = TV > MT_MAX ? TV : MT_MAX;
  int nw = 8;
  size_t lds = 0;
  for (; nw >= 1; --nw) {  // (any wave count: 6 or 7 waves where 8 do not fit)
    lds = (size_t)4 * rt::kBwdHeadWords + (size_t)a.block_words * 4 + (size_t)a.bias_words * 4 +
          rt::bwd_lds_bytes(nw, a.ht_tiles, a.dt_tiles, a.ct_tiles);
    if (lds <= 160 * 1024) break;
  }
  if (nw < 1) return MNF_ERR_UNSUPPORTED;
  auto kernel = nsf_bwd_rt_kernel<MT_MAX>;
  static DeviceMemo attr;
  attr.get([&](int) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nsf_bwd_rt_kernel<MT_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    return 1;
  });
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nw * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const int64_t need = (rows + 16 * nw - 1) / (16 * nw);
  int64_t grid = (int64_t)per_cu * device_cus(current_device());
  if (grid > need) grid = need;
  tag_kernel("nsf_bwd_rt");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nw * 64), lds, (hipStream_t)stream, a);
  return check_launch();
}
