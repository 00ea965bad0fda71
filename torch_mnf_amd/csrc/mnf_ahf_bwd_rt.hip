// Gradients of AffineHalfFlow.forward / .inverse (torch_mnf/flows/affine_half_flow.py:44-66 under loss.backward(); the
// reference trains through these layers: tests/test_flows.py:14-31) for ANY conditioner shape on the f16 matrix pipe:
// run-time layer count and widths (mnf_rt.h, mnf_rt_bwd.h), weights read from the plain `flat` parameter vector.  Takes
// the calls the per-shape gradient kernels (mnf_ahf_bwd_split.hip, mnf_ahf_bwd_mfma.hip: three hidden layers of at most
// 32 units) have no instantiation for: 1 .. 4 hidden layers of widths 4 .. 64, any even dim.
//
// A workgroup owns a block of 16 NW rows, a wave one tile of it.  Per net (s, then t): the forward recompute keeps every
// hidden vector (turned, in the LDS exchange area) and the row scales; the output layer is walked two 16-column tiles at
// a time -- s or t of the tiles, the cotangents g_s / g_t from grad_y, grad_ld (and y: the inverse direction's g_s = -g y
// - g_ld needs no second net), grad_x of the transformed half, the first step of the delta chain W_out^T g and the
// tiles' dW_out products --, then the hidden layers backwards (delta chain in registers, dW per layer through the
// exchange area), then grad_x of the conditioning half and dW of the first layer input tile by input tile.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_host.h"
#include "mnf_rt_bwd.h"

namespace mnf {

struct AhfBwdRtArgs {
  const float* x;
  const float* y;  // the layer's output for the same x (inverse direction only)
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  const float* gscale_dev;  // power of two that brings the cotangents near 1
  int64_t rows;
  int dim, parity, inverse, has_scale, has_shift;
  int n_params, vec;
  int cb, bt, block_words, bias_words;  // weight stream (mnf_rt.h Source<false>)
  int ht_tiles, dt_tiles, ct_tiles;     // exchange tiles: hidden vectors of one net | one layer's deltas | a chunk
  NetDesc s_net, t_net;
};

constexpr float kLog2eB = 1.4426950408889634f;

template <int MT_MAX>
__global__ void __launch_bounds__(512) ahf_bwd_rt_kernel(AhfBwdRtArgs a) {
  using namespace rt;
  const bool VEC = a.vec != 0;  // (uniform) rows are 16-byte aligned: dwordx4 row accesses
  extern __shared__ __attribute__((aligned(16))) uint32_t rt_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
  // LDS: [scratch 16][scales: sA 8, sC 8, sH (layers + 1) x 8][weights][bias][exchange: HT | DT | CT][meta: per wave sign bits]
  float* scratch = reinterpret_cast<float*>(rt_lds);
  uint32_t* blocks = rt_lds + kBwdHeadWords;
  float* bias = reinterpret_cast<float*>(blocks + a.block_words);
  const BwdLds lds = bwd_lds(rt_lds, bias + a.bias_words, a.ht_tiles, a.dt_tiles, a.ct_tiles);
  float* const sC = lds.sC;
  float* const sH = lds.sH;
  const Exchange& exH = lds.exH;
  const Exchange& exC = lds.exC;
  const f16x4& ident = lds.ident;

  const float wmax = block_weight_max(a.flat, a.n_params, scratch);
  const int we = weight_exponent(wmax);
  const float wup = pow2f(we);
  Source<false> src{blocks, bias, a.cb, a.bt, 0, 0, 0, pow2f(-we), 0};
  const float gs = *a.gscale_dev, inv_gs = 1.f / gs;
  const int H = a.dim / 2;
  const int cond_off = a.parity ? H : 0, act_off = a.parity ? 0 : H;
  const int n_nets = (a.has_scale ? 1 : 0) + (a.has_shift ? 1 : 0);
  const int64_t n_blocks = (a.rows + 16 * nw - 1) / (16 * nw);

  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t r = blk * (16 * nw) + 16 * wave + j;
    const bool live = r < a.rows;
    const int64_t rc = live ? r : a.rows - 1;
    const float* xrow = a.x + rc * a.dim;
    const float* yrow = a.y ? a.y + rc * a.dim : xrow;
    const float* gyrow = a.grad_y ? a.grad_y + rc * a.dim : xrow;
    float* gxrow = a.grad_x + rc * a.dim;
    const float gl = a.grad_ld && live ? a.grad_ld[rc] : 0.f;
    const float rowmask = live ? 1.f : 0.f;  // rows past the end add nothing to the parameter sums

#pragma unroll 1
    for (int pass = 0; pass < n_nets; ++pass) {
      const bool is_s = a.has_scale && pass == 0;
      const NetDesc& nd = is_s ? a.s_net : a.t_net;
      const int n_hid = nd.n_lin - 1, L = n_hid;
      float* gflat = a.grad_flat;
      // ---- forward recompute: every hidden vector goes, turned, into the exchange area; its sign bits into LDS
      Hidden<MT_MAX, 1> h;
      {
        auto load_x = [&](int, int ks, f32x4& xa, f32x4& xb) {
          const int c0 = 32 * ks + 4 * q;
          xa = load4(xrow + cond_off, c0, H, VEC);
          xb = load4(xrow + cond_off, c0 + 16, H, VEC);
        };
        forward_keep<MT_MAX>(src, a.flat, nd, n_hid, -1, wup, lds, load_x, h);
      }
      // ---- output layer, two 16-column tiles (one K-step of the chain) per chunk
      const int MTh = tiles16(nd.sizes[L]), KSh = steps32(16 * MTh), M = tiles16(H);
      // (the t pass behind an s pass needs e^{-s} in the inverse direction only, and only as g e^{-s}: that is the
      //  value-half cotangent the s pass stored in grad_x -- read back by the lane that wrote it, program order)
      const bool after_s = !is_s && a.has_scale;
      const int ht_last = exH_tile_of(nd, L);
      Acc<MT_MAX, 1> accd;
      accd.zero();
      float downd = 1.f;  // the row's running scale of the chain's first product (as in net_to_hidden)
      for (int m0 = 0; m0 < M; m0 += 2) {
        const int mo = M - m0 < 2 ? M - m0 : 2;
        uint32_t* buf = src.cur_blocks();
        float* bbuf = src.cur_bias();
        stage_blocks(buf, mo * KSh, DenseMMajor{a.flat + nd.w_off[L], nd.sizes[L], H, KSh, m0, 1, 0}, src.wdown);
        stage_bias(bbuf, mo, DenseBias{a.flat + nd.b_off[L], H, m0});
        const uint32_t* bufT = buf + mo * KSh * kBlockWords;
        stage_blocks(const_cast<uint32_t*>(bufT), MTh, DenseTKMajor{a.flat + nd.w_off[L], nd.sizes[L], H, MTh, m0 >> 1},
                     src.wdown);
        src.commit();
        f32x4 g2[2];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
          g2[ml] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ml < mo) {
            const int col = 16 * (m0 + ml) + 4 * q;
            f32x4 o[1], sv[1];
            out_tile<MT_MAX, 1>(buf, ml * KSh, KSh, bbuf + 16 * ml, lane, q, h, wup, o);
            sv[0] = is_s ? o[0] : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 x1 = load4(xrow + act_off, col, H, VEC);
            const f32x4 gy1 = a.grad_y ? load4(gyrow + act_off, col, H, VEC) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 y1 = a.inverse && is_s ? load4(yrow + act_off, col, H, VEC) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 gxs = a.inverse && after_s ? load4(gxrow + act_off, col, H, VEC) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 gx1;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              // forward: y = e^s v + t      g_v = g e^s     g_s = g e^s v + g_ld     g_t = g
              // inverse: y = (v - t) e^-s   g_v = g e^-s    g_s = -g y - g_ld        g_t = -g e^-s
              const float ex = __builtin_amdgcn_exp2f((a.inverse ? -sv[0][e4] : sv[0][e4]) * kLog2eB);
              gx1[e4] = gy1[e4] * ex;
              float g;
              if (is_s) g = a.inverse ? -gy1[e4] * y1[e4] - gl : gy1[e4] * ex * x1[e4] + gl;
              else if (after_s) g = a.inverse ? -gxs[e4] : gy1[e4];
              else g = a.inverse ? -gy1[e4] * ex : gy1[e4];  // (no scale net: ex = 1)
              g2[ml][e4] = col + e4 < H ? g * gs * rowmask : 0.f;
            }
            if (pass == 0) store4(gxrow + act_off, col, H, VEC, live, gx1);
          }
        }
        // the chain's first step: accd += W_out^T-blocks x [g tile 0 | g tile 1]
        {
          f16x8 bh[1], bl[1];
          float mx = 0.f;
          split_kstep(g2[0], g2[1], downd, bh[0], bl[0], mx);
          if (__builtin_expect(wave_any(!(mx < kSplitLimit)), 0)) {
            float fm = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) fm = __builtin_fmaxf(fm, __builtin_fmaxf(finite_abs(g2[0][e4]), finite_abs(g2[1][e4])));
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
            split_kstep(g2[0], g2[1], downd, bh[0], bl[0], unused);
          }
          mac_kstep<MT_MAX, 1>(bufT, 0, MTh, lane, bh, bl, accd.main, accd.corr);
        }
        // dW_out, db_out of the two tiles: cotangents (times the hidden vector's row scale) x last hidden vector
        if (gflat) {
          f32x4 cv[MT_MAX];
#pragma unroll
          for (int m = 0; m < MT_MAX; ++m) cv[m] = m < 2 ? g2[m < 2 ? m : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
          const float sc = exchange_store<MT_MAX>(cv, mo, exC, 0, 16 * wave, lane, ident);
          if (lane == 0) sC[wave] = sc;
          lds_barrier();
          dw_phase(exC, 0, mo, exH, ht_last, MTh, sC, sH + L * 8, nw, inv_gs, gflat + nd.w_off[L], gflat + nd.b_off[L], H, nd.sizes[L], m0, 0);
        }
      }
      // ---- hidden layers backwards, then grad_x of the conditioning half and dW_0 (mnf_rt_bwd.h)
      f32x4 dv[MT_MAX];
      chain_result<MT_MAX>(accd, wup / downd, lds.meta_bits[n_hid * 64 + lane], dv);
      auto load_in = [&](int mi) { return load4(xrow + cond_off, 16 * mi + 4 * q, H, VEC); };
      auto add_gx = [&](int mi, const f32x4& g) {
        const int col = 16 * mi + 4 * q;
        const f32x4 base = pass == 0 ? (a.grad_y ? load4(gyrow + cond_off, col, H, VEC) : f32x4{0.f, 0.f, 0.f, 0.f})
                                     : load4(gxrow + cond_off, col, H, VEC);
        store4(gxrow + cond_off, col, H, VEC, live, base + g);
      };
      backward_tail<MT_MAX>(src, a.flat, gflat, nd, n_hid, -1, dv, lds, wup, inv_gs, H, load_in, add_gx);
    }
  }
}

}  // namespace mnf

using namespace mnf;

extern "C" int mnf_affine_half_bwd_rt(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x,
                                      float* grad_flat, const float* flat, const float* grad_scale_dev, int64_t rows, int dim,
                                      int parity, int inverse, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                      void* stream) {
  if (!x || !grad_x || !flat || !grad_scale_dev || rows < 0 || dim < 2 || (dim & 1) || !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (n_hidden < 1 || n_hidden > rt::kMaxBwdLayers || (!has_scale && !has_shift) || (inverse && has_scale && !y) ||
      deterministic() || rows * dim >= (1ll << 40))
    return MNF_ERR_UNSUPPORTED;
  AhfBwdRtArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.gscale_dev = grad_scale_dev; a.rows = rows; a.dim = dim; a.parity = parity != 0; a.inverse = inverse != 0;
  a.has_scale = has_scale != 0; a.has_shift = has_shift != 0;
  const int H = dim / 2;
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
  sizes[n_hidden + 1] = H;
  if (mn < 4 || mxh > 64) return MNF_ERR_UNSUPPORTED;
  int64_t off = 0;
  if (has_scale) off += fill_net(a.s_net, n_hidden + 2, sizes, off);
  if (has_shift) off += fill_net(a.t_net, n_hidden + 2, sizes, off);
  if (!has_scale) a.s_net = a.t_net;
  if (!has_shift) a.t_net = a.s_net;
  if (off >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;
  a.n_params = (int)off;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec = dim % 8 == 0 && aligned(x) && aligned(grad_x) && (!y || aligned(y)) && (!grad_y || aligned(grad_y));
  constexpr int MT_MAX = 4;
  a.bt = 8;
  a.bias_words = 2 * a.bt * 16;
  a.ht_tiles = ht;
  a.dt_tiles = 0;  // (the deltas reuse the hidden vectors' tiles: mnf_rt_bwd.h backward_tail)
  (void)dt;
  const int KS1 = (16 * ((hidden[0] + 15) / 16) + 31) / 32;
  // The LDS plan.  Rows per workgroup first (16 per wave, any wave count: the per-row-block cost is what this kernel is
  // bound by), then the roomier of two weight-stream sizes (12 blocks per buffer: fewer chunks; 8: the largest chunk
  // there is -- two output tiles' forward blocks and the transposed blocks of their K-step at 64 hidden units), then as
  // many first-layer input tiles per chunk as fit (the output-layer chunks need two).
  int nw = 8;
  size_t lds = 0;
  bool fits = false;
  for (; nw >= 1; --nw) {
    const size_t tile_bytes = (size_t)2 * 16 * (16 * nw + rt::kExPad) * 2;
    for (int cb = 12; cb >= 8 && !fits; cb -= 4) {
      int ci = cb / KS1;
      ci = ci > MT_MAX ? MT_MAX : ci < 2 ? 2 : ci;
      for (int ct = ci; ct >= 2 && !fits; ct = ct > 2 ? 2 : 0) {
        lds = (size_t)4 * rt::kBwdHeadWords + (size_t)2 * cb * rt::kBlockWords * 4 + (size_t)a.bias_words * 4 +
              (size_t)(a.ht_tiles + a.dt_tiles + ct) * tile_bytes + (size_t)nw * (rt::kMaxBwdLayers + 1) * 64 * 4;
        if (lds <= 160 * 1024) {
          a.cb = cb;
          a.ct_tiles = ct;
          fits = true;
        }
      }
    }
    if (fits) break;
  }
  a.block_words = 2 * a.cb * rt::kBlockWords;
  if (nw < 1) return MNF_ERR_UNSUPPORTED;
  auto kernel = ahf_bwd_rt_kernel<MT_MAX>;
  static DeviceMemo attr;
  attr.get([&](int) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ahf_bwd_rt_kernel<MT_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    return 1;
  });
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nw * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const int64_t need = (rows + 16 * nw - 1) / (16 * nw);
  int64_t grid = (int64_t)per_cu * device_cus(current_device());
  if (grid > need) grid = need;
  tag_kernel("ahf_bwd_rt");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nw * 64), lds, (hipStream_t)stream, a);
  return check_launch();
}
