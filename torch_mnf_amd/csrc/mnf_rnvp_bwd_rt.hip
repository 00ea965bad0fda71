// Gradients of RNVP.forward (torch_mnf/flows/rnvp.py:25-39 under loss.backward(); what layers/mnf_linear.py:58-64, 84 and
// tests/test_mnf_mnist.py:14-56 train through) for ANY conditioner shape on the f16 matrix pipe: net = MLP(dim, h_1 .. h_n)
// with 1 .. 4 layers of widths 4 .. 128, any dim; run-time shapes (mnf_rt.h, mnf_rt_bwd.h), weights read from the plain
// `flat` parameter vector.  Takes the calls the per-shape gradient kernels (mnf_rnvp_bwd.hip: one hidden layer of at most
// 64 units) have no instantiation for.
//
// A workgroup owns a block of 16 NW rows, a wave one tile of it: forward recompute of y = net(mask z) keeping every
// hidden vector; then the heads two 16-dim output tiles at a time -- shift, scale and gate of the tiles, the cotangents
//   g_t = G (1 - gate)      g_s = (G ((1 - m) z - t) gate + g_ld (1 - m)) (1 - gate)      grad_z = G ((1 - m) gate + m)
// (G = grad_x), the first step of the delta chain  W_t^T g_t + W_s^T g_s  and the tiles' dW_t, dW_s products through the
// LDS exchange area --, then the hidden layers backwards and the first layer input tile by input tile
// (grad_z += mask * W_0^T delta_1, dW_0 += delta_1 (x) (mask z)): mnf_rt_bwd.h backward_tail.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_rt_bwd.h"

namespace mnf {

struct RnvpBwdRtArgs {
  const float* z;
  const float* mask;  // nullptr: the in-kernel mask of `seed`
  const float* grad_x;
  const float* grad_ld;
  float* grad_z;
  float* grad_flat;
  const float* flat;
  const float* gscale_dev;
  int64_t rows;
  uint64_t seed;
  int dim, n_params, vec;
  int t_w, t_b, s_w, s_b;
  int cb, bt, block_words, bias_words;
  int ht_tiles, dt_tiles, ct_tiles;
  NetDesc net;
};

__device__ __forceinline__ f32x4 bwd_mask_bits4(uint32_t word, int first_bit) {
  f32x4 m;
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = (float)((word >> (first_bit + r)) & 1u);
  return m;
}

template <int MT_MAX>
__global__ void __launch_bounds__(512) rnvp_bwd_rt_kernel(RnvpBwdRtArgs a) {
  using namespace rt;
  const bool VEC = a.vec != 0;  // (uniform) rows are 16-byte aligned: dwordx4 row accesses
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
  const int d = a.dim;
  const bool seeded = a.mask == nullptr;
  const NetDesc& nd = a.net;
  const int n_hid = nd.n_lin;  // every layer of `net` ends in a hidden vector; the last one has no activation
  const int hl = nd.sizes[n_hid], MTh = tiles16(hl), KSh = steps32(16 * MTh), M = tiles16(d);
  const int ht_last = exH_tile_of(nd, n_hid);
  const int64_t n_blocks = (a.rows + 16 * nw - 1) / (16 * nw);
  float* gflat = a.grad_flat;

  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t r = blk * (16 * nw) + 16 * wave + j;
    const bool live = r < a.rows;
    const int64_t rc = live ? r : a.rows - 1;
    const float* zrow = a.z + rc * d;
    const float* mrow = a.mask ? a.mask + rc * d : zrow;
    const float* gxrow = a.grad_x ? a.grad_x + rc * d : zrow;
    float* gzrow = a.grad_z + rc * d;
    const float gl = a.grad_ld && live ? a.grad_ld[rc] : 0.f;
    const float rowmask = live ? 1.f : 0.f;
    auto mask4 = [&](int col) {  // the mask of columns col .. col + 3 (col a multiple of 4)
      return seeded ? bwd_mask_bits4(rnvp_mask_word(a.seed, rc, col >> 5), col & 31) : load4(mrow, col, d, VEC);
    };
    // ---- forward recompute of y = net(mask z), every hidden vector kept
    Hidden<MT_MAX, 1> h;
    {
      auto load_x = [&](int, int ks, f32x4& xa, f32x4& xb) {
        const int c0 = 32 * ks + 4 * q;
        xa = load4(zrow, c0, d, VEC) * mask4(c0);
        xb = load4(zrow, c0 + 16, d, VEC) * mask4(c0 + 16);
      };
      forward_keep<MT_MAX>(src, a.flat, nd, n_hid, n_hid - 1, wup, lds, load_x, h);
    }
    // ---- the heads, two 16-dim tiles per round: [shift | scale blocks] -> cotangents -> [W_t^T | W_s^T blocks] -> chain
    Acc<MT_MAX, 1> accd;
    accd.zero();
    float downd = 1.f;
    for (int m0 = 0; m0 < M; m0 += 2) {
      const int mo = M - m0 < 2 ? M - m0 : 2;
      uint32_t* buf = src.cur_blocks();
      float* bbuf = src.cur_bias();
      stage_blocks(buf, mo * 2 * KSh, DenseMMajor{a.flat + a.t_w, hl, d, KSh, m0, 2, (int64_t)a.s_w - a.t_w}, src.wdown);
      stage_bias(bbuf, mo * 2, DenseBiasHeads{a.flat + a.t_b, d, m0, 2, (int64_t)a.s_b - a.t_b});
      src.commit();
      f32x4 gt2[2], gs2[2];
#pragma unroll
      for (int ml = 0; ml < 2; ++ml) {
        gt2[ml] = f32x4{0.f, 0.f, 0.f, 0.f};
        gs2[ml] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ml < mo) {
          const int col = 16 * (m0 + ml) + 4 * q;
          f32x4 t4[1], s4[1];
          out_tile<MT_MAX, 1>(buf, (ml * 2) * KSh, KSh, bbuf + (ml * 2) * 16, lane, q, h, wup, t4);
          out_tile<MT_MAX, 1>(buf, (ml * 2 + 1) * KSh, KSh, bbuf + (ml * 2 + 1) * 16, lane, q, h, wup, s4);
          const f32x4 zz = load4(zrow, col, d, VEC), mm = mask4(col);
          const f32x4 G = a.grad_x ? load4(gxrow, col, d, VEC) : f32x4{0.f, 0.f, 0.f, 0.f};
          f32x4 gz;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gate = __builtin_amdgcn_rcpf(1.f + exp6r(-s4[0][e]));
            const float a1 = (1.f - mm[e]) * zz[e];
            gz[e] = G[e] * ((1.f - mm[e]) * gate + mm[e]);
            const bool real = col + e < d;
            gt2[ml][e] = real ? G[e] * (1.f - gate) * gs * rowmask : 0.f;
            gs2[ml][e] = real ? (G[e] * (a1 - t4[0][e]) * gate + gl * (1.f - mm[e])) * (1.f - gate) * gs * rowmask : 0.f;
          }
          store4(gzrow, col, d, VEC, live, gz);
        }
      }
      // dW_t, dW_s, db_t, db_s of the two tiles: cotangent tiles [t0 t1 s0 s1] x last hidden vector
      if (gflat) {
        f32x4 cv[MT_MAX];
#pragma unroll
        for (int m = 0; m < MT_MAX; ++m) cv[m] = m < 2 ? gt2[m & 1] : m < 4 ? gs2[m & 1] : f32x4{0.f, 0.f, 0.f, 0.f};
        const float sc = exchange_store<MT_MAX>(cv, 4, lds.exC, 0, 16 * wave, lane, lds.ident);
        if (lane == 0) lds.sC[wave] = sc;
      }
      // the chain's first step: accd += W_t^T-blocks x [g_t tiles] + W_s^T-blocks x [g_s tiles]
      uint32_t* bufT = src.cur_blocks();
      stage_blocks(bufT, MTh, DenseTKMajor{a.flat + a.t_w, hl, d, MTh, m0 >> 1}, src.wdown);
      stage_blocks(bufT + MTh * kBlockWords, MTh, DenseTKMajor{a.flat + a.s_w, hl, d, MTh, m0 >> 1}, src.wdown);
      src.commit();  // (also: the cotangent tiles are in the exchange area)
#pragma unroll
      for (int head = 0; head < 2; ++head) {
        const f32x4& g0 = head == 0 ? gt2[0] : gs2[0];
        const f32x4& g1 = head == 0 ? gt2[1] : gs2[1];
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
        mac_kstep<MT_MAX, 1>(bufT, head * MTh, MTh, lane, bh, bl, accd.main, accd.corr);
      }
      if (gflat) {
        dw_phase(lds.exC, 0, mo, lds.exH, ht_last, MTh, lds.sC, lds.sH + n_hid * 8, nw, inv_gs, gflat + a.t_w, gflat + a.t_b, d, hl, m0, 0);
        dw_phase(lds.exC, 2, mo, lds.exH, ht_last, MTh, lds.sC, lds.sH + n_hid * 8, nw, inv_gs, gflat + a.s_w, gflat + a.s_b, d, hl, m0, 0);
      }
    }
    // ---- hidden layers backwards (y has no activation: derivative 1), then the first layer: grad_z += mask * W_0^T delta_1
    f32x4 dv[MT_MAX];
    chain_result<MT_MAX>(accd, wup / downd, 0xffffffffu, dv);
    auto load_in = [&](int mi) { return load4(zrow, 16 * mi + 4 * q, d, VEC) * mask4(16 * mi + 4 * q); };
    auto add_in = [&](int mi, const f32x4& g) {
      const int col = 16 * mi + 4 * q;
      const f32x4 base = load4(gzrow, col, d, VEC);
      store4(gzrow, col, d, VEC, live, base + g * mask4(col));
    };
    backward_tail<MT_MAX>(src, a.flat, gflat, nd, n_hid, n_hid, dv, lds, wup, inv_gs, d, load_in, add_in);
  }
}

template <int MT_MAX>
static int rnvp_bwd_rt_launch_class(RnvpBwdRtArgs& a, int max_nw, hipStream_t stream) {
  // rows per workgroup first (any wave count), then as many first-layer input tiles per chunk as still fit (a.ct_tiles on
  // entry; the output-layer chunks need four: [t0 t1 s0 s1])
  int nw = max_nw;
  size_t lds = 0;
  bool fits = false;
  const int ct_wish = a.ct_tiles;
  for (; nw >= 1; --nw) {
    for (int ct = ct_wish; ct >= 4 && !fits; ct = ct > 4 ? 4 : 0) {
      lds = (size_t)4 * rt::kBwdHeadWords + (size_t)a.block_words * 4 + (size_t)a.bias_words * 4 +
            rt::bwd_lds_bytes(nw, a.ht_tiles, a.dt_tiles, ct);
      if (lds <= 160 * 1024) {
        a.ct_tiles = ct;
        fits = true;
      }
    }
    if (fits) break;
  }
  if (nw < 1) return MNF_ERR_UNSUPPORTED;
  auto kernel = rnvp_bwd_rt_kernel<MT_MAX>;
  static DeviceMemo attr;
  attr.get([&](int) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnvp_bwd_rt_kernel<MT_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    return 1;
  });
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nw * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const int64_t need = (a.rows + 16 * nw - 1) / (16 * nw);
  int64_t grid = (int64_t)per_cu * device_cus(current_device());
  if (grid > need) grid = need;
  tag_kernel("rnvp_bwd_rt");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nw * 64), lds, stream, a);
  return check_launch();
}

}  // namespace mnf

using namespace mnf;

extern "C" int mnf_rnvp_bwd_rt(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                               float* grad_z, float* grad_flat, const float* flat, const float* grad_scale_dev, int64_t rows,
                               int dim, int n_hidden, const int* hidden, void* stream) {
  if (!z || !grad_z || !flat || !grad_scale_dev || rows < 0 || dim < 1 || n_hidden < 1 || !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (n_hidden > rt::kMaxBwdLayers || deterministic() || rows * dim >= (1ll << 40)) return MNF_ERR_UNSUPPORTED;
  RnvpBwdRtArgs a;
  memset(&a, 0, sizeof(a));
  a.z = z; a.mask = mask; a.seed = seed; a.grad_x = grad_x; a.grad_ld = grad_ld; a.grad_z = grad_z; a.grad_flat = grad_flat;
  a.flat = flat; a.gscale_dev = grad_scale_dev; a.rows = rows; a.dim = dim;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = dim;
  int mn = 1 << 30, mxh = 0, ht = 0, dt = 0;
  for (int i = 0; i < n_hidden; ++i) {
    sizes[1 + i] = hidden[i];
    mn = hidden[i] < mn ? hidden[i] : mn;
    mxh = hidden[i] > mxh ? hidden[i] : mxh;
    ht += (hidden[i] + 15) / 16;
    dt = (hidden[i] + 15) / 16 > dt ? (hidden[i] + 15) / 16 : dt;
  }
  if (mn < 4 || mxh > 128) return MNF_ERR_UNSUPPORTED;
  int64_t off = fill_net(a.net, n_hidden + 1, sizes, 0);
  const int hl = hidden[n_hidden - 1];
  a.t_w = (int)off; off += (int64_t)hl * dim;
  a.t_b = (int)off; off += dim;
  a.s_w = (int)off; off += (int64_t)hl * dim;
  a.s_b = (int)off; off += dim;
  if (off >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;
  a.n_params = (int)off;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec = dim % 4 == 0 && aligned(z) && aligned(grad_z) && (!mask || aligned(mask)) && (!grad_x || aligned(grad_x));
  const int MT_MAX = mxh <= 64 ? 4 : 8;
  const int MTh = (hl + 15) / 16, KSh = (16 * MTh + 31) / 32;
  a.cb = 4 * KSh > 2 * MTh ? 4 * KSh : 2 * MTh;  // a round of the heads: 2 tiles x 2 heads x KSh blocks, then 2 x MTh turned ones
  const int MT1 = (hidden[0] + 15) / 16;
  if (a.cb < MT1) a.cb = MT1;  // (a K-step of the first layer)
  if (a.cb < 8) a.cb = 8;
  a.bt = MT_MAX > 4 ? MT_MAX : 4;
  a.block_words = 2 * a.cb * rt::kBlockWords;
  a.bias_words = 2 * a.bt * 16;
  a.ht_tiles = ht;
  a.dt_tiles = 0;  // (the deltas reuse the hidden vectors' tiles: mnf_rt_bwd.h backward_tail)
  (void)dt;
  const int KS1 = (16 * MT1 + 31) / 32;
  int ci = a.cb / KS1;
  ci = ci > MT_MAX ? MT_MAX : ci;
  a.ct_tiles = ci < 4 ? 4 : ci;  // (the wish: rnvp_bwd_rt_launch_class settles for four where that buys a larger workgroup)
  if (MT_MAX == 4) return rnvp_bwd_rt_launch_class<4>(a, 8, (hipStream_t)stream);
  return rnvp_bwd_rt_launch_class<8>(a, 8, (hipStream_t)stream);
}
