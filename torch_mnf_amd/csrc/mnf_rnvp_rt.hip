// RNVP.forward (torch_mnf/flows/rnvp.py:25-39) for ANY conditioner shape on the f16 matrix pipe: net = MLP(dim, h_1 .. h_n)
// with any number of layers of widths 4 .. 256, t and s = Linear(h_n, dim), any dim; run-time shapes (mnf_rt.h), weights
// read from the plain `flat` parameter vector.  Takes the calls the per-shape kernels (mnf_rnvp_resident.hip,
// mnf_rnvp_mfma.hip: one hidden layer of at most 64 units) have no instantiation for; the VALU kernel of mnf_generic.hip
// keeps few rows and hidden layers narrower than 4 units.
//
// A wave owns one 16-row tile: y = net(mask z) with the row streamed from memory K-step by K-step (the mask read, or
// regenerated from the counter-based hash of mnf_device.h: one 32-bit word per row and K-step), then the two heads are
// walked 16 output dims at a time: shift and scale of the tile, the gate, x, the row's log|det J| in registers.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_rt.h"

namespace mnf {

struct RnvpRtArgs {
  const float* z;
  const float* mask;  // nullptr: the in-kernel mask of `seed`
  float* x;
  float* log_det;
  const float* flat;
  int64_t rows;
  uint64_t seed;
  int dim, accumulate;
  int n_params, vec;  // vec: rows are 16-byte aligned (dwordx4 row accesses)
  int t_w, t_b, s_w, s_b;  // float offsets of the heads
  int cb, bt;
  int block_words, bias_words;
  NetDesc net;
};

__device__ __forceinline__ f32x4 mask_bits4(uint32_t word, int first_bit) {
  f32x4 m;
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = (float)((word >> (first_bit + r)) & 1u);
  return m;
}

template <int MT_MAX, int VECM, bool PREFILL, typename Src>  // VECM: 0 / 1, or 2 = a.vec
__device__ __forceinline__ void rnvp_rt_block(const RnvpRtArgs& a, Src& src, float wup, int64_t row0) {
  using namespace rt;
  const bool VEC = VECM == 2 ? a.vec != 0 : VECM == 1;
  constexpr int NTL = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const int d = a.dim;
  const int64_t r = row0 + (int64_t)wave * 16 + j;
  const bool live = !PREFILL && r < a.rows;
  const int64_t rc = r < a.rows ? r : a.rows - 1;
  const float* zrow = a.z + rc * d;
  const float* mrow = a.mask ? a.mask + rc * d : a.z + rc * d;  // (never read when the mask is generated)
  float* xrow = a.x + rc * d;
  const bool seeded = a.mask == nullptr;

  Hidden<MT_MAX, NTL> h;
  auto load_x = [&](int, int ks, f32x4& xa, f32x4& xb) {
    const int c0 = 32 * ks + 4 * q;
    const f32x4 za = load4(zrow, c0, d, VEC), zb = load4(zrow, c0 + 16, d, VEC);
    f32x4 ma, mb;
    if (seeded) {  // (uniform)
      const uint32_t word = rnvp_mask_word(a.seed, rc, ks);
      ma = mask_bits4(word, 4 * q);
      mb = mask_bits4(word, 16 + 4 * q);
    } else {
      ma = load4(mrow, c0, d, VEC);
      mb = load4(mrow, c0 + 16, d, VEC);
    }
    xa = za * ma;  // z2 = mask * z (:30)
    xb = zb * mb;
  };
  auto use_x = [&](int, int, const f32x4&, const f32x4&) {};
  const int n_hid = a.net.n_lin;  // every layer of `net` ends in a hidden vector; the last one has no activation (mlp.py:12)
  net_to_hidden<MT_MAX, NTL, PREFILL>(src, a.flat, a.net, n_hid, n_hid - 1, wup, lane, q, load_x, use_x, h);

  // ---- the heads, 16 output dims at a time: blocks [tile][t | s][K-step]
  const int hl = a.net.sizes[n_hid];
  const int KS = steps32(16 * tiles16(hl)), M = tiles16(d);
  int MO = Src::resident ? M : src.cb / (2 * KS);
  if (!Src::resident && MO > src.bt / 2) MO = src.bt / 2;
  if (MO < 1) MO = 1;
  const float* W0 = a.flat + a.t_w;
  const float* B0 = a.flat + a.t_b;
  const int64_t w_stride = (int64_t)a.s_w - a.t_w, b_stride = (int64_t)a.s_b - a.t_b;
  float ld = 0.f;
  f32x4 nz, nm;  // the next tile's columns (requested one tile ahead)
  if (!PREFILL) {
    nz = load4(zrow, 4 * q, d, VEC);
    nm = seeded ? f32x4{0.f, 0.f, 0.f, 0.f} : load4(mrow, 4 * q, d, VEC);
  }
  for (int m0 = 0; m0 < M; m0 += MO) {
    const int mo = M - m0 < MO ? M - m0 : MO;
    const Chunk c = src.template chunk<PREFILL>(mo * 2 * KS, DenseMMajor{W0, hl, d, KS, m0, 2, w_stride}, mo * 2,
                                                DenseBiasHeads{B0, d, m0, 2, b_stride});
    if (PREFILL) continue;
    for (int ml = 0; ml < mo; ++ml) {
      const int m = m0 + ml, col = 16 * m + 4 * q;
      const f32x4 zz = nz;
      f32x4 mm = nm;
      const int m_next = m + 1 < M ? m + 1 : M - 1;
      nz = load4(zrow, 16 * m_next + 4 * q, d, VEC);
      if (seeded) mm = mask_bits4(rnvp_mask_word(a.seed, rc, m >> 1), 16 * (m & 1) + 4 * q);
      else nm = load4(mrow, 16 * m_next + 4 * q, d, VEC);
      f32x4 t4[NTL], s4[NTL];
      out_tile<MT_MAX, NTL>(c.A, (ml * 2) * KS, KS, c.bias + (ml * 2) * 16, lane, q, h, wup, t4);
      out_tile<MT_MAX, NTL>(c.A, (ml * 2 + 1) * KS, KS, c.bias + (ml * 2 + 1) * 16, lane, q, h, wup, s4);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gate = __builtin_amdgcn_rcpf(1.f + exp6r(-s4[0][e]));  // sigmoid (:34)
        const float keep = mm[e] * zz[e];                                  // z2 = m z
        const float gated = (1.f - mm[e]) * zz[e];                         // z1 = (1 - m) z
        o[e] = (gated * gate + (1.f - gate) * t4[0][e]) + keep;            // (:37)
        const float lg = (1.f - mm[e]) * (__builtin_amdgcn_logf(gate) * 0.693147180559945309f);  // (:36)
        ld += col + e < d ? lg : 0.f;
      }
      store4(xrow, col, d, VEC, live, o);
    }
  }
  if (PREFILL) return;
  const float total = sum_over_q(ld);
  if (q == 0 && live && a.log_det) a.log_det[r] = a.accumulate ? a.log_det[r] + total : total;
}

template <int MT_MAX, int NW, bool RESIDENT, int VEC>
__global__ void __launch_bounds__(NW * 64) rnvp_rt_kernel(RnvpRtArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t rt_lds[];
  float* scratch = reinterpret_cast<float*>(rt_lds);
  uint32_t* blocks = rt_lds + 16;
  float* bias = reinterpret_cast<float*>(blocks + a.block_words);
  const float wmax = rt::block_weight_max(a.flat, a.n_params, scratch);
  const int e = rt::weight_exponent(wmax);
  const float wup = rt::pow2f(e);
  rt::Source<RESIDENT> src{blocks, bias, a.cb, a.bt, 0, 0, 0, rt::pow2f(-e), 0};
  if (RESIDENT) {
    rnvp_rt_block<MT_MAX, VEC, true>(a, src, wup, 0);
    __syncthreads();
  }
  const int64_t rows_per_block = (int64_t)(blockDim.x >> 6) * 16;
  const int64_t n_blocks = (a.rows + rows_per_block - 1) / rows_per_block;
  for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    src.slot = 0;
    src.btile = 0;
    rnvp_rt_block<MT_MAX, VEC, false>(a, src, wup, b * rows_per_block);
  }
}

template <typename K>
static void rnvp_rt_allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int MT_MAX, int NW>
static int rnvp_rt_launch_class(RnvpRtArgs& a, int64_t n_blocks, int64_t n_bias, int cb_stream, hipStream_t stream) {
  constexpr int kResidentBytes = 150 * 1024;
  // Rows that are not 16-byte aligned (dim not a multiple of 4, a view at an odd offset) have the resident variant only,
  // except in the widest class, whose streaming kernel takes the alignment at run time (a branch around every row access:
  // 15-20 % on the memory-bound shapes) and serves every width: rnvp_rt_launch sends such a call there.
  const bool resident = n_blocks * 2048 + n_bias * 64 <= kResidentBytes;
  constexpr int kStreamVec = MT_MAX == 16 ? 2 : 1;
  if (!resident && !a.vec && kStreamVec != 2) return MNF_ERR_UNSUPPORTED;
  if (resident) {
    a.cb = (int)n_blocks;
    a.bt = (int)n_bias;
    a.block_words = (int)n_blocks * rt::kBlockWords;
    a.bias_words = (int)n_bias * 16;
  } else {
    a.cb = cb_stream;
    a.bt = cb_stream;
    a.block_words = 2 * cb_stream * rt::kBlockWords;
    a.bias_words = 2 * cb_stream * 16;
  }
  const size_t lds = 64 + (size_t)a.block_words * 4 + (size_t)a.bias_words * 4;
  static DeviceMemo attr;
  attr.get([&](int) {
    rnvp_rt_allow_big_lds(rnvp_rt_kernel<MT_MAX, NW, true, 1>);
    rnvp_rt_allow_big_lds(rnvp_rt_kernel<MT_MAX, NW, false, kStreamVec>);
    rnvp_rt_allow_big_lds(rnvp_rt_kernel<MT_MAX, NW, true, 0>);
    return 1;
  });
  auto kernel = !resident ? rnvp_rt_kernel<MT_MAX, NW, false, kStreamVec>
                          : a.vec ? rnvp_rt_kernel<MT_MAX, NW, true, 1> : rnvp_rt_kernel<MT_MAX, NW, true, 0>;
  const int nw = NW == 8 && resident && lds <= 79 * 1024 ? 4 : NW;  // (streaming: every wave of the CU shares one conversion of the weights)
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nw * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const int64_t rows_per_block = (int64_t)nw * 16;
  const int64_t need = (a.rows + rows_per_block - 1) / rows_per_block;
  int64_t grid = (int64_t)per_cu * device_cus(current_device());
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nw * 64), lds, stream, a);
  return check_launch();
}

// MNF_ERR_UNSUPPORTED: the shape is outside the run-time-shaped kernel too (the caller runs the VALU kernel)
int rnvp_rt_launch(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                   const float* flat, int64_t rows, int dim, int n_hidden, const int* hidden, hipStream_t stream) {
  if (!flat || n_hidden < 1 || n_hidden > MNF_MAX_LINEAR || rows * dim >= (1ll << 40)) return MNF_ERR_UNSUPPORTED;
  RnvpRtArgs a;
  memset(&a, 0, sizeof(a));
  a.z = z; a.mask = mask; a.seed = seed; a.x = x; a.log_det = log_det; a.flat = flat; a.rows = rows; a.dim = dim;
  a.accumulate = accumulate != 0;
  int sizes[MNF_MAX_LINEAR + 1];
  sizes[0] = dim;
  int mn = 1 << 30, mxh = 0;
  for (int i = 0; i < n_hidden; ++i) {
    sizes[1 + i] = hidden[i];
    mn = hidden[i] < mn ? hidden[i] : mn;
    mxh = hidden[i] > mxh ? hidden[i] : mxh;
  }
  if (mn < 4 || mxh > 256) return MNF_ERR_UNSUPPORTED;
  int64_t off = fill_net(a.net, n_hidden + 1, sizes, 0);
  const int hl = hidden[n_hidden - 1];
  a.t_w = (int)off; off += (int64_t)hl * dim;
  a.t_b = (int)off; off += dim;
  a.s_w = (int)off; off += (int64_t)hl * dim;
  a.s_b = (int)off; off += dim;
  if (off >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;
  a.n_params = (int)off;
  a.vec = dim % 4 == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (!mask || (reinterpret_cast<uintptr_t>(mask) & 15) == 0);
  int64_t n_blocks = 0, n_bias = 0;
  for (int l = 0; l < n_hidden; ++l) {
    const int in_cols = l == 0 ? dim : 16 * ((sizes[l] + 15) / 16);
    n_blocks += (int64_t)((in_cols + 31) / 32) * ((sizes[l + 1] + 15) / 16);
    n_bias += (sizes[l + 1] + 15) / 16;
  }
  const int KS = (16 * ((hl + 15) / 16) + 31) / 32, M = (dim + 15) / 16;
  n_blocks += 2ll * KS * M;
  n_bias += 2ll * M;
  tag_kernel("rnvp_rt");
  int rc = MNF_ERR_UNSUPPORTED;
  if (mxh <= 64) rc = rnvp_rt_launch_class<4, 8>(a, n_blocks, n_bias, 16, stream);
  else if (mxh <= 128) rc = rnvp_rt_launch_class<8, 8>(a, n_blocks, n_bias, 16, stream);
  if (rc == MNF_ERR_UNSUPPORTED) rc = rnvp_rt_launch_class<16, 4>(a, n_blocks, n_bias, 16, stream);
  return rc;
}

}  // namespace mnf
