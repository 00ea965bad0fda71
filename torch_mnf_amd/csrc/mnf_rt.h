// Run-time-shaped conditioner on the f16 matrix pipe: the any-shape path of the three coupling layers.
//
// The specialised kernels (mnf_ahf_split.hip, mnf_nsf_mfma.hip, mnf_rnvp_*.hip) are template instantiations per shape
// with packed operand images; everything else used to run VALU dot-product kernels 10-100 x slower.  The kernels built
// on this header take the layer's plain fp32 parameter vector (`flat`, state_dict order) and ANY layer count / widths:
//
//   * a workgroup converts weights fp32 -> split f16 (hi + scaled lo, mnf_split.h) into LDS itself, as 16 x 32 A-operand
//     blocks in the lane order of v_mfma_f32_16x16x32_f16 (2 KB per block): once per workgroup when the whole conditioner
//     fits (resident mode), else chunk by chunk through two LDS buffers with one barrier per chunk (streaming mode) --
//     no operand image, no index table, no repack after a weight update;
//   * a wave owns NTL 16-row tiles; a hidden vector lives in registers as split accumulator tiles (tile m, lane (j, q),
//     register r <-> unit 16 m + 4 q + r of row j), which ARE the next layer's B operands (K-step ks = tiles 2ks, 2ks+1);
//     loops over tiles are unrolled to the class bound MT_MAX with wave-uniform guards (m < MT), so widths are run-time;
//   * the first layer streams its input from memory K-step by K-step (any width), the last one streams its output tiles
//     to the layer's epilogue (any width);
//   * range: instead of recomputing out-of-range tiles on an fp32 path, a row whose operands reach the split limit is
//     scaled by a power of two before the split and its products scaled back (exact), and weights beyond the limit are
//     staged scaled down by a power of two -- results do not depend on the input or weight range.  Rows holding
//     non-finite values give NaN (the reference: inf or NaN).
//
// Three size classes are instantiated per layer type (MT_MAX = 4 / 8 / 16 hidden tiles: widths <= 64 / 128 / 256).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mnf_device.h"
#include "mnf_split.h"

namespace mnf {
namespace rt {

constexpr int kBlockWords = 512;  // one A block: [hi | lo][lane][4 words]
constexpr int kBiasTileWords = 16;
constexpr int kRing = 1;          // K-steps / output tiles a wave's row reads run ahead of their use

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also drains the wave's outstanding global
// stores and atomics (its release fence is s_waitcnt vmcnt(0)): a wave would sit out the ~1-3 k cycles of its last row
// stores or gradient atomics at every chunk boundary.  Nothing in these kernels hands data to another wave through
// global memory, so the LDS counter is all that has to reach zero.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------------
// weight range: max |w| over the finite entries of the parameter vector, by every workgroup for itself (the vector is
// L2-resident; a pass costs microseconds), and the power of two that brings it to <= kSplitWeightLimit
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float finite_abs(float v) {
  const float a = __builtin_fabsf(v);
  return a <= 3.0e38f ? a : 0.f;  // inf, NaN -> 0
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// scratch: >= 16 floats of LDS; ends with a barrier; every thread returns the same value
__device__ __forceinline__ float block_weight_max(const float* __restrict__ flat, int n, float* scratch) {
  float mx = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mx = __builtin_fmaxf(mx, finite_abs(flat[i]));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = __builtin_fmaxf(r, scratch[w]);
  __syncthreads();
  return r;
}

// 2^-e with the smallest e >= 0 such that v 2^-e < limit (limit a power of two, log2 = lg)
__device__ __forceinline__ int down_exponent(float v, int lg) {
  const int ex = (int)((__builtin_bit_cast(uint32_t, v) >> 23) & 0xffu) - 127;  // floor(log2 v) for normal v
  const int e = ex - (lg - 1);
  return e > 0 ? e : 0;
}
// the signed e that brings v 2^-e into [2^14, 2^15) (0 for v = 0): the staged weights use the top of f16's range, so that
// small weights stay normal f16 numbers as far down as possible
__device__ __forceinline__ int weight_exponent(float v) {
  if (!(v > 0.f)) return 0;
  int e = (int)((__builtin_bit_cast(uint32_t, v) >> 23) & 0xffu) - 127 - 14;
  return e < -100 ? -100 : e > 100 ? 100 : e;
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (uint32_t)(127 + e) << 23); }

// ---------------------------------------------------------------------------------------------------------------------
// staging: fp32 weights -> split-f16 A blocks in LDS
// ---------------------------------------------------------------------------------------------------------------------
// Block b of a chunk: A operand of one (output tile, K-step).  Lane (i, q) of the block holds, in slot e of its 8 halves,
// the weight of output row i against K index k = (e < 4 ? 4 q + e : 16 + 4 q + (e - 4)) of the step -- the order in
// which two accumulator tiles form a B operand (pair_operand, mnf_split.h).  A wave converts whole blocks (block index =
// a mixed-radix number with digits (d0, d1, d2), d0 fastest, radices fetch.R0, fetch.R1: kept as counters, no division);
// fetch.load(d0, d1, d2, i, q, lo4, hi4) returns the lane's eight weights (0 for padding); every weight is multiplied by
// `wdown` (a power of two) first.
__device__ __forceinline__ void convert_block(uint32_t* dst, int lane, f32x4 va, f32x4 vb, float wdown) {
  va *= wdown;
  vb *= wdown;
  uint32_t hi[4], lo[4];
  float unused = 0.f;
  split_pair(va[0], va[1], hi[0], lo[0], unused);
  split_pair(va[2], va[3], hi[1], lo[1], unused);
  split_pair(vb[0], vb[1], hi[2], lo[2], unused);
  split_pair(vb[2], vb[3], hi[3], lo[3], unused);
  u32x4v* d = reinterpret_cast<u32x4v*>(dst);
  d[lane] = u32x4v{hi[0], hi[1], hi[2], hi[3]};
  d[64 + lane] = u32x4v{lo[0], lo[1], lo[2], lo[3]};
}

template <typename Fetch>
__device__ __forceinline__ void stage_blocks(uint32_t* dst, int n_blocks, const Fetch& fetch, float wdown) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, i = lane & 15, q = lane >> 4;
  // a wave converts a contiguous range of blocks: the digits are set once and then counted up; two blocks per trip, both
  // blocks' loads ahead of the arithmetic (a block's eight weights are L2 reads: their latency is the cost of this loop)
  const int per = (n_blocks + nw - 1) / nw, b0 = wave * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
  int d0 = b0 % fetch.R0, rest = b0 / fetch.R0, d1 = rest % fetch.R1, d2 = rest / fetch.R1;
  auto advance = [&](int& e0, int& e1, int& e2) {
    if (++e0 == fetch.R0) {
      e0 = 0;
      if (++e1 == fetch.R1) {
        e1 = 0;
        ++e2;
      }
    }
  };
  for (int b = b0; b < b1; b += 2) {
    const bool two = b + 1 < b1;  // (uniform)
    int e0 = d0, e1 = d1, e2 = d2;
    if (two) advance(e0, e1, e2);
    f32x4 va, vb, wa, wb;
    fetch.load(d0, d1, d2, i, q, va, vb);
    fetch.load(e0, e1, e2, i, q, wa, wb);  // (the last odd block: the same block again, not stored)
    convert_block(dst + b * kBlockWords, lane, va, vb, wdown);
    if (two) convert_block(dst + (b + 1) * kBlockWords, lane, wa, wb, wdown);
    d0 = e0;
    d1 = e1;
    d2 = e2;
    advance(d0, d1, d2);
  }
}

// bias tiles: tile t, entry u (0..15) = bias(t, u) -- fp32, NOT scaled (added after the products are scaled back)
template <typename Bias>
__device__ __forceinline__ void stage_bias(float* dst, int n_tiles, const Bias& bias) {
  for (int u = threadIdx.x; u < n_tiles * 16; u += blockDim.x) dst[u] = bias(u >> 4, u & 15);
}

// the eight weights of lane (i, q) out of row `row` (n_cols wide, valid when row_ok) of a row-major matrix: columns
// c0 + 4 q + e and c0 + 16 + 4 q + e, zeros beyond n_cols.  aligned: every row starts 16-byte aligned and n_cols % 4 == 0
// (two dwordx4 per lane), else element by element.  No load sits under a divergent branch.
__device__ __forceinline__ void load_row8(const float* __restrict__ row, bool row_ok, int c0, int n_cols, bool aligned, int q,
                                          f32x4& va, f32x4& vb) {
  const int ca = c0 + 4 * q, cb = ca + 16;
  if (aligned) {  // (uniform)
    const bool oka = row_ok && ca < n_cols, okb = row_ok && cb < n_cols;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + (oka ? ca : 0));
    const f32x4 b = *reinterpret_cast<const f32x4*>(row + (okb ? cb : 0));
    va = oka ? a : f32x4{0.f, 0.f, 0.f, 0.f};
    vb = okb ? b : f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool oka = row_ok && ca + e < n_cols, okb = row_ok && cb + e < n_cols;
    const float xa = row[oka ? ca + e : 0], xb = row[okb ? cb + e : 0];
    va[e] = oka ? xa : 0.f;
    vb[e] = okb ? xb : 0.f;
  }
}

// A dense Linear W (n_out x n_in, row-major) walked [K-step][tile] from K-step ks0 (K-major stages: the first layer and
// the hidden layers): digits (tile, K-step - ks0)
struct DenseKMajor {
  const float* W;
  int n_in, n_out, R0, ks0;  // R0 = MT
  static constexpr int R1 = 1 << 30;
  __device__ __forceinline__ void load(int m, int ksl, int, int i, int q, f32x4& va, f32x4& vb) const {
    const int o = 16 * m + i, c0 = 32 * (ks0 + ksl);
    const bool aligned = (n_in & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    load_row8(W + (int64_t)(o < n_out ? o : 0) * n_in, o < n_out, c0, n_in, aligned, q, va, vb);
  }
};
// the same matrix walked [tile][head][K-step] from output tile m0 (M-major stages: the output layer), `heads` matrices of
// one shape interleaved per tile, head h at W + h * head_stride: digits (K-step, head, tile - m0)
struct DenseMMajor {
  const float* W;
  int n_in, n_out, R0, m0, R1;  // R0 = KS, R1 = heads
  int64_t head_stride;
  __device__ __forceinline__ void load(int ks, int head, int ml, int i, int q, f32x4& va, f32x4& vb) const {
    const int o = 16 * (m0 + ml) + i, c0 = 32 * ks;
    const float* Wh = W + head * head_stride;
    const bool aligned = (n_in & 3) == 0 && (reinterpret_cast<uintptr_t>(Wh) & 15) == 0;
    load_row8(Wh + (int64_t)(o < n_out ? o : 0) * n_in, o < n_out, c0, n_in, aligned, q, va, vb);
  }
};
struct DenseBias {
  const float* b;
  int n_out, m0;
  __device__ __forceinline__ float operator()(int t, int u) const {
    const int o = 16 * (m0 + t) + u;
    const float v = b[o < n_out ? o : 0];
    return o < n_out ? v : 0.f;
  }
};
struct DenseBiasHeads {  // tile t = (m - m0) * heads + head
  const float* b;
  int n_out, m0, heads;
  int64_t head_stride;
  __device__ __forceinline__ float operator()(int t, int u) const {
    const int head = t % heads, o = 16 * (m0 + t / heads) + u;
    const float v = b[o < n_out ? head * head_stride + o : 0];
    return o < n_out ? v : 0.f;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Where a kernel's A blocks come from: resident (staged once, absolute slots) or streamed (two buffers, one barrier per
// chunk).  chunk() is called by every wave of the workgroup at the same points.
// ---------------------------------------------------------------------------------------------------------------------
struct Chunk {
  const uint32_t* A;  // block b of the chunk at A + b * kBlockWords
  const float* bias;  // bias tile t at bias + 16 t
};

template <bool RESIDENT>
struct Source {
  static constexpr bool resident = RESIDENT;
  uint32_t* blocks;   // LDS: resident: all blocks; streaming: 2 buffers of cb blocks
  float* bias;        // LDS: resident: all bias tiles; streaming: 2 buffers of bt tiles
  int cb, bt, cur;
  int slot, btile;    // running position in the resident image
  float wdown;
  int spare;

  // streaming mode, chunks built from several parts: stage into cur_blocks() / cur_bias(), then commit()
  __device__ __forceinline__ uint32_t* cur_blocks() const { return blocks + cur * cb * kBlockWords; }
  __device__ __forceinline__ float* cur_bias() const { return bias + cur * bt * 16; }
  __device__ __forceinline__ void commit() {
    lds_barrier();
    cur ^= 1;
  }

  // PREFILL (resident mode's first pass): stage to the absolute position, no barrier, nothing is computed
  template <bool PREFILL, typename Fetch, typename Bias>
  __device__ __forceinline__ Chunk chunk(int n_blocks, const Fetch& fetch, int n_bias, const Bias& bias_fn) {
    Chunk c;
    if (RESIDENT) {
      uint32_t* a = blocks + slot * kBlockWords;
      float* b = bias + btile * 16;
      if (PREFILL) {
        stage_blocks(a, n_blocks, fetch, wdown);
        stage_bias(b, n_bias, bias_fn);
      }
      c.A = a;
      c.bias = b;
      slot += n_blocks;
      btile += n_bias;
    } else {
      uint32_t* a = blocks + cur * cb * kBlockWords;
      float* b = bias + cur * bt * 16;
      stage_blocks(a, n_blocks, fetch, wdown);
      stage_bias(b, n_bias, bias_fn);
      lds_barrier();  // (the other buffer is free once every wave is here: see the header comment)
      cur ^= 1;
      c.A = a;
      c.bias = b;
    }
    return c;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// register-resident hidden vectors
// ---------------------------------------------------------------------------------------------------------------------
template <int MT_MAX, int NTL>
struct Hidden {
  u32x2 hi[NTL][MT_MAX], lo[NTL][MT_MAX];  // split accumulator tiles (tiles >= MT hold zeros)
  float up[NTL];                           // 2^e of the row's scaling (1 normally): multiplies the next layer's products
};

template <int MT_MAX, int NTL>
struct Acc {
  f32x4 main[NTL][MT_MAX], corr[NTL][MT_MAX];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int m = 0; m < MT_MAX; ++m) {
        main[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        corr[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
};

__device__ __forceinline__ void read_block(const uint32_t* A, int b, int lane, f16x8& ah, f16x8& al) {
  const f16x8* p = reinterpret_cast<const f16x8*>(A + b * kBlockWords) + lane;
  ah = p[0];
  al = p[64];
}

// one K-step into the accumulators of tiles 0 .. MT-1: blocks b0 + m
template <int MT_MAX, int NTL>
__device__ __forceinline__ void mac_kstep(const uint32_t* A, int b0, int MT, int lane, const f16x8 (&bh)[NTL],
                                          const f16x8 (&bl)[NTL], f32x4 (&main)[NTL][MT_MAX], f32x4 (&corr)[NTL][MT_MAX]) {
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m)
    if (m < MT) {
      f16x8 ah, al;
      read_block(A, b0 + m, lane, ah, al);
#pragma unroll
      for (int t = 0; t < NTL; ++t) main[t][m] = mfma_h(ah, bh[t], main[t][m]);
#pragma unroll
      for (int t = 0; t < NTL; ++t) corr[t][m] = mfma_h(ah, bl[t], corr[t][m]);
#pragma unroll
      for (int t = 0; t < NTL; ++t) corr[t][m] = mfma_h(al, bh[t], corr[t][m]);
    }
}

// max over the four lanes (q = 0..3) that hold one row
__device__ __forceinline__ float max_over_q(float v) {
  v = __builtin_fmaxf(v, __shfl_xor(v, 16, 64));
  v = __builtin_fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// Two fp32 tiles (the K-step's halves) -> B operands, the row scaled by `down` first (1 normally)
__device__ __forceinline__ void split_kstep(const f32x4& a, const f32x4& b, float down, f16x8& bh, f16x8& bl, float& mx) {
  u32x2 ah, al, bh2, bl2;
  split_tile(a * down, ah, al, mx);
  split_tile(b * down, bh2, bl2, mx);
  bh = pair_operand(ah, bh2);
  bl = pair_operand(al, bl2);
}

// Accumulators -> activations -> the next layer's split tiles.  pre = (main + corr 2^-11) * scale + bias; LeakyReLU when
// `act`.  scale[t] = wup * (the input rows' factor); tiles >= MT come out as zeros.
template <int MT_MAX, int NTL>
__device__ __forceinline__ void finish_layer(const Acc<MT_MAX, NTL>& acc, const float* bias_tiles, int MT, int q,
                                             const float (&scale)[NTL], bool act, Hidden<MT_MAX, NTL>& h) {
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto value = [&](int t, int m, const f32x4& bias) {
    const f32x4 p = (acc.corr[t][m] * kSplitInvScale + acc.main[t][m]) * scale[t] + bias;
    return act ? __builtin_elementwise_max(p, p * kLeakySlope) : p;
  };
  float mx = 0.f;
#pragma unroll
  for (int m = 0; m < MT_MAX; ++m) {
    f32x4 bias = zero4;
    if (m < MT) bias = *reinterpret_cast<const f32x4*>(bias_tiles + 16 * m + 4 * q);
#pragma unroll
    for (int t = 0; t < NTL; ++t) split_tile(value(t, m, bias), h.hi[t][m], h.lo[t][m], mx);
  }
#pragma unroll
  for (int t = 0; t < NTL; ++t) h.up[t] = 1.f;
  if (__builtin_expect(wave_any(!(mx < kSplitLimit)), 0)) {  // rare: scale the rows that need it (exact: powers of two) and split again
    float fm[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) fm[t] = 0.f;
#pragma unroll
    for (int m = 0; m < MT_MAX; ++m) {
      f32x4 bias = zero4;
      if (m < MT) bias = *reinterpret_cast<const f32x4*>(bias_tiles + 16 * m + 4 * q);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const f32x4 v = value(t, m, bias);
#pragma unroll
        for (int r = 0; r < 4; ++r) fm[t] = __builtin_fmaxf(fm[t], finite_abs(v[r]));
      }
    }
    float down[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int e = down_exponent(max_over_q(fm[t]), 13);
      down[t] = pow2f(-e);
      h.up[t] = pow2f(e);
    }
    float unused = 0.f;
#pragma unroll
    for (int m = 0; m < MT_MAX; ++m) {
      f32x4 bias = zero4;
      if (m < MT) bias = *reinterpret_cast<const f32x4*>(bias_tiles + 16 * m + 4 * q);
#pragma unroll
      for (int t = 0; t < NTL; ++t) split_tile(value(t, m, bias) * down[t], h.hi[t][m], h.lo[t][m], unused);
    }
  }
}

// a hidden vector's K-step ks as B operands (static ks)
template <int MT_MAX, int NTL>
__device__ __forceinline__ void hidden_operand(const Hidden<MT_MAX, NTL>& h, int ks, f16x8 (&bh)[NTL], f16x8 (&bl)[NTL]) {
  const u32x2 zero2 = u32x2{0u, 0u};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    bh[t] = pair_operand(h.hi[t][2 * ks], 2 * ks + 1 < MT_MAX ? h.hi[t][2 * ks + 1 < MT_MAX ? 2 * ks + 1 : 0] : zero2);
    bl[t] = pair_operand(h.lo[t][2 * ks], 2 * ks + 1 < MT_MAX ? h.lo[t][2 * ks + 1 < MT_MAX ? 2 * ks + 1 : 0] : zero2);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// One conditioner net up to its LAST HIDDEN vector: Linear(in -> hid[0]) streamed over the input, then the
// hidden -> hidden layers in registers.  nd = MLP(sizes[0] .. sizes[n_lin]); the layers 0 .. n_hid-1 are evaluated here
// (n_hid = number of hidden vectors), each followed by LeakyReLU unless it is layer `no_act_layer`.
//   load_x(t, ks, a, b): the fp32 input of row tile t for K-step ks: a = columns 32 ks + 4 q + r, b = 32 ks + 16 + 4 q + r
//   (zeros beyond the input width); a pure read -- it is issued one K-step ahead, past the end the last step again.
//   use_x(t, ks, a, b): called once per K-step when its data is consumed (the caller's side effects: copies, sums).
//   hook(i, h): called with every finished hidden vector H_i, i = 1 .. n_hid (the gradient kernels keep them).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tiles16(int n) { return (n + 15) >> 4; }
__device__ __forceinline__ int steps32(int n) { return (n + 31) >> 5; }

struct NoLayerHook {
  template <typename H>
  __device__ __forceinline__ void operator()(int, const H&) const {}
};

// NN nets of ONE shape reading the same input (AffineHalfFlow's s- and t-net): a chunk holds, per K-step, the MT blocks of
// net 0, then those of net 1: digits (tile + MT * net, K-step - ks0); bias tiles likewise.
template <int NN>
struct DenseKMajorN {
  const float* W[NN];
  int n_in, n_out, MT, R0, ks0;  // R0 = NN * MT
  static constexpr int R1 = 1 << 30;
  __device__ __forceinline__ void load(int mm, int ksl, int, int i, int q, f32x4& va, f32x4& vb) const {
    const int net = mm >= MT ? 1 : 0, m = mm - net * MT;  // (NN <= 2)
    const float* Wn = W[NN > 1 ? net : 0];
    const int o = 16 * m + i, c0 = 32 * (ks0 + ksl);
    const bool aligned = (n_in & 3) == 0 && (reinterpret_cast<uintptr_t>(Wn) & 15) == 0;
    load_row8(Wn + (int64_t)(o < n_out ? o : 0) * n_in, o < n_out, c0, n_in, aligned, q, va, vb);
  }
};
template <int NN>
struct DenseBiasN {
  const float* b[NN];
  int n_out, MT;
  __device__ __forceinline__ float operator()(int t, int u) const {
    const int net = t >= MT ? 1 : 0, o = 16 * (t - net * MT) + u;
    const float v = b[NN > 1 ? net : 0][o < n_out ? o : 0];
    return o < n_out ? v : 0.f;
  }
};

// Layer 0 of NN nets of one shape over the same input, K-streamed: the input is loaded and split ONCE per K-step and
// multiplied into every net's accumulators (NN = 2: AffineHalfFlow's s- and t-net at hidden widths <= 64 -- the input
// side of a wide layer is most of its forward pass).  h[n] = H_1 of net n.
template <int MT_MAX, int NTL, bool PREFILL, int NN, typename Src, typename LoadX, typename UseX, typename Hook = NoLayerHook>
__device__ __forceinline__ void first_layer(Src& src, const float* __restrict__ flat, const NetDesc* const (&nds)[NN], bool act,
                                            float wup, int lane, int q, const LoadX& load_x, const UseX& use_x,
                                            Hidden<MT_MAX, NTL> (&h)[NN], const Hook& hook = Hook()) {
  Acc<MT_MAX, NTL> acc[NN];
  const NetDesc& nd = *nds[0];
  const int n_in = nd.sizes[0], n_out = nd.sizes[1];
  const int KS = steps32(n_in), MT = tiles16(n_out);
  int KC = src.cb / (NN * MT);  // K-steps per chunk
  if (KC < 1) KC = 1;
  if (Src::resident) KC = KS;
  if (!PREFILL) {
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n].zero();
  }
  float down[NTL];  // the rows' running scale (a power of two <= 1)
#pragma unroll
  for (int t = 0; t < NTL; ++t) down[t] = 1.f;
  // the input runs kRing K-steps ahead of the arithmetic in a register ring (a wave's 16 rows are 2 KB per K-step: a
  // CU needs tens of KB in flight to cover HBM latency); past the end the last step is read again and dropped
  f32x4 ra[kRing][NTL], rb[kRing][NTL];
  if (!PREFILL) {
#pragma unroll
    for (int u = 0; u < kRing; ++u)
#pragma unroll
      for (int t = 0; t < NTL; ++t) load_x(t, u < KS ? u : KS - 1, ra[u][t], rb[u][t]);
  }
  DenseKMajorN<NN> fetch;
  DenseBiasN<NN> bias_fn;
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    fetch.W[n] = flat + nds[n]->w_off[0];
    bias_fn.b[n] = flat + nds[n]->b_off[0];
  }
  fetch.n_in = n_in; fetch.n_out = n_out; fetch.MT = MT; fetch.R0 = NN * MT;
  bias_fn.n_out = n_out; bias_fn.MT = MT;
  Chunk c{nullptr, nullptr};
  int next_start = 0, chunk_start = 0;
  for (int ks_base = 0; ks_base < KS; ks_base += kRing) {
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
      const int ks = ks_base + u;
      if (ks < KS) {
        if (ks == next_start) {  // (uniform) a new chunk of A blocks starts at this K-step
          const int kc = KS - ks < KC ? KS - ks : KC;
          fetch.ks0 = ks;
          c = src.template chunk<PREFILL>(kc * NN * MT, fetch, ks + kc == KS ? NN * MT : 0, bias_fn);
          chunk_start = ks;
          next_start = ks + kc;
        }
        if (!PREFILL) {
          f16x8 bh[NTL], bl[NTL];
          f32x4 xa[NTL], xb[NTL];
          float mx = 0.f;
          const int ks_ahead = ks + kRing < KS ? ks + kRing : KS - 1;
#pragma unroll
          for (int t = 0; t < NTL; ++t) {
            xa[t] = ra[u][t];
            xb[t] = rb[u][t];
            load_x(t, ks_ahead, ra[u][t], rb[u][t]);
            use_x(t, ks, xa[t], xb[t]);
            split_kstep(xa[t], xb[t], down[t], bh[t], bl[t], mx);
          }
          if (__builtin_expect(wave_any(!(mx < kSplitLimit)), 0)) {
            // rare: a row at or beyond the split range (or non-finite).  The row's accumulators and every later K-step
            // of it move to a smaller power-of-two scale (exact); finish_layer multiplies the layer's result back
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
              float fm = 0.f;
#pragma unroll
              for (int r = 0; r < 4; ++r) fm = __builtin_fmaxf(fm, __builtin_fmaxf(finite_abs(xa[t][r]), finite_abs(xb[t][r])));
              const float want = pow2f(-down_exponent(max_over_q(fm), 13));
              if (want < down[t]) {
                const float f = want / down[t];
#pragma unroll
                for (int n = 0; n < NN; ++n)
#pragma unroll
                  for (int m = 0; m < MT_MAX; ++m) {
                    acc[n].main[t][m] *= f;
                    acc[n].corr[t][m] *= f;
                  }
                down[t] = want;
              }
              float unused = 0.f;
              split_kstep(xa[t], xb[t], down[t], bh[t], bl[t], unused);
            }
          }
#pragma unroll
          for (int n = 0; n < NN; ++n)
            mac_kstep<MT_MAX, NTL>(c.A, ((ks - chunk_start) * NN + n) * MT, MT, lane, bh, bl, acc[n].main, acc[n].corr);
        }
      }
    }
  }
  if (!PREFILL) {
    float scale[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) scale[t] = wup / down[t];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      finish_layer<MT_MAX, NTL>(acc[n], c.bias + 16 * n * MT, MT, q, scale, act, h[n]);
      if (n == 0) hook(1, h[0]);
    }
  }
}

// The hidden -> hidden layers 1 .. n_hid - 1 of one net in registers, from h = H_1 to h = H_{n_hid}.
template <int MT_MAX, int NTL, bool PREFILL, typename Src, typename Hook = NoLayerHook>
__device__ __forceinline__ void hidden_layers(Src& src, const float* __restrict__ flat, const NetDesc& nd, int n_hid,
                                              int no_act_layer, float wup, int lane, int q, Hidden<MT_MAX, NTL>& h,
                                              const Hook& hook = Hook()) {
  constexpr int KS_MAX = MT_MAX / 2;
  Acc<MT_MAX, NTL> acc;
  for (int l = 1; l < n_hid; ++l) {
    const int n_in = nd.sizes[l], n_out = nd.sizes[l + 1];
    const int KS = steps32(16 * tiles16(n_in)), MT = tiles16(n_out);
    int KC = src.cb / MT;
    if (KC < 1) KC = 1;
    if (Src::resident) KC = KS;
    if (!PREFILL) acc.zero();
    Chunk c{nullptr, nullptr};
    int next_start = 0, chunk_start = 0;
#pragma unroll
    for (int ks = 0; ks < KS_MAX; ++ks)
      if (ks < KS) {
        if (ks == next_start) {  // (uniform) a new chunk starts at this K-step
          const int kc = KS - ks < KC ? KS - ks : KC;
          c = src.template chunk<PREFILL>(kc * MT, DenseKMajor{flat + nd.w_off[l], n_in, n_out, MT, ks}, ks + kc == KS ? MT : 0,
                                 DenseBias{flat + nd.b_off[l], n_out, 0});
          chunk_start = ks;
          next_start = ks + kc;
        }
        if (!PREFILL) {
          f16x8 bh[NTL], bl[NTL];
          hidden_operand<MT_MAX, NTL>(h, ks, bh, bl);
          mac_kstep<MT_MAX, NTL>(c.A, (ks - chunk_start) * MT, MT, lane, bh, bl, acc.main, acc.corr);
        }
      }
    if (!PREFILL) {
      float scale[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) scale[t] = wup * h.up[t];
      finish_layer<MT_MAX, NTL>(acc, c.bias, MT, q, scale, l != no_act_layer, h);
      hook(l + 1, h);
    }
  }
}

template <int MT_MAX, int NTL, bool PREFILL, typename Src, typename LoadX, typename UseX, typename Hook = NoLayerHook>
__device__ __forceinline__ void net_to_hidden(Src& src, const float* __restrict__ flat, const NetDesc& nd, int n_hid,
                                              int no_act_layer, float wup, int lane, int q, const LoadX& load_x,
                                              const UseX& use_x, Hidden<MT_MAX, NTL>& h, const Hook& hook = Hook()) {
  const NetDesc* const nds[1] = {&nd};
  Hidden<MT_MAX, NTL> h1[1];
  first_layer<MT_MAX, NTL, PREFILL, 1>(src, flat, nds, no_act_layer != 0, wup, lane, q, load_x, use_x, h1, hook);
  h = h1[0];
  hidden_layers<MT_MAX, NTL, PREFILL>(src, flat, nd, n_hid, no_act_layer, wup, lane, q, h, hook);
}

// One output tile from a last-hidden vector: the KS blocks at A + b0, bias tile `bias16` (16 floats), scale as in
// finish_layer.  KS <= KS_MAX.
template <int MT_MAX, int NTL>
__device__ __forceinline__ void out_tile(const uint32_t* A, int b0, int KS, const float* bias16, int lane, int q,
                                         const Hidden<MT_MAX, NTL>& h, float wup, f32x4 (&out)[NTL]) {
  constexpr int KS_MAX = MT_MAX / 2;
  f32x4 mn[NTL], cr[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    mn[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    cr[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ks = 0; ks < KS_MAX; ++ks)
    if (ks < KS) {
      f16x8 ah, al, bh[NTL], bl[NTL];
      read_block(A, b0 + ks, lane, ah, al);
      hidden_operand<MT_MAX, NTL>(h, ks, bh, bl);
#pragma unroll
      for (int t = 0; t < NTL; ++t) mn[t] = mfma_h(ah, bh[t], mn[t]);
#pragma unroll
      for (int t = 0; t < NTL; ++t) cr[t] = mfma_h(ah, bl[t], cr[t]);
#pragma unroll
      for (int t = 0; t < NTL; ++t) cr[t] = mfma_h(al, bh[t], cr[t]);
    }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bias16 + 4 * q);
#pragma unroll
  for (int t = 0; t < NTL; ++t) out[t] = (cr[t] * kSplitInvScale + mn[t]) * (wup * h.up[t]) + bias;
}

// ---------------------------------------------------------------------------------------------------------------------
// row access: 4 consecutive columns of one row, zeros beyond `limit` columns; vec = the row's columns are 16-byte
// aligned and limit % 4 == 0 (one dwordx4), else element by element.  No load sits under a divergent branch (an
// out-of-range piece reads column 0 of the same row and is zeroed by a select).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 load4(const float* __restrict__ row, int col, int limit, bool vec) {
  if (vec) {
    const bool ok = col < limit;
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + (ok ? col : 0));
    return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = col + r < limit;
    const float x = row[ok ? col + r : 0];
    v[r] = ok ? x : 0.f;
  }
  return v;
}
__device__ __forceinline__ void store4(float* __restrict__ row, int col, int limit, bool vec, bool live, const f32x4& v) {
  if (!live) return;
  if (vec) {
    if (col < limit) *reinterpret_cast<f32x4*>(row + col) = v;
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (col + r < limit) row[col + r] = v[r];
}

}  // namespace rt
}  // namespace mnf
