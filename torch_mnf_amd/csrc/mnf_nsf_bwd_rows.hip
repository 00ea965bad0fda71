// NSF_CL gradients with one lane per (row, element): d = 32, three hidden layers of one width <= 8, K = 5 or 8
// (SURVEY.md 8f rank 1 for config 3; the generic kernel in mnf_backward.hip covers every other shape).
//
// A wave owns 4 rows at a time: lane (r = lane >> 4, j = lane & 15) is element j of both halves of row r.
//   * Hidden layers (16 -> 8 -> 8 -> 8): lane j computes unit j & 7 (both 8-lane halves of a row hold the same
//     units).  The inputs of a layer sit one per lane, so the dot product runs over DPP row rotations:
//     sum_n W[unit][source(j, n)] * rot_n(h), with the rotated weights read per lane from LDS.
//   * Output layer (8 -> 16 (3K-1)): the lane gathers the row's 8 h3 values (8 rotations) and computes the
//     3K-1 spline parameters of ITS element with weights read from LDS (stored rotated per element, so that
//     h3all[n] -- the unit rot_n delivers -- meets its weight), then differentiates the spline in registers.
//   * W4^T g_p: every lane forms its element's share for the 8 units (rotated order), a rotation butterfly
//     (1, 2, 4, 8 lanes, the unit index shifting along) sums the 16 elements of the row in every lane.
//   * Weight gradients are sums over rows: dW[m][n] += sum_r A[r][m] B[r][n] is one v_mfma_f32_16x16x4_f32 per
//     tile with the wave's 4 rows on the K axis -- lane (r, j) supplies A[r][j] and B[r][j] straight from its
//     registers (A: g_p[k] or a delta, B: the layer input with a column of ones for the bias).  The 2 (3K+3) tiles
//     stay in accumulator registers for the whole launch; one LDS sum over the 4 waves and one atomic per
//     parameter per workgroup at the end.
// No activations in LDS, no cross-lane traffic other than DPP, no atomics inside the row loop.
// Two kernels: nsf_bwd_rows_kernel (a wave does a tile's three steps itself: both nets' accumulators, one wave per
// SIMD) and nsf_bwd_pairs_kernel (the default: a pair of waves per tile, one net's accumulators each, two waves
// per SIMD; MNF_NSF_BWD_PAIRS=0 selects the former).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_nsf_spline_grad.h"

namespace mnf {
namespace {

constexpr int kNrWaves = 4;
constexpr int kNrHalf = 16;    // elements per half row
constexpr int kNrUnits = 8;    // hidden units per layer (narrower nets get structural-zero units)
constexpr int kNrHidRec = 59;  // per-lane hidden-layer record in LDS (odd pitch: conflict free)
// record: [0,16) W1 rotated, [16,24) W2 rotated, [24,32) W3 rotated, 32..34 biases,
//         [35,43) W3^T rotated, [43,51) W2^T rotated, [51,59) W1^T rotated

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// value of lane (j - N) or (j + N) of the same 16-lane row (whichever way row_ror turns: the weight tables are
// built from the rotation applied to the lane index itself)
template <int N>
__device__ __forceinline__ float rot(float x) {
  constexpr int S = N & 15;
  if constexpr (S == 0) {
    return x;
  } else {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + S, 0xf, 0xf, false));
  }
}
template <int N>
__device__ __forceinline__ int rot_int(int x) {
  constexpr int S = N & 15;
  if constexpr (S == 0) {
    return x;
  } else {
    return __builtin_amdgcn_update_dpp(0, x, 0x120 + S, 0xf, 0xf, false);
  }
}

using namespace nsfgrad;

template <int K>
struct NrShape {
  static constexpr int P = 3 * K - 1;
  static constexpr int REC = 8 * P + 4;  // floats per element record of the rotated output weights (+4: banks)
  static constexpr int NET_FLOATS = kNrHalf * REC + kNrHalf * P + kNrHalf * kNrHidRec;
  static constexpr int TILES = P + 4;    // dW4 per parameter index, dW3, dW2, dW1, db1
  static constexpr int RED_FLOATS = 2 * TILES * 256;
  static constexpr int LDS_FLOATS = RED_FLOATS > 2 * NET_FLOATS ? RED_FLOATS : 2 * NET_FLOATS;
};


struct NrArgs {
  const float* x;
  const float* grad_y;
  const float* grad_ld;
  float* grad_x;
  float* grad_flat;
  const float* flat;
  int64_t rows;
  float T;
  int hid;
  NetDesc f1, f2;
};

// The element's output-layer weights, 8 per spline parameter, streamed from LDS in groups of 4 parameters: the next
// group's reads are issued before the current group is consumed, and a scheduling barrier per group keeps the
// compiler from hoisting all 2 (3K-1) reads (184 registers) to the top.  f(k, weights 0..3, weights 4..7, bias).
template <int P, bool BIAS, typename F>
__device__ __forceinline__ void w4_stream(const float* w4, const float* b4, F&& f) {
  // (groups of 2: 1,752 us per launch against 1,800 with groups of 4, and no spills.  Round 4 also tried the two dot
  // products over the 8 units as v_pk_fma_f32 on adjacent weight pairs -- 490 packed instead of 1,053 plain fma in the
  // kernel -- and measured it SLOWER, 1,830 - 1,950 us with and without spills: on gfx950 a v_pk_fma_f32 costs more
  // issue time than the two v_fma_f32 it replaces.)
  constexpr int GS = 2, NG = (P + GS - 1) / GS;
  f32x4 wa[2][GS], wb[2][GS];
  float bs[2][GS];
  auto load = [&](auto g) {
    constexpr int gi = decltype(g)::value;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      const int k = gi * GS + i;
      if (k < P) {
        wa[gi & 1][i] = *reinterpret_cast<const f32x4*>(w4 + 8 * k);
        wb[gi & 1][i] = *reinterpret_cast<const f32x4*>(w4 + 8 * k + 4);
        bs[gi & 1][i] = BIAS ? b4[k] : 0.f;
      }
    }
  };
  load(std::integral_constant<int, 0>{});
  static_for<0, NG>([&](auto g) {
    constexpr int gi = decltype(g)::value;
    if constexpr (gi + 1 < NG) load(std::integral_constant<int, gi + 1>{});
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      const int k = gi * GS + i;
      if (k < P) f(k, wa[gi & 1][i], wb[gi & 1][i], bs[gi & 1][i]);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// conditioner forward for this lane's row: hidden activations (unit j & 7), the row's h3 in rotated order, and the
// 3K-1 raw spline parameters of element j
template <int K>
__device__ __forceinline__ void net_forward(const float* hw, const float* w4, const float* b4, float cond, float& h1,
                                            float& h2, float& h3, float (&h3all)[kNrUnits], float (&p)[3 * K - 1]) {
  float acc = hw[32];
  static_for<0, 16>([&](auto n) { acc = fmaf(hw[n], rot<decltype(n)::value>(cond), acc); });
  h1 = leaky(acc);
  acc = hw[33];
  static_for<0, 8>([&](auto n) { acc = fmaf(hw[16 + n], rot<decltype(n)::value>(h1), acc); });
  h2 = leaky(acc);
  acc = hw[34];
  static_for<0, 8>([&](auto n) { acc = fmaf(hw[24 + n], rot<decltype(n)::value>(h2), acc); });
  h3 = leaky(acc);
  static_for<0, 8>([&](auto n) { h3all[n] = rot<decltype(n)::value>(h3); });
  w4_stream<3 * K - 1, true>(w4, b4, [&](int k, const f32x4& wa, const f32x4& wb, float bias) {
    float a = bias;
#pragma unroll
    for (int n = 0; n < 4; ++n) a = fmaf(wa[n], h3all[n], a);
#pragma unroll
    for (int n = 0; n < 4; ++n) a = fmaf(wb[n], h3all[4 + n], a);
    p[k] = a;
  });
}

// One half-step backwards.  cond: the conditioning half's element; val: the transformed half's element BEFORE the
// spline; g_val: cotangent of the element after it (becomes the cotangent of val); g_cond gets the net's share.
template <int K, bool INV>
__device__ __forceinline__ void half_backward(const float* hw, const float* w4, const float* b4, int j, float T,
                                              float cond, float val, float g_ld, float& g_val, float& g_cond,
                                              f32x4 (&acc)[3 * K + 3]) {
  constexpr int P = 3 * K - 1;
  float h1, h2, h3, h3all[kNrUnits], p[P], g_p[P];
  net_forward<K>(hw, w4, b4, cond, h1, h2, h3, h3all, p);
  float g_v;
  rqs_grad<K, INV>(val, T, p, g_val, g_ld, g_v, g_p);
  g_val = g_v;
  // B operands: the layer input with a ones column (n = 8) for the bias
  const float one8 = j == kNrUnits ? 1.f : 0.f;
  const float hb3 = j < kNrUnits ? h3 : one8, hb2 = j < kNrUnits ? h2 : one8, hb1 = j < kNrUnits ? h1 : one8;
#pragma unroll
  for (int k = 0; k < P; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_p[k], hb3, acc[k], 0, 0, 0);
  // W4^T g_p: this element's share per unit (rotated order), then the sum over the row's 16 elements
  float y[kNrUnits];
#pragma unroll
  for (int n = 0; n < kNrUnits; ++n) y[n] = 0.f;
  w4_stream<P, false>(w4, b4, [&](int k, const f32x4& wa, const f32x4& wb, float) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      y[n] = fmaf(wa[n], g_p[k], y[n]);
      y[4 + n] = fmaf(wb[n], g_p[k], y[4 + n]);
    }
  });
  static_for<0, 3>([&](auto s) {
    constexpr int t = 1 << decltype(s)::value;
    float z[kNrUnits];
    static_for<0, kNrUnits>([&](auto n) { z[n] = y[n] + rot<t>(y[(decltype(n)::value - t) & 7]); });
#pragma unroll
    for (int n = 0; n < kNrUnits; ++n) y[n] = z[n];
  });
#pragma unroll
  for (int n = 0; n < kNrUnits; ++n) y[n] += rot<8>(y[n]);
  // delta3 for every unit (rotated order; index 0 is this lane's own unit)
#pragma unroll
  for (int n = 0; n < kNrUnits; ++n) y[n] = h3all[n] > 0.f ? y[n] : kLeakySlope * y[n];
  acc[P] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[0], hb2, acc[P], 0, 0, 0);
  float g2 = 0.f;
#pragma unroll
  for (int n = 0; n < kNrUnits; ++n) g2 = fmaf(hw[35 + n], y[n], g2);
  const float d2 = h2 > 0.f ? g2 : kLeakySlope * g2;
  acc[P + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d2, hb1, acc[P + 1], 0, 0, 0);
  float g1 = 0.f;
  static_for<0, 8>([&](auto n) { g1 = fmaf(hw[43 + n], rot<decltype(n)::value>(d2), g1); });
  const float d1 = h1 > 0.f ? g1 : kLeakySlope * g1;
  acc[P + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, cond, acc[P + 2], 0, 0, 0);
  acc[P + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, 1.f, acc[P + 3], 0, 0, 0);
  float gc = 0.f;
  static_for<0, 8>([&](auto n) { gc = fmaf(hw[51 + n], rot<decltype(n)::value>(d1), gc); });
  g_cond += gc;
}

// flat-parameter offset of accumulator element (tile t, lane, reg) of a net, -1 if it is padding
template <int K>
__device__ __forceinline__ int flush_offset(const NetDesc& nd, int hid, int t, int lane, int reg) {
  constexpr int P = 3 * K - 1;
  const int n = lane & 15, m = 4 * (lane >> 4) + reg;
  if (t < P) {  // output layer: row (element m, parameter t), column n = hidden unit, n = 8: bias
    if (n < hid) return nd.w_off[3] + (m * P + t) * hid + n;
    return n == kNrUnits ? nd.b_off[3] + m * P + t : -1;
  }
  if (m >= hid) return -1;
  if (t == P || t == P + 1) {
    const int l = t == P ? 2 : 1;
    if (n < hid) return nd.w_off[l] + m * hid + n;
    return n == kNrUnits ? nd.b_off[l] + m : -1;
  }
  if (t == P + 2) return nd.w_off[0] + m * kNrHalf + n;
  return n == 0 ? nd.b_off[0] + m : -1;
}

// the inverse: accumulator element (t * 256 + lane * 4 + reg) that holds the gradient of flat parameter p of the net
template <int K>
__device__ __forceinline__ int flush_source(const NetDesc& nd, int hid, int p) {
  constexpr int P = 3 * K - 1;
  int t, m, n;
  if (p >= nd.b_off[3]) {
    const int q = p - nd.b_off[3];
    m = q / P, t = q - m * P, n = kNrUnits;
  } else if (p >= nd.w_off[3]) {
    const int q = p - nd.w_off[3], mt = q / hid;
    n = q - mt * hid, m = mt / P, t = mt - m * P;
  } else if (p >= nd.b_off[2]) {
    m = p - nd.b_off[2], t = P, n = kNrUnits;
  } else if (p >= nd.w_off[2]) {
    const int q = p - nd.w_off[2];
    m = q / hid, n = q - m * hid, t = P;
  } else if (p >= nd.b_off[1]) {
    m = p - nd.b_off[1], t = P + 1, n = kNrUnits;
  } else if (p >= nd.w_off[1]) {
    const int q = p - nd.w_off[1];
    m = q / hid, n = q - m * hid, t = P + 1;
  } else if (p >= nd.b_off[0]) {
    m = p - nd.b_off[0], t = P + 3, n = 0;
  } else {
    const int q = p - nd.w_off[0];
    m = q / kNrHalf, n = q - m * kNrHalf, t = P + 2;
  }
  return t * 256 + (16 * (m >> 2) + n) * 4 + (m & 3);
}

// The workgroup's sums (lds: [net][tile][lane][reg]) added to grad_flat in PARAMETER order: consecutive threads add to
// consecutive addresses.  (In accumulator order the same 7,184 atomics per workgroup were scattered over the buffer and
// cost 123 us of the launch's 1,858 at 2^20 rows: scattered atomics run ~10 x slower at the memory side, section 3.6.)
template <int K>
__device__ __forceinline__ void flush_nets(const NrArgs& a, const float* lds) {
  using S = NrShape<K>;
  for (int net = 0; net < 2; ++net) {
    const NetDesc& nd = net ? a.f2 : a.f1;
    const int base = nd.w_off[0], count = nd.b_off[3] + kNrHalf * S::P - base;
    for (int i = threadIdx.x; i < count; i += blockDim.x)
      atomicAdd(a.grad_flat + base + i, lds[net * S::TILES * 256 + flush_source<K>(nd, a.hid, base + i)]);
  }
}

// LDS images of both nets (rotated output weights, biases, per-lane hidden records); the caller syncs
template <int K>
__device__ __forceinline__ void fill_net_images(float* lds, const NrArgs& a, int sgn) {
  using S = NrShape<K>;
  constexpr int P = S::P;
  const int hid = a.hid;
  for (int net = 0; net < 2; ++net) {
    const NetDesc& nd = net ? a.f2 : a.f1;
    float* base = lds + net * S::NET_FLOATS;
    const float* W4 = a.flat + nd.w_off[3];
    for (int i = threadIdx.x; i < kNrHalf * P * 8; i += blockDim.x) {
      const int jj = i / (P * 8), rem = i - jj * (P * 8), k = rem >> 3, n = rem & 7;
      const int unit = (jj + sgn * n) & 7;
      base[jj * S::REC + k * 8 + n] = unit < hid ? W4[(jj * P + k) * hid + unit] : 0.f;
    }
    float* b4 = base + kNrHalf * S::REC;
    for (int i = threadIdx.x; i < kNrHalf * P; i += blockDim.x) b4[i] = a.flat[nd.b_off[3] + i];
    float* hwr = b4 + kNrHalf * P;
    for (int i = threadIdx.x; i < kNrHalf * kNrHidRec; i += blockDim.x) {
      const int jj = i / kNrHidRec, e = i - jj * kNrHidRec, u = jj & 7;
      const bool uv = u < hid;
      float v = 0.f;
      if (e < 16) {
        v = uv ? a.flat[nd.w_off[0] + u * kNrHalf + ((jj + sgn * e) & 15)] : 0.f;
      } else if (e < 32) {
        const int l = e < 24 ? 1 : 2, un = (jj + sgn * (e & 7)) & 7;
        v = uv && un < hid ? a.flat[nd.w_off[l] + u * hid + un] : 0.f;
      } else if (e < 35) {
        v = uv ? a.flat[nd.b_off[e - 32] + u] : 0.f;
      } else if (e < 51) {
        const int l = e < 43 ? 2 : 1, un = (jj + sgn * ((e - 35) & 7)) & 7;
        v = uv && un < hid ? a.flat[nd.w_off[l] + un * hid + u] : 0.f;
      } else {
        const int un = (jj + sgn * (e - 51)) & 7;
        v = un < hid ? a.flat[nd.w_off[0] + un * kNrHalf + jj] : 0.f;
      }
      hwr[i] = v;
    }
  }
}

template <int K, bool INV>
__global__ void __launch_bounds__(kNrWaves * 64, 1) nsf_bwd_rows_kernel(NrArgs a) {
  using S = NrShape<K>;
  constexpr int P = S::P, dim = 2 * kNrHalf;
  __shared__ __attribute__((aligned(16))) float lds[S::LDS_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, r = lane >> 4;
  // which way the rotation turns: rot<1> of the lane's own element index is j - 1 or j + 1
  const int sgn = ((rot_int<1>(j) - j) & 15) == 1 ? 1 : -1;

  fill_net_images<K>(lds, a, sgn);
  __syncthreads();

  f32x4 acc1[S::TILES], acc2[S::TILES];
#pragma unroll
  for (int t = 0; t < S::TILES; ++t) acc1[t] = acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t n_tiles = (a.rows + 3) >> 2, stride = (int64_t)gridDim.x * kNrWaves;
  struct RowIn {
    float lo, up, g_lo, g_up, g_ld;
  };
  auto load_rows = [&](int64_t tile) {
    const int64_t row = tile * 4 + r;
    const bool live = row < a.rows;
    const int64_t rowc = live ? row : a.rows - 1;
    RowIn in;
    in.lo = a.x[rowc * dim + j];
    in.up = a.x[rowc * dim + kNrHalf + j];
    in.g_lo = (a.grad_y && live) ? a.grad_y[rowc * dim + j] : 0.f;
    in.g_up = (a.grad_y && live) ? a.grad_y[rowc * dim + kNrHalf + j] : 0.f;
    in.g_ld = (a.grad_ld && live) ? a.grad_ld[rowc] : 0.f;
    return in;
  };
  int64_t tile = (int64_t)blockIdx.x * kNrWaves + wave;
  RowIn next = load_rows(tile < n_tiles ? tile : 0);
  for (; tile < n_tiles; tile += stride) {
    const RowIn cur = next;
    next = load_rows(tile + stride < n_tiles ? tile + stride : tile);  // one trip ahead: a wave has no one to hide behind
    __builtin_amdgcn_sched_barrier(0);
    const int64_t row = tile * 4 + r;
    const bool live = row < a.rows;
    const int64_t rowc = live ? row : a.rows - 1;
    const float lo0 = cur.lo, up0 = cur.up, gl = cur.g_ld;
    float g_lo = cur.g_lo, g_up = cur.g_up;

    int off1 = j, off2 = S::NET_FLOATS + j;
    asm volatile("" : "+v"(off1), "+v"(off2));  // keep the weight reads inside the row loop
    const float* n1 = lds + (off1 - j), * n2 = lds + (off2 - j);
    // per-lane pointers into a net's image: element record, biases, hidden record
    auto w4_of = [&](const float* nb) { return nb + j * S::REC; };
    auto b4_of = [&](const float* nb) { return nb + kNrHalf * S::REC + j * P; };
    auto hw_of = [&](const float* nb) { return nb + kNrHalf * S::REC + kNrHalf * P + j * kNrHidRec; };
    // forward:  up1 = S(up0; f1(lo0)),  lo1 = S(lo0; f2(up1))       -> reverse: f2's step, then f1's
    // inverse:  lo1 = S^-1(lo0; f2(up0)), up1 = S^-1(up0; f1(lo1))  -> reverse: f1's step, then f2's
    const float* na = INV ? n2 : n1;  // the net of the first half-step
    const float* nb = INV ? n1 : n2;
    const float cond_a = INV ? up0 : lo0, val_a = INV ? lo0 : up0, val_b = INV ? up0 : lo0;
    float mid;
    {
      float h1, h2, h3, h3all[kNrUnits], p[P];
      net_forward<K>(hw_of(na), w4_of(na), b4_of(na), cond_a, h1, h2, h3, h3all, p);
      mid = rqs_value<K, INV>(val_a, a.T, p);
    }
    float& g_a = INV ? g_lo : g_up;  // cotangent of the first half-step's output half
    float& g_b = INV ? g_up : g_lo;
    half_backward<K, INV>(hw_of(nb), w4_of(nb), b4_of(nb), j, a.T, mid, val_b, gl, g_b, g_a, INV ? acc1 : acc2);
    half_backward<K, INV>(hw_of(na), w4_of(na), b4_of(na), j, a.T, cond_a, val_a, gl, g_a, g_b, INV ? acc2 : acc1);
    if (live) {
      a.grad_x[rowc * dim + j] = g_lo;
      a.grad_x[rowc * dim + kNrHalf + j] = g_up;
    }
  }

  // ------------------------------------------------------------------ flush: sum over the waves, one atomic per parameter
  if (a.grad_flat == nullptr) return;
  __syncthreads();
  for (int w = 0; w < kNrWaves; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < S::TILES; ++t) {
        f32x4* p1 = reinterpret_cast<f32x4*>(lds + t * 256 + lane * 4);
        f32x4* p2 = reinterpret_cast<f32x4*>(lds + (S::TILES + t) * 256 + lane * 4);
        *p1 = w == 0 ? acc1[t] : *p1 + acc1[t];
        *p2 = w == 0 ? acc2[t] : *p2 + acc2[t];
      }
    }
    __syncthreads();
  }
  flush_nets<K>(a, lds);
}

// Two waves per SIMD: the accumulators of BOTH nets (216 registers) are what keeps nsf_bwd_rows_kernel at one wave per
// SIMD, where nothing hides a dependent instruction's latency (vector unit busy 51 % of the time).  Here a pair of
// waves shares a 4-row tile: wave X (waves 0..3 of the workgroup) owns the first half-step's net and its
// accumulators, wave Y (waves 4..7) the second's.  Per tile: X runs the first net forward and hands the half it
// produced to Y; Y differentiates the second half-step and hands both cotangents back; X differentiates the first.
// Software-pipelined over the pair's tiles, one workgroup barrier per slot: in slot s, X does step 3 of tile s - 2
// and step 1 of tile s while Y does step 2 of tile s - 1.  (Step 1 holds no accumulators and could alternate between
// the waves to even their load out: measured slower, 2.04 vs 1.87 ms at 2^20 rows -- both code paths in both waves.)  Hand-over through a double-buffered 3-float mailbox per lane.
constexpr int kNpPairs = 4;

template <int K, bool INV>
__global__ void __launch_bounds__(2 * kNpPairs * 64, 1) nsf_bwd_pairs_kernel(NrArgs a) {
  using S = NrShape<K>;
  constexpr int P = S::P, dim = 2 * kNrHalf;
  __shared__ __attribute__((aligned(16))) float lds[S::LDS_FLOATS];
  __shared__ float mail[2][kNpPairs][3][64];  // [slot parity][pair][first step's output | g_a | g_b][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = wave & (kNpPairs - 1), role = wave / kNpPairs;  // role 0: X, 1: Y (one of each per SIMD)
  const int j = lane & 15, r = lane >> 4;
  const int sgn = ((rot_int<1>(j) - j) & 15) == 1 ? 1 : -1;
  fill_net_images<K>(lds, a, sgn);
  __syncthreads();

  f32x4 acc[S::TILES];
#pragma unroll
  for (int t = 0; t < S::TILES; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (32-bit tile numbers and element offsets -- the launcher admits rows x dim < 2^31 --: the 64-bit forms of these cost
  //  the registers the row prefetch below needs)
  const int n_rows = (int)a.rows;
  const int n_tiles = (n_rows + 3) >> 2, stride = (int)gridDim.x * kNpPairs;
  const int first0 = (int)blockIdx.x * kNpPairs, first = first0 + pair;
  // workgroup-uniform slot count (pair 0 has the most tiles), + 2 slots to drain the pipeline
  const int n_slots = (first0 < n_tiles ? (n_tiles - first0 + stride - 1) / stride : 0) + 2;
  // forward:  up1 = S(up0; f1(lo0)),  lo1 = S(lo0; f2(up1))       -> reverse: f2's step, then f1's
  // inverse:  lo1 = S^-1(lo0; f2(up0)), up1 = S^-1(up0; f1(lo1))  -> reverse: f1's step, then f2's
  constexpr int net_a = INV ? 1 : 0, net_b = 1 - net_a;      // f1 = 0, f2 = 1
  constexpr int col_a = INV ? 0 : kNrHalf, col_b = kNrHalf - col_a;  // columns of the half each step transforms
  // Row values a slot AHEAD (round 4): every step used to load its x / grad_y / grad_ld values where it needed them -- up
  // to three exposed global-memory latencies per slot, the third behind the slot's grad_x stores -- with both waves of a
  // SIMD (one pair) waiting at the same time.  Now slot s requests what slot s + 1 will use: five loads per wave from
  // role-dependent addresses, no branch around them (behind a load under a branch hipcc's wait counts fall back to
  // vmcnt(0)); a missing cotangent reads x, a step without a tile the nearest tile: both are not used.
  //   X (role 0): 0 cond, 1 value, 2 grad_ld of step 3's tile; 3 cond, 4 value of step 1's tile
  //   Y (role 1): 0 value, 1 g_a, 2 grad_ld, 3 g_b of step 2's tile
  const float* const gy_or_x = a.grad_y ? a.grad_y : a.x;
  const float* const gl_or_x = a.grad_ld ? a.grad_ld : a.x;
  auto tile_row = [&](int t) -> uint32_t {
    t = t < 0 ? 0 : (t < n_tiles ? t : n_tiles - 1);
    const int row = t * 4 + r;
    return (uint32_t)(row < n_rows ? row : n_rows - 1);
  };
  float nx[5];
  auto request_rows = [&](int sn) {
    const uint32_t ra = tile_row(first + (sn - (role ? 1 : 2)) * stride);  // step 3's (X) / step 2's (Y) tile
    const uint32_t rb = role ? ra : tile_row(first + sn * stride);          // step 1's tile (X)
    const float* p0 = a.x + ra * dim + col_b + j;
    const float* p1 = (role ? gy_or_x : a.x) + ra * dim + col_a + j;
    const float* p2 = gl_or_x + ra;
    const float* p3 = (role ? gy_or_x : a.x) + rb * dim + col_b + j;
    const float* p4 = a.x + rb * dim + col_a + j;
    nx[0] = *p0;
    nx[1] = *p1;
    nx[2] = *p2;
    nx[3] = *p3;
    nx[4] = *p4;
  };
  request_rows(0);
  for (int s = 0; s < n_slots; ++s) {
    float cur[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) cur[i] = nx[i];
    request_rows(s + 1);
    int off = (role ? net_b : net_a) * S::NET_FLOATS + j;
    asm volatile("" : "+v"(off));  // keep the weight reads inside the slot loop
    const float* nb = lds + (off - j);
    const float* w4 = nb + j * S::REC;
    const float* b4 = nb + kNrHalf * S::REC + j * P;
    const float* hw = nb + kNrHalf * S::REC + kNrHalf * P + j * kNrHidRec;
    if (role == 0) {
      const int t3 = first + (s - 2) * stride, t1 = first + s * stride;
      if (s >= 2 && t3 < n_tiles) {  // step 3: the first half-step backwards
        const int row = t3 * 4 + r;
        const bool live = row < n_rows;
        const uint32_t rowc = (uint32_t)(live ? row : n_rows - 1);
        const float cond_a = cur[0], val_a = cur[1];
        const float gl = (a.grad_ld && live) ? cur[2] : 0.f;
        float g_a = mail[(s - 1) & 1][pair][1][lane], g_b = mail[(s - 1) & 1][pair][2][lane];
        half_backward<K, INV>(hw, w4, b4, j, a.T, cond_a, val_a, gl, g_a, g_b, acc);
        if (live) {
          a.grad_x[rowc * dim + col_a + j] = g_a;
          a.grad_x[rowc * dim + col_b + j] = g_b;
        }
      }
      if (t1 < n_tiles) {  // step 1: the first net forward, the half it produces
        const float cond_a = cur[3], val_a = cur[4];
        float h1, h2, h3, h3all[kNrUnits], p[P];
        net_forward<K>(hw, w4, b4, cond_a, h1, h2, h3, h3all, p);
        mail[s & 1][pair][0][lane] = rqs_value<K, INV>(val_a, a.T, p);
      }
    } else {
      const int t2 = first + (s - 1) * stride;
      if (s >= 1 && t2 < n_tiles) {  // step 2: the second half-step backwards, conditioned on X's output
        const int row = t2 * 4 + r;
        const bool live = row < n_rows;
        const float val_b = cur[0];
        float g_a = (a.grad_y && live) ? cur[1] : 0.f;
        float g_b = (a.grad_y && live) ? cur[3] : 0.f;
        const float gl = (a.grad_ld && live) ? cur[2] : 0.f;
        const float mid = mail[(s - 1) & 1][pair][0][lane];
        half_backward<K, INV>(hw, w4, b4, j, a.T, mid, val_b, gl, g_b, g_a, acc);
        mail[s & 1][pair][1][lane] = g_a;
        mail[s & 1][pair][2][lane] = g_b;
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ flush: X waves hold the first net's sums, Y waves the second's
  if (a.grad_flat == nullptr) return;
  for (int w = 0; w < kNpPairs; ++w) {
    if (pair == w) {
      float* area = lds + (role ? net_b : net_a) * S::TILES * 256;
#pragma unroll
      for (int t = 0; t < S::TILES; ++t) {
        f32x4* q = reinterpret_cast<f32x4*>(area + t * 256 + lane * 4);
        *q = w == 0 ? acc[t] : *q + acc[t];
      }
    }
    __syncthreads();
  }
  flush_nets<K>(a, lds);
}

template <int K>
int launch_rows(const NrArgs& a, int inverse, hipStream_t stream) {
  const int dev = current_device();
  const int64_t n_tiles = (a.rows + 3) / 4;
  int64_t blocks = (n_tiles + kNrWaves - 1) / kNrWaves;
  const int cus = device_cus(dev);
  if (blocks > cus) blocks = cus;  // one persistent workgroup per CU: the accumulators take the register file
  const char* e = getenv("MNF_NSF_BWD_PAIRS");  // (read per call: the tests run both kernels in one process)
  if (!e || e[0] != '0') {  // (same tiles per workgroup and trip: 4 pairs of waves instead of 4 waves)
    if (inverse)
      hipLaunchKernelGGL((nsf_bwd_pairs_kernel<K, true>), dim3((unsigned)blocks), dim3(2 * kNpPairs * 64), 0, stream, a);
    else
      hipLaunchKernelGGL((nsf_bwd_pairs_kernel<K, false>), dim3((unsigned)blocks), dim3(2 * kNpPairs * 64), 0, stream, a);
    return check_launch();
  }
  if (inverse)
    hipLaunchKernelGGL((nsf_bwd_rows_kernel<K, true>), dim3((unsigned)blocks), dim3(kNrWaves * 64), 0, stream, a);
  else
    hipLaunchKernelGGL((nsf_bwd_rows_kernel<K, false>), dim3((unsigned)blocks), dim3(kNrWaves * 64), 0, stream, a);
  return check_launch();
}

}  // namespace
}  // namespace mnf

extern "C" {

int mnf_nsf_cl_bwd_rows_supported(int dim, int K, int n_hidden, const int* hidden) {
  if (dim != 2 * mnf::kNrHalf || (K != 5 && K != 8) || n_hidden != 3 || !hidden) return 0;
  return hidden[0] >= 1 && hidden[0] <= mnf::kNrUnits && hidden[1] == hidden[0] && hidden[2] == hidden[0];
}

int mnf_nsf_cl_bwd_rows(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                        const float* flat, int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden,
                        const int* hidden, void* stream) {
  if (!x || !grad_x || !flat || rows < 0 || dim < 2 || (dim & 1) || K < 2 || !(tail_bound > 0.f) ||
      !mnf::hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (!mnf_nsf_cl_bwd_rows_supported(dim, K, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
  if (rows * dim >= (1ll << 31)) return MNF_ERR_UNSUPPORTED;  // (32-bit element offsets: the caller's generic kernel)
  if (rows == 0) return MNF_OK;
  mnf::NrArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.grad_y = grad_y; a.grad_ld = grad_ld; a.grad_x = grad_x; a.grad_flat = grad_flat; a.flat = flat;
  a.rows = rows; a.T = tail_bound; a.hid = hidden[0];
  int sizes[5] = {mnf::kNrHalf, hidden[0], hidden[1], hidden[2], (3 * K - 1) * mnf::kNrHalf};
  const int64_t off = mnf::fill_net(a.f1, 5, sizes, 0);
  mnf::fill_net(a.f2, 5, sizes, off);
  return K == 8 ? mnf::launch_rows<8>(a, inverse, (hipStream_t)stream)
                : mnf::launch_rows<5>(a, inverse, (hipStream_t)stream);
}

}  // extern "C"
