// AffineHalfFlow gradients on fp32 MFMAs (SURVEY.md 8f rank 1 at matrix-pipe rate).
//
// One wave owns 16 rows at a time.  Per tile it
//   1. recomputes the two conditioner nets (as one concatenated net, block-diagonal hidden layers) with
//      v_mfma_f32_16x16x4_f32, parking every activation tile in LDS,
//   2. forms the output deltas from grad_y / grad_ld and the transform,
//   3. back-propagates them through the transposed weights (same MFMA scheme: the accumulator layout of
//      one product is the B-operand layout of the next), and
//   4. accumulates the weight gradients dW_l += delta_l^T h_{l-1} with the 16 rows on the MFMA K axis.
//      That product needs rows on K where everything else has them on N, so delta and h tiles go through
//      a per-wave LDS scratch ([row][unit], 20-float pitch: conflict-free float4 writes) and are read back
//      transposed; the bias gradients are the sums of those same operand reads.
// The 28 weight-gradient tiles (112 VGPRs at d = 64) stay in registers across all tiles of a wave; at the
// end the four waves of a workgroup add them up in LDS and issue one atomic add per parameter.
// Operand images (forward and transposed weights, 57 KB at d = 64) are gathered from `flat` into LDS by every
// workgroup through an index table the caller builds once per shape (mnf_affine_half_bwd_index).
// Tile halves of 16, 32, 64 and 128 columns (d = 256: the layer-1 / output-layer weights once in the image, read
// transposed by the backward products -- BwdShape::SHARED_T); a coupling half narrower
// than its tile is padded (RAG: zero operands, masked row accesses) -- d = 2, the reference's half-moons model, runs here.
#include <hip/hip_runtime.h>

#include "mnf_ahf_bwd_shape.h"
#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {

// D-layout tile (lane (j, q) holds units 4q..4q+3 of row j) -> LDS scratch [row][unit]
__device__ __forceinline__ void tile_to_lds(float* tile, int j, int q, const f32x4& v) {
  *reinterpret_cast<f32x4*>(tile + j * kTilePitch + 4 * q) = v;
}
__device__ __forceinline__ f32x4 tile_from_lds(const float* tile, int j, int q) {
  return *reinterpret_cast<const f32x4*>(tile + j * kTilePitch + 4 * q);
}
// transposed read: element [row 4 s + kq][unit i] for the four K-steps s (rows on the MFMA K axis)
__device__ __forceinline__ void tile_rows_on_k(const float* tile, int i, int kq, float (&out)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) out[s] = tile[(4 * s + kq) * kTilePitch + i];
}

// RAG: the coupling half is narrower than the tile (real_h < H: d = 2 ... ): the operand images carry zeros in the padded
// columns (build_bwd_index), rows are read and written element by element under a column mask.
constexpr int kBwdBlocksMode = -1;  // list_capacity of a launch whose grad_flat is the per-workgroup block workspace

template <int H, int HID, bool INV, bool RAG>
__global__ void __launch_bounds__((BwdShape<H, HID>::WAVES * 64), 1)
ahf_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ grad_y, const float* __restrict__ grad_ld,
                    float* __restrict__ grad_x, float* __restrict__ grad_flat, const float* __restrict__ flat,
                    const int32_t* __restrict__ index, int64_t rows, int parity, const int32_t* __restrict__ tile_list,
                    int list_capacity, int real_h) {
  using S = BwdShape<H, HID>;
  // tile_list = [count, tile, tile, ...]: only those 16-row tiles (the ones mnf_affine_half_bwd_split handed back
  // because an operand left the split range; count < 0: all of them -- weights beyond the split range); nullptr:
  // every tile
  const int listed = tile_list ? tile_list[0] : -1;
  if (listed == 0) return;
  constexpr int G = S::G, NT = S::NT;
  const int hh = RAG ? real_h : H, dim = 2 * hh;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // (eight independent index -> weight load chains per thread in flight: at a few hundred rows the gather is what the
  //  launch costs)
  for (int i0 = threadIdx.x; i0 < S::IMAGE_FLOATS; i0 += 8 * blockDim.x) {
    int32_t src[8];
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      src[u] = i < S::IMAGE_FLOATS ? index[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = src[u] < 0 ? 0.f : flat[src[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      if (i < S::IMAGE_FLOATS) lds[i] = w[u];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int cond_off = parity ? hh : 0, act_off = parity ? 0 : hh;
  float* scratch = lds + S::IMAGE_FLOATS + wave * S::SCRATCH_TILES * kTileFloats;
  float* TH = scratch;                             // h1, h2, h3: 3 NT tiles
  float* TD = TH + 3 * NT * kTileFloats;           // deltas of the layer in flight (the output layer's: one net at a time)

  f32x4 dW[S::DW_TILES];
  float db[S::DB_TILES];
#pragma unroll
  for (int t = 0; t < S::DW_TILES; ++t) dW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < S::DB_TILES; ++t) db[t] = 0.f;

  const int n_tiles = (int)((rows + 15) >> 4);
  const int n_items = listed < 0 ? n_tiles : (listed < list_capacity ? listed : list_capacity);
  for (int item = (int)blockIdx.x * S::WAVES + wave; item < n_items; item += (int)gridDim.x * S::WAVES) {
    const int tile = listed < 0 ? item : tile_list[1 + item];
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* xr = x + rowc * dim + 4 * q;
    f32x4 cnd[G], act[G], gc[G], ga[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RAG) {
        cnd[g] = act[g] = gc[g] = ga[g] = zero;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = 16 * g + 4 * q + e;
          if (col < hh) {
            cnd[g][e] = xr[cond_off + 16 * g + e];
            act[g][e] = xr[act_off + 16 * g + e];
            if (grad_y && live) {
              gc[g][e] = grad_y[rowc * dim + cond_off + col];
              ga[g][e] = grad_y[rowc * dim + act_off + col];
            }
          }
        }
      } else {
        cnd[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
        act[g] = *reinterpret_cast<const f32x4*>(xr + act_off + 16 * g);
        gc[g] = (grad_y && live) ? *reinterpret_cast<const f32x4*>(grad_y + rowc * dim + 4 * q + cond_off + 16 * g) : zero;
        ga[g] = (grad_y && live) ? *reinterpret_cast<const f32x4*>(grad_y + rowc * dim + 4 * q + act_off + 16 * g) : zero;
      }
    }
    const float gl = (grad_ld && live) ? grad_ld[rowc] : 0.f;

    // SHARED_T: the layer-1 / output-layer blocks are stored with lane (i, kq) in slot 16 kq + 4 (i >> 2) + ((i + kq) & 3);
    // element (row 4 kq + r, column i) of such a block -- the transposed operand -- is then at t_off[r]
    int a_off = lane * 4, b_off = S::A_FLOATS + q * 4;
    int p_off = S::SHARED_T ? (16 * q + 4 * (j >> 2) + ((j + q) & 3)) * 4 : lane * 4;
    int t_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t_off[r] = 64 * (j >> 2) + 16 * q + 4 * ((r + (j >> 2)) & 3) + (j & 3);
    asm volatile("" : "+v"(a_off), "+v"(b_off), "+v"(p_off), "+v"(t_off[0]), "+v"(t_off[1]), "+v"(t_off[2]),
                 "+v"(t_off[3]));  // keep the operand reads inside the tile loop
    const f32x4* A4 = reinterpret_cast<const f32x4*>(lds + a_off);
    const f32x4* P4 = reinterpret_cast<const f32x4*>(lds + p_off);
    const f32x4* B4 = reinterpret_cast<const f32x4*>(lds + b_off);
    int n = 0, bt = 0, dwt = 0, dbt = 0;
    f32x4 a4;
    auto mfma_from = [&](const f32x4* base, float b, f32x4& acc) {
      if ((n & 3) == 0) a4 = base[64 * (n >> 2)];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], b, acc, 0, 0, 0);
      ++n;
    };
    auto mfma = [&](float b, f32x4& acc) { mfma_from(A4, b, acc); };
    auto mfma_w = [&](float b, f32x4& acc) { mfma_from(P4, b, acc); };  // layer-1 / output-layer weights
    auto mfma_t = [&](int group, int r, float b, f32x4& acc) {          // ... read transposed (no image order)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[group * 256 + t_off[r]], b, acc, 0, 0, 0);
    };

    // ------------------------------------------------------------------ forward recompute
    f32x4 h[3][NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      h[0][m] = B4[4 * (bt++)];
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) mfma_w(cnd[g][r], h[0][m]);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h[l][m][r] = leaky2(h[l][m][r]);
        tile_to_lds(TH + (l * NT + m) * kTileFloats, j, q, h[l][m]);
      }
      if (l < 2) {
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          h[l + 1][m] = B4[4 * (bt++)];
#pragma unroll
          for (int mt = 0; mt < NT; ++mt)
            if (S::needs(m, mt))
#pragma unroll
              for (int r = 0; r < 4; ++r) mfma(h[l][mt][r], h[l + 1][m]);
        }
      }
    }
    f32x4 st[2][G];  // raw s (net 0) and t (net 1) in the row's float4 layout
#pragma unroll
    for (int net = 0; net < 2; ++net)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        st[net][g] = B4[4 * (bt++)];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
          if ((S::tile_nets(mt) >> net) & 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) mfma_w(h[2][mt][r], st[net][g]);
      }

    // ------------------------------------------------------------------ output deltas, grad of the transformed half
    //   forward: y = e^s v + t          g_v = g e^s      g_s = g e^s v + g_ld      g_t = g
    //   inverse: y = (v - t) e^-s       g_v = g e^-s     g_s = -g y - g_ld         g_t = -g e^-s
    f32x4 d4[2][G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      f32x4 gv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = st[0][g][r], t = st[1][g][r], gy = ga[g][r], v = act[g][r];
        const float e = exp6(INV ? -s : s);
        gv[r] = gy * e;
        const bool real = live && (!RAG || 16 * g + 4 * q + r < hh);  // (a padded column has no s to collect grad_ld)
        d4[0][g][r] = real ? (INV ? -gy * ((v - t) * e) - gl : gy * e * v + gl) : 0.f;
        d4[1][g][r] = INV ? -gy * e : gy;
      }
      if constexpr (RAG) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (live && 16 * g + 4 * q + r < hh) grad_x[rowc * dim + act_off + 16 * g + 4 * q + r] = gv[r];
      } else {
        if (live) *reinterpret_cast<f32x4*>(grad_x + rowc * dim + 4 * q + act_off + 16 * g) = gv;
      }
    }

    // weight gradients of one layer: out tiles' deltas in TD[0..n_out), in tile mi with its rows on K from `load_in`;
    // (mo, mi) pairs by `used`
    auto weight_grads = [&](int n_out, int n_in, auto load_in, auto used) {
      float aop[S::D_TILES][4], bop[NT > G ? NT : G][4];
#pragma unroll
      for (int mo = 0; mo < S::D_TILES; ++mo)
        if (mo < n_out) {
          tile_rows_on_k(TD + mo * kTileFloats, j, q, aop[mo]);
          db[dbt++] += (aop[mo][0] + aop[mo][1]) + (aop[mo][2] + aop[mo][3]);
        }
#pragma unroll
      for (int mi = 0; mi < (NT > G ? NT : G); ++mi)
        if (mi < n_in) load_in(mi, bop[mi]);
#pragma unroll
      for (int mo = 0; mo < S::D_TILES; ++mo)
#pragma unroll
        for (int mi = 0; mi < (NT > G ? NT : G); ++mi)
          if (mo < n_out && mi < n_in && used(mo, mi)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              dW[dwt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[mo][s], bop[mi][s], dW[dwt], 0, 0, 0);
            ++dwt;
          }
    };

    // ------------------------------------------------------------------ output layer: dW4, then delta3 = W4^T delta4
    // (dW tile order: layer 1, hidden 1, hidden 2, output -- the flush table follows the same order, so the
    //  counters are set per layer instead of running)
    constexpr int DW_L1 = 0, DW_H1 = NT * G, DW_H2 = DW_H1 + S::PAIRS, DW_OUT = DW_H2 + S::PAIRS;
    constexpr int DB_L1 = 0, DB_H1 = NT, DB_H2 = 2 * NT, DB_OUT = 3 * NT;
    auto from_lds = [&](const float* tiles) {
      return [=](int mi, float (&o)[4]) { tile_rows_on_k(tiles + mi * kTileFloats, j, q, o); };
    };
    dwt = DW_OUT;
    dbt = DB_OUT;
#pragma unroll
    for (int net = 0; net < 2; ++net) {  // (one net's deltas in the scratch tiles at a time: s tiles, then t tiles,
      constexpr int DC = G < S::D_TILES ? G : S::D_TILES;  //  at most D_TILES of them per trip)
#pragma unroll
      for (int g0 = 0; g0 < G; g0 += DC) {
#pragma unroll
        for (int g = 0; g < DC; ++g) tile_to_lds(TD + g * kTileFloats, j, q, d4[net][g0 + g]);
        weight_grads(DC, NT, from_lds(TH + 2 * NT * kTileFloats),
                     [net](int, int mi) { return ((S::tile_nets(mi) >> net) & 1) != 0; });
      }
    }
    f32x4 dl[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      dl[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int net = 0; net < 2; ++net)
        if ((S::tile_nets(m) >> net) & 1)
#pragma unroll
          for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (S::SHARED_T)
                mfma_t(S::F4_GROUP0 + S::f4_rank(net, g, m), r, d4[net][g][r], dl[m]);
              else
                mfma(d4[net][g][r], dl[m]);
            }
    }
    // ------------------------------------------------------------------ hidden layers 3 and 2
#pragma unroll
    for (int l = 2; l >= 1; --l) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const f32x4 hv = tile_from_lds(TH + (l * NT + m) * kTileFloats, j, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) dl[m][r] = hv[r] > 0.f ? dl[m][r] : kLeakySlope * dl[m][r];
        tile_to_lds(TD + m * kTileFloats, j, q, dl[m]);
      }
      dwt = l == 2 ? DW_H2 : DW_H1;
      dbt = l == 2 ? DB_H2 : DB_H1;
      weight_grads(NT, NT, from_lds(TH + (l - 1) * NT * kTileFloats), [](int mo, int mi) { return S::needs(mo, mi); });
      f32x4 dn[NT];
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        dn[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
          if (S::needs(m, mt))
#pragma unroll
            for (int r = 0; r < 4; ++r) mfma(dl[mt][r], dn[m]);
      }
#pragma unroll
      for (int m = 0; m < NT; ++m) dl[m] = dn[m];
    }
    // ------------------------------------------------------------------ layer 1
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const f32x4 hv = tile_from_lds(TH + m * kTileFloats, j, q);
#pragma unroll
      for (int r = 0; r < 4; ++r) dl[m][r] = hv[r] > 0.f ? dl[m][r] : kLeakySlope * dl[m][r];
      tile_to_lds(TD + m * kTileFloats, j, q, dl[m]);
    }
    dwt = DW_L1;
    dbt = DB_L1;
    // layer 1's input is x0 itself, rows on K straight from global memory (element [row 4 s + q][dim 16 mi + j]: the
    // lines this wave read as row float4s a moment ago): no scratch tiles for it
    weight_grads(NT, G,
                 [&](int mi, float (&o)[4]) {
#pragma unroll
                   for (int s4 = 0; s4 < 4; ++s4) {
                     const int64_t rk = (int64_t)tile * 16 + 4 * s4 + q;
                     const int col = 16 * mi + j;
                     o[s4] = (!RAG || col < hh) ? x[(rk < rows ? rk : rows - 1) * dim + cond_off + col] : 0.f;
                   }
                 },
                 [](int, int) { return true; });
#pragma unroll
    for (int g = 0; g < G; ++g) {
      f32x4 gx0 = gc[g];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (S::SHARED_T)
            mfma_t(mt * G + g, r, dl[mt][r], gx0);
          else
            mfma(dl[mt][r], gx0);
        }
      if constexpr (RAG) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (live && 16 * g + 4 * q + r < hh) grad_x[rowc * dim + cond_off + 16 * g + 4 * q + r] = gx0[r];
      } else {
        if (live) *reinterpret_cast<f32x4*>(grad_x + rowc * dim + 4 * q + cond_off + 16 * g) = gx0;
      }
    }
  }

  // ------------------------------------------------------------------ flush: sum over the waves in LDS, one atomic per parameter
  if (grad_flat == nullptr) return;
  __syncthreads();
  float* red = lds;  // the images are no longer needed
  constexpr int DW_FLOATS = S::DW_TILES * 256;
  for (int w = 0; w < S::WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < S::DW_TILES; ++t) {
        f32x4* p = reinterpret_cast<f32x4*>(red + t * 256 + lane * 4);
        *p = w == 0 ? dW[t] : *p + dW[t];
      }
#pragma unroll
      for (int t = 0; t < S::DB_TILES; ++t) {
        float* p = red + DW_FLOATS + t * 64 + lane;
        *p = w == 0 ? db[t] : *p + db[t];
      }
    }
    __syncthreads();
  }
  // MNF_DETERMINISTIC (no tile list, list_capacity == kBwdBlocksMode): grad_flat is the workspace -- the workgroup's sums
  // go there as one block and ahf_bwd_mfma_reduce_kernel adds the blocks in order.  (Signalled through an existing
  // argument: one more kernel argument sent hipcc's AGPR-copy rewrite pass into a segmentation fault on the <128, 24>
  // ragged instantiation.)
  if (!tile_list && list_capacity == kBwdBlocksMode) {
    float* dst = grad_flat + (int64_t)blockIdx.x * (DW_FLOATS + S::DB_TILES * 16);
    for (int i = threadIdx.x; i < DW_FLOATS; i += blockDim.x) dst[i] = red[i];
    for (int i = threadIdx.x; i < S::DB_TILES * 16; i += blockDim.x) {
      const float* p = red + DW_FLOATS + (i >> 4) * 64 + (i & 15);
      dst[DW_FLOATS + i] = (p[0] + p[16]) + (p[32] + p[48]);
    }
    return;
  }
  const int32_t* flush_w = index + S::IMAGE_FLOATS;
  const int32_t* flush_b = flush_w + DW_FLOATS;
  for (int i0 = threadIdx.x; i0 < DW_FLOATS; i0 += 4 * blockDim.x) {
    int32_t dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      dst[u] = i < DW_FLOATS ? flush_w[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dst[u] >= 0) atomicAdd(grad_flat + dst[u], red[i0 + u * (int)blockDim.x]);
  }
  // db: lane (i = unit, kq) holds a quarter of the rows' sum
  for (int i = threadIdx.x; i < S::DB_TILES * 16; i += blockDim.x) {
    const int t = i >> 4, u = i & 15;
    const int32_t dst = flush_b[i];
    if (dst >= 0) {
      const float* p = red + DW_FLOATS + t * 64 + u;
      atomicAdd(grad_flat + dst, (p[0] + p[16]) + (p[32] + p[48]));
    }
  }
}

// ---------------------------------------------------------------- host: index table
// rh: the real half width (<= H; columns rh .. H-1 of the tile are padding: -1 = zero operand, no flush);
// hs: the three real hidden widths (<= HID each; the unit slots above them are structural zeros the same way)
// has_s / has_t: a NICE-style layer has only one of the two nets (affine_half_flow.py:38: the other is the zero
// function): the absent net's operands, biases and flush entries all stay -1, so it computes s = 0 (or t = 0) and
// receives no gradient
template <int H, int HID>
static void build_bwd_index(int32_t* idx, int rh, const int* hs, bool has_s, bool has_t) {
  using S = BwdShape<H, HID>;
  constexpr int G = S::G, NT = S::NT;
  int sizes[5] = {rh, hs[0], hs[1], hs[2], rh};
  NetDesc net[2];
  const int64_t off = fill_net(net[0], 5, sizes, 0);
  fill_net(net[1], 5, sizes, has_s ? off : 0);  // flat = [s_net if present][t_net if present]
  const bool present[2] = {has_s, has_t};
  for (int i = 0; i < S::INDEX_INTS; ++i) idx[i] = -1;
  auto netof = [&](int u) { return u / HID; };
  // slot u of the concatenated net is a real unit of hidden layer `layer` (0, 1, 2)
  auto valid = [&](int u, int layer) { return u < 2 * HID && u % HID < hs[layer] && present[u / HID]; };
  int n = 0;
  auto put = [&](int lane, int32_t src) { idx[(n >> 2) * 256 + lane * 4 + (n & 3)] = src; };
  // layer-1 / output-layer blocks: with SHARED_T lane (i, kq) sits in slot 16 kq + 4 (i >> 2) + ((i + kq) & 3), which
  // keeps the forward float4 reads and the transposed single-float reads of the backward products conflict free
  auto put_perm = [&](int lane, int32_t src) {
    const int i = lane & 15, kq = lane >> 4;
    put(S::SHARED_T ? 16 * kq + 4 * (i >> 2) + ((i + kq) & 3) : lane, src);
  };
  // forward operands: A[i][kq] = W[out unit of row i][input behind K slot kq of step r]
  for (int m = 0; m < NT; ++m)
    for (int g = 0; g < G; ++g)
      for (int r = 0; r < 4; ++r, ++n)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
          if (valid(u, 0) && 16 * g + 4 * kq + r < rh)
            put_perm(lane, net[netof(u)].w_off[0] + (u % HID) * rh + 16 * g + 4 * kq + r);
        }
  for (int l = 1; l <= 2; ++l)
    for (int m = 0; m < NT; ++m)
      for (int mt = 0; mt < NT; ++mt) {
        if (!S::needs(m, mt)) continue;
        for (int r = 0; r < 4; ++r, ++n)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, uo = 16 * m + i, ui = 16 * mt + 4 * kq + r;
            if (valid(uo, l) && valid(ui, l - 1) && netof(uo) == netof(ui))
              put(lane, net[netof(uo)].w_off[l] + (uo % HID) * hs[l - 1] + ui % HID);
          }
      }
  for (int nn = 0; nn < 2; ++nn)
    for (int g = 0; g < G; ++g)
      for (int mt = 0; mt < NT; ++mt) {
        if (!((S::tile_nets(mt) >> nn) & 1)) continue;
        for (int r = 0; r < 4; ++r, ++n)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, ui = 16 * mt + 4 * kq + r;
            if (valid(ui, 2) && netof(ui) == nn && 16 * g + i < rh)
              put_perm(lane, net[nn].w_off[3] + (16 * g + i) * hs[2] + ui % HID);
          }
      }
  // transposed operands (the layer-1 and output-layer ones only without SHARED_T)
  for (int m = 0; m < NT; ++m)  // delta3[unit 16 m + i] += W4[dim 16 g + 4 kq + r][unit] delta4[dim]
    for (int nn = 0; nn < 2; ++nn) {
      if (S::SHARED_T || !((S::tile_nets(m) >> nn) & 1)) continue;
      for (int g = 0; g < G; ++g)
        for (int r = 0; r < 4; ++r, ++n)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
            if (valid(u, 2) && netof(u) == nn && 16 * g + 4 * kq + r < rh)
              put(lane, net[nn].w_off[3] + (16 * g + 4 * kq + r) * hs[2] + u % HID);
          }
    }
  for (int l = 2; l >= 1; --l)  // delta_{l}[unit 16 m + i] += W_{l+1}... here: W_l[out 16 mt + 4 kq + r][in unit]
    for (int m = 0; m < NT; ++m)
      for (int mt = 0; mt < NT; ++mt) {
        if (!S::needs(m, mt)) continue;
        for (int r = 0; r < 4; ++r, ++n)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4, ui = 16 * m + i, uo = 16 * mt + 4 * kq + r;
            if (valid(uo, l) && valid(ui, l - 1) && netof(uo) == netof(ui))
              put(lane, net[netof(uo)].w_off[l] + (uo % HID) * hs[l - 1] + ui % HID);
          }
      }
  for (int g = 0; g < (S::SHARED_T ? 0 : G); ++g)  // grad x0[dim 16 g + i] += W1[unit 16 mt + 4 kq + r][dim] delta1[unit]
    for (int mt = 0; mt < NT; ++mt)
      for (int r = 0; r < 4; ++r, ++n)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = 16 * mt + 4 * kq + r;
          if (valid(u, 0) && 16 * g + i < rh) put(lane, net[netof(u)].w_off[0] + (u % HID) * rh + 16 * g + i);
        }
  // biases of the forward recompute
  int32_t* b = idx + S::A_FLOATS;
  int bt = 0;
  for (int l = 0; l < 3; ++l)
    for (int m = 0; m < NT; ++m, ++bt)
      for (int i = 0; i < 16; ++i)
        if (valid(16 * m + i, l)) b[bt * 16 + i] = net[netof(16 * m + i)].b_off[l] + (16 * m + i) % HID;
  for (int nn = 0; nn < 2; ++nn)
    for (int g = 0; g < G; ++g, ++bt)
      for (int i = 0; i < 16; ++i)
        if (16 * g + i < rh && present[nn]) b[bt * 16 + i] = net[nn].b_off[3] + 16 * g + i;

  // flush tables.  dW tile (mo, mi): lane (j = in index b, q), reg r <-> out index a = 4 q + r
  int32_t* fw = idx + S::IMAGE_FLOATS;
  int t = 0;
  auto put_w = [&](int lane, int r, int32_t dst) { fw[t * 256 + lane * 4 + r] = dst; };
  for (int mo = 0; mo < NT; ++mo)  // layer 1: W1[unit 16 mo + a][dim 16 mi + b]
    for (int mi = 0; mi < G; ++mi, ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int bq = lane & 15, a = 4 * (lane >> 4) + r, u = 16 * mo + a;
          if (valid(u, 0) && 16 * mi + bq < rh) put_w(lane, r, net[netof(u)].w_off[0] + (u % HID) * rh + 16 * mi + bq);
        }
  for (int l = 1; l <= 2; ++l)
    for (int mo = 0; mo < NT; ++mo)
      for (int mi = 0; mi < NT; ++mi) {
        if (!S::needs(mo, mi)) continue;
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int bq = lane & 15, a = 4 * (lane >> 4) + r, uo = 16 * mo + a, ui = 16 * mi + bq;
            if (valid(uo, l) && valid(ui, l - 1) && netof(uo) == netof(ui))
              put_w(lane, r, net[netof(uo)].w_off[l] + (uo % HID) * hs[l - 1] + ui % HID);
          }
        ++t;
      }
  for (int mo = 0; mo < 2 * G; ++mo)  // output: delta tile mo = net * G + g
    for (int mi = 0; mi < NT; ++mi) {
      const int nn = mo / G, g = mo % G;
      if (!((S::tile_nets(mi) >> nn) & 1)) continue;
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int bq = lane & 15, a = 4 * (lane >> 4) + r, ui = 16 * mi + bq;
          if (valid(ui, 2) && netof(ui) == nn && 16 * g + a < rh) put_w(lane, r, net[nn].w_off[3] + (16 * g + a) * hs[2] + ui % HID);
        }
      ++t;
    }
  int32_t* fb = fw + S::DW_TILES * 256;
  t = 0;
  for (int l = 0; l < 3; ++l)
    for (int m = 0; m < NT; ++m, ++t)
      for (int i = 0; i < 16; ++i)
        if (valid(16 * m + i, l)) fb[t * 16 + i] = net[netof(16 * m + i)].b_off[l] + (16 * m + i) % HID;
  for (int nn = 0; nn < 2; ++nn)
    for (int g = 0; g < G; ++g, ++t)
      for (int i = 0; i < 16; ++i)
        if (16 * g + i < rh && present[nn]) fb[t * 16 + i] = net[nn].b_off[3] + 16 * g + i;
}

// shapes: 16, 24 or 32 hidden-unit slots at tile halves 16, 32 (d <= 64; narrower halves and layers padded), 24 at 64, 128
#define MNF_AHF_BWD_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16) X(64, 24) X(16, 32) X(32, 32) X(128, 24)

// second stage of the deterministic flush: entry i of the flush tables (weights, then biases) += the blocks' entries i, in
// block order; every parameter appears in exactly one entry
__global__ void __launch_bounds__(256)
ahf_bwd_mfma_reduce_kernel(const float* __restrict__ partials, int n_blocks, int red_floats,
                           const int32_t* __restrict__ flush, float* __restrict__ grad_flat) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= red_floats) return;
  const int32_t dst = flush[i];
  if (dst < 0) return;
  float s = 0.f;
  for (int b = 0; b < n_blocks; ++b) s += partials[(int64_t)b * red_floats + i];
  grad_flat[dst] += s;
}

template <int H, int HID>
static int64_t bwd_blocks(int64_t rows) {
  using S = BwdShape<H, HID>;
  const int cus = device_cus(current_device());
  const int64_t n_tiles = (rows + 15) / 16, blocks = (n_tiles + S::WAVES - 1) / S::WAVES;
  return blocks > cus ? cus : blocks;
}

template <int H, int HID, bool RAG>
static int launch_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                      const float* flat, const int32_t* index, int64_t rows, int parity, int inverse,
                      const int32_t* tile_list, int list_capacity, int real_h, hipStream_t stream,
                      float* partials = nullptr) {
  using S = BwdShape<H, HID>;
  constexpr size_t lds_bytes = S::LDS_FLOATS * sizeof(float);
  static DeviceMemo memo;  // per device: CU count once the dynamic-LDS attribute is set there, -1 if it cannot be
  const int cus = memo.get([](int dev) {
    const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(ahf_bwd_mfma_kernel<H, HID, true, RAG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(ahf_bwd_mfma_kernel<H, HID, false, RAG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess;
    return ok ? device_cus(dev) : -1;
  });
  if (cus <= 0) return MNF_ERR_UNSUPPORTED;
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + S::WAVES - 1) / S::WAVES;
  if (blocks > cus) blocks = cus;  // one persistent workgroup per CU (LDS bound)
  // (a fix-up pass gets the full grid too: the list is usually empty and every workgroup returns at once, but it may
  //  also name every tile)
  // MNF_DETERMINISTIC: the list in ascending order and ONE workgroup -- its waves take the tiles in list order and add up
  // in wave order (mnf_host.h det_sort_ids_async)
  if (tile_list && !partials && deterministic() &&
      det_sort_ids_async(const_cast<int32_t*>(tile_list) + 1, tile_list, list_capacity, n_tiles, stream) == MNF_OK)
    blocks = 1;
  const dim3 grid((unsigned)blocks), block(S::WAVES * 64);
  if (!tile_list) tag_kernel("ahf_bwd_mfma_fp32");  // (as the split kernel's fix-up pass it keeps that kernel's name)
  if (inverse)
    hipLaunchKernelGGL((ahf_bwd_mfma_kernel<H, HID, true, RAG>), grid, block, lds_bytes, stream, x, grad_y, grad_ld,
                       grad_x, partials ? partials : grad_flat, flat, index, rows, parity, tile_list,
                       partials ? kBwdBlocksMode : list_capacity, real_h);
  else
    hipLaunchKernelGGL((ahf_bwd_mfma_kernel<H, HID, false, RAG>), grid, block, lds_bytes, stream, x, grad_y, grad_ld,
                       grad_x, partials ? partials : grad_flat, flat, index, rows, parity, tile_list,
                       partials ? kBwdBlocksMode : list_capacity, real_h);
  if (int rc = check_launch()) return rc;
  if (!partials || !grad_flat) return MNF_OK;
  constexpr int RED = S::DW_TILES * 256 + S::DB_TILES * 16;
  hipLaunchKernelGGL(ahf_bwd_mfma_reduce_kernel, dim3((RED + 255) / 256), dim3(256), 0, stream, partials, (int)blocks, RED,
                     index + S::IMAGE_FLOATS, grad_flat);
  return check_launch();
}

// the tile half width a layer of `dim` columns runs at (0: none): halves narrower than a tile are padded
static int bwd_padded_half(int dim) {
  if (dim < 2 || (dim & 1)) return 0;
  const int h = dim / 2;
  return h <= 16 ? 16 : h <= 32 ? 32 : h <= 64 ? 64 : h <= 128 ? 128 : 0;
}

// the hidden width the kernel runs three hidden layers of widths hidden[0..2] at: 16, 24 or 32, whichever holds the widest
// (narrower layers get structural-zero units: zero operands, LeakyReLU(0) = 0, no flush); false: none
static bool bwd_uniform3(int n_hidden, const int* hidden, int& hid) {
  if (n_hidden != 3 || !hidden) return false;
  int mx = 0;
  for (int i = 0; i < 3; ++i) mx = hidden[i] > mx ? hidden[i] : mx;
  hid = mx <= 16 ? 16 : mx <= 24 ? 24 : mx <= 32 ? 32 : 0;
  return hid != 0;
}

}  // namespace mnf

extern "C" {

int64_t mnf_affine_half_bwd_index_ints(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift) {
  int hid = 0;
  if ((!has_scale && !has_shift) || !mnf::hidden_ok(n_hidden, hidden) || !mnf::bwd_uniform3(n_hidden, hidden, hid))
    return 0;
  const int ph = mnf::bwd_padded_half(dim);
  if (ph >= 64) hid = hid <= 24 ? 24 : 0;  // (the 64- and 128-column tiles exist with 24 hidden units only)
#define X(HH, HD) \
  if (ph == HH && hid == HD) return mnf::BwdShape<HH, HD>::INDEX_INTS;
  MNF_AHF_BWD_SHAPES(X)
#undef X
  return 0;
}

int mnf_affine_half_bwd_index(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                              int32_t* idx_host) {
  int hid = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if ((!has_scale && !has_shift) || !mnf::bwd_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  const int ph = mnf::bwd_padded_half(dim);
  if (ph >= 64) hid = hid <= 24 ? 24 : 0;  // (the 64- and 128-column tiles exist with 24 hidden units only)
#define X(HH, HD)                                       \
  if (ph == HH && hid == HD) {                          \
    mnf::build_bwd_index<HH, HD>(idx_host, dim / 2, hidden, has_scale != 0, has_shift != 0);    \
    return MNF_OK;                                      \
  }
  MNF_AHF_BWD_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_affine_half_bwd_mfma(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                             float* grad_flat, const float* flat, const int32_t* index_dev, int64_t rows, int dim,
                             int parity, int inverse, int n_hidden, const int* hidden, void* stream) {
  return mnf_affine_half_bwd_mfma_tiles(x, grad_y, grad_ld, grad_x, grad_flat, flat, index_dev, rows, dim, parity,
                                        inverse, n_hidden, hidden, nullptr, 0, stream);
}

int mnf_affine_half_bwd_mfma_tiles(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                                   float* grad_flat, const float* flat, const int32_t* index_dev, int64_t rows, int dim,
                                   int parity, int inverse, int n_hidden, const int* hidden, const int32_t* tile_list,
                                   int list_capacity, void* stream) {
  int hid = 0;
  if (list_capacity < 0) return MNF_ERR_INVALID_ARG;
  if (!x || !grad_x || !flat || !index_dev || rows < 0 || dim < 2 || (dim & 1) || !mnf::hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (!mnf::bwd_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  const int ph = mnf::bwd_padded_half(dim);
  if (ph >= 64) hid = hid <= 24 ? 24 : 0;  // (the 64- and 128-column tiles exist with 24 hidden units only)
  const bool ragged = ph != dim / 2;  // (element-wise row accesses: no alignment condition)
  if (!ragged &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) & 15))
    return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                                                                      \
  if (ph == HH && hid == HD)                                                                                           \
    return ragged ? mnf::launch_bwd<HH, HD, true>(x, grad_y, grad_ld, grad_x, grad_flat, flat, index_dev, rows,       \
                                                  parity != 0, inverse != 0, tile_list, list_capacity, dim / 2,        \
                                                  (hipStream_t)stream)                                                 \
                  : mnf::launch_bwd<HH, HD, false>(x, grad_y, grad_ld, grad_x, grad_flat, flat, index_dev, rows,      \
                                                   parity != 0, inverse != 0, tile_list, list_capacity, dim / 2,       \
                                                   (hipStream_t)stream);
  MNF_AHF_BWD_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int64_t mnf_affine_half_bwd_mfma_workspace(int64_t rows, int dim, int n_hidden, const int* hidden) {
  int hid = 0;
  if (rows < 1 || dim < 2 || (dim & 1) || !mnf::hidden_ok(n_hidden, hidden) || !mnf::bwd_uniform3(n_hidden, hidden, hid))
    return 0;
  const int ph = mnf::bwd_padded_half(dim);
  if (ph >= 64) hid = hid <= 24 ? 24 : 0;
#define X(HH, HD) \
  if (ph == HH && hid == HD) \
    return mnf::bwd_blocks<HH, HD>(rows) * (mnf::BwdShape<HH, HD>::DW_TILES * 256 + mnf::BwdShape<HH, HD>::DB_TILES * 16);
  MNF_AHF_BWD_SHAPES(X)
#undef X
  return 0;
}

int mnf_affine_half_bwd_mfma_det(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                                 float* grad_flat, const float* flat, const int32_t* index_dev, int64_t rows, int dim,
                                 int parity, int inverse, int n_hidden, const int* hidden, float* workspace,
                                 int64_t workspace_floats, void* stream) {
  int hid = 0;
  if (!x || !grad_x || !flat || !index_dev || !workspace || rows < 0 || dim < 2 || (dim & 1) ||
      !mnf::hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  if (!mnf::bwd_uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  const int64_t need = mnf_affine_half_bwd_mfma_workspace(rows, dim, n_hidden, hidden);
  if (need == 0) return MNF_ERR_UNSUPPORTED;
  if (workspace_floats < need) return MNF_ERR_INVALID_ARG;
  const int ph = mnf::bwd_padded_half(dim);
  if (ph >= 64) hid = hid <= 24 ? 24 : 0;
  const bool ragged = ph != dim / 2;
  if (!ragged &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) & 15))
    return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                                                                       \
  if (ph == HH && hid == HD)                                                                                            \
    return ragged ? mnf::launch_bwd<HH, HD, true>(x, grad_y, grad_ld, grad_x, grad_flat, flat, index_dev, rows,        \
                                                  parity != 0, inverse != 0, nullptr, 0, dim / 2, (hipStream_t)stream, \
                                                  workspace)                                                            \
                  : mnf::launch_bwd<HH, HD, false>(x, grad_y, grad_ld, grad_x, grad_flat, flat, index_dev, rows,       \
                                                   parity != 0, inverse != 0, nullptr, 0, dim / 2, (hipStream_t)stream,\
                                                   workspace);
  MNF_AHF_BWD_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
