// Gradients of MNFLinear.forward (layers/mnf_linear.py:46-56 under loss.backward(); what tests/test_mnf_mnist.py:14-56
// and examples/mnf_mnist.ipynb train through) on the f16 matrix pipe in split (hi + lo) fp32 arithmetic, gfx950.
//
//   forward   mean = (x z) Wm^T + bm;   var = x^2 Wv^T + bv   (Wv = exp(W_log_var), bv = exp(b_log_var));
//             out = mean + sd eps,  sd = sqrt(var)
//   backward  g_m = G;   g_v = G eps / (2 sd)                                         [G = grad_out, (rows, n_out)]
//             a = g_m Wm,  b = g_v Wv   (rows, n_in);   grad_x = z a + 2 x b;   grad_z = x a
//             dWm = g_m^T (x z);   dW_log_var = Wv (.) g_v^T x^2;   dbm = sum g_m;   db_log_var = bv sum g_v
//
// The same two-launch shape as the RNVP gradient kernels (mnf_rnvp_bwd.hip), with the roles of "hidden units" played
// by the n_out <= 64 outputs: the per-row vectors g_m and g_v are small, the weight gradients are 2 n_out n_in sums over
// all rows.  A row-parallel PROLOGUE turns G, eps and the saved sd into g_m, g_v as ready-made split MFMA operands in
// both orientations (HandoverShape: 1 KB per row at n_out = 50) and sums the bias gradients; the SLAB launch gives a
// workgroup 32 input dims and a range of rows: per 32 rows a, b (K = outputs), grad_x and grad_z (the only row data
// written; x and z are read once), and dWm, dWv as K = 32-ROW products into 4 YT accumulator tiles that stay in
// registers over the whole row range.  Everything in the slab launch is computed transposed (rows on the MFMA M axis),
// so an accumulator holds four ROWS of one dim per lane -- the operand layout of a sum over rows; rows are read and
// written as 8-byte pieces (dims 2 j, 2 j + 1 of the slab).
//
// The image's exp(W_log_var) carries the power of two of the forward image; `var_unscale` undoes it on the way OUT of
// every product with it (folded into the operand g_v it would push that operand to ~1e-4, where the unscaled f16
// residuals of the row sums have no bits left).  G takes the gradient scale on the way in (mnf_affine_half_grad_scale:
// cotangents of a mean are ~1 / rows, below f16's normal range), undone on the way out.  128-row groups
// the FORWARD pass flagged (|x z| or x^2 beyond the split range, or weights beyond the weight limit) and groups whose
// cotangents leave the range are skipped by the slab launch and redone in fp32 by ml_bwd_fixup_kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_split.h"

namespace mnf {

constexpr int kMlbGroupRows = 128;  // rows per flag: the forward kernel's 8-wave group
constexpr int kMlbWaves = 4;        // waves of a slab workgroup
// MNF_DETERMINISTIC (mnf_host.h): the prologue's workgroups (at most kMlDetBlocks) leave 128 floats each, the slab
// launch's row parts (at most kMlDetParts) a block of the layer's parameter count each, behind the workspace's tiles
constexpr int kMlDetBlocks = 1024, kMlDetParts = 64;
__host__ __device__ inline int64_t ml_params(int n_in, int n_out) { return 2 * (int64_t)n_out * n_in + 2 * n_out; }

template <int YT>
struct MlBwdShape {
  using H = HandoverShape<YT>;
  static constexpr int NKS = H::NKS;
  // backward image, per 32-dim slab: (dim tile E/O) x (Wm, Wv') x NKS x (hi, lo) B operands with the outputs on K
  static constexpr int SLAB_WORDS = 2 * 2 * NKS * 512;
  static constexpr int64_t n_slabs(int n_in) { return (n_in + 31) / 32; }
  static constexpr int64_t split_words(int n_in) { return n_slabs(n_in) * SLAB_WORDS; }
};

// ------------------------------------------------------------------------------------------------ prologue
// one 8-wave workgroup per 128-row group (the forward pass's flag granularity), a wave per 16-row tile
template <int YT>
__global__ void __launch_bounds__(8 * 64)
ml_bwd_prologue_kernel(const float* __restrict__ gout, const float* __restrict__ sd, const float* __restrict__ eps,
                       uint64_t seed, const float* __restrict__ flat, const int32_t* __restrict__ fwd_flags,
                       uint32_t* __restrict__ side, int32_t* __restrict__ flags, int32_t* __restrict__ list,
                       const float* __restrict__ gscale_dev, float* __restrict__ grad_flat, int64_t rows, int n_in,
                       int n_out, float var_unscale, float wmax_limit_ok, float* __restrict__ det_b) {
  using H = HandoverShape<YT>;
  constexpr int kPitch = 16 * YT + 1;  // (odd: the 16 rows a lane group reads land on 16 different banks)
  __shared__ float stage[8 * 2 * 16 * kPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float gscale = gscale_dev[0];
  const int n_groups = (int)((rows + kMlbGroupRows - 1) / kMlbGroupRows);
  f32x4 bm_acc[YT], bv_acc[YT];
#pragma unroll
  for (int m = 0; m < YT; ++m) bm_acc[m] = bv_acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t tile = (int64_t)grp * 8 + wave;
    const int64_t row = tile * 16 + j;
    const bool live = row < rows;
    // the tile's 16 x n_out cotangents, sd and noise are contiguous in memory: read them coalesced, form g_m and g_v per
    // element, and turn them into the (row, 4 outputs per lane) layout through LDS (read per lane as 4-byte pieces with
    // a 200-byte row stride they took 0.55 ms per 256,000 rows)
    float* tm_ = stage + wave * (2 * 16 * kPitch);
    float* tv_ = tm_ + 16 * kPitch;
    {
      // (every load of the tile requested up front, without a branch -- elements past the tile or the last row read a
      //  clamped address and are zeroed by a select: as a loop of guarded loads each of the ~13 passes waited for its own
      //  three loads, 180 us per 256,000 x 50 launch)
      constexpr int IT = 4 * YT;  // ceil(16 n_out / 64) at n_out = 16 YT
      const int64_t base = tile * 16 * n_out, last = rows * n_out - 1;
      const int n_el = (int)min((int64_t)16 * n_out, max((int64_t)0, rows * n_out - base));
      const float* eps_or_g = eps ? eps : gout;
      float gl[IT], sl[IT], el[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int64_t at = min(base + lane + 64 * it, last);
        gl[it] = gout[at];
        sl[it] = sd[at];
        el[it] = eps_or_g[at];
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int i = lane + 64 * it;
        const int r_ = i / n_out, o = i - r_ * n_out;
        const bool on = i < n_el;
        float e = el[it];
        if (!eps) e = ml_normal(seed, tile * 16 + r_, o);  // (wave-uniform; no memory operation inside)
        const float g = on ? gl[it] : 0.f;
        const float v = on ? g * e / (2.f * sl[it]) : 0.f;  // d out / d var = eps / (2 sqrt(var))   (mnf_linear.py:56)
        if (i < 16 * n_out) {
          tm_[r_ * kPitch + o] = g;
          tv_[r_ * kPitch + o] = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the wave's own LDS writes, read back by other lanes of the same wave below)
    f32x4 gm[YT], gv[YT], gvt[YT];
    u32x2 mh[YT], ml[YT], vh[YT], vl[YT];
    float mx = wmax_limit_ok > 0.f ? 0.f : __builtin_inff();
#pragma unroll
    for (int m = 0; m < YT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * m + 4 * q + r;
        const bool in = live && o < n_out;
        const float g = in ? tm_[j * kPitch + o] : 0.f, v = in ? tv_[j * kPitch + o] : 0.f;
        gm[m][r] = g * gscale;
        gvt[m][r] = v;
        gv[m][r] = v * gscale;  // (O(1) like g_m: the unscaled residuals of the row sums need operands near 1)
      }
      split_tile(gm[m], mh[m], ml[m], mx);
      split_tile(gv[m], vh[m], vl[m], mx);
    }
    const bool bad = __syncthreads_or(!(mx <= kSplitLimit) ? 1 : 0) != 0 || fwd_flags[grp] != 0;
    if (threadIdx.x == 0) {
      flags[grp] = bad ? 1 : 0;
      if (bad) list[1 + atomicAdd(list, 1)] = grp;  // (list[0] zeroed by the launcher)
    }
    __builtin_amdgcn_wave_barrier();  // (the next group overwrites the staging area)
    if (bad || tile * 16 >= rows) continue;
#pragma unroll
    for (int m = 0; m < YT; ++m) {
      bm_acc[m] += gm[m];
      bv_acc[m] += gvt[m];
    }
    store_handover<YT>(side + tile * H::TILE_WORDS, mh, ml, vh, vl, lane, j, q);
  }
  // db_mean, db_log_var = exp(b_log_var) sum g_v: sums over the wave's rows (the 16 lanes j of a q), over the
  // workgroup's waves in LDS, then one atomic per output per workgroup (one per output per WAVE -- 410,000 atomics on
  // 100 addresses -- serialised at the memory side for 0.5 ms)
  if (grad_flat) {
    const float inv = 1.f / gscale;
    const int64_t bmo = 2 * (int64_t)n_out * n_in, bvo = bmo + n_out;
    float* bsum = stage;  // [0, 64): mean, [64, 128): var
    __syncthreads();
    if (threadIdx.x < 128) bsum[threadIdx.x] = 0.f;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < YT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = bm_acc[m][r], b = bv_acc[m][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          a += __shfl_xor(a, off, 64);
          b += __shfl_xor(b, off, 64);
        }
        bm_acc[m][r] = a;
        bv_acc[m][r] = b;
      }
    const bool det = det_b != nullptr;  // (the waves add in turn, the workgroup's sums go out as its own block of 128)
    lds_wave_add<8>(det, wave, [&](auto op) {
#pragma unroll
      for (int m = 0; m < YT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j == 0) {
            op(bsum + 16 * m + 4 * q + r, bm_acc[m][r]);
            op(bsum + 64 + 16 * m + 4 * q + r, bv_acc[m][r]);
          }
    });
    __syncthreads();
    if ((int)threadIdx.x < n_out) {
      const int o = threadIdx.x;
      const float gm_ = bsum[o] * inv, gv_ = bsum[64 + o] * flat[bvo + o];
      if (det) {
        det_b[(int64_t)blockIdx.x * 128 + o] = gm_;
        det_b[(int64_t)blockIdx.x * 128 + 64 + o] = gv_;
      } else {
        if (bsum[o] != 0.f) atomicAdd(grad_flat + bmo + o, gm_);
        if (bsum[64 + o] != 0.f) atomicAdd(grad_flat + bvo + o, gv_);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ slab launch
// lane (c, q): column c of a 16-column dim tile, rows 4 q .. 4 q + 3 of a 16-row tile in its registers
template <int YT, bool RAG>
__global__ void __launch_bounds__(kMlbWaves * 64, 2)
ml_bwd_slab_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ grad_x,
                   float* __restrict__ grad_z, float* __restrict__ grad_flat, const float* __restrict__ flat,
                   const uint32_t* __restrict__ bimage, const uint32_t* __restrict__ side, const int32_t* __restrict__ flags,
                   const float* __restrict__ gscale_dev, int64_t rows, int n_in, int n_out, int n_slabs, int row_parts,
                   int vec2, float var_unscale, float* __restrict__ det_part) {
  using S = MlBwdShape<YT>;
  using H = HandoverShape<YT>;
  constexpr int NKS = S::NKS;
  constexpr int UP = 33;  // padded row of the flush area ([output][dim of the slab])
  constexpr int LDS_WORDS = S::SLAB_WORDS > 2 * 16 * YT * UP ? S::SLAB_WORDS : 2 * 16 * YT * UP;
  __shared__ __attribute__((aligned(16))) uint32_t w_lds[LDS_WORDS];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const float inv_gscale = 1.f / gscale_dev[0];
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};
  const int64_t n_tiles = (rows + 15) / 16, n_pairs = (n_tiles + 1) / 2;
  const int64_t per_part = (n_pairs + row_parts - 1) / row_parts;
  const SlabItems items(n_slabs, row_parts);
  for (int item = items.first; item < items.n_items; item += items.step) {
    const int slab = items.slab(item), part = items.part(item);
    __syncthreads();  // the previous item's flush area is no longer read
    {
      const uint32_t* src = bimage + (int64_t)slab * S::SLAB_WORDS;
      for (int i = threadIdx.x; i < S::SLAB_WORDS / 4; i += blockDim.x)
        reinterpret_cast<uint4*>(w_lds)[i] = reinterpret_cast<const uint4*>(src)[i];
    }
    __syncthreads();
    const int dim0 = 32 * slab + 2 * j;  // the lane's even dim; + 1: its odd dim
    const bool in0 = dim0 < n_in, in1 = dim0 + 1 < n_in;
    // operand numbering in LDS: [(dt * 2 + which) * NKS + ks][part], which = 0: W_mean, 1: exp(W_log_var) (scaled)
    int w_lane = lane;  // (opaque, refreshed per pair: keeps the operand reads inside the pair loop)
    const f16x8* W8 = reinterpret_cast<const f16x8*>(w_lds);
    auto wop = [&](int dt, int which, int ks, int part_) { return W8[w_lane + 64 * (2 * ((dt * 2 + which) * NKS + ks) + part_)]; };
    const uint32_t lane_off = (uint32_t)(4 * q) * (uint32_t)n_in + (uint32_t)dim0;

    f32x4 aWm[2][YT], aWv[2][YT];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int m = 0; m < YT; ++m) aWm[dt][m] = aWv[dt][m] = zero4;

    const int64_t p_end = min(n_pairs, (int64_t)(part + 1) * per_part);
    for (int64_t p = (int64_t)part * per_part + wave; p < p_end; p += kMlbWaves) {
      if (flags[(p * 32) / kMlbGroupRows]) continue;  // the fp32 kernel redoes flagged groups
      asm volatile("" : "+v"(w_lane));
      const bool has1 = 2 * p + 1 < n_tiles;
      u32x2 ph[2][2], pl[2][2], sh[2][2], sl[2][2];  // [row tile][dim tile]: x z and x^2 as B operands of the row sums
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const bool has = tt == 0 || has1;
        const int64_t tbase = (has ? 2 * p + tt : 2 * p) * 16;              // wave-uniform
        const int n_live = has ? (int)min((int64_t)16, rows - tbase) : 0;  // rows of the tile that exist
        const float* xt = x + tbase * n_in;
        const float* zt = z + tbase * n_in;
        const uint32_t* sd_ = side + (tbase >> 4) * H::TILE_WORDS;
        f32x2 xx[4], zz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool live = 4 * q + r < n_live;
          const uint32_t off = live ? lane_off + (uint32_t)r * (uint32_t)n_in : (uint32_t)dim0;
          f32x2 xv = {0.f, 0.f}, zv = {0.f, 0.f};
          if (!RAG) {
            xv = *reinterpret_cast<const f32x2*>(xt + off);
            zv = *reinterpret_cast<const f32x2*>(zt + off);
          } else if (vec2) {
            if (in0) {
              xv = *reinterpret_cast<const f32x2*>(xt + off);
              zv = *reinterpret_cast<const f32x2*>(zt + off);
            }
          } else {
            if (in0) {
              xv[0] = xt[off];
              zv[0] = zt[off];
            }
            if (in1) {
              xv[1] = xt[off + 1];
              zv[1] = zt[off + 1];
            }
          }
          const float keep = live ? 1.f : 0.f;  // (the cotangents of a row past the end are zero; keep its x z finite)
          xx[r] = xv * keep;
          zz[r] = zv * keep;
        }
        f16x8 moh[NKS], mol[NKS], voh[NKS], vol[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          moh[ks] = *reinterpret_cast<const f16x8*>(sd_ + H::A_OP + ((2 * ks) * 64 + lane) * 4);
          mol[ks] = *reinterpret_cast<const f16x8*>(sd_ + H::A_OP + ((2 * ks + 1) * 64 + lane) * 4);
          voh[ks] = *reinterpret_cast<const f16x8*>(sd_ + H::B_OP + ((2 * ks) * 64 + lane) * 4);
          vol[ks] = *reinterpret_cast<const f16x8*>(sd_ + H::B_OP + ((2 * ks + 1) * 64 + lane) * 4);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          // a^T, b^T [row][dim] = g [row][output] W [output][dim]
          f32x4 am = zero4, ac = zero4, bm = zero4, bc = zero4;
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            split_mac(moh[ks], mol[ks], wop(dt, 0, ks, 0), wop(dt, 0, ks, 1), am, ac);
            split_mac(voh[ks], vol[ks], wop(dt, 1, ks, 0), wop(dt, 1, ks, 1), bm, bc);
          }
          const f32x4 a4 = (ac * kSplitInvScale + am) * inv_gscale;
          const f32x4 b4 = (bc * kSplitInvScale + bm) * (inv_gscale * var_unscale);
          f32x4 pz, x2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xv = xx[r][dt], zv = zz[r][dt];
            pz[r] = xv * zv;                                   // the forward pass's operands (mnf_linear.py:48,53)
            x2[r] = xv * xv;
            xx[r][dt] = zv * a4[r] + 2.f * (xv * b4[r]);       // grad_x takes x's register, grad_z z's
            zz[r][dt] = xv * a4[r];
          }
          split_plain(pz, ph[tt][dt], pl[tt][dt]);
          split_plain(x2, sh[tt][dt], sl[tt][dt]);
        }
        if (has) {
          float* ox = grad_x + tbase * n_in;
          float* oz = grad_z + tbase * n_in;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (4 * q + r < n_live) {
              const uint32_t off = lane_off + (uint32_t)r * (uint32_t)n_in;
              if (!RAG) {
                *reinterpret_cast<f32x2*>(ox + off) = xx[r];
                *reinterpret_cast<f32x2*>(oz + off) = zz[r];
              } else if (vec2) {  // (n_in even: in0 implies in1)
                if (in0) {
                  *reinterpret_cast<f32x2*>(ox + off) = xx[r];
                  *reinterpret_cast<f32x2*>(oz + off) = zz[r];
                }
              } else {
                if (in0) {
                  ox[off] = xx[r][0];
                  oz[off] = zz[r][0];
                }
                if (in1) {
                  ox[off + 1] = xx[r][1];
                  oz[off + 1] = zz[r][1];
                }
              }
            }
          }
        }
        if (!has) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) ph[tt][dt] = pl[tt][dt] = sh[tt][dt] = sl[tt][dt] = zero2;
        }
      }
      if (grad_flat) {
        // sums over the 32 rows: D [output][dim] += A [output][row] B [row][dim], three partial products, one accumulator
        const uint32_t* s0 = side + (2 * p) * H::TILE_WORDS + lane * 2;
        const uint32_t* s1 = side + (2 * p + (has1 ? 1 : 0)) * H::TILE_WORDS + lane * 2;
#pragma unroll
        for (int m = 0; m < YT; ++m) {
          const f16x8 mh8 = pair_operand(*reinterpret_cast<const u32x2*>(s0 + H::A_TR + (2 * m) * 128),
                                         *reinterpret_cast<const u32x2*>(s1 + H::A_TR + (2 * m) * 128));
          const f16x8 ml8 = pair_operand(*reinterpret_cast<const u32x2*>(s0 + H::A_TR + (2 * m + 1) * 128),
                                         *reinterpret_cast<const u32x2*>(s1 + H::A_TR + (2 * m + 1) * 128));
          const f16x8 vh8 = pair_operand(*reinterpret_cast<const u32x2*>(s0 + H::B_TR + (2 * m) * 128),
                                         *reinterpret_cast<const u32x2*>(s1 + H::B_TR + (2 * m) * 128));
          const f16x8 vl8 = pair_operand(*reinterpret_cast<const u32x2*>(s0 + H::B_TR + (2 * m + 1) * 128),
                                         *reinterpret_cast<const u32x2*>(s1 + H::B_TR + (2 * m + 1) * 128));
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {  // (no second tile: the B operands are zero there)
            const f16x8 pH = pair_operand(ph[0][dt], ph[1][dt]), pL = pair_operand(pl[0][dt], pl[1][dt]);
            const f16x8 sH = pair_operand(sh[0][dt], sh[1][dt]), sL = pair_operand(sl[0][dt], sl[1][dt]);
            aWm[dt][m] = mfma_h(mh8, pH, aWm[dt][m]);
            aWv[dt][m] = mfma_h(vh8, sH, aWv[dt][m]);
            aWm[dt][m] = mfma_h(mh8, pL, aWm[dt][m]);
            aWv[dt][m] = mfma_h(vh8, sL, aWv[dt][m]);
            aWm[dt][m] = mfma_h(ml8, pH, aWm[dt][m]);
            aWv[dt][m] = mfma_h(vl8, sH, aWv[dt][m]);
          }
        }
      }
    }
    if (!grad_flat) continue;
    // flush through LDS (the operand area is free now): the four waves add their tiles up as [tensor][output][dim of
    // the slab]; the workgroup then adds 128 contiguous bytes of a weight row per half wave to grad_flat (scattered
    // atomics run an order of magnitude slower at the memory side).  dW_log_var = exp(W_log_var) (.) dWv: the image's
    // scaled exp(W_log_var) (flat) times the scaled sum is the true product.
    float* red = reinterpret_cast<float*>(w_lds);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 16 * YT * UP; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    lds_wave_add<kMlbWaves>(det_part != nullptr, wave, [&](auto op) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int m = 0; m < YT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            op(red + (16 * m + 4 * q + r) * UP + 2 * j + dt, aWm[dt][m][r]);
            op(red + 16 * YT * UP + (16 * m + 4 * q + r) * UP + 2 * j + dt, aWv[dt][m][r]);
          }
    });
    __syncthreads();
    const int n_dims = min(32, n_in - 32 * slab);
    const int64_t wv = (int64_t)n_out * n_in;
    float* const out = det_part ? det_part + (int64_t)part * ml_params(n_in, n_out) : grad_flat;
    for (int e = threadIdx.x; e < n_out * 32; e += blockDim.x) {
      const int o = e >> 5, dl = e & 31;
      if (dl < n_dims) {
        const int64_t at = (int64_t)o * n_in + 32 * slab + dl;
        const float gm_ = red[o * UP + dl] * inv_gscale;
        const float gv_ = red[16 * YT * UP + o * UP + dl] * flat[wv + at] * (inv_gscale * var_unscale);
        if (det_part) {
          out[at] = gm_;
          out[wv + at] = gv_;
        } else {
          atomicAdd(out + at, gm_);
          atomicAdd(out + wv + at, gv_);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ slab launch, shared hand-over
// The kernel above re-reads the prologue's hand-over (1 KB per row) once per 32-dim slab -- twice the bytes of the slab's
// own x and z -- and each of its waves waits on its loads in turn.  Here, as in rnvp_bwd_ts_shared_kernel
// (mnf_rnvp_bwd.hip), a workgroup of EIGHT waves owns FOUR adjacent slabs and walks its row pairs together -- wave
// (slab s, tile t) computes tile 2 p + t of pair p for slab s --; the pair's hand-over is copied once per workgroup into
// a double-buffered LDS window by LDS-DMA (buffer_load ... lds) while the previous pair is computed, every wave reads its
// MFMA operands out of it, and the next pair's x and z rows are prefetched into registers.  The sums over a tile's 16
// rows are two K = 32 MFMAs per (output tile, tensor): A = [g_hi | g_lo'] against B = [p_hi | p_hi] and [p_lo' | 0].
// LDS at 64 outputs: 4 x 16 KB of operands + 2 x 2 x 16 KB of hand-over = 128 KB: one workgroup per CU.
typedef __attribute__((address_space(3))) void* lds_void_ptr_m;
constexpr int kMlsSlabs = 4;
constexpr int kMlsWaves = 2 * kMlsSlabs;

template <int YT>
struct MlSharedShape {
  using S = MlBwdShape<YT>;
  using H = HandoverShape<YT>;
  static constexpr int W_WORDS = S::SLAB_WORDS;
  static constexpr int HT_WORDS = H::TILE_WORDS;  // the whole tile: g_m, g_v with outputs on K and with rows on K
  static_assert(HT_WORDS % 256 == 0, "whole 1 KB pieces");
  static constexpr int HT_PIECES = HT_WORDS / 256;
  static constexpr int N_DMA = (2 * HT_PIECES + kMlsWaves - 1) / kMlsWaves;
  static constexpr int H_OFF = kMlsSlabs * W_WORDS;
  static constexpr int LDS_WORDS = H_OFF + 2 * 2 * HT_WORDS;
  static constexpr int UP = 33;                          // padded row of the flush area ([output][dim of the slab])
  static constexpr int RED_SLAB = 2 * 16 * YT * UP;      // both tensors of one slab
  static_assert(kMlsSlabs * RED_SLAB <= LDS_WORDS, "the flush area fits (it may run on into the hand-over window: every "
                                                   "piece has landed behind the barrier in front of the flush)");
  static_assert(LDS_WORDS * 4 <= 160 * 1024, "fits the CU's LDS");
};

template <int YT, bool RAG>
__global__ void __launch_bounds__(kMlsWaves * 64, 2)
ml_bwd_slab_shared_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ grad_x,
                          float* __restrict__ grad_z, float* __restrict__ grad_flat, const float* __restrict__ flat,
                          const uint32_t* __restrict__ bimage, const uint32_t* __restrict__ side,
                          const int32_t* __restrict__ flags, const float* __restrict__ gscale_dev, int64_t rows, int n_in,
                          int n_out, int n_slabs, int row_parts, int vec2, float var_unscale,
                          float* __restrict__ det_part) {
  using S = MlBwdShape<YT>;
  using H = HandoverShape<YT>;
  using T = MlSharedShape<YT>;
  constexpr int NKS = S::NKS;
  extern __shared__ __attribute__((aligned(16))) uint32_t ms_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sl = wave & (kMlsSlabs - 1), tt = wave / kMlsSlabs;  // (waves w and w + 4 share a SIMD: same slab, other tile)
  const int j = lane & 15, q = lane >> 4;
  const float inv_gscale = 1.f / gscale_dev[0];
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t n_tiles = (rows + 15) / 16, n_pairs = (n_tiles + 1) / 2;
  const int64_t per_part = (n_pairs + row_parts - 1) / row_parts;
  const int n_groups = (n_slabs + kMlsSlabs - 1) / kMlsSlabs;
  const SlabItems items(n_groups, row_parts);
  for (int item = items.first; item < items.n_items; item += items.step) {
    const int sg = items.slab(item), part = items.part(item);
    const int slab = sg * kMlsSlabs + sl;
    const bool slab_ok = slab < n_slabs;  // (wave-uniform: the last group may hold fewer than four slabs)
    const int64_t p0 = (int64_t)part * per_part, p_end = min(n_pairs, (int64_t)(part + 1) * per_part);
    if (p0 >= p_end) continue;  // (workgroup-uniform)
    __syncthreads();  // the previous item's flush is over
#pragma unroll 1
    for (int s_ = 0; s_ < kMlsSlabs; ++s_) {
      const int sb = sg * kMlsSlabs + s_;
      if (sb >= n_slabs) break;
      const uint4* src = reinterpret_cast<const uint4*>(bimage + (int64_t)sb * S::SLAB_WORDS);
      uint4* w = reinterpret_cast<uint4*>(ms_lds + s_ * T::W_WORDS);
      for (int i = threadIdx.x; i < S::SLAB_WORDS / 4; i += blockDim.x) w[i] = src[i];
    }
    // (visible to every wave behind the pair loop's first barrier)
    const int dim0 = 32 * (slab_ok ? slab : n_slabs - 1) + 2 * j;  // the lane's even dim; + 1: its odd dim
    const bool in0 = dim0 < n_in, in1 = dim0 + 1 < n_in;
    const uint32_t lane_off = (uint32_t)(4 * q) * (uint32_t)n_in + (uint32_t)dim0;
    // operand numbering in LDS: [(dt * 2 + which) * NKS + ks][part], which = 0: W_mean, 1: exp(W_log_var) (scaled)
    int w_lane = lane;  // (opaque, refreshed per pair: keeps the operand reads inside the pair loop)
    const f16x8* W8 = reinterpret_cast<const f16x8*>(ms_lds + sl * T::W_WORDS);
    auto wop = [&](int dt, int which, int ks, int part_) { return W8[w_lane + 64 * (2 * ((dt * 2 + which) * NKS + ks) + part_)]; };

    f32x4 aWm[2][YT], aWv[2][YT];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int m = 0; m < YT; ++m) aWm[dt][m] = aWv[dt][m] = zero4;

    // a tile's rows as LOADED (nothing here waits for a load: the values are first touched by compute(), one pair later)
    struct RowsIn {
      f32x2 xx[4], zz[4];
      bool active;
      int n_live;
      int64_t tbase;
    };
    auto load_rows = [&](int64_t p, RowsIn& in) {
      const int64_t tile = 2 * p + tt;
      const bool has = tile < n_tiles;
      in.tbase = (has ? tile : 2 * p) * 16;  // wave-uniform
      in.n_live = has ? (int)min((int64_t)16, rows - in.tbase) : 0;
      in.active = has && slab_ok && flags[(p * 32) / kMlbGroupRows] == 0;  // the fp32 kernel redoes flagged groups
      // (an idle wave loads all the same, from its clamped slab: no branch around the loads, so that the compiler's
      // count of the operations in flight stays exact)
      const float* xt = x + in.tbase * n_in;
      const float* zt = z + in.tbase * n_in;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool live = 4 * q + r < in.n_live;
        // a row past the end reads the tile's first row instead (multiplied by zero, nothing of it is stored)
        const uint32_t ob = (live ? lane_off + (uint32_t)r * (uint32_t)n_in : (uint32_t)dim0) * 4u;
        auto at = [&](const float* base) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + ob); };
        f32x2 xv = {0.f, 0.f}, zv = {0.f, 0.f};
        if (!RAG) {
          xv = *reinterpret_cast<const f32x2*>(at(xt));
          zv = *reinterpret_cast<const f32x2*>(at(zt));
        } else if (vec2) {
          if (in0) {
            xv = *reinterpret_cast<const f32x2*>(at(xt));
            zv = *reinterpret_cast<const f32x2*>(at(zt));
          }
        } else {
          if (in0) {
            xv[0] = at(xt)[0];
            zv[0] = at(zt)[0];
          }
          if (in1) {
            xv[1] = at(xt)[1];
            zv[1] = at(zt)[1];
          }
        }
        in.xx[r] = xv;
        in.zz[r] = zv;
      }
    };
    // the pair's hand-over -> LDS buffer `buf`: 2 x HT_PIECES pieces of 1 KB, dealt to the eight waves (no branches: a
    // piece past the end, or of a second tile that does not exist, repeats a valid one)
    auto request_handover = [&](int64_t p, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource builtins do not exist in hipcc's host pass over this file)
      const int has1 = 2 * p + 1 < n_tiles ? 1 : 0;
      const __amdgpu_buffer_rsrc_t pair_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint32_t*>(side + (2 * p) * H::TILE_WORDS), 0, 2 * H::TILE_WORDS * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < T::N_DMA; ++i) {
        const int piece = min(i * kMlsWaves + wave, 2 * T::HT_PIECES - 1);  // wave-uniform
        const int t = piece / T::HT_PIECES, k = piece - t * T::HT_PIECES;
        uint32_t* dst = ms_lds + T::H_OFF + (buf * 2 + t) * T::HT_WORDS + k * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pair_rsrc, (lds_void_ptr_m)dst, 16, lane * 16,
                                                 (t & has1) * (H::TILE_WORDS * 4) + k * 1024, 0, 0);
      }
#endif
      asm volatile("" ::: "memory");  // (the pair's stores stay behind the pieces: landed_barrier() counts on it)
    };
    // Vector-memory operations complete in issue order: with `stores` operations known to have been issued behind this
    // wave's pieces, all but the youngest `stores` being complete means the pieces are in LDS (the stores stay in flight).
    auto landed_barrier = [&](int stores) {
      if (stores == 8)
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // returns the number of vector-memory instructions it issued for certain (8 or 0)
    auto compute = [&](RowsIn& in, int buf) -> int {
      if (!in.active) return 0;
      asm volatile("" : "+v"(w_lane));
      uint32_t h_off = (uint32_t)(T::H_OFF + (buf * 2 + tt) * T::HT_WORDS) * 4u + (uint32_t)lane * 16u;
      asm volatile("" : "+v"(h_off));
      const char* hb = reinterpret_cast<const char*>(ms_lds) + h_off;                   // 16 B per lane: + operand * 1 KB
      const char* hb_tr = reinterpret_cast<const char*>(ms_lds) + h_off - lane * 8u;  // 8 B per lane: + operand * 512 B
      auto op = [&](int base_words, int ks, int part_) {
        return *reinterpret_cast<const f16x8*>(hb + base_words * 4 + (2 * ks + part_) * 1024);
      };
      auto tr = [&](int base_words, int m, int part_) {
        return *reinterpret_cast<const u32x2*>(hb_tr + base_words * 4 + (2 * m + part_) * 512);
      };
      f32x4 a4[2], b4[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {  // a^T, b^T [row][dim] = g [row][output] W [output][dim]
        f32x4 am = zero4, ac = zero4, bm = zero4, bc = zero4;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          split_mac(op(H::A_OP, ks, 0), op(H::A_OP, ks, 1), wop(dt, 0, ks, 0), wop(dt, 0, ks, 1), am, ac);
          split_mac(op(H::B_OP, ks, 0), op(H::B_OP, ks, 1), wop(dt, 1, ks, 0), wop(dt, 1, ks, 1), bm, bc);
        }
        a4[dt] = (ac * kSplitInvScale + am) * inv_gscale;
        b4[dt] = (bc * kSplitInvScale + bm) * (inv_gscale * var_unscale);
      }
      // the element-wise part on f32x2 values (the lane's two dims of a row): packed fp32 instructions
      f32x4 pz[2], x2[2];
      f32x2 gxo[4], gzo[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float keep = 4 * q + r < in.n_live ? 1.f : 0.f;  // (the cotangents of a row past the end are zero; keep
        const f32x2 xv = in.xx[r] * keep, zv = in.zz[r] * keep;  //  its x z finite)
        const f32x2 a2 = f32x2{a4[0][r], a4[1][r]}, b2 = f32x2{b4[0][r], b4[1][r]};
        const f32x2 p2 = xv * zv, s2 = xv * xv;  // the forward pass's operands (mnf_linear.py:48,53)
        gxo[r] = zv * a2 + (xv * b2) * 2.f;
        gzo[r] = xv * a2;
        pz[0][r] = p2[0];
        pz[1][r] = p2[1];
        x2[0][r] = s2[0];
        x2[1][r] = s2[1];
      }
      float* ox = grad_x + in.tbase * n_in;
      float* oz = grad_z + in.tbase * n_in;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * q + r < in.n_live) {
          const uint32_t off = lane_off + (uint32_t)r * (uint32_t)n_in;
          if (!RAG) {
            *reinterpret_cast<f32x2*>(ox + off) = gxo[r];
            *reinterpret_cast<f32x2*>(oz + off) = gzo[r];
          } else if (vec2) {  // (n_in even: in0 implies in1)
            if (in0) {
              *reinterpret_cast<f32x2*>(ox + off) = gxo[r];
              *reinterpret_cast<f32x2*>(oz + off) = gzo[r];
            }
          } else {
            if (in0) {
              ox[off] = gxo[r][0];
              oz[off] = gzo[r][0];
            }
            if (in1) {
              ox[off + 1] = gxo[r][1];
              oz[off + 1] = gzo[r][1];
            }
          }
        }
      }
      if (grad_flat) {
        const u32x2 zero2 = u32x2{0u, 0u};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          u32x2 ph, pl, sh, sl_;
          split_plain(pz[dt], ph, pl);
          split_plain(x2[dt], sh, sl_);
          const f16x8 p_hh = pair_operand(ph, ph), p_l0 = pair_operand(pl, zero2);
          const f16x8 s_hh = pair_operand(sh, sh), s_l0 = pair_operand(sl_, zero2);
#pragma unroll
          for (int m = 0; m < YT; ++m) {  // D [output][dim] += A [output][row] B [row][dim] over the tile's 16 rows
            const f16x8 m_hl = pair_operand(tr(H::A_TR, m, 0), tr(H::A_TR, m, 1));
            const f16x8 v_hl = pair_operand(tr(H::B_TR, m, 0), tr(H::B_TR, m, 1));
            aWm[dt][m] = mfma_h(m_hl, p_hh, aWm[dt][m]);
            aWv[dt][m] = mfma_h(v_hl, s_hh, aWv[dt][m]);
            aWm[dt][m] = mfma_h(m_hl, p_l0, aWm[dt][m]);
            aWv[dt][m] = mfma_h(v_hl, s_l0, aWv[dt][m]);
          }
        }
      }
      return (in.n_live == 16 && (!RAG || vec2)) ? 8 : 0;
    };

    {
      // two register sets, alternating roles (no copies: a copy would wait for the loads it moves); the last pair
      // requests itself once more into the idle buffer rather than branching around the requests
      RowsIn ra, rb;
      load_rows(p0, ra);
      request_handover(p0, 0);
      int stores = 0;
      int64_t p = p0;
      while (true) {
        landed_barrier(stores);  // every wave's pieces of pair p are in LDS, pair p - 1 (the other buffer) is consumed
        {
          const int64_t pn = min(p + 1, p_end - 1);
          load_rows(pn, rb);
          request_handover(pn, 1);
        }
        stores = compute(ra, 0);
        if (++p >= p_end) break;
        landed_barrier(stores);
        {
          const int64_t pn = min(p + 1, p_end - 1);
          load_rows(pn, ra);
          request_handover(pn, 0);
        }
        stores = compute(rb, 1);
        if (++p >= p_end) break;
      }
    }
    if (!grad_flat) continue;
    // flush through LDS: the waves add their tiles up as [slab][tensor][output][dim of the slab]; the workgroup then adds
    // 128 contiguous bytes of a weight row per half wave to grad_flat.  dW_log_var = exp(W_log_var) (.) dWv: the image's
    // scaled exp(W_log_var) (flat) times the scaled sum is the true product.
    float* red = reinterpret_cast<float*>(ms_lds);
    __syncthreads();  // (drains every wave's outstanding pieces too: the area may run on into the hand-over window)
    for (int i = threadIdx.x; i < kMlsSlabs * T::RED_SLAB; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    if (slab_ok) {
      float* rs = red + sl * T::RED_SLAB;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int m = 0; m < YT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(rs + (16 * m + 4 * q + r) * T::UP + 2 * j + dt, aWm[dt][m][r]);
            atomicAdd(rs + 16 * YT * T::UP + (16 * m + 4 * q + r) * T::UP + 2 * j + dt, aWv[dt][m][r]);
          }
    }
    __syncthreads();
    const int dim_g = 32 * kMlsSlabs * sg;                   // first dim of the group
    const int n_dims = min(32 * kMlsSlabs, n_in - dim_g);    // dims of this group that exist
    const int64_t wv = (int64_t)n_out * n_in;
    for (int e = threadIdx.x; e < n_out * 32 * kMlsSlabs; e += blockDim.x) {
      const int o = e / (32 * kMlsSlabs), dl = e - o * (32 * kMlsSlabs);
      if (dl < n_dims) {
        const float* rs = red + (dl >> 5) * T::RED_SLAB + o * T::UP + (dl & 31);
        const int64_t at = (int64_t)o * n_in + dim_g + dl;
        const float gm_ = rs[0] * inv_gscale, gv_ = rs[16 * YT * T::UP] * flat[wv + at] * (inv_gscale * var_unscale);
        // (two waves per slab added into the zeroed area: either order gives the same number.  det_part: plain stores
        // into the row part's block, det_reduce_async adds the blocks in a fixed order)
        if (det_part) {
          float* const out = det_part + (int64_t)part * ml_params(n_in, n_out);
          out[at] = gm_;
          out[wv + at] = gv_;
        } else {
          atomicAdd(grad_flat + at, gm_);
          atomicAdd(grad_flat + wv + at, gv_);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ fp32 fix-up
// the listed 128-row groups from the flat parameters: grad_x, grad_z per element, weight gradients by atomics (rare path)
__global__ void __launch_bounds__(256)
ml_bwd_fixup_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ gout,
                    const float* __restrict__ sd, const float* __restrict__ eps, uint64_t seed, float* __restrict__ grad_x,
                    float* __restrict__ grad_z, float* __restrict__ grad_flat, const float* __restrict__ flat,
                    const int32_t* __restrict__ list, int64_t rows, int n_in, int n_out, float var_unscale) {
  __shared__ float gm[64], gv[64];
  const float* wm = flat;
  const float* wvs = flat + (int64_t)n_out * n_in;  // exp(W_log_var) / var_unscale
  const int64_t bmo = 2 * (int64_t)n_out * n_in, bvo = bmo + n_out;
  const int n = list[0];
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int64_t row0 = (int64_t)list[1 + e] * kMlbGroupRows;
    for (int64_t row = row0; row < row0 + kMlbGroupRows && row < rows; ++row) {
      __syncthreads();
      if ((int)threadIdx.x < n_out) {
        const int o = threadIdx.x;
        const float g = gout[row * n_out + o];
        const float ev = eps ? eps[row * n_out + o] : ml_normal(seed, row, o);
        const float v = g * ev / (2.f * sd[row * n_out + o]);
        gm[o] = g;
        gv[o] = v;
        if (grad_flat) {
          atomicAdd(grad_flat + bmo + o, g);
          atomicAdd(grad_flat + bvo + o, v * flat[bvo + o]);
        }
      }
      __syncthreads();
      for (int c = threadIdx.x; c < n_in; c += blockDim.x) {
        const float xv = x[row * n_in + c], zv = z[row * n_in + c];
        float a = 0.f, b = 0.f;
        for (int o = 0; o < n_out; ++o) {
          a = fmaf(gm[o], wm[(int64_t)o * n_in + c], a);
          b = fmaf(gv[o], wvs[(int64_t)o * n_in + c], b);
        }
        b *= var_unscale;
        grad_x[row * n_in + c] = zv * a + 2.f * (xv * b);
        grad_z[row * n_in + c] = xv * a;
        if (grad_flat)
          for (int o = 0; o < n_out; ++o) {
            atomicAdd(grad_flat + (int64_t)o * n_in + c, gm[o] * (xv * zv));
            atomicAdd(grad_flat + (int64_t)n_out * n_in + (int64_t)o * n_in + c,
                      gv[o] * (xv * xv) * (wvs[(int64_t)o * n_in + c] * var_unscale));
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
// flat: W_mean (n_out, n_in) | exp(W_log_var) * 2^shift (n_out, n_in) | b_mean | exp(b_log_var)   (as for the forward image)
template <int YT>
static void build_mlb_index(int n_in, int n_out, int32_t* idx) {
  using S = MlBwdShape<YT>;
  constexpr int NKS = S::NKS;
  const int64_t wm = 0, wv = (int64_t)n_out * n_in;
  const int64_t n_entries = 2 * S::split_words(n_in);
  for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  auto put = [&](int64_t base_words, int op, int lane, int e, int64_t src) {
    for (int part = 0; part < 2; ++part)
      idx[2 * base_words + (((int64_t)(2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] =
          (int32_t)src | (part ? kSplitLoBit : 0);
  };
  // slot 8 kq + e of K-step ks <-> output 16 (2 ks + (e >> 2)) + 4 kq + (e & 3)   (as the hand-over's A operands)
  for (int sl = 0; sl < S::n_slabs(n_in); ++sl)
    for (int dt = 0; dt < 2; ++dt)
      for (int lane = 0; lane < 64; ++lane) {
        const int c = lane & 15, kq = lane >> 4, dim = 32 * sl + 2 * c + dt;
        if (dim >= n_in) continue;
        for (int ks = 0; ks < NKS; ++ks)
          for (int e = 0; e < 8; ++e) {
            const int o = 16 * (2 * ks + (e >> 2)) + 4 * kq + (e & 3);
            if (o >= n_out || 2 * ks + (e >> 2) >= YT) continue;
            for (int which = 0; which < 2; ++which)
              put((int64_t)sl * S::SLAB_WORDS, (dt * 2 + which) * NKS + ks, lane, e, (which ? wv : wm) + (int64_t)o * n_in + dim);
          }
      }
}

static int mlb_tiles(int n_out) { return n_out < 1 || n_out > 64 ? 0 : (n_out + 15) / 16; }

// the slab launch on the one-slab-per-workgroup kernel (round 3) is kept as the fallback when the shared kernel's LDS
// request is refused (return true here for a same-box A/B of the two)
static bool mlb_slab_split_forced() { return false; }

static int64_t mlb_tiles_end(int64_t rows, int64_t tile_words);
static int64_t mlb_header_bytes(int64_t rows) {
  const int64_t n_groups = (rows + kMlbGroupRows - 1) / kMlbGroupRows;
  return (((1 + 2 * n_groups) * 4 + 255) & ~(int64_t)255);
}

static int64_t mlb_tiles_end(int64_t rows, int64_t tile_words) {
  return (mlb_header_bytes(rows) + ((rows + 15) / 16) * tile_words * 4 + 255) & ~(int64_t)255;
}

template <int YT, bool RAG>
static int launch_mlb(const float* x, const float* z, const float* gout, const float* sd, const float* eps, uint64_t seed,
                      float* grad_x, float* grad_z, float* grad_flat, const float* flat, const uint32_t* bimage,
                      float var_unscale, const int32_t* fwd_flags, const float* gscale, void* work, int64_t rows, int n_in,
                      int n_out, int vec2, hipStream_t stream) {
  using S = MlBwdShape<YT>;
  const int64_t n_groups = (rows + kMlbGroupRows - 1) / kMlbGroupRows;
  int32_t* list = static_cast<int32_t*>(work);
  int32_t* flags = list + 1 + n_groups;
  uint32_t* side = reinterpret_cast<uint32_t*>(static_cast<char*>(work) + mlb_header_bytes(rows));
  if (int rc = zero_word_async(list, stream)) return rc;
  // as many workgroups as are resident at once (YT = 4 holds a tile's 48 loads in registers: one per CU; a second one
  // per CU queued behind the first measured 99 us against 89 at 256,000 x 50)
  static DeviceMemo memo_p;
  const int resident_p =
      memo_p.get([](int dev) { return resident_by_occupancy(ml_bwd_prologue_kernel<YT>, 8 * 64, dev, 1); });
  int64_t blocks_p = n_groups < resident_p ? n_groups : resident_p;
  const bool det = deterministic() && grad_flat != nullptr;
  float* const det_b =
      det ? reinterpret_cast<float*>(static_cast<char*>(work) + mlb_tiles_end(rows, HandoverShape<YT>::TILE_WORDS)) : nullptr;
  float* const det_part = det ? det_b + (int64_t)kMlDetBlocks * 128 : nullptr;
  const int64_t n_params = ml_params(n_in, n_out), bmo = 2 * (int64_t)n_out * n_in;
  const int max_l = det ? kMlDetParts / 8 : 32;
  if (det && blocks_p > kMlDetBlocks) blocks_p = kMlDetBlocks;
  tag_kernel("mnf_linear_bwd");
  hipLaunchKernelGGL((ml_bwd_prologue_kernel<YT>), dim3((unsigned)blocks_p), dim3(8 * 64), 0, stream, gout, sd, eps, seed,
                     flat, fwd_flags, side, flags, list, gscale, grad_flat, rows, n_in, n_out, var_unscale, 1.f, det_b);
  if (int rc = check_launch()) return rc;
  if (det) {
    if (int rc = det_reduce_async(det_b, (int)blocks_p, 128, n_out, grad_flat + bmo, stream)) return rc;
    if (int rc = det_reduce_async(det_b + 64, (int)blocks_p, 128, n_out, grad_flat + bmo + n_out, stream)) return rc;
  }
  const int n_slabs = (int)S::n_slabs(n_in);
  const int64_t n_pairs = ((rows + 15) / 16 + 1) / 2;
  int row_parts, grid;
  bool launched = false;
  if (!mlb_slab_split_forced()) {
    // the shared-hand-over kernel: four slabs per workgroup, one workgroup per CU
    using T = MlSharedShape<YT>;
    static DeviceMemo memo_s;
    void (*const kernel)(const float*, const float*, float*, float*, float*, const float*, const uint32_t*, const uint32_t*,
                         const int32_t*, const float*, int64_t, int, int, int, int, int, float, float*) =
        ml_bwd_slab_shared_kernel<YT, RAG>;  // (named out here: a kernel first named inside a lambda gets no host stub)
    const int resident_s = memo_s.get([kernel](int dev) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS_WORDS * 4) != hipSuccess)
        return -1;
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kMlsWaves * 64, T::LDS_WORDS * 4) != hipSuccess ||
          per_cu < 1)
        per_cu = 1;
      return per_cu * device_cus(dev);
    });
    if (resident_s > 0) {
      plan_slab_launch(n_pairs, 1, (n_slabs + kMlsSlabs - 1) / kMlsSlabs, resident_s, row_parts, grid, max_l);
      if (det)  // (a row part without pairs writes nothing)
        if (int rc = zero_floats_async(det_part, row_parts * n_params, stream)) return rc;
      hipLaunchKernelGGL((ml_bwd_slab_shared_kernel<YT, RAG>), dim3((unsigned)grid), dim3(kMlsWaves * 64), T::LDS_WORDS * 4,
                         stream, x, z, grad_x, grad_z, grad_flat, flat, bimage, side, flags, gscale, rows, n_in, n_out,
                         n_slabs, row_parts, vec2, var_unscale, det_part);
      if (int rc = check_launch()) return rc;
      launched = true;
    }
  }
  if (!launched) {
    static DeviceMemo memo;
    const int resident = memo.get(
        [](int dev) { return resident_by_occupancy(ml_bwd_slab_kernel<YT, RAG>, kMlbWaves * 64, dev, 2); });
    plan_slab_launch(n_pairs, kMlbWaves, n_slabs, resident, row_parts, grid, max_l);
    if (det)
      if (int rc = zero_floats_async(det_part, row_parts * n_params, stream)) return rc;
    hipLaunchKernelGGL((ml_bwd_slab_kernel<YT, RAG>), dim3((unsigned)grid), dim3(kMlbWaves * 64), 0, stream, x, z, grad_x,
                       grad_z, grad_flat, flat, bimage, side, flags, gscale, rows, n_in, n_out, n_slabs, row_parts, vec2,
                       var_unscale, det_part);
    if (int rc = check_launch()) return rc;
  }
  if (det)  // the weight blocks of the row parts, in order (the bias sums went out behind the prologue)
    if (int rc = det_reduce_async(det_part, row_parts, n_params, bmo, grad_flat, stream)) return rc;
  int fix_blocks = 256;
  if (deterministic()) {  // the groups in ascending order, one workgroup: mnf_host.h det_sort_ids_async
    const int64_t n_groups = (rows + kMlbGroupRows - 1) / kMlbGroupRows;
    if (det_sort_ids_async(list + 1, list, (int)n_groups, n_groups, stream) == MNF_OK) fix_blocks = 1;
  }
  hipLaunchKernelGGL(ml_bwd_fixup_kernel, dim3((unsigned)fix_blocks), dim3(256), 0, stream, x, z, gout, sd, eps, seed, grad_x,
                     grad_z, grad_flat, flat, list, rows, n_in, n_out, var_unscale);
  return check_launch();
}

}  // namespace mnf

extern "C" {

using namespace mnf;

int64_t mnf_mnf_linear_bwd_workspace_bytes(int64_t rows, int n_in, int n_out) {
  const int yt = mlb_tiles(n_out);
  if (rows < 0 || n_in < 1 || yt == 0) return 0;
  const int64_t r = rows < 1 ? 1 : rows;
  int64_t tile_words = 0;
#define X(YT) if (yt == YT) tile_words = HandoverShape<YT>::TILE_WORDS;
  X(1) X(2) X(3) X(4)
#undef X
  const int64_t det = deterministic() ? ((int64_t)kMlDetBlocks * 128 + kMlDetParts * ml_params(n_in, n_out)) * 4 : 0;
  return mlb_tiles_end(r, tile_words) + det;
}

int mnf_mnf_linear_bwd_layout(int n_in, int n_out, int64_t* n_split_words, int64_t* n_plain_words) {
  if (!n_split_words || !n_plain_words || n_in < 1) return MNF_ERR_INVALID_ARG;
  const int yt = mlb_tiles(n_out);
  if (yt == 0 || (int64_t)n_in * 64 * 8 >= (1ll << 30)) return MNF_ERR_UNSUPPORTED;
#define X(YT) if (yt == YT) *n_split_words = MlBwdShape<YT>::split_words(n_in);
  X(1) X(2) X(3) X(4)
#undef X
  *n_plain_words = 0;
  return MNF_OK;
}

int mnf_mnf_linear_bwd_index(int n_in, int n_out, int32_t* idx_host) {
  if (!idx_host || n_in < 1) return MNF_ERR_INVALID_ARG;
  const int yt = mlb_tiles(n_out);
#define X(YT)                                   \
  if (yt == YT) {                               \
    build_mlb_index<YT>(n_in, n_out, idx_host); \
    return MNF_OK;                              \
  }
  X(1) X(2) X(3) X(4)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_mnf_linear_bwd(const float* x, const float* z, const float* grad_out, const float* sd, const float* eps,
                       uint64_t seed, float* grad_x, float* grad_z, float* grad_flat, const float* flat,
                       const void* bwd_image, float var_unscale, const int32_t* fwd_flags, const float* grad_scale_dev,
                       void* workspace, int64_t workspace_bytes, int64_t rows, int n_in, int n_out, void* stream) {
  if (!x || !z || !grad_out || !sd || !grad_x || !grad_z || !flat || !bwd_image || !fwd_flags || !grad_scale_dev ||
      !workspace || rows < 0 || n_in < 1 || n_out < 1 || !(var_unscale > 0.f))
    return MNF_ERR_INVALID_ARG;
  const int yt = mlb_tiles(n_out);
  if (yt == 0) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  if (workspace_bytes < mnf_mnf_linear_bwd_workspace_bytes(rows, n_in, n_out)) return MNF_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(bwd_image) | reinterpret_cast<uintptr_t>(workspace)) & 15) return MNF_ERR_UNSUPPORTED;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(z) |
                         reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(grad_z);
  const bool ragged = (n_in & 31) != 0 || (ptrs & 7) != 0;
  const int vec2 = (ptrs & 7) == 0 && (n_in & 1) == 0;
  const uint32_t* bi = static_cast<const uint32_t*>(bwd_image);
  hipStream_t st = (hipStream_t)stream;
#define X(YT)                                                                                                          \
  if (yt == YT)                                                                                                        \
    return ragged ? launch_mlb<YT, true>(x, z, grad_out, sd, eps, seed, grad_x, grad_z, grad_flat, flat, bi, var_unscale, \
                                         fwd_flags, grad_scale_dev, workspace, rows, n_in, n_out, vec2, st)            \
                  : launch_mlb<YT, false>(x, z, grad_out, sd, eps, seed, grad_x, grad_z, grad_flat, flat, bi,          \
                                          var_unscale, fwd_flags, grad_scale_dev, workspace, rows, n_in, n_out, vec2, st);
  X(1) X(2) X(3) X(4)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
