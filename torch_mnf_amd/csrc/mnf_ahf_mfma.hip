// AffineHalfFlow coupling layer on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// One wave owns a tile of 16 consecutive rows (samples).  Everything is computed transposed,
// activations^T = W . x^T, so that the batch sits on the MFMA's N axis (lane & 15) and the
// accumulator of one Linear is, register for register, the B operand of the next one:
//
//   D layout of 16x16x4:  lane (j = lane & 15, q = lane >> 4), reg r  holds  D[row 4q + r][col j]
//   B operand of a K-step: lane (j, q) supplies B[k = q][col j]
//   => register r of an accumulator tile IS a K-step operand whose four k's are the rows
//      {r, 4 + r, 8 + r, 12 + r} of that tile.  Hidden units are therefore numbered so that
//      "quad" c (= K-step) of the concatenated [s_net ; t_net] hidden vector holds units
//      4c .. 4c+3, quad c lives in register c % 4 of tile c / 4, and no lane shuffle or LDS
//      round trip is needed between layers: bias is the initial accumulator, LeakyReLU is two
//      VALU ops on the accumulator registers.
//
// s_net and t_net share their input, so layer 1 is one (2*HID x H) product; the hidden layers
// are block diagonal and only the K-steps of the nets present in an output tile are issued
// (for HID = 24: 6 + 12 + 6 MFMAs instead of 32 for two padded nets).  The last layer produces
// s and t for 16 output dims at a time in two accumulators with the same lane layout as a
// float4 of the row, so the affine transform, the store and the log|det J| partial sum run
// straight out of registers; the per-row sum is finished with two cross-lane adds.
//
// Weights: the host builds an "image" in MFMA A-operand order ([group of 4 MFMAs][lane][4]
// floats, then biases); each workgroup copies it to LDS once and every wave re-reads it with
// conflict-free ds_read_b128 (one read feeds four MFMAs).  Workgroups are persistent and
// stride over the tiles.
//
// HBM traffic per row: read 4*dim, write 4*dim, log_det 4 (+4 when accumulating): the
// algorithmic minimum for an out-of-place layer (8d + 8 bytes).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {

// Full-line stores.  In the operand layout one store instruction writes 64 bytes of each of 16
// rows, i.e. it touches 16 cache lines, and store cost on this chip goes with the number of lines
// an instruction touches (about 9 cycles each; tools/ahf_microbench.hip: the stores cost 20 us of
// a 140 us layer).  Rows j and j^8 sit in the same 16-lane DPP row, so one row_ror:8 exchange per
// register turns two float4 slots (g, g+1) of 16 rows into  A: rows 0-7, 128 contiguous bytes
// each  and  B: rows 8-15  -- 8 lines per instruction instead of 16, same bytes.
__device__ __forceinline__ float dpp_ror8(float keep, float src, int bank_mask_hi) {
  // lanes 8..15 of every 16-lane row (bank_mask 0xC) or lanes 0..7 (0x3) take src from lane j^8
  const int r = bank_mask_hi
                    ? __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, src), 0x128, 0xF, 0xC, false)
                    : __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, src), 0x128, 0xF, 0x3, false);
  return __builtin_bit_cast(float, r);
}
// v0 = this lane's float4 of slot g, v1 = of slot g+1 (row j, floats 16g+4q.. and 16(g+1)+4q..).
// ya / yb: this lane's destination in instruction A (row j&7) / B (row 8 + (j&7)).
__device__ __forceinline__ void store_pair_wide(float* ya, float* yb, bool live_a, bool live_b, const f32x4& v0,
                                                const f32x4& v1) {
  f32x4 a, b;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a[r] = dpp_ror8(v0[r], v1[r], 1);  // j < 8: own slot g ; j >= 8: slot g+1 of row j-8
    b[r] = dpp_ror8(v1[r], v0[r], 0);  // j < 8: slot g of row j+8 ; j >= 8: own slot g+1
  }
  if (live_a) *reinterpret_cast<f32x4*>(ya) = a;
  if (live_b) *reinterpret_cast<f32x4*>(yb) = b;
}

template <bool NT>
__device__ __forceinline__ void st4(float* p, const f32x4& v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ f32x4 ld4(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}

// (ablation builds replace the MFMA by a pass-through of the accumulator)
#define MNF_MFMA(a, b, c, x0, x1, x2) \
  ((ABL == 1 || ABL == 5) ? (c) : __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), (x0), (x1), (x2)))


// Waves per workgroup.  Up to d = 64 the 25 KB image allows six 256-thread workgroups per CU (one
// wave per SIMD each, so residency moves in steps of one wave/SIMD).  From d = 128 the image is
// 38-63 KB and LDS, not registers, caps residency: 512-thread workgroups share one image between
// twice as many waves.
template <int H>
constexpr int ahf_waves() { return H >= 64 ? 8 : 4; }

// ABL != 0 only in tools/ahf_microbench.hip (ablation builds: 1 = no MFMA chain, 2 = no HBM
// traffic, 3 = no exp/divide, 4 = A operands not re-read from LDS, 5 = copy only, 6 = no stores,
// 7-9 = single stores off, 12 = full-line stores through a DPP row exchange);
// the library uses ABL = 0.
template <int H, int HID, bool INV, bool PREFETCH, int ABL = 0>
__global__ void __launch_bounds__(ahf_waves<H>() * 64)
ahf_mfma_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ log_det,
                float* __restrict__ ysq, const float* __restrict__ image, int64_t rows, int parity,
                int accumulate) {
  using S = AhfShape<H, HID>;
  constexpr int G = S::G, QN = S::QN, NQ = S::NQ, NT = S::NT;
  constexpr int dim = 2 * H;
  __shared__ __attribute__((aligned(16))) float lds[S::IMAGE_FLOATS];

  {  // stage the operand image (L2-resident) into LDS
    const float4* src = reinterpret_cast<const float4*>(image);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < S::IMAGE_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int cond_off = parity ? H : 0, act_off = parity ? 0 : H;

  // tiles are counted in 32 bits (rows < 2^35); only the row offset is 64-bit
  const int n_tiles = (int)((rows + 15) >> 4);
  constexpr int kAhfWaves = ahf_waves<H>();
  const int tile_stride = (int)gridDim.x * kAhfWaves;
  int tile = (int)blockIdx.x * kAhfWaves + wave;

  // rows past the end are clamped to the last row for loads and masked for stores
  auto row_ptr = [&](int t) -> const float* {
    if (ABL == 2) t = blockIdx.x * kAhfWaves + wave;  // ablation: stay on one cached tile
    const int64_t r = (int64_t)t * 16 + j;
    if (ABL == 10) return x + (int64_t)t * 16 * dim + lane * 4 - cond_off;  // ablation: contiguous 1-KiB accesses (wrong data)
    return x + (r < rows ? r : rows - 1) * dim + 4 * q;
  };

  f32x4 cnd[G], act[G];
  if (PREFETCH && tile < n_tiles) {
    const float* xr = row_ptr(tile);
#pragma unroll
    for (int g = 0; g < G; ++g) cnd[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
#pragma unroll
    for (int g = 0; g < G; ++g) act[g] = *reinterpret_cast<const f32x4*>(xr + act_off + 16 * g);
  }

  for (; tile < n_tiles; tile += tile_stride) {
    const int64_t row = (int64_t)tile * 16 + j;
    bool live = row < rows;
    if (ABL == 2 || ABL == 6) live = live && (cnd[0][0] == 1.2345e30f);  // ablation: never true, keeps the math alive
    const int64_t rowc = live ? row : rows - 1;
    float* yr = y + rowc * dim + 4 * q;
    if (ABL == 10) yr = y + (int64_t)tile * 16 * dim + lane * 4 - cond_off;
    // wide stores: instruction A serves row (j & 7) of the tile, B row 8 + (j & 7); the lanes with
    // j >= 8 carry the second 64 bytes of the line
    // Measured (interleaved rounds, d = 64): wide 141.7 us vs narrow 135.4 us -- the extra DPP moves
    // cost more than the halved line count saves, so the library uses the narrow form (ABL 12 = wide).
    constexpr bool WIDE = (G % 2 == 0) && ABL == 12;
    const int64_t row_a = (int64_t)tile * 16 + (j & 7), row_b = row_a + 8;
    const bool live_a = row_a < rows && (ABL != 2 && ABL != 6), live_b = row_b < rows && (ABL != 2 && ABL != 6);
    float* ya = y + (live_a ? row_a : rows - 1) * dim + 16 * (j >> 3) + 4 * q;
    float* yb = y + (live_b ? row_b : rows - 1) * dim + 16 * (j >> 3) + 4 * q;
    // PREFETCH: the next tile's rows are requested into the SAME registers as soon as their
    // last reader of this tile has issued (cnd: after layer 1; act[m]: after output step m), so
    // the loads fly under the remaining MFMA chain at no register cost.  One tile past the end
    // re-reads the last tile: harmless, and keeps the loop branch-free.
    const float* xn = PREFETCH ? row_ptr(tile + tile_stride < n_tiles ? tile + tile_stride : n_tiles - 1)
                               : row_ptr(tile);
    if (!PREFETCH) {
#pragma unroll
      for (int g = 0; g < G; ++g) cnd[g] = ld4<ABL == 14 || ABL == 15>(xn + cond_off + 16 * g);
#pragma unroll
      for (int g = 0; g < G; ++g) act[g] = ld4<ABL == 14 || ABL == 15>(xn + act_off + 16 * g);
    }
    if (WIDE) {
      if (ABL != 8) {
#pragma unroll
        for (int g = 0; g + 1 < G; g += 2)
          store_pair_wide(ya + cond_off + 16 * g, yb + cond_off + 16 * g, live_a, live_b, cnd[g], cnd[g + 1]);
      }
    } else if (live && ABL != 8) {
#pragma unroll
      for (int g = 0; g < G; ++g) st4<ABL == 13 || ABL == 14>(yr + cond_off + 16 * g, cnd[g]);
    }
    float ld = 0.f, sq = 0.f;
    if (ysq) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = fmaf(cnd[g][r], cnd[g][r], sq);
    }

    int n = 0;       // MFMA sequence number (compile-time after unrolling)
    int btile = 0;   // bias tile number
    f32x4 a4;
    // The operand image never changes, so hipcc would hoist all ~100 A-operand reads out of
    // the tile loop into VGPRs (2 waves/SIMD).  Re-reading them from LDS costs one
    // ds_read_b128 per four MFMAs and keeps the kernel at 8 waves/SIMD, which is what hides
    // the HBM latency here; making the base pointers opaque per tile blocks the hoist.
    // (the opaque value is an integer offset, not the pointer: an opaque pointer loses its LDS
    // address space and turns every read into a flat_load that also waits on vmcnt.)
    int a_off = lane * 4, b_off = S::A_FLOATS + q * 4;
    asm volatile("" : "+v"(a_off), "+v"(b_off));
    const f32x4* A4 = reinterpret_cast<const f32x4*>(lds + a_off);  // + 64 * group
    const f32x4* B4 = reinterpret_cast<const f32x4*>(lds + b_off);  // + 4 * tile

    // ---- layer 1: [s;t] hidden (2*HID) <- cond (H)
    f32x4 h1[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) h1[m] = B4[4 * (btile++)];
#pragma unroll
    for (int c1 = 0; c1 < H / 4; ++c1) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        if ((n & 3) == 0 && (ABL != 4 || n == 0)) a4 = A4[64 * (n >> 2)];
        h1[m] = MNF_MFMA(a4[n & 3], cnd[c1 >> 2][c1 & 3], h1[m], 0, 0, 0);
        ++n;
      }
    }
    if (PREFETCH) {
#pragma unroll
      for (int g = 0; g < G; ++g) cnd[g] = *reinterpret_cast<const f32x4*>(xn + cond_off + 16 * g);
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h1[m][r] = leaky2(h1[m][r]);

    // ---- layers 2 and 3: block-diagonal (HID <- HID) per net
    f32x4 h2[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) h2[m] = B4[4 * (btile++)];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        if (S::tile_has_net(m, c >= QN ? 1 : 0)) {
          if ((n & 3) == 0 && (ABL != 4 || n == 0)) a4 = A4[64 * (n >> 2)];
          h2[m] = MNF_MFMA(a4[n & 3], h1[c >> 2][c & 3], h2[m], 0, 0, 0);
          ++n;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h2[m][r] = leaky2(h2[m][r]);

    f32x4 h3[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) h3[m] = B4[4 * (btile++)];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        if (S::tile_has_net(m, c >= QN ? 1 : 0)) {
          if ((n & 3) == 0 && (ABL != 4 || n == 0)) a4 = A4[64 * (n >> 2)];
          h3[m] = MNF_MFMA(a4[n & 3], h2[c >> 2][c & 3], h3[m], 0, 0, 0);
          ++n;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h3[m][r] = leaky2(h3[m][r]);

    // ---- layer 4 + affine transform, 16 output dims per step
    f32x4 o_even = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < G; ++m) {
      f32x4 s4 = B4[4 * (btile++)];
      f32x4 t4 = B4[4 * (btile++)];
#pragma unroll
      for (int c = 0; c < QN; ++c) {
        if ((n & 3) == 0 && (ABL != 4 || n == 0)) a4 = A4[64 * (n >> 2)];
        s4 = MNF_MFMA(a4[n & 3], h3[c >> 2][c & 3], s4, 0, 0, 0);
        ++n;
        if ((n & 3) == 0 && (ABL != 4 || n == 0)) a4 = A4[64 * (n >> 2)];
        t4 = MNF_MFMA(a4[n & 3], h3[(QN + c) >> 2][(QN + c) & 3], t4, 0, 0, 0);
        ++n;
      }
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // inverse: (v - t) / exp(s) evaluated as (v - t) * exp(-s): one multiply instead of a
        // ~10-instruction IEEE divide; same limits (0, inf, NaN) and <= 2 ulp from the quotient
        const float e = (ABL == 3 || ABL == 5) ? s4[r] : exp6(INV ? -s4[r] : s4[r]);
        o[r] = (ABL == 3 || ABL == 5) ? (act[m][r] - t4[r]) + e : INV ? (act[m][r] - t4[r]) * e : __builtin_fmaf(e, act[m][r], t4[r]);
        ld += s4[r];
        sq = fmaf(o[r], o[r], sq);
      }
      if (WIDE) {
        if (m & 1) {
          const bool keep = (ABL != 9 || o[0] == 1.2345e30f);
          store_pair_wide(ya + act_off + 16 * (m - 1), yb + act_off + 16 * (m - 1), live_a && keep, live_b && keep,
                          o_even, o);
        } else {
          o_even = o;
        }
      } else if (live && (ABL != 9 || o[0] == 1.2345e30f)) {
        st4<ABL == 13 || ABL == 14>(yr + act_off + 16 * m, o);
      }
      if (PREFETCH) act[m] = *reinterpret_cast<const f32x4*>(xn + act_off + 16 * m);
    }
    if (log_det) {
      ld = sum_over_q(ld);
      if (INV) ld = -ld;
      if (live && q == 0 && (ABL != 7 || ld == 1.2345e30f)) log_det[row] = accumulate ? log_det[row] + ld : ld;
    }
    if (ysq) {  // |y_row|^2 for the base log-prob epilogue: saves re-reading y (4*dim bytes/row)
      sq = sum_over_q(sq);
      if (live && q == 0) ysq[row] = sq;
    }
  }
}

// ---------------------------------------------------------------- host: image index table
// h <= H: real half width (the image is zero in the padded input columns / output rows)
// has[0] / has[1]: the s / t net exists (scale=False / shift=False, affine_half_flow.py:38: an absent net is the
// zero function -- here an all-zero operand set, so s = 0 or t = 0 exactly)
template <int H, int HID>
static void build_index(int32_t* idx, int h, bool has_s = true, bool has_t = true, const int* widths = nullptr) {
  using S = AhfShape<H, HID>;
  constexpr int QN = S::QN, NQ = S::NQ, NT = S::NT, G = S::G;
  const int w[3] = {widths ? widths[0] : HID, widths ? widths[1] : HID, widths ? widths[2] : HID};  // real widths <= HID
  int sizes[5] = {h, w[0], w[1], w[2], h};
  NetDesc net[2];
  const bool has[2] = {has_s, has_t};
  int64_t off = 0;
  if (has_s) off += fill_net(net[0], 5, sizes, off);
  if (has_t) off += fill_net(net[1], 5, sizes, off);
  for (int64_t i = 0; i < S::IMAGE_FLOATS; ++i) idx[i] = -1;
  int n = 0;
  auto put = [&](int lane, int32_t src) { idx[(n >> 2) * 256 + lane * 4 + (n & 3)] = src; };
  // hidden-vector unit of accumulator row i of tile m
  auto unit_of = [&](int m, int i) { return 16 * m + 4 * (i & 3) + (i >> 2); };

  for (int c1 = 0; c1 < H / 4; ++c1) {
    const int g = c1 >> 2, e = c1 & 3;
    for (int m = 0; m < NT; ++m) {
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, u = unit_of(m, i);
        if (u < 2 * HID) {
          const int nn = u / HID, unit = u % HID;
          const int col = 16 * g + 4 * kq + e;
          if (col < h && has[nn] && unit < w[0]) put(lane, net[nn].w_off[0] + unit * h + col);
        }
      }
      ++n;
    }
  }
  for (int l = 1; l <= 2; ++l) {
    for (int c = 0; c < NQ; ++c) {
      const int cn = c >= QN ? 1 : 0;
      for (int m = 0; m < NT; ++m) {
        if (!S::tile_has_net(m, cn)) continue;
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = unit_of(m, i);
          const int in_unit = 4 * c + kq - cn * HID;
          if (u < 2 * HID && u / HID == cn && has[cn] && u % HID < w[l] && in_unit < w[l - 1])
            put(lane, net[cn].w_off[l] + (u % HID) * w[l - 1] + in_unit);
        }
        ++n;
      }
    }
  }
  for (int m = 0; m < G; ++m) {
    for (int c = 0; c < QN; ++c) {
      for (int nn = 0; nn < 2; ++nn) {
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4;
          if (16 * m + i < h && has[nn] && 4 * c + kq < w[2]) put(lane, net[nn].w_off[3] + (16 * m + i) * w[2] + 4 * c + kq);
        }
        ++n;
      }
    }
  }
  // biases: [tile][row i]
  int32_t* b = idx + S::A_FLOATS;
  int bt = 0;
  for (int l = 0; l < 3; ++l)
    for (int m = 0; m < NT; ++m, ++bt)
      for (int i = 0; i < 16; ++i) {
        const int u = unit_of(m, i);
        if (u < 2 * HID && has[u / HID] && u % HID < w[l]) b[bt * 16 + i] = net[u / HID].b_off[l] + u % HID;
      }
  for (int m = 0; m < G; ++m)
    for (int nn = 0; nn < 2; ++nn, ++bt)
      for (int i = 0; i < 16; ++i)
        if (16 * m + i < h && has[nn]) b[bt * 16 + i] = net[nn].b_off[3] + 16 * m + i;
}

template <int H, int HID>
static int launch(const float* x, float* y, float* log_det, float* ysq, int accumulate,
                  const float* image, int64_t rows, int parity, int inverse, hipStream_t stream) {
  // Measured (tools/ahf_microbench.hip, d = 64, 6 workgroups/CU): 139.5 us without the in-place
  // prefetch, 145 us with it -- six resident waves per SIMD already cover the HBM latency.
  constexpr bool kPrefetch = false;
  constexpr int kAhfWaves = ahf_waves<H>();
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + kAhfWaves - 1) / kAhfWaves;
  // persistent grid: as many workgroups as are resident at once (registers and the LDS image
  // bound it), each striding over the tiles
  static DeviceMemo memo;
  const int resident = memo.get([](int dev) {
    return resident_by_occupancy(ahf_mfma_kernel<H, HID, true, kPrefetch>, kAhfWaves * 64, dev, 4);
  });
  const int cus = device_cus(current_device());
  blocks = balanced_grid(n_tiles, kAhfWaves, resident, cus);
  // Non-temporal loads/stores are a compile-time experiment switch, off: in the
  // isolated microbench they gain 5 % at d = 64, but inside the 9-layer pass (each layer re-reads
  // what the previous one just wrote) they are neutral at d = 64 and cost 13 % at d = 256.
  constexpr bool nt = false;  // (rebuild with true for the A/B)
  constexpr int kNt = 14;  // loads + stores non-temporal (see the ABL list above the kernel)
  const dim3 grid((unsigned)blocks), block(kAhfWaves * 64);
  tag_kernel("ahf_mfma_fp32");
  if constexpr (nt) {
    if (inverse)
      hipLaunchKernelGGL((ahf_mfma_kernel<H, HID, true, kPrefetch, kNt>), grid, block, 0, stream, x, y, log_det,
                         ysq, image, rows, parity, accumulate);
    else
      hipLaunchKernelGGL((ahf_mfma_kernel<H, HID, false, kPrefetch, kNt>), grid, block, 0, stream, x, y, log_det,
                         ysq, image, rows, parity, accumulate);
  } else {
    if (inverse)
      hipLaunchKernelGGL((ahf_mfma_kernel<H, HID, true, kPrefetch>), grid, block, 0, stream, x, y, log_det, ysq,
                         image, rows, parity, accumulate);
    else
      hipLaunchKernelGGL((ahf_mfma_kernel<H, HID, false, kPrefetch>), grid, block, 0, stream, x, y, log_det, ysq,
                         image, rows, parity, accumulate);
  }
  return check_launch();
}

// (H, HID) pairs with an instantiated kernel
// reference default hidden width 24 at d = 32..256, plus widths 16 and 32 at the small dims
#define MNF_AHF_SHAPES(X) X(16, 24) X(32, 24) X(64, 24) X(128, 24) X(16, 16) X(32, 16) X(16, 32) X(32, 32) X(64, 32) X(16, 64) X(32, 64) X(64, 64)

// three hidden layers of at most 64 units: hid = the width the kernels run them at (see ahf_padded_hidden)
static bool uniform_hidden(int n_hidden, const int* hidden, int& hid) {
  hid = ahf_padded_hidden(n_hidden, hidden);
  return hid != 0;
}

int ahf_mfma_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate,
                    const float* image, int64_t rows, int dim, int parity, int inverse, int n_hidden,
                    const int* hidden, int has_scale, int has_shift, hipStream_t stream) {
  int hid = 0;
  if ((!has_scale && !has_shift) || !uniform_hidden(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
       reinterpret_cast<uintptr_t>(image)) & 15)
    return MNF_ERR_UNSUPPORTED;  // float4 accesses need 16-byte aligned bases
#define X(HH, HD) \
  if (dim == 2 * HH && hid == HD) \
    return launch<HH, HD>(x, y, log_det, ysq, accumulate, image, rows, parity != 0, inverse != 0, stream);
  MNF_AHF_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf

extern "C" {

int64_t mnf_affine_half_image_floats(int dim, int n_hidden, const int* hidden, int has_scale,
                                     int has_shift) {
  int hid = 0;
  if ((!has_scale && !has_shift) || !mnf::hidden_ok(n_hidden, hidden) ||
      !mnf::uniform_hidden(n_hidden, hidden, hid))
    return 0;
  const int h = dim / 2, hp = (dim & 1) ? 0 : mnf::ahf_padded_half(h);
  if (h != hp && hid != 24 && hid != 16 && hid != 32) return 0;  // a narrow half only has the stack kernel
#define X(HH, HD) \
  if (hp == HH && hid == HD) return mnf::AhfShape<HH, HD>::IMAGE_FLOATS;
  MNF_AHF_SHAPES(X)
#undef X
  return 0;
}

int mnf_affine_half_image_index(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                int32_t* idx_host) {
  int hid = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if ((!has_scale && !has_shift) || !mnf::uniform_hidden(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  const int h = dim / 2, hp = (dim & 1) ? 0 : mnf::ahf_padded_half(h);
  if (h != hp && hid != 24 && hid != 16 && hid != 32) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                           \
  if (hp == HH && hid == HD) {              \
    mnf::build_index<HH, HD>(idx_host, h, has_scale != 0, has_shift != 0, hidden);  \
    return MNF_OK;                          \
  }
  MNF_AHF_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
