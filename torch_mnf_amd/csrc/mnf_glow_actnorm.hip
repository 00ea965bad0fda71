// Glow.inverse followed by ActNormFlow.inverse -- the pair every [ActNormFlow, Glow, NSF_CL] block applies after its
// spline layer on the way x -> z, i.e. in every log_prob / training pass (torch_mnf/flows/glow.py:33-37,
// affine_constant_flow.py:22-26) -- as ONE forward launch and ONE gradient launch for d = 16, 32 or 64:
//
//   forward    z = (u @ M - t) e^-s                                  M = W^-1 (Glow's assembled inverse)
//   gradients  g_v = g_z e^-s      g_u = g_v @ M^T      g_M = u^T g_v
//              g_s = -sum_r g_z z  g_t = -sum_r g_v                   (z is recomputed from u: never read)
//
// As separate layers the training step moved 14 d floats per row through HBM for this pair (ActNorm and Glow each
// read and write the rows forward, then Glow's two gradient launches and ActNorm's one read 2 + 2 + 2 arrays and write
// 2); fused it is 2 d forward (u in, z out) and 3 d backward (u, g_z in, g_u out).  Both launches are HBM bound; the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products) do all three products.
//
// Two register layouts of the same rows meet in the gradient kernel:
//   * "row on the lane" (mnf_linear_mfma.hip): lane (j, q) holds dims 16 g + 4 q + e of row j of a 16-row tile as
//     float4s; u @ M and g_v @ M^T run in it, and so do the elementwise gradients and the column sums.
//   * "rows on the K axis" (xtg32_mfma_kernel in mnf_backward.hip): lane (c, k) holds dims c and c + 16 of row k of a
//     4-row group; u^T g_v is a sum over rows and needs them on K.  The wave turns its tile through a private 6 KB of
//     LDS (row pitch 48 floats: both the float4 writes and the per-row-group reads are conflict free).  Loading the
//     second layout from memory again was measured first: 106-111 us per launch at 2^20 rows, and SLOWER with more
//     waves resident (139 us at four per SIMD) -- the re-read comes an iteration after the first touch, by which time
//     the other waves' streams have pushed the lines out of the XCD's L2.
// Every load and store of the row loop is issued unconditionally (rows past the end read / rewrite the last row and
// are masked out of the sums): hipcc's vmcnt counts stay exact, the next tile's rows are in flight under this one's
// arithmetic.
#include <hip/hip_runtime.h>

#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGaWaves = 4;

template <int D>
struct GaShape {
  static_assert(D == 16 || D == 32 || D == 64, "a whole number of 16-dim tiles, at most four");
  static constexpr int G = D / 16, NK = D / 4;
  // floats per row of a wave's turn buffer: the four rows 4 g + k of a group land 16 banks apart (48, 80 = 16 mod 64
  // up to order), and a row of float4 writes stays 16-byte aligned
  static constexpr int PITCH = D == 64 ? 80 : 48;
};

template <int G>
__device__ __forceinline__ void zero_tiles(f32x4 (&a)[G]) {
#pragma unroll
  for (int m = 0; m < G; ++m) a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc[m] += image (A operand order of mnf_linear_rows_image_index) x rows held as xv (B operand)
template <int D>
__device__ __forceinline__ void rows_times_image(const float* lds_image, int lane, const f32x4 (&xv)[D / 16],
                                                 f32x4 (&acc)[D / 16]) {
  constexpr int G = D / 16, NK = D / 4;
  int a_off = lane * 4;
  asm volatile("" : "+v"(a_off));  // keep the operand reads in the loop (see mnf_ahf_mfma.hip)
  const f32x4* A4 = reinterpret_cast<const f32x4*>(lds_image + a_off);
  int n = 0;
  f32x4 a4;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int m = 0; m < G; ++m) {
      if ((n & 3) == 0) a4 = A4[64 * (n >> 2)];
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[n & 3], xv[kk >> 2][kk & 3], acc[m], 0, 0, 0);
      ++n;
    }
}

// M (row-major, or its transpose when T) into LDS in A-operand order -- the layout mnf_linear_rows_image_index describes:
// word (n >> 2) * 256 + lane * 4 + (n & 3) of MFMA n = (K-step kk, output tile m) is M[16 g + 4 (lane >> 4) + e][16 m +
// (lane & 15)] with kk = 4 g + e.  Done here, per workgroup, from the few KB of the matrix in L2: no packing launch per step.
template <int D, bool T>
__device__ __forceinline__ void stage_matrix(float* lds, const float* M) {
  constexpr int G = D / 16;
  for (int p = threadIdx.x; p < D * D; p += blockDim.x) {
    const int n = 4 * (p >> 8) + (p & 3), lane = (p >> 2) & 63;
    const int kk = n / G, m = n - kk * G;
    const int k = 16 * (kk >> 2) + 4 * (lane >> 4) + (kk & 3), o = 16 * m + (lane & 15);
    lds[p] = T ? M[o * D + k] : M[k * D + o];
  }
}

// [e^-s | t] of the lane's columns 16 m + 4 q + r
template <int G>
__device__ __forceinline__ void load_post(const float* s, const float* t, int q, f32x4 (&es)[G], f32x4 (&tt)[G]) {
#pragma unroll
  for (int m = 0; m < G; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // (scalar loads: s and t may sit anywhere in a flat parameter buffer)
      es[m][r] = expf(-s[16 * m + 4 * q + r]);
      tt[m][r] = t[16 * m + 4 * q + r];
    }
}

template <int G>
__device__ __forceinline__ void load_row(const float* p, f32x4 (&v)[G]) {
#pragma unroll
  for (int m = 0; m < G; ++m) v[m] = *reinterpret_cast<const f32x4*>(p + 16 * m);
}

// z = (u @ M - t) e^-s.  LP (the pair closes a density pass under a standard-normal base, core.py:46-49): z is not
// stored; the launch writes log p(row) = log_det_rows[row] + (ld_glow - sum s) - |z|^2 / 2 - d log(2 pi) / 2 instead.
template <int D, bool LP>
__global__ void __launch_bounds__(kGaWaves * 64)
glow_actnorm_inv_kernel(const float* __restrict__ u, const float* __restrict__ M, const float* __restrict__ s,
                        const float* __restrict__ t, float* __restrict__ z, const float* __restrict__ ld_glow,
                        float* __restrict__ ld_out, const float* __restrict__ ld_rows, float* __restrict__ lp_out,
                        int64_t rows) {
  constexpr int G = D / 16;
  __shared__ __attribute__((aligned(16))) float lds[D * D];
  __shared__ float ld_pair;
  stage_matrix<D, false>(lds, M);
  if (threadIdx.x < 64) {  // the pair's log|det J|: Glow's (given) - sum s
    float v = threadIdx.x < D ? -s[threadIdx.x] : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    v += ld_glow ? ld_glow[0] : 0.f;
    if (threadIdx.x == 0) {
      ld_pair = v;
      if (ld_out && blockIdx.x == 0) ld_out[0] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  f32x4 es[G], tt[G];
  load_post<G>(s, t, q, es, tt);
  const int n_tiles = (int)((rows + 15) >> 4), step = (int)gridDim.x * kGaWaves;
  auto row_of = [&](int tile) {
    const int64_t row = (int64_t)tile * 16 + j;
    return row < rows ? row : rows - 1;
  };
  int tile = (int)blockIdx.x * kGaWaves + wave;
  if (tile >= n_tiles) return;
  const float ldc = ld_pair - (float)D * kHalfLog2Pi;
  f32x4 nx[G];
  float nl = 0.f;
  load_row<G>(u + row_of(tile) * D + 4 * q, nx);
  if (LP) nl = ld_rows[row_of(tile)];
  for (; tile < n_tiles; tile += step) {
    f32x4 xv[G];
#pragma unroll
    for (int m = 0; m < G; ++m) xv[m] = nx[m];
    const float ldr = nl;
    {  // the next tile's rows (past the end: the last tile's again)
      const int nt = tile + step < n_tiles ? tile + step : n_tiles - 1;
      load_row<G>(u + row_of(nt) * D + 4 * q, nx);
      if (LP) nl = ld_rows[row_of(nt)];
    }
    f32x4 acc[G];
    zero_tiles<G>(acc);
    rows_times_image<D>(lds, lane, xv, acc);
    if (LP) {
      float sq = 0.f;
#pragma unroll
      for (int m = 0; m < G; ++m) {
        const f32x4 zz = (acc[m] - tt[m]) * es[m];
        sq += (zz[0] * zz[0] + zz[1] * zz[1]) + (zz[2] * zz[2] + zz[3] * zz[3]);
      }
      sq = sum_over_q(sq);
      lp_out[row_of(tile)] = ldr + (ldc - 0.5f * sq);  // (every q lane of a row, and rows past the end, store the same value)
    } else {
      float* zr = z + row_of(tile) * D + 4 * q;  // (rows past the end rewrite the last row with its own values)
#pragma unroll
      for (int m = 0; m < G; ++m) *reinterpret_cast<f32x4*>(zr + 16 * m) = (acc[m] - tt[m]) * es[m];
    }
  }
}

// LP: gz holds d loss / d log p per ROW; grad_z = -z gz[row] is formed from the recomputed z, and the row sums of gz --
// the cotangent of the pair's log|det J| -- leave through grad_ld_out (added to) and enter grad_s.
template <int D, int WAVES, bool LP>
__global__ void __launch_bounds__(WAVES * 64)
glow_actnorm_inv_bwd_kernel(const float* __restrict__ u, const float* __restrict__ gz, const float* __restrict__ M,
                            const float* __restrict__ s, const float* __restrict__ t, float* __restrict__ gu,
                            float* __restrict__ grad_m, float* __restrict__ grad_s, float* __restrict__ grad_t,
                            const float* __restrict__ grad_ld, float* __restrict__ grad_ld_out, int64_t rows,
                            float* __restrict__ partials) {
  constexpr int G = D / 16, PITCH = GaShape<D>::PITCH;
  __shared__ __attribute__((aligned(16))) float lds_m[D * D], lds_mt[D * D];
  __shared__ __attribute__((aligned(16))) float red[G * G * 256];
  __shared__ __attribute__((aligned(16))) float turn[WAVES][2][16 * PITCH];
  __shared__ float red_st[2 * D + 1];
  stage_matrix<D, false>(lds_m, M);
  stage_matrix<D, true>(lds_mt, M);
  if (threadIdx.x < 2 * D + 1) red_st[threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;  // row-on-the-lane layout
  const int c = lane & 15, k = lane >> 4;  // rows-on-K layout
  f32x4 es[G], tt[G];
  load_post<G>(s, t, q, es, tt);
  float* const my_u = &turn[wave][0][0];
  float* const my_g = &turn[wave][1][0];
  f32x4 wacc[G][G];  // g_M tiles, summed over this wave's rows
#pragma unroll
  for (int a = 0; a < G; ++a) zero_tiles<G>(wacc[a]);
  f32x4 sacc[G], tacc[G];
  zero_tiles<G>(sacc);
  zero_tiles<G>(tacc);
  const int n_tiles = (int)((rows + 15) >> 4), step = (int)gridDim.x * WAVES;
  auto row_of = [&](int tile) {
    const int64_t row = (int64_t)tile * 16 + j;
    return row < rows ? row : rows - 1;
  };
  int tile = (int)blockIdx.x * WAVES + wave;
  f32x4 nu[G], ng[G];
  float nl = 0.f, lacc = 0.f;
  {
    const int t0 = tile < n_tiles ? tile : n_tiles - 1;
    const int64_t off = row_of(t0) * D + 4 * q;
    load_row<G>(u + off, nu);
    if (LP)
      nl = gz[row_of(t0)];
    else
      load_row<G>(gz + off, ng);
  }
  for (; tile < n_tiles; tile += step) {
    f32x4 uv[G], gv_in[G];
    const float glp = nl;
#pragma unroll
    for (int m = 0; m < G; ++m) {
      uv[m] = nu[m];
      if (!LP) gv_in[m] = ng[m];
    }
    {  // the next tile's rows in the first layout
      const int nt = tile + step < n_tiles ? tile + step : n_tiles - 1;
      const int64_t off = row_of(nt) * D + 4 * q;
      load_row<G>(u + off, nu);
      if (LP)
        nl = gz[row_of(nt)];
      else
        load_row<G>(gz + off, ng);
    }
    const float live = (int64_t)tile * 16 + j < rows ? 1.f : 0.f;
    if (LP && q == 0) lacc += glp * live;  // (one lane per row)
    // z = (u @ M - t) e^-s, recomputed; g_s -= g_z z; g_v = g_z e^-s; g_t -= g_v
    f32x4 acc[G];
    zero_tiles<G>(acc);
    rows_times_image<D>(lds_m, lane, uv, acc);
    f32x4 gv[G];
#pragma unroll
    for (int m = 0; m < G; ++m) {
      const f32x4 zz = (acc[m] - tt[m]) * es[m];
      if (LP) gv_in[m] = zz * -glp;  // d log N(z) / d z = -z
      gv[m] = gv_in[m] * es[m];
      sacc[m] -= gv_in[m] * zz * live;
      tacc[m] -= gv[m] * live;
      // the tile for the second layout (rows past the end: zeros on the u side)
      *reinterpret_cast<f32x4*>(my_u + j * PITCH + 16 * m + 4 * q) = uv[m] * live;
      *reinterpret_cast<f32x4*>(my_g + j * PITCH + 16 * m + 4 * q) = gv[m];
    }
    // g_u = g_v @ M^T
    f32x4 gacc[G];
    zero_tiles<G>(gacc);
    rows_times_image<D>(lds_mt, lane, gv, gacc);
    float* gr = gu + row_of(tile) * D + 4 * q;
#pragma unroll
    for (int m = 0; m < G; ++m) *reinterpret_cast<f32x4*>(gr + 16 * m) = gacc[m];
    // g_M += u^T g_v, 4 rows per MFMA (LDS operations of a wave run in order: its reads see its lanes' writes)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int at = (4 * g + k) * PITCH + c;
      float ua[G], ga[G];
#pragma unroll
      for (int a = 0; a < G; ++a) ua[a] = my_u[at + 16 * a], ga[a] = my_g[at + 16 * a];
#pragma unroll
      for (int a = 0; a < G; ++a)
#pragma unroll
        for (int b = 0; b < G; ++b)
          wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[a], ga[b], wacc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // column sums: the 16 row lanes of a q group hold the same columns -- a fixed-order tree over those lanes, then the
  // waves add in turn (below): no LDS atomics, a workgroup's sums repeat bit for bit
#pragma unroll
  for (int m = 0; m < G; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        sacc[m][r] += __shfl_xor(sacc[m][r], off, 64);
        tacc[m][r] += __shfl_xor(tacc[m][r], off, 64);
      }
  if (LP) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lacc += __shfl_xor(lacc, m, 64);
  }
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int tl = 0; tl < G * G; ++tl) {
        f32x4* p = reinterpret_cast<f32x4*>(red + tl * 256 + lane * 4);
        *p = w == 0 ? wacc[tl / G][tl % G] : *p + wacc[tl / G][tl % G];
      }
      if (j == 0) {
#pragma unroll
        for (int m = 0; m < G; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            red_st[16 * m + 4 * q + r] += sacc[m][r];
            red_st[D + 16 * m + 4 * q + r] += tacc[m][r];
          }
      }
      if (LP && lane == 0) red_st[2 * D] += lacc;
    }
    __syncthreads();
  }
  if (partials) {
    // two-stage flush (MNF_DETERMINISTIC=1): this workgroup's sums as one block -- [g_M in destination order | s | t | l] --,
    // glow_actnorm_reduce_kernel adds the blocks up in a fixed order
    float* dst = partials + (int64_t)blockIdx.x * (D * D + 2 * D + 1);
    for (int e = threadIdx.x; e < G * G * 256; e += blockDim.x) {
      const int tl = e >> 8, l = (e >> 2) & 63, reg = e & 3;
      const int i = 16 * (tl / G) + 4 * (l >> 4) + reg, jj = 16 * (tl % G) + (l & 15);
      dst[i * D + jj] = red[e];
    }
    if (threadIdx.x < 2 * D + 1) dst[D * D + threadIdx.x] = red_st[threadIdx.x];
    return;
  }
  for (int e = threadIdx.x; e < G * G * 256; e += blockDim.x) {
    const int tl = e >> 8, l = (e >> 2) & 63, reg = e & 3;
    const int i = 16 * (tl / G) + 4 * (l >> 4) + reg, jj = 16 * (tl % G) + (l & 15);
    atomicAdd(grad_m + i * D + jj, red[e]);
  }
  if (threadIdx.x < D) {
    // (d log|det J| / d s = -1 per column: once, from the first workgroup)
    if (grad_s)
      atomicAdd(grad_s + threadIdx.x, red_st[threadIdx.x] - (grad_ld && blockIdx.x == 0 ? grad_ld[0] : 0.f) -
                                          (LP ? red_st[2 * D] : 0.f));
    if (grad_t) atomicAdd(grad_t + threadIdx.x, red_st[D + threadIdx.x]);
    if (LP && grad_ld_out && threadIdx.x == 0) atomicAdd(grad_ld_out, red_st[2 * D]);
  }
}

// second stage of the deterministic flush: entry i of every workgroup's block, added up in block order
__global__ void __launch_bounds__(256) glow_actnorm_reduce_kernel(const float* __restrict__ partials, int n_blocks, int D,
                                                                  float* __restrict__ grad_m, float* __restrict__ grad_s,
                                                                  float* __restrict__ grad_t,
                                                                  const float* __restrict__ grad_ld,
                                                                  float* __restrict__ grad_ld_out, int lp) {
  const int n = D * D + 2 * D + 1, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto total = [&](int e) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // (four interleaved chains, each in block order)
    int b = 0;
    for (; b + 3 < n_blocks; b += 4) {
      a0 += partials[(int64_t)(b + 0) * n + e];
      a1 += partials[(int64_t)(b + 1) * n + e];
      a2 += partials[(int64_t)(b + 2) * n + e];
      a3 += partials[(int64_t)(b + 3) * n + e];
    }
    for (; b < n_blocks; ++b) a0 += partials[(int64_t)b * n + e];
    return (a0 + a1) + (a2 + a3);
  };
  const float v = total(i);
  if (i < D * D) {
    grad_m[i] += v;
  } else if (i < D * D + D) {
    if (grad_s) grad_s[i - D * D] += v - (grad_ld ? grad_ld[0] : 0.f) - (lp ? total(n - 1) : 0.f);
  } else if (i < D * D + 2 * D) {
    if (grad_t) grad_t[i - D * D - D] += v;
  } else if (lp && grad_ld_out) {
    grad_ld_out[0] += v;
  }
}

int64_t grid_for_tiles(int64_t rows, int per_cu, int waves = kGaWaves) {
  const int64_t n_tiles = (rows + 15) / 16;
  int64_t blocks = (n_tiles + waves - 1) / waves;
  const int64_t cap = (int64_t)per_cu * device_cus(current_device());
  return blocks > cap ? cap : blocks;
}

bool aligned16(const void* a, const void* b, const void* c2) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c2)) & 15) == 0;
}

#ifndef MNF_GA_BWD_WAVES
#define MNF_GA_BWD_WAVES 4
#endif
#ifndef MNF_GA_BWD_PER_CU
#define MNF_GA_BWD_PER_CU 2
#endif

template <int D, bool LP>
int launch_fwd(const float* u, const float* M, const float* s, const float* t, float* z, const float* ld_glow, float* ld_out,
               const float* ld_rows, float* lp_out, int64_t rows, hipStream_t stream) {
  tag_kernel("glow_actnorm_inv");
  hipLaunchKernelGGL((glow_actnorm_inv_kernel<D, LP>), dim3((unsigned)grid_for_tiles(rows, 8)), dim3(kGaWaves * 64), 0,
                     stream, u, M, s, t, z, ld_glow, ld_out, ld_rows, lp_out, rows);
  return check_launch();
}

// eight waves per CU measured best at d = 32, 2^20 rows (workgroups of 4 waves: 1 per CU 95 us, 2: 89, 3: 95, 4: 101; 8
// waves x 2: 101, 16 x 1: 101 -- the same 8 waves as 8 x 1: 90, so it is not the count of closing atomics; the operand
// images held in registers instead of re-read from LDS per tile: 101); d = 64 holds 88 KB of LDS: one workgroup per CU
template <int D, bool LP>
int launch_bwd(const float* u, const float* g, const float* M, const float* s, const float* t, float* grad_u, float* grad_m,
               float* grad_s, float* grad_t, const float* grad_ld, float* grad_ld_out, int64_t rows, hipStream_t stream,
               float* workspace = nullptr, int64_t workspace_floats = 0) {
  tag_kernel("glow_actnorm_inv_bwd");
  const int64_t grid = grid_for_tiles(rows, D == 64 ? 1 : MNF_GA_BWD_PER_CU, MNF_GA_BWD_WAVES);
  constexpr int N = D * D + 2 * D + 1;
  if (workspace && workspace_floats < grid * N) return MNF_ERR_INVALID_ARG;  // (not: a silent fall-back to the atomic flush)
  float* partials = workspace;
  hipLaunchKernelGGL((glow_actnorm_inv_bwd_kernel<D, MNF_GA_BWD_WAVES, LP>), dim3((unsigned)grid),
                     dim3(MNF_GA_BWD_WAVES * 64), 0, stream, u, g, M, s, t, grad_u, grad_m, grad_s, grad_t, grad_ld,
                     grad_ld_out, rows, partials);
  if (int rc = check_launch()) return rc;
  if (!partials) return MNF_OK;
  hipLaunchKernelGGL(glow_actnorm_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, partials, (int)grid, D, grad_m,
                     grad_s, grad_t, grad_ld, grad_ld_out, LP ? 1 : 0);
  return check_launch();
}

bool dim_ok(int dim) { return dim == 16 || dim == 32 || dim == 64; }

}  // namespace
}  // namespace mnf

using namespace mnf;

#define MNF_GA_DISPATCH(call16, call32, call64) (dim == 16 ? (call16) : dim == 32 ? (call32) : (call64))

extern "C" {

int mnf_glow_actnorm_inv(const float* u, const float* M, const float* s, const float* t, float* z, const float* ld_glow,
                         float* ld_out, int64_t rows, int dim, void* stream) {
  if (!u || !M || !s || !t || !z || u == z || rows < 0) return MNF_ERR_INVALID_ARG;
  if (!dim_ok(dim) || !aligned16(u, z, z) || rows >= (int64_t)1 << 31) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  hipStream_t st = (hipStream_t)stream;
  const float* no_rows = nullptr;
  float* no_lp = nullptr;
  return MNF_GA_DISPATCH((launch_fwd<16, false>(u, M, s, t, z, ld_glow, ld_out, no_rows, no_lp, rows, st)),
                         (launch_fwd<32, false>(u, M, s, t, z, ld_glow, ld_out, no_rows, no_lp, rows, st)),
                         (launch_fwd<64, false>(u, M, s, t, z, ld_glow, ld_out, no_rows, no_lp, rows, st)));
}

int mnf_glow_actnorm_inv_logprob(const float* u, const float* M, const float* s, const float* t, const float* ld_glow,
                                 const float* log_det_rows, float* log_prob, int64_t rows, int dim, void* stream) {
  if (!u || !M || !s || !t || !log_det_rows || !log_prob || rows < 0) return MNF_ERR_INVALID_ARG;
  if (!dim_ok(dim) || !aligned16(u, u, u) || rows >= (int64_t)1 << 31) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  hipStream_t st = (hipStream_t)stream;
  float* no_z = nullptr;
  float* no_ld = nullptr;
  return MNF_GA_DISPATCH((launch_fwd<16, true>(u, M, s, t, no_z, ld_glow, no_ld, log_det_rows, log_prob, rows, st)),
                         (launch_fwd<32, true>(u, M, s, t, no_z, ld_glow, no_ld, log_det_rows, log_prob, rows, st)),
                         (launch_fwd<64, true>(u, M, s, t, no_z, ld_glow, no_ld, log_det_rows, log_prob, rows, st)));
}

int64_t mnf_glow_actnorm_inv_bwd_workspace(int64_t rows, int dim) {
  if (rows < 0 || !dim_ok(dim)) return 0;
  return grid_for_tiles(rows, dim == 64 ? 1 : MNF_GA_BWD_PER_CU, MNF_GA_BWD_WAVES) * ((int64_t)dim * dim + 2 * dim + 1);
}

int mnf_glow_actnorm_inv_bwd(const float* u, const float* grad_z, const float* M, const float* s, const float* t,
                             float* grad_u, float* grad_m, float* grad_s, float* grad_t, const float* grad_ld,
                             int64_t rows, int dim, void* stream) {
  return mnf_glow_actnorm_inv_bwd_det(u, grad_z, M, s, t, grad_u, grad_m, grad_s, grad_t, grad_ld, rows, dim, nullptr, 0,
                                      stream);
}

int mnf_glow_actnorm_inv_bwd_det(const float* u, const float* grad_z, const float* M, const float* s, const float* t,
                                 float* grad_u, float* grad_m, float* grad_s, float* grad_t, const float* grad_ld,
                                 int64_t rows, int dim, float* workspace, int64_t workspace_floats, void* stream) {
  if (!u || !grad_z || !M || !s || !t || !grad_u || !grad_m || grad_u == u || grad_u == grad_z || rows < 0)
    return MNF_ERR_INVALID_ARG;
  if (!dim_ok(dim) || !aligned16(u, grad_z, grad_u) || rows >= (int64_t)1 << 31) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  hipStream_t st = (hipStream_t)stream;
  float* no_out = nullptr;
  return MNF_GA_DISPATCH(
      (launch_bwd<16, false>(u, grad_z, M, s, t, grad_u, grad_m, grad_s, grad_t, grad_ld, no_out, rows, st, workspace, workspace_floats)),
      (launch_bwd<32, false>(u, grad_z, M, s, t, grad_u, grad_m, grad_s, grad_t, grad_ld, no_out, rows, st, workspace, workspace_floats)),
      (launch_bwd<64, false>(u, grad_z, M, s, t, grad_u, grad_m, grad_s, grad_t, grad_ld, no_out, rows, st, workspace, workspace_floats)));
}

int mnf_glow_actnorm_inv_logprob_bwd(const float* u, const float* grad_log_prob, const float* M, const float* s,
                                     const float* t, float* grad_u, float* grad_m, float* grad_s, float* grad_t,
                                     float* grad_ld_glow, int64_t rows, int dim, void* stream) {
  return mnf_glow_actnorm_inv_logprob_bwd_det(u, grad_log_prob, M, s, t, grad_u, grad_m, grad_s, grad_t, grad_ld_glow, rows,
                                              dim, nullptr, 0, stream);
}

int mnf_glow_actnorm_inv_logprob_bwd_det(const float* u, const float* grad_log_prob, const float* M, const float* s,
                                         const float* t, float* grad_u, float* grad_m, float* grad_s, float* grad_t,
                                         float* grad_ld_glow, int64_t rows, int dim, float* workspace,
                                         int64_t workspace_floats, void* stream) {
  if (!u || !grad_log_prob || !M || !s || !t || !grad_u || !grad_m || grad_u == u || rows < 0) return MNF_ERR_INVALID_ARG;
  if (!dim_ok(dim) || !aligned16(u, grad_u, grad_u) || rows >= (int64_t)1 << 31) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  hipStream_t st = (hipStream_t)stream;
  const float* no_ld = nullptr;
  return MNF_GA_DISPATCH(
      (launch_bwd<16, true>(u, grad_log_prob, M, s, t, grad_u, grad_m, grad_s, grad_t, no_ld, grad_ld_glow, rows, st, workspace, workspace_floats)),
      (launch_bwd<32, true>(u, grad_log_prob, M, s, t, grad_u, grad_m, grad_s, grad_t, no_ld, grad_ld_glow, rows, st, workspace, workspace_floats)),
      (launch_bwd<64, true>(u, grad_log_prob, M, s, t, grad_u, grad_m, grad_s, grad_t, no_ld, grad_ld_glow, rows, st, workspace, workspace_floats)));
}

}  // extern "C"
