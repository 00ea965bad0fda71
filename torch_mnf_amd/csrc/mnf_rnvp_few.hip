// RNVP on a handful of rows (torch_mnf/flows/rnvp.py:25-39, one hidden layer): what the MNF layers' kl_div runs -- ONE
// row through flow_q and flow_r per layer per step (mnf_linear.py:67/84, mnf_conv.py:101/127) -- and what MNFConv2d's
// sample_z runs in forward (mnf_conv.py:90-98).  There the streaming kernels are all latency: the matrix-core kernel
// spends 86 us on one row of 800 dims (a tile of 16 rows, 15 of them padding, one wave walking all dims), the generic
// gradient kernel 150 us (50 threads each walking 800 dims; 120,000 atomics from one workgroup).
//
// Here one workgroup of 1,024 threads takes ALL rows (one or two) and every phase is laid out for coalesced weight reads:
//   y = W1 kept + b1        a wave per hidden unit, lanes along the dims (W1 is (h, d): contiguous), wave reduction
//   t, s = Wt y + bt, ...   a thread per dim (Wt is (d, h): the thread's own 200-byte row), all rows per weight load
//   gradients               one workgroup owns every parameter gradient: plain read-modify-write into grad_flat, no
//                           atomics; d W1 and the gradient of kept read / write along the dims (coalesced)
// Every sum is taken in a fixed order: results repeat bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mnf_hip.h"
#include "mnf_host.h"

namespace mnf {

#ifndef MNF_FEW_ABL  // timing experiments: 1 stop after the hidden layer, 2 no hidden layer, 4 no x / grad stores
#define MNF_FEW_ABL 0
#endif
#ifndef MNF_FEW_THREADS
#define MNF_FEW_THREADS 1024
#endif
constexpr int kFewThreads = MNF_FEW_THREADS;
constexpr int kFewWaves = kFewThreads / 64;
constexpr int kFewRows = MNF_RNVP_FEW_ROWS;
constexpr int kFewFwdRows = MNF_RNVP_FEW_FWD_ROWS;
constexpr int kFewFwdRowsSeeded = 64;
constexpr int kFewBwdRows = MNF_RNVP_FEW_BWD_ROWS;
constexpr int kFewLdsFloats = 36 * 1024;  // 144 KB

struct RnvpFewArgs {
  const float* z;
  const float* mask;
  uint64_t seed;
  float* x;
  float* log_det;
  int accumulate;
  const float* grad_x;
  const float* grad_ld;
  float* grad_z;
  float* grad_flat;
  const float* flat;
  int rows, dim, hid;
  int64_t row0;  // absolute index of this workgroup's first row (the in-kernel mask is a function of it)
  float* partial;  // gradients launched as a grid: workgroup w WRITES its parameter gradients to partial + w * n_par
  int64_t n_par;
};

// Sum over the wave, returned to every lane, on the DPP path (row-internal butterflies, row broadcasts, one readlane): 9
// VALU-rate instructions instead of six ds_bpermute round trips through the LDS crossbar -- the per-dim phase takes two
// such sums per (row, dim).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float few_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float few_wave_sum(float v) {
  v += few_dpp<0xB1>(v);         // quad_perm [1,0,3,2]
  v += few_dpp<0x4E>(v);         // quad_perm [2,3,0,1]
  v += few_dpp<0x141>(v);        // row_half_mirror
  v += few_dpp<0x140>(v);        // row_mirror: every lane holds its row's sum
  v += few_dpp<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v += few_dpp<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// kept = mask z into LDS; y = W1 kept + b1 into LDS ([R][hid]); both published by the trailing barrier
template <int R>
__device__ __forceinline__ void few_hidden(const RnvpFewArgs& a, float* kept, float* y) {
  const int d = a.dim, h = a.hid, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < R * d; idx += kFewThreads) {
    const int r = idx / d, j = idx - r * d;
    float v = 0.f;  // (rows past the end of the batch: zeros, their cotangents are zero too)
    if (r < a.rows) v = (a.mask ? a.mask[idx] : rnvp_mask_bit(a.seed, a.row0 + r, j)) * a.z[idx];
    kept[idx] = v;
  }
  __syncthreads();
  const float* W1 = a.flat;
  const float* b1 = a.flat + (size_t)h * d;
  // a wave takes 4 hidden units at a time and 4 x 64 dims per step: 16 independent weight loads in flight per lane
  // (issued one at a time, each load's round trip to L2 was the kernel)
  for (int kb = wave; kb < h; kb += 4 * kFewWaves) {
    float acc[4][R];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[u][r] = 0.f;
    for (int j0 = lane; j0 < d; j0 += 256) {
      float w[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = kb + kFewWaves * u, j = j0 + 64 * v;
          w[u][v] = (k < h && j < d) ? W1[(size_t)k * d + j] : 0.f;
        }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int j = j0 + 64 * v;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float kp = j < d ? kept[r * d + j] : 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u][r] = fmaf(w[u][v], kp, acc[u][r]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + kFewWaves * u;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float sum = few_wave_sum(acc[u][r]);
        if (lane == 0 && k < h) y[r * h + k] = sum + b1[k];
      }
    }
  }
  __syncthreads();
}

constexpr int kFewJB = 8;  // dims a wave has in flight per buffer in the lanes-along-the-hidden-units passes

// Pass A, lanes along the hidden units (h <= 64): wave w takes dims j = w, w + 16, ...; lane k reads Wt[j][k], Ws[j][k] --
// the dim's 200-byte rows, coalesced (a thread per dim instead reads 64 different cache lines per instruction: 64
// cycles each in the texture unit, 35 us for one row of 800 dims) -- and a DPP wave sum gives t[r][j], s[r][j] without
// their biases, which lane 0 leaves in LDS for the elementwise pass (a thread per dim: no redundant transcendentals).
// Two register buffers of 8 dims each: the next buffer's loads are in flight while this one is summed.
template <int R>
__device__ __forceinline__ void few_shift_scale(const RnvpFewArgs& a, const float* y, const float* Wt, const float* Ws,
                                                float* sh, float* sc) {
  const int d = a.dim, h = a.hid, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool kin = lane < h;
  constexpr int step = kFewWaves * kFewJB;
  float yk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) yk[r] = kin ? y[r * h + lane] : 0.f;
  auto load = [&](float (&wt)[kFewJB], float (&ws)[kFewJB], int j0) {
#pragma unroll
    for (int u = 0; u < kFewJB; ++u) {
      const int j = j0 + kFewWaves * u;
      const bool in = kin && j < d;
      wt[u] = in ? Wt[(size_t)j * h + lane] : 0.f;
      ws[u] = in ? Ws[(size_t)j * h + lane] : 0.f;
    }
  };
  auto sums = [&](const float (&wt)[kFewJB], const float (&ws)[kFewJB], int j0) {
#pragma unroll
    for (int u = 0; u < kFewJB; ++u) {
      const int j = j0 + kFewWaves * u;
      if (j < d) {  // (wave-uniform)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float t = few_wave_sum(wt[u] * yk[r]), sv = few_wave_sum(ws[u] * yk[r]);
          if (lane == 0) sh[r * d + j] = t, sc[r * d + j] = sv;
        }
      }
    }
  };
  float wtA[kFewJB], wsA[kFewJB], wtB[kFewJB], wsB[kFewJB];
  load(wtA, wsA, wave);
  for (int j0 = wave; j0 < d; j0 += 2 * step) {
    load(wtB, wsB, j0 + step);
    sums(wtA, wsA, j0);
    load(wtA, wsA, j0 + 2 * step);
    sums(wtB, wsB, j0 + step);
  }
  __syncthreads();
}

// (forward only: also launched as a grid, a workgroup per R rows, for batches of up to a few hundred rows -- the reference's
//  training batch is 128 -- where the matrix-core kernel has 8 waves on the whole chip walking all dims: 103 us)
template <int R>
__global__ void __launch_bounds__(kFewThreads) rnvp_few_fwd_kernel(const RnvpFewArgs a_in) {
  extern __shared__ float few_lds[];
  RnvpFewArgs a = a_in;
  {
    const int64_t row0 = (int64_t)blockIdx.x * R;
    a.row0 = row0;
    a.rows = (int)min((int64_t)R, (int64_t)a_in.rows - row0);
    a.z += row0 * a.dim;
    a.x += row0 * a.dim;
    if (a.mask) a.mask += row0 * a.dim;
    if (a.log_det) a.log_det += row0;
  }
  const int d = a.dim, h = a.hid, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* kept = few_lds;          // [R][d]
  float* sh = kept + R * d;       // [R][d]  shift, scale without their biases
  float* sc = sh + R * d;
  float* y = sc + R * d;          // [R][h]
  float* red = y + R * h;         // [waves][R]
  if (!(MNF_FEW_ABL & 2)) few_hidden<R>(a, kept, y);
  if (MNF_FEW_ABL & 1) return;
  const float* Wt = a.flat + (size_t)h * d + h;
  const float* bt = Wt + (size_t)h * d;
  const float* Ws = bt + d;
  const float* bs = Ws + (size_t)h * d;
  few_shift_scale<R>(a, y, Wt, Ws, sh, sc);
  float lad[R];
#pragma unroll
  for (int r = 0; r < R; ++r) lad[r] = 0.f;
  for (int j = tid; j < d; j += kFewThreads) {
    const float b_t = bt[j], b_s = bs[j];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < a.rows) {
        const int idx = r * d + j;
        const float m = a.mask ? a.mask[idx] : rnvp_mask_bit(a.seed, a.row0 + r, j), z = a.z[idx];
        const float gate = sigmoidf(sc[idx] + b_s), shift = sh[idx] + b_t;
        a.x[idx] = ((1.f - m) * z * gate + (1.f - gate) * shift) + m * z;  // rnvp.py:37
        lad[r] += (1.f - m) * logf(gate);                                  // :36 (0 * -inf = NaN, as there)
      }
    }
  }
  if (a.log_det) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float sum = few_wave_sum(lad[r]);
      if (lane == 0) red[wave * R + r] = sum;
    }
    __syncthreads();
    if (tid < a.rows) {
      float sum = 0.f;
      for (int w = 0; w < kFewWaves; ++w) sum += red[w * R + tid];
      a.log_det[tid] = a.accumulate ? a.log_det[tid] + sum : sum;
    }
  }
}

// Launched as a grid (a workgroup per R rows) for batch-sized calls, every workgroup WRITES its own copy of the parameter
// gradients (each parameter exactly once: no zeroing, no read-modify-write) and rnvp_few_reduce_kernel adds the copies
// to grad_flat; launched as one workgroup it adds to grad_flat itself.
template <int R>
__global__ void __launch_bounds__(kFewThreads) rnvp_few_bwd_kernel(const RnvpFewArgs a_in) {
  extern __shared__ float few_lds[];
  RnvpFewArgs a = a_in;
  const bool rmw = a_in.partial == nullptr;  // add to what the buffer holds (one workgroup) or write (grid)
  {
    const int64_t row0 = (int64_t)blockIdx.x * R;
    a.row0 = row0;
    a.rows = (int)min((int64_t)R, (int64_t)a_in.rows - row0);
    a.z += row0 * a.dim;
    a.grad_z += row0 * a.dim;
    if (a.mask) a.mask += row0 * a.dim;
    if (a.grad_x) a.grad_x += row0 * a.dim;
    if (a.grad_ld) a.grad_ld += row0;
    if (a_in.partial) a.grad_flat = a_in.partial + (int64_t)blockIdx.x * a_in.n_par;
  }
  const int d = a.dim, h = a.hid, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* kept = few_lds;                  // [R][d]
  float* g_t = kept + R * d;              // [R][d]  shift (pass A), then its cotangent (pass B)
  float* g_s = g_t + R * d;               // [R][d]  scale, then its cotangent
  float* y = g_s + R * d;                 // [R][h]
  float* dA = y + R * h;                  // [R][h]   gradient wrt y
  float* part = dA + R * h;               // [waves][R][64]  the waves' shares of it; later [groups][R][width] of g_kept
  few_hidden<R>(a, kept, y);
  const size_t w1 = 0, b1o = (size_t)h * d, tw = b1o + h, tb = tw + (size_t)h * d, sw = tb + d, sb = sw + (size_t)h * d;
  const float* Wt = a.flat + tw;
  const float* Ws = a.flat + sw;
  float* gf = a.grad_flat;
  few_shift_scale<R>(a, y, Wt, Ws, g_t, g_s);
  // pass B, a thread per dim: cotangents of shift and scale, the direct part of grad_z, d bt, d bs
  for (int j = tid; j < d; j += kFewThreads) {
    const float b_t = a.flat[tb + j], b_s = a.flat[sb + j];
    float sum_t = 0.f, sum_s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float gt = 0.f, gs = 0.f;
      const int idx = r * d + j;
      if (r < a.rows) {
        const float m = a.mask ? a.mask[idx] : rnvp_mask_bit(a.seed, a.row0 + r, j), z = a.z[idx];
        const float G = a.grad_x ? a.grad_x[idx] : 0.f, gl = a.grad_ld ? a.grad_ld[r] : 0.f;
        const float gate = sigmoidf(g_s[idx] + b_s), shift = g_t[idx] + b_t;
        // x = (1-m) z g + (1-g) t + m z ; ld = sum (1-m) log g
        gt = G * (1.f - gate);
        gs = (G * ((1.f - m) * z - shift) * gate + gl * (1.f - m)) * (1.f - gate);
        a.grad_z[idx] = G * ((1.f - m) * gate + m);  // (+ m g_kept in the last pass)
      }
      g_t[idx] = gt;
      g_s[idx] = gs;
      sum_t += gt;
      sum_s += gs;
    }
    if (gf) {
      gf[tb + j] = (rmw ? gf[tb + j] : 0.f) + sum_t;
      gf[sb + j] = (rmw ? gf[sb + j] : 0.f) + sum_s;
    }
  }
  __syncthreads();
  // pass C, lanes along the hidden units again: d Wt[j][k] += sum_r g_t[r][j] y[r][k] (ditto d Ws), read-modify-write
  // of the dim's coalesced row, and lane k's share of g_y[r][k] = sum_j Wt[j][k] g_t[r][j] + Ws[j][k] g_s[r][j]
  float gy[R];
#pragma unroll
  for (int r = 0; r < R; ++r) gy[r] = 0.f;
  {
    const bool kin = lane < h;
    constexpr int JC = R == 1 ? 4 : 2, step = kFewWaves * JC;  // (R = 2 at JC = 4: 12 registers spilled)
    float yk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) yk[r] = kin ? y[r * h + lane] : 0.f;
    auto load = [&](float (&wt)[JC], float (&ws)[JC], float (&ot)[JC], float (&os)[JC], int j0) {
#pragma unroll
      for (int u = 0; u < JC; ++u) {
        const int j = j0 + kFewWaves * u;
        const bool in = kin && j < d;
        wt[u] = in ? Wt[(size_t)j * h + lane] : 0.f;
        ws[u] = in ? Ws[(size_t)j * h + lane] : 0.f;
        ot[u] = in && gf && rmw ? gf[tw + (size_t)j * h + lane] : 0.f;
        os[u] = in && gf && rmw ? gf[sw + (size_t)j * h + lane] : 0.f;
      }
    };
    auto work = [&](const float (&wt)[JC], const float (&ws)[JC], const float (&ot)[JC], const float (&os)[JC], int j0) {
#pragma unroll
      for (int u = 0; u < JC; ++u) {
        const int j = j0 + kFewWaves * u;
        if (j < d) {
          float dwt = ot[u], dws = os[u];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float gt = g_t[r * d + j], gs = g_s[r * d + j];  // (LDS broadcast reads)
            dwt = fmaf(gt, yk[r], dwt);
            dws = fmaf(gs, yk[r], dws);
            gy[r] = fmaf(wt[u], gt, fmaf(ws[u], gs, gy[r]));
          }
          if (gf && kin) {
            gf[tw + (size_t)j * h + lane] = dwt;
            gf[sw + (size_t)j * h + lane] = dws;
          }
        }
      }
    };
    float wtA[JC], wsA[JC], otA[JC], osA[JC], wtB[JC], wsB[JC], otB[JC], osB[JC];
    load(wtA, wsA, otA, osA, wave);
    for (int j0 = wave; j0 < d; j0 += 2 * step) {
      load(wtB, wsB, otB, osB, j0 + step);
      work(wtA, wsA, otA, osA, j0);
      load(wtA, wsA, otA, osA, j0 + 2 * step);
      work(wtB, wsB, otB, osB, j0 + step);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) part[(wave * R + r) * 64 + lane] = gy[r];
  __syncthreads();
  for (int idx = tid; idx < R * h; idx += kFewThreads) {
    const int r = idx / h, k = idx - r * h;
    float sum = 0.f;
    for (int w = 0; w < kFewWaves; ++w) sum += part[(w * R + r) * 64 + k];
    dA[idx] = sum;
  }
  __syncthreads();
  if (gf && tid < h) {  // y is the net's (linear) output layer: d b1 = sum_r g_y
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) sum += dA[r * h + tid];
    gf[b1o + tid] = (rmw ? gf[b1o + tid] : 0.f) + sum;
  }
  // d W1[k][j] = sum_r g_y[r][k] kept[r][j];  g_kept[r][j] = sum_k W1[k][j] g_y[r][k]: threads (unit group, dim),
  // reads and writes along the dims; a narrow layer spreads the hidden units over the groups
  for (int jb = 0; jb < d; jb += kFewThreads) {
    const int width = min(d - jb, kFewThreads), groups = min(kFewThreads / width, h);
    const int grp = tid / width, jj = tid - grp * width, j = jb + jj;
    float gk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) gk[r] = 0.f;
    if (grp < groups) {
      float kp[R];
#pragma unroll
      for (int r = 0; r < R; ++r) kp[r] = kept[r * d + j];
      for (int k0 = grp; k0 < h; k0 += 8 * groups) {
        float w[8], old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + groups * u;
          w[u] = k < h ? a.flat[w1 + (size_t)k * d + j] : 0.f;
          old[u] = k < h && gf && rmw ? gf[w1 + (size_t)k * d + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + groups * u;
          if (k < h) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const float g = dA[r * h + k];
              gk[r] = fmaf(w[u], g, gk[r]);
              old[u] = fmaf(g, kp[r], old[u]);
            }
            if (gf) gf[w1 + (size_t)k * d + j] = old[u];
          }
        }
      }
    }
    __syncthreads();  // (part: the previous contents are no longer read)
    if (grp < groups) {
#pragma unroll
      for (int r = 0; r < R; ++r) part[(grp * R + r) * width + jj] = gk[r];
    }
    __syncthreads();
    if (tid < width) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < a.rows) {
          float sum = 0.f;
          for (int g = 0; g < groups; ++g) sum += part[(g * R + r) * width + tid];
          const int idx = r * d + j;
          const float m = a.mask ? a.mask[idx] : rnvp_mask_bit(a.seed, a.row0 + r, j);
          a.grad_z[idx] += m * sum;
        }
      }
    }
  }
}

// grad_flat[i] += sum over the workgroups' copies (coalesced: consecutive threads, consecutive parameters)
__global__ void __launch_bounds__(256) rnvp_few_reduce_kernel(const float* __restrict__ partial, int n_copies, int64_t n_par,
                                                              float* __restrict__ grad_flat) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_par) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= n_copies; c += 4) {
    s0 += partial[(int64_t)c * n_par + i];
    s1 += partial[(int64_t)(c + 1) * n_par + i];
    s2 += partial[(int64_t)(c + 2) * n_par + i];
    s3 += partial[(int64_t)(c + 3) * n_par + i];
  }
  for (; c < n_copies; ++c) s0 += partial[(int64_t)c * n_par + i];
  grad_flat[i] += (s0 + s1) + (s2 + s3);
}

static inline int few_rows_class(int64_t rows) { return rows <= 1 ? 1 : 2; }

static bool few_shape_ok(int64_t rows, int64_t max_rows, int dim, int n_hidden, const int* hidden) {
  static const bool off = [] {  // MNF_RNVP_FEW=0: the streaming kernels at every row count (A/B runs)
    const char* e = getenv("MNF_RNVP_FEW");
    return e && e[0] == '0';
  }();
  if (off || rows < 1 || rows > max_rows || n_hidden != 1 || !hidden || hidden[0] < 1 || hidden[0] > 64) return false;
  return 3 * 2 * (int64_t)dim + 2 * 2 * hidden[0] + 2 * 1024 <= kFewLdsFloats;  // (the gradient kernel's layout at R = 2)
}

// gradients: one workgroup owns every parameter gradient -- one or two rows
bool rnvp_few_ok(int64_t rows, int dim, int n_hidden, const int* hidden) {
  return few_shape_ok(rows, kFewRows, dim, n_hidden, hidden);
}

// forward: a workgroup per two rows.  With an explicit mask tensor up to MNF_RNVP_FEW_FWD_ROWS rows (the streaming
// kernel for explicit masks takes 103 us at 128 rows of 800 dims, this one 35 up to 512 rows); with the in-kernel mask
// the register-resident kernels take over at ~100 rows (31 us at 128): up to 64 rows.  (tools/time_rnvp_fwd_rows.py;
// include/mnf_hip.h MNF_RNVP_FEW_FWD_ROWS.)
bool rnvp_few_fwd_ok(int64_t rows, int dim, int n_hidden, const int* hidden, bool explicit_mask) {
  const int64_t max_rows = explicit_mask ? kFewFwdRows : kFewFwdRowsSeeded;
  return few_shape_ok(rows, max_rows, dim, n_hidden, hidden);
}

template <typename K>
static int few_attr(K kernel) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             kFewLdsFloats * (int)sizeof(float)) == hipSuccess
             ? 1
             : -1;
}

int rnvp_few_fwd_launch(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                        const float* flat, int64_t rows, int dim, int hid, hipStream_t stream) {
  RnvpFewArgs a{z, mask, seed, x, log_det, accumulate, nullptr, nullptr, nullptr, nullptr, flat, (int)rows, dim, hid, 0,
                nullptr, 0};
  const int R = few_rows_class(rows);
  const unsigned grid = (unsigned)((rows + R - 1) / R);
  const size_t lds = (3 * (size_t)R * dim + (size_t)R * hid + kFewWaves * R) * sizeof(float);
  static DeviceMemo attrs;
  if (attrs.get([&](int) {
        return few_attr(rnvp_few_fwd_kernel<1>) > 0 && few_attr(rnvp_few_fwd_kernel<2>) > 0 ? 1 : -1;
      }) < 0)
    return MNF_ERR_LAUNCH;
  tag_kernel("rnvp_few");
  switch (R) {
    case 1: hipLaunchKernelGGL(rnvp_few_fwd_kernel<1>, dim3(grid), dim3(kFewThreads), lds, stream, a); break;
    default: hipLaunchKernelGGL(rnvp_few_fwd_kernel<2>, dim3(grid), dim3(kFewThreads), lds, stream, a); break;
  }
  return check_launch();
}

static inline int64_t few_n_par(int dim, int hid) { return 3 * (int64_t)hid * dim + hid + 2 * (int64_t)dim; }

// gradients as a grid: up to MNF_RNVP_FEW_BWD_ROWS rows (include/mnf_hip.h)
bool rnvp_few_bwd_grid_ok(int64_t rows, int dim, int n_hidden, const int* hidden) {
  return rows > kFewRows && few_shape_ok(rows, kFewBwdRows, dim, n_hidden, hidden);
}

// partial == nullptr: one workgroup (rows <= 2) adding to grad_flat; else a workgroup per two rows writing its copy
// of the parameter gradients to partial, summed into grad_flat by a second launch
int rnvp_few_bwd_launch(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                        float* grad_z, float* grad_flat, const float* flat, float* partial, int64_t rows, int dim,
                        int hid, hipStream_t stream) {
  const int64_t n_par = few_n_par(dim, hid);
  RnvpFewArgs a{z, mask, seed, nullptr, nullptr, 0, grad_x, grad_ld, grad_z, grad_flat, flat, (int)rows, dim, hid, 0,
                grad_flat ? partial : nullptr, n_par};
  const int R = partial ? 2 : few_rows_class(rows);
  const unsigned grid = partial ? (unsigned)((rows + 1) / 2) : 1u;
  const size_t lds = (3 * (size_t)R * dim + 2 * (size_t)R * hid + (size_t)R * 1024) * sizeof(float);
  static DeviceMemo attrs;
  if (attrs.get([&](int) {
        return few_attr(rnvp_few_bwd_kernel<1>) > 0 && few_attr(rnvp_few_bwd_kernel<2>) > 0 ? 1 : -1;
      }) < 0)
    return MNF_ERR_LAUNCH;
  tag_kernel("rnvp_bwd_few");
  switch (R) {
    case 1: hipLaunchKernelGGL(rnvp_few_bwd_kernel<1>, dim3(grid), dim3(kFewThreads), lds, stream, a); break;
    default: hipLaunchKernelGGL(rnvp_few_bwd_kernel<2>, dim3(grid), dim3(kFewThreads), lds, stream, a); break;
  }
  if (int rc = check_launch()) return rc;
  if (partial && grad_flat) {
    hipLaunchKernelGGL(rnvp_few_reduce_kernel, dim3((unsigned)((n_par + 255) / 256)), dim3(256), 0, stream, partial,
                       (int)grid, n_par, grad_flat);
    return check_launch();
  }
  return MNF_OK;
}

}  // namespace mnf

extern "C" int mnf_rnvp_few_rows_ok(int64_t rows, int dim, int n_hidden, const int* hidden_host, int explicit_mask) {
  return mnf::rnvp_few_fwd_ok(rows, dim, n_hidden, hidden_host, explicit_mask != 0) ? 1 : 0;
}

extern "C" int64_t mnf_rnvp_bwd_few_workspace_floats(int64_t rows, int dim, int n_hidden, const int* hidden_host) {
  if (!mnf::rnvp_few_bwd_grid_ok(rows, dim, n_hidden, hidden_host)) return 0;
  return ((rows + 1) / 2) * mnf::few_n_par(dim, hidden_host[0]);
}

extern "C" int mnf_rnvp_bwd_few(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                                float* grad_z, float* grad_flat, const float* flat, float* workspace, int64_t rows, int dim,
                                int n_hidden, const int* hidden_host, void* stream) {
  if (!z || !grad_z || !flat || !workspace || rows < 0 || dim < 1) return MNF_ERR_INVALID_ARG;
  if (!mnf::rnvp_few_bwd_grid_ok(rows, dim, n_hidden, hidden_host)) return MNF_ERR_UNSUPPORTED;
  return mnf::rnvp_few_bwd_launch(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, flat, workspace, rows, dim,
                                  hidden_host[0], (hipStream_t)stream);
}
