// Dense layers of the any-shape kernels (mnf_generic.hip, mnf_backward.hip): a 256-thread workgroup holds R rows of
// activations in LDS and walks a Linear layer as R x n_out dot products, one per thread at a time.  The weights used to
// be read where they lie -- W[o][k] with the lanes of a wave along o: a stride of n_in floats between lanes, one L2
// transaction (or, staged as they lie, one 32-way bank conflict) per multiply-add; at 64 hidden units the kernels ran at
// ~2 multiply-adds per clock and CU.  Here a tile of the layer is first copied into LDS in the orientation the loop
// reads it in -- TRANSPOSED, wt[k][o] with an odd row stride, for y = W a (lanes along o), as it lies, wn[o][k], for
// W^T delta (lanes along k) -- so that every weight read is a conflict-free LDS read and the activation read a broadcast.
// Arithmetic is unchanged: the same fmaf chain in the same order per output, so results are bit-identical to the old
// loops (the matrix-core kernels are tested against these).
#pragma once
#include <hip/hip_runtime.h>

#include "mnf_device.h"

namespace mnf {

constexpr int kGemmStageFloats = 4608;  // 18 KB of static LDS per kernel that uses these helpers (60 + 18 KB: two workgroups per CU)
__device__ __forceinline__ float* gemm_stage() {
  __shared__ float stage[kGemmStageFloats];
  return stage;
}

// out[r][o] = act(b[o] + sum_k W[o][k] in[r][k]) for r < R, o < n_out; in: [R][ld_in], out: [R][ld_out] in LDS; W, b
// anywhere.  Ends with a barrier.  (A layer wider than the stage on the K axis falls back to direct weight reads.)
__device__ __forceinline__ void staged_linear(const float* __restrict__ W, const float* __restrict__ b, const float* in,
                                              int ld_in, float* out, int ld_out, int n_in, int n_out, int R, bool act) {
  float* const wt = gemm_stage();
  int tile = n_in <= kGemmStageFloats / 2 ? kGemmStageFloats / n_in : 0;  // outputs per tile, stride forced odd below
  if ((tile & 1) == 0) --tile;
  if (tile > n_out) tile = n_out;
  // a long K axis leaves room for a few outputs per tile only: with few rows in the workgroup (wide layers: the rows'
  // own LDS footprint) a tile would keep a fraction of the threads busy -- RNVP's 800 -> 100 layer measured 252 ns per
  // row staged against 223 direct -- so such a layer reads its weights where they lie
  if (tile < 1 || (tile < n_out && R * tile < 128)) {
    for (int idx = threadIdx.x; idx < R * n_out; idx += blockDim.x) {
      const int r = idx / n_out, o = idx - r * n_out;
      const float* w = W + (size_t)o * n_in;
      const float* a = in + r * ld_in;
      float acc = b[o];
      for (int k = 0; k < n_in; ++k) acc = fmaf(w[k], a[k], acc);
      out[r * ld_out + o] = act ? leaky(acc) : acc;
    }
    __syncthreads();
    return;
  }
  for (int o0 = 0; o0 < n_out; o0 += tile) {
    const int tw = min(tile, n_out - o0), ldt = tw | 1;
    // (ldt = tw | 1 <= tile when tile is odd; tw == tile even only if tile was clipped to n_out: then n_in * (n_out | 1)
    //  may exceed the stage by n_in floats -- shrink the tile by one output in that case)
    if (n_in * ldt > kGemmStageFloats) {
      tile = tw - 1;
      o0 -= tile;  // (redo this position with the smaller tile: the loop adds `tile` back)
      continue;
    }
    for (int idx = threadIdx.x; idx < tw * n_in; idx += blockDim.x) {
      const int oo = idx / n_in, k = idx - oo * n_in;  // global reads along k (coalesced), LDS writes ldt apart (odd: no conflicts)
      wt[k * ldt + oo] = W[(size_t)(o0 + oo) * n_in + k];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < R * tw; idx += blockDim.x) {
      const int r = idx / tw, oo = idx - r * tw;
      const float* a = in + r * ld_in;
      const float* w = wt + oo;
      float acc = b[o0 + oo];
      for (int k = 0; k < n_in; ++k) acc = fmaf(w[k * ldt], a[k], acc);
      out[r * ld_out + o0 + oo] = act ? leaky(acc) : acc;
    }
    __syncthreads();
  }
}

// The weights of one layer as they lie, W[o][k], in LDS when the whole layer fits the stage (returns W itself
// otherwise): the W^T delta loops of the gradient kernels read them with the lanes along k.  Ends with a barrier when it
// copied; the caller must place a barrier before the stage is overwritten again.
__device__ __forceinline__ const float* staged_weights(const float* __restrict__ W, int n_in, int n_out) {
  float* const stage = gemm_stage();
  if (n_in * n_out > kGemmStageFloats) return W;
  for (int idx = threadIdx.x; idx < n_in * n_out; idx += blockDim.x) stage[idx] = W[idx];
  __syncthreads();
  return stage;
}

}  // namespace mnf
