// AffineHalfFlow with the conditioner nets on the f16 matrix pipe in split (hi + lo) fp32
// arithmetic -- see mnf_split.h for the number format, its error bound and the range guard.
//
// Per 16-row tile at d = 64: 45 v_mfma_f32_16x16x32_f16 (16 cycles each, half of them free for
// VALU issue) instead of 96 v_mfma_f32_16x16x4_f32 (32 cycles each, none free), which moves the
// layer from issue bound to HBM bound.  Two kernels share the conditioner:
//   ahf_split_kernel        one coupling layer per launch (a single AffineHalfFlow call)
//   ahf_split_stack_kernel  L layers per launch, rows kept in registers, every intermediate written once
//                           (what NormalizingFlow does with a run of equal layers; FusedAffineStack: no
//                           intermediates); its ragged variant also serves halves narrower than an MFMA tile
// A tile whose operands leave the f16 range is recomputed with fp32 MFMAs from the fp32 operand image
// (read from global memory: the cold path is about correctness, not speed).
#include <hip/hip_runtime.h>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_split.h"

#ifndef MNF_LP_EPILOGUE
#define MNF_LP_EPILOGUE 1  // experiment switch: 0 compiles the fused log-prob epilogue out of the stack kernel
#endif

namespace mnf {

constexpr int kSplitWaves = 8;

template <int H, int HID>
__device__ __forceinline__ void stage_split_image(uint32_t* lds, const uint32_t* image) {
  using S = SplitShape<H, HID>;
  const uint4* src = reinterpret_cast<const uint4*>(image);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < S::IMAGE_WORDS / 4; i += blockDim.x) dst[i] = src[i];
}

// fp32 conditioner for one tile, out of line: the stack kernel's layer loop must stay free of ordinary
// global loads -- with the cold path inlined the compiler puts a vmcnt(0) at the join of the two paths,
// which on the hot path waits for the previous layer's intermediate-tensor stores (an HBM write latency
// per layer).  A call is a clean boundary for the wait-count tracking.
template <int G>
struct CondOut {
  f32x4 s[G], t[G];
};
template <int G>
struct CondIn {
  f32x4 c[G];
};
// (arguments and result by value: by reference the rows would have to live in scratch memory)
template <int H, int HID>
__device__ __attribute__((noinline)) CondOut<H / 16> ahf_cond_f32_cold(const float* image_f32, int lane, int q,
                                                                      CondIn<H / 16> in) {
  CondOut<H / 16> out;
  ahf_cond_f32<H, HID>(image_f32, lane, q, in.c, out.s, out.t);
  return out;
}

// s, t for NTL tiles: split path, then the fp32 path if any operand was out of range
template <int H, int HID, int NTL, typename Hook>
__device__ __forceinline__ void ahf_cond_guarded(const uint32_t* lds, const float* image_f32, int lane, int q,
                                                 const f32x4 (&cnd)[NTL][H / 16], f32x4 (&s4)[NTL][H / 16],
                                                 f32x4 (&t4)[NTL][H / 16], Hook at_stage) {
  using S = SplitShape<H, HID>;
  float mx = split_guard_seed(__builtin_bit_cast(float, lds[S::SPLIT_WORDS + S::PLAIN_WORDS]));  // max |weight|
  split_conditioner<H, HID, NTL, Hook>(lds, lane, q, cnd, s4, t4, mx, at_stage);
  if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      CondIn<H / 16> in;
#pragma unroll
      for (int g = 0; g < H / 16; ++g) in.c[g] = cnd[t][g];
      const CondOut<H / 16> out = ahf_cond_f32_cold<H, HID>(image_f32, lane, q, in);
#pragma unroll
      for (int g = 0; g < H / 16; ++g) {
        s4[t][g] = out.s[g];
        t4[t][g] = out.t[g];
      }
    }
  }
}

// ABL != 0 only in tools/split_microbench.hip (2 = no HBM traffic, 5 = no range guard, 6 = copy only,
// 7 = no prefetch; 1 and 3: see split_conditioner); the library uses ABL = 0.
template <int H, int HID, bool INV, int ABL = 0>
__global__ void __launch_bounds__(kSplitWaves * 64, H <= 32 ? 4 : 1)  // d <= 64: 4 waves/SIMD (<= 128 VGPRs)
ahf_split_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ log_det,
                 float* __restrict__ ysq, const uint32_t* __restrict__ simage, const float* __restrict__ image_f32,
                 int64_t rows, int parity, int accumulate) {
  using S = SplitShape<H, HID>;
  constexpr int G = S::G, dim = 2 * H;
  __shared__ __attribute__((aligned(16))) uint32_t lds[S::IMAGE_WORDS];
  stage_split_image<H, HID>(lds, simage);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int cond_off = parity ? H : 0, act_off = parity ? 0 : H;
  const int n_tiles = (int)((rows + 15) >> 4);
  const int tile_stride = (int)gridDim.x * kSplitWaves;

  // rows past the end are clamped to the last row for loads and masked for stores
  auto row_ptr = [&](int t) -> const float* {
    if (ABL == 2) t = blockIdx.x * kSplitWaves + wave;  // ablation: stay on one cached tile
    const int64_t r = (int64_t)t * 16 + j;
    return x + (r < rows ? r : rows - 1) * dim + 4 * q;
  };
  int tile = (int)blockIdx.x * kSplitWaves + wave;
  f32x4 cnd[G];
  if (tile < n_tiles) {
    const float* xr = row_ptr(tile);
#pragma unroll
    for (int g = 0; g < G; ++g) cnd[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
  }
  for (; tile < n_tiles; tile += tile_stride) {
    const int64_t row = (int64_t)tile * 16 + j;
    bool live = row < rows;
    if (ABL == 2) live = live && cnd[0][0] == 1.2345e30f;  // ablation: never true, keeps the math alive
    const float* xr = row_ptr(tile);
    float* yr = y + (xr - x);
    if (ABL == 7) {
#pragma unroll
      for (int g = 0; g < G; ++g) cnd[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
    }
    // the transformed half is first needed after the conditioner, which hides its latency
    f32x4 act[G];
#pragma unroll
    for (int g = 0; g < G; ++g) act[g] = *reinterpret_cast<const f32x4*>(xr + act_off + 16 * g);
    if (live) {
#pragma unroll
      for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(yr + cond_off + 16 * g) = cnd[g];
    }
    float sq = 0.f;
    if (ysq) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = fmaf(cnd[g][r], cnd[g][r], sq);
    }
    // Software prefetch: the next tile's conditioning half is requested into the same registers as
    // soon as this tile's has been split into MFMA operands, and flies under the conditioner (at
    // 4 waves/SIMD the other waves alone do not cover the HBM latency).  One tile past the end
    // re-reads the last tile: harmless, and keeps the loop branch-free.
    f32x4 cur[1][G];
#pragma unroll
    for (int g = 0; g < G; ++g) cur[0][g] = cnd[g];
    const float* xn = row_ptr(tile + tile_stride < n_tiles ? tile + tile_stride : n_tiles - 1);
    f32x4 s4[1][G], t4[1][G];
    {
      using SS = SplitShape<H, HID>;
      float mx = split_guard_seed(__builtin_bit_cast(float, lds[SS::SPLIT_WORDS + SS::PLAIN_WORDS]));  // max |weight|
      auto prefetch = [&](int stage = 0) {
        if (ABL == 7 || stage != 0) return;
#pragma unroll
        for (int g = 0; g < G; ++g) cnd[g] = *reinterpret_cast<const f32x4*>(xn + cond_off + 16 * g);
      };
      if (ABL == 6) {
        prefetch();
#pragma unroll
        for (int g = 0; g < G; ++g) s4[0][g] = t4[0][g] = cur[0][g];
      } else {
        split_conditioner<H, HID, 1, decltype(prefetch), ABL>(lds, lane, q, cur, s4, t4, mx, prefetch);
      }
      if (ABL != 5 && ABL != 6 && __builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {  // out of f16 range: fp32 MFMAs, operands from L2
        f32x4 again[G];
#pragma unroll
        for (int g = 0; g < G; ++g) again[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
        ahf_cond_f32<H, HID>(image_f32, lane, q, again, s4[0], t4[0]);
      }
    }
    float ld = ahf_transform<H, INV, true>(s4[0], t4[0], act);
    if (live) {
#pragma unroll
      for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(yr + act_off + 16 * g) = act[g];
    }
    if (log_det) {
      ld = sum_over_q(ld);
      if (INV) ld = -ld;
      if (live && q == 0) log_det[row] = accumulate ? log_det[row] + ld : ld;
    }
    if (ysq) {  // |y_row|^2 for the base log-prob epilogue: saves re-reading y
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = fmaf(act[g][r], act[g][r], sq);
      sq = sum_over_q(sq);
      if (live && q == 0) ysq[row] = sq;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// L layers per launch, rows kept in registers across layers.  Compute bound (no per-layer HBM read), and
// inside the conditioner the co-bottleneck is LDS: a wave re-reads 30 KB of operands per 16 rows per
// layer.  So (d <= 64) a wave owns two row tiles that share every operand read, a workgroup has
// four waves -- one per SIMD, two workgroups per CU -- and the next layer's split image is
// copied L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers) into the other half of
// a double buffer while the current layer computes; one barrier per layer.
// mid != nullptr: the output of every layer but the last goes to mid[li] (application order): each
// intermediate tensor is written once and never re-read.
// ------------------------------------------------------------------------------------------------
// d <= 64: two row tiles per wave (shared operand reads), four waves, two workgroups per CU.
// d >= 128: the rows of ONE tile already take 32-64 VGPRs and the double-buffered image 90-150 KB of LDS:
// one tile per wave, eight waves sharing the image, one workgroup per CU (still two waves per SIMD).
#ifndef MNF_STACK_ABL
#define MNF_STACK_ABL 0
#endif
#ifndef MNF_STACK_TILES  // experiment switches for d <= 64 (tools/kernel_variants.sh)
#define MNF_STACK_TILES 2
#define MNF_STACK_WAVES 4
#define MNF_STACK_WPS 2
#endif
template <int H>
constexpr int stack_tiles() { return H <= 32 ? MNF_STACK_TILES : 1; }
template <int H>
constexpr int stack_waves() { return H <= 32 ? MNF_STACK_WAVES : 8; }
template <int H>
constexpr int stack_waves_per_simd() { return H <= 32 ? MNF_STACK_WPS : 2; }
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// Image copy L2 -> LDS by LDS-DMA (no staging registers).  One wave-instruction copies 64 x 16 B; the LDS
// destination is the wave-uniform base + lane * 16.  Completion: s_waitcnt vmcnt(0) by the issuing wave,
// then a barrier.
template <int IMG4, int WAVES>
__device__ __forceinline__ void image_to_lds_async(const uint4* src, uint32_t* dst, int lane, int wave) {
  constexpr int PIECES = (IMG4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < (PIECES + WAVES - 1) / WAVES; ++i) {
    const int piece = i * WAVES + wave;  // wave-uniform
    if (piece < PIECES && piece * 64 + lane < IMG4)
      __builtin_amdgcn_global_load_lds(src + piece * 64 + lane, (lds_void_ptr)(dst + piece * 256), 16, 0, 0);
  }
}

// Row accesses of the stack kernel: four columns col..col+3 of a coupling half that is `h` columns wide.
// RAG = false: h == H, one 16-byte access.  RAG = true (h < H): columns >= h are zero on load and skipped on
// store; `vec` (h % 4 == 0 and 16-byte aligned bases) keeps the 16-byte access, else element by element.
template <bool RAG>
__device__ __forceinline__ f32x4 half_load4(const float* p, int col, int h, bool vec) {
  if (!RAG) return *reinterpret_cast<const f32x4*>(p + col);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    if (col < h) v = *reinterpret_cast<const f32x4*>(p + col);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float e = 0.f;
      if (col + r < h) e = p[col + r];
      v[r] = e;
    }
  }
  return v;
}
// The stack kernel's row stores carry the non-temporal hint: every intermediate is written once and not read again
// by this launch, and without the hint the 2.4 GB write stream (C2) competes with the operand images for L2.  Measured
// (same box, A/B): C2 798-802 -> 783 us per launch, C4 1,411 -> 1,375; the single-layer kernel, whose output the
// next launch reads right away, is better off without it (+8 us).  MNF_STACK_NT=0: A/B switch.
#ifndef MNF_STACK_NT
#define MNF_STACK_NT 1
#endif
template <bool RAG>
__device__ __forceinline__ void half_store4(float* p, int col, int h, bool vec, f32x4 v) {
  if (!RAG) {
    if (MNF_STACK_NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + col));
    else *reinterpret_cast<f32x4*>(p + col) = v;
  } else if (vec) {
    if (col < h) *reinterpret_cast<f32x4*>(p + col) = v;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col + r < h) p[col + r] = v[r];
  }
}

template <int H, int HID, bool INV, bool RAG>
__global__ void __launch_bounds__(stack_waves<H>() * 64, stack_waves_per_simd<H>())
ahf_split_stack_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mid,
                       float* __restrict__ log_det, float* __restrict__ ysq, const uint32_t* __restrict__ simages,
                       const float* __restrict__ images_f32, uint32_t parity_bits, int n_layers, int64_t rows,
                       int accumulate, float* __restrict__ log_prob, double* __restrict__ log_prob_sum, int h_ragged,
                       int vec_ok) {
  using S = SplitShape<H, HID>;
  constexpr int G = S::G, NTL = stack_tiles<H>(), kStackWaves = stack_waves<H>();
  const int h = RAG ? h_ragged : H, dim = 2 * h;  // the rows in memory are 2 h wide; the tiles H
  const bool vec = vec_ok != 0;
  constexpr int IMG4 = S::IMAGE_WORDS / 4;
  constexpr int F32_FLOATS = AhfShape<H, HID>::IMAGE_FLOATS;
  constexpr int GROUP_ROWS = 16 * NTL * kStackWaves;
  // two images: static LDS while they fit the 64 KB static limit (the compiler folds constant LDS addresses
  // into the operand reads: measured 755 vs 900 us for the d = 64 pass), dynamic LDS above it
  constexpr bool kStaticLds = 2 * S::IMAGE_WORDS * sizeof(uint32_t) <= 64 * 1024;
  __shared__ __attribute__((aligned(16))) uint32_t lds_static[kStaticLds ? 2 * S::IMAGE_WORDS : 4];
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  uint32_t* const lds_base = kStaticLds ? lds_static : lds_dyn;
  auto lds_buf = [&](int k) -> uint32_t* { return lds_base + (k & 1) * S::IMAGE_WORDS; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const uint4* img4 = reinterpret_cast<const uint4*>(simages);
  auto layer_at = [&](int li) { return INV ? n_layers - 1 - li : li; };  // application order

  double lp_acc = 0.0;
  const int n_groups = (int)((rows + GROUP_ROWS - 1) / GROUP_ROWS);
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    int64_t row[NTL], rowc[NTL];
    bool live[NTL];
    f32x4 lo[NTL][G], hi[NTL][G];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      row[t] = (int64_t)grp * GROUP_ROWS + (wave * NTL + t) * 16 + j;
      live[t] = row[t] < rows;
      rowc[t] = live[t] ? row[t] : rows - 1;  // rows past the end: clamped loads, masked stores
      const float* xr = x + rowc[t] * dim;
#pragma unroll
      for (int g = 0; g < G; ++g) lo[t][g] = half_load4<RAG>(xr, 16 * g + 4 * q, h, vec);
#pragma unroll
      for (int g = 0; g < G; ++g) hi[t][g] = half_load4<RAG>(xr + h, 16 * g + 4 * q, h, vec);
    }
    __syncthreads();  // the previous group's last layer is fully consumed
    image_to_lds_async<IMG4, kStackWaves>(img4 + (int64_t)layer_at(0) * IMG4, lds_buf(0), lane, wave);
    __syncthreads();  // (hipcc drains vmcnt before a barrier: image and rows have landed)
    float ld[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) ld[t] = 0.f;
    for (int li = 0; li < n_layers; ++li) {
      const int layer = layer_at(li);
      if (li + 1 < n_layers)  // next layer's image into the other buffer, in flight under this layer's math
        image_to_lds_async<IMG4, kStackWaves>(img4 + (int64_t)layer_at(li + 1) * IMG4, lds_buf(li + 1), lane, wave);
      const uint32_t* img = lds_buf(li);
      const float* f32img = images_f32 + (int64_t)layer * F32_FLOATS;
      f32x4 s4[NTL][G], t4[NTL][G];
      // The rows as they stand at the top of this layer are the previous layer's output, i.e. intermediate
      // tensor li - 1.  Its NTL * 2G stores are spread over the four stages of this layer's conditioner
      // (the registers change only in the transform after it): issued as one burst in front of the layer
      // barrier they held up both workgroups of the CU at the same moment.
      auto store_previous = [&](int stage) {
        if (mid == nullptr || li == 0) return;
        constexpr int TOTAL = NTL * 2 * G, PER = (TOTAL + 3) / 4;
#pragma unroll
        for (int k = stage * PER; k < (stage + 1) * PER && k < TOTAL; ++k) {
          const int t = k / (2 * G), half = (k / G) & 1, g = k % G;
          if (live[t]) {
            // (MNF_STACK_ABL=1, timing only: every wave stores to the first rows -- the store instructions without
            //  their HBM traffic)
            float* mr = mid + (MNF_STACK_ABL ? (int64_t)(threadIdx.x & 15) * dim
                                             : ((int64_t)(li - 1) * rows + rowc[t]) * dim) + (half ? h : 0);
            half_store4<RAG>(mr, 16 * g + 4 * q, h, vec, half ? hi[t][g] : lo[t][g]);
          }
        }
      };
      // one coupling layer on the rows in registers: cnd_rows condition, act_rows are transformed
      auto run_layer = [&](const f32x4 (&cnd_rows)[NTL][G], f32x4 (&act_rows)[NTL][G]) {
        if constexpr (G > S::GC) {
          // d = 256: s and t are consumed chunk by chunk (all sixteen tiles of a row would take 128 VGPRs)
          bool cold = false;
          float mx = split_guard_seed(__builtin_bit_cast(float, img[S::SPLIT_WORDS + S::PLAIN_WORDS]));
          auto emit = make_chunk_emit(
              [&](float m) { return cold = wave_any(!(m <= kSplitLimit)); },
              [&](int g0, const f32x4 (&sc)[NTL][S::GC], const f32x4 (&tc)[NTL][S::GC]) {
#pragma unroll
                for (int t = 0; t < NTL; ++t)
#pragma unroll
                  for (int g = 0; g < S::GC; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                      const float sv = sc[t][g][r], tv = tc[t][g][r];
                      const float e = exp2x(INV ? -sv : sv);
                      const float a = act_rows[t][g0 + g][r];
                      act_rows[t][g0 + g][r] = INV ? (a - tv) * e : __builtin_fmaf(e, a, tv);
                      ld[t] += sv;
                    }
              });
          split_conditioner<H, HID, NTL, decltype(store_previous), 0, decltype(emit)>(img, lane, q, cnd_rows, s4, t4, mx,
                                                                                   store_previous, emit);
          if (__builtin_expect(cold, 0)) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
              CondIn<G> in;
#pragma unroll
              for (int g = 0; g < G; ++g) in.c[g] = cnd_rows[t][g];
              const CondOut<G> out = ahf_cond_f32_cold<H, HID>(f32img, lane, q, in);
              ld[t] += ahf_transform<H, INV>(out.s, out.t, act_rows[t]);
            }
          }
        } else {
          ahf_cond_guarded<H, HID, NTL>(img, f32img, lane, q, cnd_rows, s4, t4, store_previous);
#pragma unroll
          for (int t = 0; t < NTL; ++t) ld[t] += ahf_transform<H, INV, true>(s4[t], t4[t], act_rows[t]);
        }
      };
      if ((parity_bits >> layer) & 1u) run_layer(hi, lo);  // conditioner = upper half
      else run_layer(lo, hi);
      if (li + 1 < n_layers) {
        // End of layer: the next image (LDS-DMA, issued before this layer's math) must have landed before the
        // barrier.  vmcnt(0) also covers this layer's staged stores; the last of them was issued a quarter of
        // a layer ago.  (Inline asm with a memory clobber, not __builtin_amdgcn_s_barrier(): the builtin is no
        // compiler barrier for memory operations, and the next layer's first operand reads must not be
        // hoisted above it.)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      if (live[t]) {
        float* yr = y + rowc[t] * dim;
#pragma unroll
        for (int g = 0; g < G; ++g) half_store4<RAG>(yr, 16 * g + 4 * q, h, vec, lo[t][g]);
#pragma unroll
        for (int g = 0; g < G; ++g) half_store4<RAG>(yr + h, 16 * g + 4 * q, h, vec, hi[t][g]);
      }
      float ld_row = 0.f;
      if (log_det) {
        float l = sum_over_q(ld[t]);
        if (INV) l = -l;
        ld_row = (accumulate && live[t]) ? log_det[row[t]] + l : l;
        if (live[t] && q == 0) log_det[row[t]] = ld_row;
      }
      if (ysq || (MNF_LP_EPILOGUE && log_prob)) {
        float sq = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) sq = fmaf(lo[t][g][r], lo[t][g][r], fmaf(hi[t][g][r], hi[t][g][r], sq));
        sq = sum_over_q(sq);
        if (ysq && live[t] && q == 0) ysq[row[t]] = sq;
        if (MNF_LP_EPILOGUE && log_prob) {  // standard-normal base: log p = log_det - |z|^2 / 2 - d/2 log(2 pi)   (core.py:46-49)
          const float lp = ld_row + (-0.5f * sq - (float)dim * kHalfLog2Pi);
          if (live[t] && q == 0) {
            log_prob[row[t]] = lp;
            lp_acc += (double)lp;
          }
        }
      }
    }
  }
  if (MNF_LP_EPILOGUE && log_prob_sum) {  // fp64 sum over the rows: wave shuffle, one native fp64 atomic per wave
    for (int off = 32; off > 0; off >>= 1) lp_acc += __shfl_down(lp_acc, off, 64);
    if (lane == 0) atomicAdd(log_prob_sum, lp_acc);
  }
}

// ---------------------------------------------------------------- host: split image index table
// 2 entries per split word (low half, high half), then 1 entry per plain (fp32 bias) word.
// h <= H: real half width (zero operands in the padded input columns / output rows)
template <int H, int HID>
static void build_split_index(int32_t* idx, int h, bool has_s = true, bool has_t = true, const int* widths = nullptr) {
  using S = SplitShape<H, HID>;
  constexpr int G = S::G, NT = S::NT, NKS = S::NKS, KS1 = S::KS1;
  // real widths of the three hidden layers (<= HID: the rest of a tile's units are structural zeros)
  const int w[3] = {widths ? widths[0] : HID, widths ? widths[1] : HID, widths ? widths[2] : HID};
  int sizes[5] = {h, w[0], w[1], w[2], h};
  NetDesc net[2];
  const bool has[2] = {has_s, has_t};  // an absent net (scale=False / shift=False) is an all-zero operand set
  int64_t off = 0;
  if (has_s) off += fill_net(net[0], 5, sizes, off);
  if (has_t) off += fill_net(net[1], 5, sizes, off);
  const int64_t n_entries = 2 * (int64_t)S::SPLIT_WORDS + S::PLAIN_WORDS;
  for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  int op = 0;
  // element e of lane (i, kq) of the A operand of `op`: weight(row i of the output tile, K slot 8 kq + e)
  auto put = [&](int lane, int e, int32_t src) {
    for (int part = 0; part < 2; ++part)
      idx[(((int64_t)(2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] = src | (part ? kSplitLoBit : 0);
  };
  // hidden unit behind K slot 8 kq + e of hidden K-step ks, or -1
  auto unit_in = [&](int ks, int kq, int e, int& tile) {
    tile = e < 4 ? S::ks_a(ks) : S::ks_b(ks);
    if (tile < 0) return -1;
    const int u = 16 * tile + 4 * kq + (e & 3);
    return u < 2 * HID ? u : -1;
  };
  for (int ks = 0; ks < KS1; ++ks)
    for (int m = 0; m < NT; ++m, ++op)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
        if (u >= 2 * HID) continue;
        for (int e = 0; e < 8; ++e) {
          const int g = 2 * ks + (e >> 2), col = 16 * g + 4 * kq + (e & 3);
          if (g < G && col < h && has[u / HID] && u % HID < w[0])
            put(lane, e, net[u / HID].w_off[0] + (u % HID) * h + col);
        }
      }
  for (int l = 1; l <= 2; ++l)
    for (int ks = 0; ks < NKS; ++ks)
      for (int m = 0; m < NT; ++m) {
        if (!S::uses(S::tile_nets(m), ks)) continue;
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = 16 * m + i;
          if (u >= 2 * HID) continue;
          for (int e = 0; e < 8; ++e) {
            int tile;
            const int ui = unit_in(ks, kq, e, tile);
            if (ui < 0 || ui / HID != u / HID || S::assigned_ks(S::tile_nets(m), tile) != ks || !has[u / HID]) continue;
            if (u % HID >= w[l] || ui % HID >= w[l - 1]) continue;
            put(lane, e, net[u / HID].w_off[l] + (u % HID) * w[l - 1] + ui % HID);
          }
        }
        ++op;
      }
  for (int g0 = 0; g0 < G; g0 += S::GC)  // output operands: chunk, net, K-step, tile (split_conditioner's order)
    for (int nn = 0; nn < 2; ++nn)
      for (int ks = 0; ks < NKS; ++ks) {
        if (!S::uses(1 << nn, ks)) continue;
        for (int g = g0; g < g0 + S::GC; ++g, ++op)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kq = lane >> 4;
            for (int e = 0; e < 8; ++e) {
              int tile;
              const int ui = unit_in(ks, kq, e, tile);
              if (ui < 0 || ui / HID != nn || S::assigned_ks(1 << nn, tile) != ks || 16 * g + i >= h || !has[nn]) continue;
              if (ui % HID >= w[2]) continue;
              put(lane, e, net[nn].w_off[3] + (16 * g + i) * w[2] + ui % HID);
            }
          }
      }
  // biases: [tile][row i], fp32
  int32_t* b = idx + 2 * (int64_t)S::SPLIT_WORDS;
  int bt = 0;
  for (int l = 0; l < 3; ++l)
    for (int m = 0; m < NT; ++m, ++bt)
      for (int i = 0; i < 16; ++i) {
        const int u = 16 * m + i;
        if (u < 2 * HID && has[u / HID] && u % HID < w[l]) b[bt * 16 + i] = net[u / HID].b_off[l] + u % HID;
      }
  for (int nn = 0; nn < 2; ++nn)
    for (int g = 0; g < G; ++g, ++bt)
      for (int i = 0; i < 16; ++i)
        if (16 * g + i < h && has[nn]) b[bt * 16 + i] = net[nn].b_off[3] + 16 * g + i;
}

template <int H, int HID>
static int launch_split(const float* x, float* y, float* log_det, float* ysq, int accumulate, const uint32_t* simage,
                        const float* image, int64_t rows, int parity, int inverse, hipStream_t stream) {
  static DeviceMemo memo;
  const int resident = memo.get([](int dev) {
    return resident_by_occupancy(ahf_split_kernel<H, HID, true>, kSplitWaves * 64, dev, 1);
  });
  const int cus = device_cus(current_device());
  const int64_t n_tiles = (rows + 15) / 16;
  const dim3 grid((unsigned)balanced_grid(n_tiles, kSplitWaves, resident, cus)), block(kSplitWaves * 64);
  tag_kernel("ahf_split");
  if (inverse)
    hipLaunchKernelGGL((ahf_split_kernel<H, HID, true>), grid, block, 0, stream, x, y, log_det, ysq, simage, image,
                       rows, parity, accumulate);
  else
    hipLaunchKernelGGL((ahf_split_kernel<H, HID, false>), grid, block, 0, stream, x, y, log_det, ysq, simage, image,
                       rows, parity, accumulate);
  return check_launch();
}

template <int H, int HID, bool RAG>
static int launch_split_stack(const float* x, float* y, float* mid, float* log_det, float* ysq, int accumulate,
                              const uint32_t* simages, const float* images, uint32_t parity_bits, int n_layers,
                              int64_t rows, int inverse, float* log_prob, double* log_prob_sum, int h, int vec_ok,
                              hipStream_t stream) {
  constexpr int kStackWaves = stack_waves<H>(), kStackTiles = stack_tiles<H>();
  constexpr size_t image_bytes = 2 * SplitShape<H, HID>::IMAGE_WORDS * sizeof(uint32_t);
  constexpr size_t lds_bytes = image_bytes <= 64 * 1024 ? 0 : image_bytes;  // dynamic part (see the kernel)
  // per device (the dynamic-LDS attribute is a per-device setting of the function; a process may drive several GPUs):
  // 0 = not set up yet, -1 = the attribute could not be set, else the number of resident workgroups
  static DeviceMemo memo;
  const int resident = memo.get([](int dev) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ahf_split_stack_kernel<H, HID, true, RAG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(ahf_split_stack_kernel<H, HID, false, RAG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return -1;
    // two waves per SIMD by registers (launch bounds), i.e. 8 waves per CU; the double-buffered image fits twice
    // for d <= 64 (the occupancy query under-reports kernels with dynamic LDS, so this is computed here)
    int per_cu = 4 * stack_waves_per_simd<H>() / kStackWaves;
    while (per_cu > 1 && per_cu * image_bytes > 160 * 1024) --per_cu;
    return per_cu * device_cus(dev);
  });
  if (resident <= 0) return MNF_ERR_UNSUPPORTED;
  constexpr int GROUP_ROWS = 16 * kStackTiles * kStackWaves;
  const int64_t n_groups = (rows + GROUP_ROWS - 1) / GROUP_ROWS;
  const dim3 grid((unsigned)(n_groups < resident ? n_groups : resident)), block(kStackWaves * 64);
  tag_kernel("ahf_split_stack");
  if (inverse)
    hipLaunchKernelGGL((ahf_split_stack_kernel<H, HID, true, RAG>), grid, block, lds_bytes, stream, x, y, mid, log_det,
                       ysq, simages, images, parity_bits, n_layers, rows, accumulate, log_prob, log_prob_sum, h, vec_ok);
  else
    hipLaunchKernelGGL((ahf_split_stack_kernel<H, HID, false, RAG>), grid, block, lds_bytes, stream, x, y, mid, log_det,
                       ysq, simages, images, parity_bits, n_layers, rows, accumulate, log_prob, log_prob_sum, h, vec_ok);
  return check_launch();
}

// (H, HID) pairs with a split kernel, and those of the stack kernel (hidden width 32 at d <= 64 spills at two tiles
// per wave and gains nothing over nine single-layer launches: there it only serves narrow halves, see below)
#define MNF_AHF_SPLIT_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16) X(64, 24) X(128, 24) X(16, 32) X(32, 32) X(64, 32) X(16, 64) X(32, 64) X(64, 64)
#define MNF_AHF_SPLIT_STACK_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16) X(64, 24) X(128, 24) X(16, 32) X(32, 32) X(64, 32)

// three hidden layers of at most 64 units: hid = the width the kernels run them at (see ahf_padded_hidden)
static bool uniform3(int n_hidden, const int* hidden, int& hid) {
  hid = ahf_padded_hidden(n_hidden, hidden);
  return hid != 0;
}

static bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

int ahf_split_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate, const void* split_image,
                     const float* image, int64_t rows, int dim, int parity, int inverse, int n_hidden,
                     const int* hidden, int has_scale, int has_shift, hipStream_t stream) {
  int hid = 0;
  if (!split_image || !image || (!has_scale && !has_shift) || !uniform3(n_hidden, hidden, hid))
    return MNF_ERR_UNSUPPORTED;
  if (!aligned16(x, y, split_image, image)) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                                                            \
  if (dim == 2 * HH && hid == HD)                                                                            \
    return launch_split<HH, HD>(x, y, log_det, ysq, accumulate, static_cast<const uint32_t*>(split_image), image, \
                                rows, parity != 0, inverse != 0, stream);
  MNF_AHF_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int ahf_split_stack_launch(const float* x, float* y, float* mid, float* log_det, float* ysq, int accumulate,
                           const void* split_images, const float* images, uint32_t parity_bits, int n_layers,
                           int64_t rows, int dim, int inverse, int hid, float* log_prob, double* log_prob_sum,
                           hipStream_t stream) {
  if (!split_images || !images || (dim & 1) || !aligned16(split_images, images, nullptr, nullptr))
    return MNF_ERR_UNSUPPORTED;
  const int h = dim / 2, hp = ahf_padded_half(h);
  const bool rows_aligned = aligned16(x, y, mid, nullptr);
  if (h == hp && !rows_aligned) return MNF_ERR_UNSUPPORTED;
  const int vec_ok = rows_aligned && (h & 3) == 0;
  if (hid == 32 && h == hp && hp <= 32 && n_layers > 1) return MNF_ERR_UNSUPPORTED;  // measured: no faster than layer by layer
  const uint32_t* simages = static_cast<const uint32_t*>(split_images);
#define X(HH, HD)                                                                                                    \
  if (hp == HH && hid == HD)                                                                                         \
    return h == HH ? launch_split_stack<HH, HD, false>(x, y, mid, log_det, ysq, accumulate, simages, images,         \
                                                       parity_bits, n_layers, rows, inverse != 0, log_prob,         \
                                                       log_prob_sum, h, vec_ok, stream)                              \
                   : launch_split_stack<HH, HD, true>(x, y, mid, log_det, ysq, accumulate, simages, images,          \
                                                      parity_bits, n_layers, rows, inverse != 0, log_prob,          \
                                                      log_prob_sum, h, vec_ok, stream);
  MNF_AHF_SPLIT_STACK_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf

extern "C" {

int mnf_affine_half_split_layout(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                 int64_t* n_split_words, int64_t* n_plain_words) {
  int hid = 0;
  if (!n_split_words || !n_plain_words || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if ((!has_scale && !has_shift) || !mnf::uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  const int h = dim / 2, hp = (dim & 1) ? 0 : mnf::ahf_padded_half(h);
#define X(HH, HD)                                            \
  if (hp == HH && hid == HD) {                               \
    *n_split_words = mnf::SplitShape<HH, HD>::SPLIT_WORDS;   \
    *n_plain_words = mnf::SplitShape<HH, HD>::PLAIN_WORDS;   \
    return MNF_OK;                                           \
  }
  if (h == hp) {  // full tiles: every split shape; a ragged half: the shapes of the stack kernel (its only kernel)
    MNF_AHF_SPLIT_SHAPES(X)
  } else {
    MNF_AHF_SPLIT_STACK_SHAPES(X)
  }
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_affine_half_split_index(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                int32_t* idx_host) {
  int hid = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if ((!has_scale && !has_shift) || !mnf::uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
  const int h = dim / 2, hp = (dim & 1) ? 0 : mnf::ahf_padded_half(h);
#define X(HH, HD)                                 \
  if (hp == HH && hid == HD) {                    \
    mnf::build_split_index<HH, HD>(idx_host, h, has_scale != 0, has_shift != 0, hidden);  \
    return MNF_OK;                                \
  }
  if (h == hp) {
    MNF_AHF_SPLIT_SHAPES(X)
  } else {
    MNF_AHF_SPLIT_STACK_SHAPES(X)
  }
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
