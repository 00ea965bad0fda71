// AffineHalfFlow with the conditioner nets on the f16 matrix pipe in split (hi + lo) fp32
// arithmetic -- see mnf_split.h for the number format, its error bound and the range guard.
//
// Per 16-row tile at d = 64: 45 v_mfma_f32_16x16x32_f16 (16 cycles each, half of them free for
// VALU issue) instead of 96 v_mfma_f32_16x16x4_f32 (32 cycles each, none free), which moves the
// layer from issue bound to HBM bound.  Two kernels share the conditioner:
//   ahf_split_kernel        one coupling layer per launch (the default AffineHalfFlow path)
//   ahf_split_stack_kernel  L layers per launch, rows kept in registers (opt-in, FusedAffineStack)
// A tile whose operands leave the f16 range is recomputed with fp32 MFMAs from the fp32 operand image
// (read from global memory: the cold path is about correctness, not speed).
#include <hip/hip_runtime.h>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_split.h"

namespace mnf {

constexpr int kSplitWaves = 8;

template <int H, int HID>
__device__ __forceinline__ void stage_split_image(uint32_t* lds, const uint32_t* image) {
  using S = SplitShape<H, HID>;
  const uint4* src = reinterpret_cast<const uint4*>(image);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < S::IMAGE_WORDS / 4; i += blockDim.x) dst[i] = src[i];
}

// s, t for one tile: split path, then the fp32 path if any operand was out of range
template <int H, int HID>
__device__ __forceinline__ void ahf_cond_guarded(const uint32_t* lds, const float* image_f32, int lane, int q,
                                                 const f32x4 (&cnd)[H / 16], f32x4 (&s4)[H / 16],
                                                 f32x4 (&t4)[H / 16]) {
  using S = SplitShape<H, HID>;
  float mx = __builtin_bit_cast(float, lds[S::SPLIT_WORDS + S::PLAIN_WORDS]);  // max |weight|
  split_conditioner<H, HID>(lds, lane, q, cnd, s4, t4, mx);
  if (__builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) ahf_cond_f32<H, HID>(image_f32, lane, q, cnd, s4, t4);
}

// ABL != 0 only in tools/split_microbench.hip (2 = no HBM traffic, 5 = no range guard, 6 = copy only,
// 7 = no prefetch; 1 and 3: see split_conditioner); the library uses ABL = 0.
template <int H, int HID, bool INV, int ABL = 0>
__global__ void __launch_bounds__(kSplitWaves * 64)
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
    f32x4 cur[G];
#pragma unroll
    for (int g = 0; g < G; ++g) cur[g] = cnd[g];
    const float* xn = row_ptr(tile + tile_stride < n_tiles ? tile + tile_stride : n_tiles - 1);
    f32x4 s4[G], t4[G];
    {
      using SS = SplitShape<H, HID>;
      float mx = __builtin_bit_cast(float, lds[SS::SPLIT_WORDS + SS::PLAIN_WORDS]);  // max |weight|
      auto prefetch = [&]() {
        if (ABL == 7) return;
#pragma unroll
        for (int g = 0; g < G; ++g) cnd[g] = *reinterpret_cast<const f32x4*>(xn + cond_off + 16 * g);
      };
      if (ABL == 6) {
        prefetch();
#pragma unroll
        for (int g = 0; g < G; ++g) s4[g] = t4[g] = cur[g];
      } else {
        split_conditioner<H, HID, decltype(prefetch), ABL>(lds, lane, q, cur, s4, t4, mx, prefetch);
      }
      if (ABL != 5 && ABL != 6 && __builtin_expect(wave_any(!(mx <= kSplitLimit)), 0)) {  // out of f16 range: fp32 MFMAs, operands from L2
        f32x4 again[G];
#pragma unroll
        for (int g = 0; g < G; ++g) again[g] = *reinterpret_cast<const f32x4*>(xr + cond_off + 16 * g);
        ahf_cond_f32<H, HID>(image_f32, lane, q, again, s4, t4);
      }
    }
    float ld = ahf_transform<H, INV>(s4, t4, act);
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

// L layers per launch; the split images are streamed through a double-buffered LDS window: the
// image of layer l+1 is requested into registers before layer l is computed and handed over at
// one barrier per layer.
template <int H, int HID, bool INV>
__global__ void __launch_bounds__(kSplitWaves * 64)
ahf_split_stack_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ log_det,
                       float* __restrict__ ysq, const uint32_t* __restrict__ simages,
                       const float* __restrict__ images_f32, uint32_t parity_bits, int n_layers, int64_t rows,
                       int accumulate) {
  using S = SplitShape<H, HID>;
  constexpr int G = S::G, dim = 2 * H;
  constexpr int IMG4 = S::IMAGE_WORDS / 4;
  constexpr int STAGE = (IMG4 + kSplitWaves * 64 - 1) / (kSplitWaves * 64);
  constexpr int F32_FLOATS = AhfShape<H, HID>::IMAGE_FLOATS;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][S::IMAGE_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const uint4* img4 = reinterpret_cast<const uint4*>(simages);
  auto layer_at = [&](int li) { return INV ? n_layers - 1 - li : li; };  // application order

  const int n_groups = (int)((rows + 16 * kSplitWaves - 1) / (16 * kSplitWaves));
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t row = (int64_t)grp * (16 * kSplitWaves) + wave * 16 + j;
    const bool live = row < rows;
    const int64_t rowc = live ? row : rows - 1;
    const float* xr = x + rowc * dim + 4 * q;
    f32x4 lo[G], hi[G];
#pragma unroll
    for (int g = 0; g < G; ++g) lo[g] = *reinterpret_cast<const f32x4*>(xr + 16 * g);
#pragma unroll
    for (int g = 0; g < G; ++g) hi[g] = *reinterpret_cast<const f32x4*>(xr + H + 16 * g);
    __syncthreads();  // the previous group's last layer is fully consumed
    {
      const uint4* src = img4 + (int64_t)layer_at(0) * IMG4;
      uint4* dst = reinterpret_cast<uint4*>(lds[0]);
      for (int k = threadIdx.x; k < IMG4; k += kSplitWaves * 64) dst[k] = src[k];
    }
    __syncthreads();
    float ld = 0.f;
    for (int li = 0; li < n_layers; ++li) {
      const int layer = layer_at(li);
      // request the next layer's image (after the last layer: the same one again, branch-free)
      const uint4* src = img4 + (int64_t)layer_at(li + 1 < n_layers ? li + 1 : li) * IMG4;
      uint4 st[STAGE];
#pragma unroll
      for (int i = 0; i < STAGE; ++i) {
        const int k = threadIdx.x + i * (kSplitWaves * 64);
        st[i] = src[k < IMG4 ? k : 0];
      }
      const uint32_t* img = lds[li & 1];
      const float* f32img = images_f32 + (int64_t)layer * F32_FLOATS;
      f32x4 s4[G], t4[G];
      if ((parity_bits >> layer) & 1u) {  // conditioner = upper half
        ahf_cond_guarded<H, HID>(img, f32img, lane, q, hi, s4, t4);
        ld += ahf_transform<H, INV>(s4, t4, lo);
      } else {
        ahf_cond_guarded<H, HID>(img, f32img, lane, q, lo, s4, t4);
        ld += ahf_transform<H, INV>(s4, t4, hi);
      }
      uint4* dst = reinterpret_cast<uint4*>(lds[(li + 1) & 1]);
#pragma unroll
      for (int i = 0; i < STAGE; ++i) {
        const int k = threadIdx.x + i * (kSplitWaves * 64);
        if (k < IMG4) dst[k] = st[i];
      }
      __syncthreads();
    }
    if (live) {
      float* yr = y + rowc * dim + 4 * q;
#pragma unroll
      for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(yr + 16 * g) = lo[g];
#pragma unroll
      for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(yr + H + 16 * g) = hi[g];
    }
    if (log_det) {
      ld = sum_over_q(ld);
      if (INV) ld = -ld;
      if (live && q == 0) log_det[row] = accumulate ? log_det[row] + ld : ld;
    }
    if (ysq) {
      float sq = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = fmaf(lo[g][r], lo[g][r], fmaf(hi[g][r], hi[g][r], sq));
      sq = sum_over_q(sq);
      if (live && q == 0) ysq[row] = sq;
    }
  }
}

// ---------------------------------------------------------------- host: split image index table
// 2 entries per split word (low half, high half), then 1 entry per plain (fp32 bias) word.
template <int H, int HID>
static void build_split_index(int32_t* idx) {
  using S = SplitShape<H, HID>;
  constexpr int G = S::G, NT = S::NT, NKS = S::NKS, KS1 = S::KS1;
  int sizes[5] = {H, HID, HID, HID, H};
  NetDesc net[2];
  int64_t off = fill_net(net[0], 5, sizes, 0);
  fill_net(net[1], 5, sizes, off);
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
          const int g = 2 * ks + (e >> 2);
          if (g < G) put(lane, e, net[u / HID].w_off[0] + (u % HID) * H + 16 * g + 4 * kq + (e & 3));
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
            if (ui < 0 || ui / HID != u / HID || S::assigned_ks(S::tile_nets(m), tile) != ks) continue;
            put(lane, e, net[u / HID].w_off[l] + (u % HID) * HID + ui % HID);
          }
        }
        ++op;
      }
  for (int nn = 0; nn < 2; ++nn)
    for (int ks = 0; ks < NKS; ++ks) {
      if (!S::uses(1 << nn, ks)) continue;
      for (int g = 0; g < G; ++g, ++op)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4;
          for (int e = 0; e < 8; ++e) {
            int tile;
            const int ui = unit_in(ks, kq, e, tile);
            if (ui < 0 || ui / HID != nn || S::assigned_ks(1 << nn, tile) != ks) continue;
            put(lane, e, net[nn].w_off[3] + (16 * g + i) * HID + ui % HID);
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
        if (u < 2 * HID) b[bt * 16 + i] = net[u / HID].b_off[l] + u % HID;
      }
  for (int nn = 0; nn < 2; ++nn)
    for (int g = 0; g < G; ++g, ++bt)
      for (int i = 0; i < 16; ++i) b[bt * 16 + i] = net[nn].b_off[3] + 16 * g + i;
}

template <typename K>
static int resident_blocks(K kernel, int& cus) {
  int per_cu = 0, dev = 0;
  cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kSplitWaves * 64, 0) != hipSuccess || per_cu < 1)
    per_cu = 1;
  return per_cu * cus;
}

template <int H, int HID>
static int launch_split(const float* x, float* y, float* log_det, float* ysq, int accumulate, const uint32_t* simage,
                        const float* image, int64_t rows, int parity, int inverse, hipStream_t stream) {
  static int cus = 256;
  static const int resident = resident_blocks(ahf_split_kernel<H, HID, true>, cus);
  const int64_t n_tiles = (rows + 15) / 16;
  const dim3 grid((unsigned)balanced_grid(n_tiles, kSplitWaves, resident, cus)), block(kSplitWaves * 64);
  if (inverse)
    hipLaunchKernelGGL((ahf_split_kernel<H, HID, true>), grid, block, 0, stream, x, y, log_det, ysq, simage, image,
                       rows, parity, accumulate);
  else
    hipLaunchKernelGGL((ahf_split_kernel<H, HID, false>), grid, block, 0, stream, x, y, log_det, ysq, simage, image,
                       rows, parity, accumulate);
  return check_launch();
}

template <int H, int HID>
static int launch_split_stack(const float* x, float* y, float* log_det, float* ysq, int accumulate,
                              const uint32_t* simages, const float* images, uint32_t parity_bits, int n_layers,
                              int64_t rows, int inverse, hipStream_t stream) {
  static int cus = 256;
  static const int resident = resident_blocks(ahf_split_stack_kernel<H, HID, true>, cus);
  const int64_t n_groups = (rows + 16 * kSplitWaves - 1) / (16 * kSplitWaves);
  const dim3 grid((unsigned)(n_groups < resident ? n_groups : resident)), block(kSplitWaves * 64);
  if (inverse)
    hipLaunchKernelGGL((ahf_split_stack_kernel<H, HID, true>), grid, block, 0, stream, x, y, log_det, ysq, simages,
                       images, parity_bits, n_layers, rows, accumulate);
  else
    hipLaunchKernelGGL((ahf_split_stack_kernel<H, HID, false>), grid, block, 0, stream, x, y, log_det, ysq, simages,
                       images, parity_bits, n_layers, rows, accumulate);
  return check_launch();
}

// (H, HID) pairs with a split kernel; the stack kernel exists for the first four
#define MNF_AHF_SPLIT_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16) X(64, 24) X(128, 24) X(16, 32) X(32, 32) X(64, 32)
#define MNF_AHF_SPLIT_STACK_SHAPES(X) X(16, 24) X(32, 24) X(16, 16) X(32, 16)

static bool uniform3(int n_hidden, const int* hidden, int& hid) {
  if (n_hidden != 3 || !hidden) return false;
  hid = hidden[0];
  return hidden[1] == hid && hidden[2] == hid;
}

static bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

int ahf_split_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate, const void* split_image,
                     const float* image, int64_t rows, int dim, int parity, int inverse, int n_hidden,
                     const int* hidden, int has_scale, int has_shift, hipStream_t stream) {
  int hid = 0;
  if (!split_image || !image || !has_scale || !has_shift || !uniform3(n_hidden, hidden, hid))
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

int ahf_split_stack_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate,
                           const void* split_images, const float* images, uint32_t parity_bits, int n_layers,
                           int64_t rows, int dim, int inverse, int hid, hipStream_t stream) {
  if (!split_images || !images || !aligned16(x, y, split_images, images)) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                                                                               \
  if (dim == 2 * HH && hid == HD)                                                                               \
    return launch_split_stack<HH, HD>(x, y, log_det, ysq, accumulate, static_cast<const uint32_t*>(split_images), \
                                      images, parity_bits, n_layers, rows, inverse != 0, stream);
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
  if (!has_scale || !has_shift || !mnf::uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                                            \
  if (dim == 2 * HH && hid == HD) {                          \
    *n_split_words = mnf::SplitShape<HH, HD>::SPLIT_WORDS;   \
    *n_plain_words = mnf::SplitShape<HH, HD>::PLAIN_WORDS;   \
    return MNF_OK;                                           \
  }
  MNF_AHF_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_affine_half_split_index(int dim, int n_hidden, const int* hidden, int has_scale, int has_shift,
                                int32_t* idx_host) {
  int hid = 0;
  if (!idx_host || !mnf::hidden_ok(n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  if (!has_scale || !has_shift || !mnf::uniform3(n_hidden, hidden, hid)) return MNF_ERR_UNSUPPORTED;
#define X(HH, HD)                              \
  if (dim == 2 * HH && hid == HD) {            \
    mnf::build_split_index<HH, HD>(idx_host);  \
    return MNF_OK;                             \
  }
  MNF_AHF_SPLIT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
