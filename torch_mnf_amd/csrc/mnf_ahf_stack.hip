// Whole-stack fusion of consecutive AffineHalfFlow layers: the C entry point
// mnf_affine_half_stack (which tries the split kernel of mnf_ahf_split.hip first) and the fp32-MFMA
// stack kernel behind it.
//
// mnf_affine_half_stack runs L coupling layers in ONE launch: a wave keeps its 16 rows in
// registers across all layers, so HBM sees the rows once in and once out (8d + 8 bytes per row
// for the whole stack instead of per layer) and the stack becomes compute bound.  The per-layer
// operand images (25 KB each at d = 64) are streamed from L2 through a double-buffered LDS window
// shared by the eight waves of a workgroup: the image of layer l+1 is requested into registers
// before layer l is computed and handed over at one barrier per layer.
//
// With `intermediates` the output of every layer is written once (never re-read): that is what
// NormalizingFlow.forward/inverse return (the reference's API keeps all L+1 tensors, core.py:20-24).
// Without it every intermediate tensor is dropped: callers opt in through torch_mnf_amd.FusedAffineStack
// and that number is reported separately from the headline metric.
#include <hip/hip_runtime.h>

#include "mnf_ahf_shape.h"
#include "mnf_device.h"
#include "mnf_host.h"

namespace mnf {

constexpr int kStackWaves = 8;

// one coupling layer on rows held in registers: act <- exp(s) act + t (or its inverse), returns sum_j s_j
template <int H, int HID, bool INV>
__device__ __forceinline__ float ahf_layer_regs(const float* img, int lane, int q, const f32x4 (&cnd)[H / 16],
                                                f32x4 (&act)[H / 16]) {
  f32x4 s4[H / 16], t4[H / 16];
  ahf_cond_f32<H, HID>(img, lane, q, cnd, s4, t4);
  return ahf_transform<H, INV>(s4, t4, act);
}

template <int H, int HID, bool INV>
__global__ void __launch_bounds__(kStackWaves * 64, 6)  // three 8-wave workgroups per CU: <= 80 VGPRs
ahf_stack_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mid,
                 float* __restrict__ log_det, float* __restrict__ ysq, const float* __restrict__ images, uint32_t parity_bits, int n_layers,
                 int64_t rows, int accumulate) {
  using S = AhfShape<H, HID>;
  constexpr int G = S::G, dim = 2 * H;
  constexpr int IMG4 = S::IMAGE_FLOATS / 4;
  constexpr int STAGE = (IMG4 + kStackWaves * 64 - 1) / (kStackWaves * 64);
  __shared__ __attribute__((aligned(16))) float lds[2][S::IMAGE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float4* img4 = reinterpret_cast<const float4*>(images);
  auto layer_at = [&](int li) { return INV ? n_layers - 1 - li : li; };  // application order

  const int n_groups = (int)((rows + 16 * kStackWaves - 1) / (16 * kStackWaves));
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t row = (int64_t)grp * (16 * kStackWaves) + wave * 16 + j;
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
      const float4* src = img4 + (int64_t)layer_at(0) * IMG4;
      float4* dst = reinterpret_cast<float4*>(lds[0]);
      for (int k = threadIdx.x; k < IMG4; k += kStackWaves * 64) dst[k] = src[k];
    }
    __syncthreads();
    float ld = 0.f;
    for (int li = 0; li < n_layers; ++li) {
      const int layer = layer_at(li);
      // request the next layer's image (after the last layer: the same one again, branch-free)
      const float4* src = img4 + (int64_t)layer_at(li + 1 < n_layers ? li + 1 : li) * IMG4;
      float4 st[STAGE];
#pragma unroll
      for (int i = 0; i < STAGE; ++i) {
        const int k = threadIdx.x + i * (kStackWaves * 64);
        st[i] = src[k < IMG4 ? k : 0];
      }
      const float* img = lds[li & 1];
      if ((parity_bits >> layer) & 1u)  // conditioner = upper half
        ld += ahf_layer_regs<H, HID, INV>(img, lane, q, hi, lo);
      else
        ld += ahf_layer_regs<H, HID, INV>(img, lane, q, lo, hi);
      if (mid && li + 1 < n_layers && live) {  // the intermediate tensor after this layer
        float* mr = mid + ((int64_t)li * rows + rowc) * dim + 4 * q;
#pragma unroll
        for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(mr + 16 * g) = lo[g];
#pragma unroll
        for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(mr + H + 16 * g) = hi[g];
      }
      float4* dst = reinterpret_cast<float4*>(lds[(li + 1) & 1]);
#pragma unroll
      for (int i = 0; i < STAGE; ++i) {
        const int k = threadIdx.x + i * (kStackWaves * 64);
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

template <int H, int HID>
static int launch_stack(const float* x, float* y, float* mid, float* log_det, float* ysq, int accumulate,
                        const float* images,
                        uint32_t parity_bits, int n_layers, int64_t rows, int inverse, hipStream_t stream) {
  const int64_t n_groups = (rows + 16 * kStackWaves - 1) / (16 * kStackWaves);
  static DeviceMemo memo;
  const int resident = memo.get([](int dev) {
    return resident_by_occupancy(ahf_stack_kernel<H, HID, true>, kStackWaves * 64, dev, 1);
  });
  const int64_t blocks = n_groups < resident ? n_groups : resident;
  const dim3 grid((unsigned)blocks), block(kStackWaves * 64);
  tag_kernel("ahf_stack_fp32");
  if (inverse)
    hipLaunchKernelGGL((ahf_stack_kernel<H, HID, true>), grid, block, 0, stream, x, y, mid, log_det, ysq, images,
                       parity_bits, n_layers, rows, accumulate);
  else
    hipLaunchKernelGGL((ahf_stack_kernel<H, HID, false>), grid, block, 0, stream, x, y, mid, log_det, ysq, images,
                       parity_bits, n_layers, rows, accumulate);
  return check_launch();
}

}  // namespace mnf

extern "C" {

// images: n_layers operand images (mnf_affine_half_image_floats each) back to back, layer 0 first;
// parity_host[l] as in mnf_affine_half.  Layers are applied 0..L-1 (forward) or L-1..0 (inverse).
int mnf_affine_half_stack(const float* x, float* y, float* intermediates, float* log_det, float* y_sqnorm,
                          float* log_prob, double* log_prob_sum, int accumulate, const float* images, const void* split_images, const int* parity_host, int n_layers,
                          int64_t rows, int dim, int inverse, int n_hidden, const int* hidden, void* stream) {
  if (!x || !y || x == y || !images || !parity_host || n_layers < 1 || n_layers > 32 || rows < 0 || dim < 2 ||
      (dim & 1) || !mnf::hidden_ok(n_hidden, hidden) || ((log_prob || log_prob_sum) && !log_det))
    return MNF_ERR_INVALID_ARG;
  if (rows == 0) return MNF_OK;
  const int hid = mnf::ahf_padded_hidden(n_hidden, hidden);  // three hidden layers of <= 32 units, run at 16 / 24 / 32
  if (hid == 0) return MNF_ERR_UNSUPPORTED;
  uint32_t bits = 0;
  for (int l = 0; l < n_layers; ++l) bits |= (parity_host[l] ? 1u : 0u) << l;
  if (split_images) {
    const int rc = mnf::ahf_split_stack_launch(x, y, intermediates, log_det, y_sqnorm, accumulate, split_images, images, bits,
                                               n_layers, rows, dim, inverse, hid, log_prob, log_prob_sum,
                                               (hipStream_t)stream);
    if (rc != MNF_ERR_UNSUPPORTED) return rc;
  }
  if (log_prob || log_prob_sum) return MNF_ERR_UNSUPPORTED;  // only the split kernel has the fused log-prob epilogue
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(images) |
       reinterpret_cast<uintptr_t>(intermediates)) & 15)
    return MNF_ERR_UNSUPPORTED;  // 16-byte row accesses (the split launcher has its own check: narrow halves need none)
#define X(HH, HD) \
  if (dim == 2 * HH && hid == HD) \
    return mnf::launch_stack<HH, HD>(x, y, intermediates, log_det, y_sqnorm, accumulate, images, bits, n_layers, rows, \
                                     inverse != 0, \
                                     (hipStream_t)stream);
  X(16, 24) X(32, 24) X(16, 16) X(32, 16)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // extern "C"
