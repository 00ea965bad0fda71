// Host-side declarations shared between the translation units of libmnf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "mnf_device.h"

namespace mnf {

extern thread_local int g_last_hip_error;
// the kernel family of the process's most recent layer launch (mnf_last_kernel()): a static string set at the launch
// site -- "*_generic" names are the any-shape kernels, 20-40 x slower than the matrix-core ones at large batches.
// Process-wide, not per thread: autograd runs the gradient launches on its own thread, and the caller asks from another.
extern std::atomic<const char*> g_last_kernel;
inline void tag_kernel(const char* name) { g_last_kernel.store(name, std::memory_order_relaxed); }
int check_launch();
int64_t fill_net(NetDesc& nd, int n_sizes, const int* sizes, int64_t base);
bool hidden_ok(int n_hidden, const int* hidden);

// Half width the AffineHalfFlow MFMA kernels pad a coupling half of `h` columns to (0: none).  A layer
// whose half is narrower than its tile runs on the stack kernel's ragged variant: zero operands in the
// padded columns, element-wise masked row accesses.
// Hidden width the AffineHalfFlow MFMA kernels run three hidden layers of widths hidden[0..2] at: the smallest of
// 16 / 24 / 32 / 64 that holds the widest one (0: none; 64: the single-layer forward kernels at dim = 32, 64 and 128 only).  Narrower layers get structural-zero units (zero weights and
// bias: LeakyReLU(0) = 0, so they contribute nothing).
inline int ahf_padded_hidden(int n_hidden, const int* hidden) {
  if (n_hidden != 3 || !hidden) return 0;
  int mx = 0;
  for (int i = 0; i < 3; ++i) {
    if (hidden[i] < 1) return 0;
    mx = hidden[i] > mx ? hidden[i] : mx;
  }
  return mx <= 16 ? 16 : mx <= 24 ? 24 : mx <= 32 ? 32 : mx <= 64 ? 64 : 0;
}
inline int ahf_padded_half(int h) { return h < 1 ? 0 : h <= 16 ? 16 : h <= 32 ? 32 : h <= 64 ? 64 : h <= 128 ? 128 : 0; }

// Launch parameters that depend on the device (CU count, occupancy, "dynamic-LDS attribute set") are cached PER
// DEVICE: a process may drive several GPUs, and hipFuncSetAttribute is a per-device setting of the function.
constexpr int kMaxDevices = 64;
inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
inline int device_cus(int dev) {
  static std::atomic<int> cus[kMaxDevices];
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    hipDeviceProp_t prop;
    v = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}
// One non-zero int per device, computed on first use on that device by `compute(dev)` (idempotent: two threads racing
// here compute the same value).  Every call site owns one static DeviceMemo.
struct DeviceMemo {
  std::atomic<int> v[kMaxDevices];
  template <typename F>
  int get(F compute) {
    const int dev = current_device();
    int r = v[dev].load(std::memory_order_relaxed);
    if (r == 0) {
      r = compute(dev);
      v[dev].store(r, std::memory_order_relaxed);
    }
    return r;
  }
};
// resident workgroups of `kernel` on device `dev` by the occupancy query (`fallback` per CU if it fails)
template <typename K>
inline int resident_by_occupancy(K kernel, int threads, int dev, int fallback) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1)
    per_cu = fallback;
  return per_cu * device_cus(dev);
}

// Persistent-grid sizing: among grids of whole workgroups-per-CU steps between resident/2 and
// resident, pick the one whose static tile striding wastes the fewest wave-rounds
// (tiles / (rounds * waves)); e.g. 65,536 tiles on 1,536 resident 4-wave workgroups run 11 rounds at
// 97 % fill, on 1,024 workgroups 16 rounds at 100 %.
inline int64_t balanced_grid(int64_t n_tiles, int waves_per_block, int resident_blocks, int cus) {
  const int64_t need = (n_tiles + waves_per_block - 1) / waves_per_block;
  if (need <= resident_blocks) return need;
  int64_t best = resident_blocks;
  double best_fill = 0.0;
  for (int64_t b = resident_blocks; b >= resident_blocks / 2 && b >= cus; b -= cus) {
    const int64_t waves = b * waves_per_block;
    const int64_t rounds = (n_tiles + waves - 1) / waves;
    const double fill = (double)n_tiles / (double)(rounds * waves);
    if (fill > best_fill + 1e-9) {
      best_fill = fill;
      best = b;
    }
  }
  return best;
}

// The run-time-shaped matrix-core kernels (mnf_rt.h: any layer count and widths, weights from `flat`) take a call from this
// many rows on; below, the VALU any-shape kernels of mnf_generic.hip (one workgroup per few rows) have the lower latency.
constexpr int64_t kRtMinRows = 2048;
int ahf_rt_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate, const float* flat, int64_t rows,
                  int dim, int parity, int inverse, int n_hidden, const int* hidden, int has_scale, int has_shift,
                  hipStream_t stream);

int nsf_rt_launch(const float* x, float* y, float* log_det, int accumulate, const float* flat, int64_t rows, int dim, int K,
                  float tail_bound, int inverse, int n_hidden, const int* hidden, hipStream_t stream);

int rnvp_rt_launch(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                   const float* flat, int64_t rows, int dim, int n_hidden, const int* hidden, hipStream_t stream);

// Specialised launchers: return MNF_ERR_UNSUPPORTED when the shape has no MFMA kernel, in
// which case the caller falls through to the generic kernel.
int ahf_mfma_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate,
                    const float* image, int64_t rows, int dim, int parity, int inverse, int n_hidden,
                    const int* hidden, int has_scale, int has_shift, hipStream_t stream);
int ahf_split_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate, const void* split_image,
                     const float* image, int64_t rows, int dim, int parity, int inverse, int n_hidden,
                     const int* hidden, int has_scale, int has_shift, hipStream_t stream);
int ahf_split_stack_launch(const float* x, float* y, float* mid, float* log_det, float* ysq, int accumulate,
                           const void* split_images, const float* images, uint32_t parity_bits, int n_layers,
                           int64_t rows, int dim, int inverse, int hid, float* log_prob, double* log_prob_sum,
                           hipStream_t stream);
int nsf_mfma_launch(const float* x, float* y, float* log_det, int accumulate, const float* image,
                    const void* split_image, int64_t rows, int dim, int K, float tail_bound, int inverse,
                    int n_hidden, const int* hidden, hipStream_t stream);
// mask == nullptr: the mask is generated in-kernel from `seed`
// split_image != nullptr: the split (f16 hi + lo) kernel, with `image` behind it for out-of-range groups
int rnvp_mfma_launch(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                     const float* image, const void* split_image, int64_t rows, int dim, int n_hidden,
                     const int* hidden, uint64_t seed, hipStream_t stream, const float* q0_mean = nullptr,
                     const float* q0_log_var = nullptr, float* y_out = nullptr, int* y_written = nullptr);

// the register-resident kernel (mnf_rnvp_resident.hip): in-kernel mask only, selected shapes; MNF_ERR_UNSUPPORTED
// sends the caller on to the streaming kernels
// y_out != nullptr (training): additionally writes y = Wn (m z) + bn, rows x 16 * ceil(hn_pad / 16) floats, for the
// gradient pass (NaN rows for groups that took the fp32 body)
int rnvp_resident_launch(const float* z, float* x, float* log_det, int accumulate, const void* split_image,
                         const float* image, int64_t rows, int dim, int hn_pad, uint64_t seed, const float* q0_mean,
                         const float* q0_log_var, int vec, hipStream_t stream, float* y_out = nullptr);

// RNVP on <= MNF_RNVP_FEW_ROWS rows with one hidden layer (mnf_rnvp_few.hip): one workgroup, no atomics.  `flat` is the
// layer's plain parameter buffer (the state_dict order of mnf_rnvp); gradients are ADDED to grad_flat.
bool rnvp_few_ok(int64_t rows, int dim, int n_hidden, const int* hidden);      // gradients: <= MNF_RNVP_FEW_ROWS
bool rnvp_few_fwd_ok(int64_t rows, int dim, int n_hidden, const int* hidden, bool explicit_mask);  // forward: a workgroup per two rows
int rnvp_few_fwd_launch(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                        const float* flat, int64_t rows, int dim, int hid, hipStream_t stream);
int rnvp_few_bwd_launch(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                        float* grad_z, float* grad_flat, const float* flat, float* partial, int64_t rows, int dim,
                        int hid, hipStream_t stream);

// one 32-bit word := 0 on the stream, as a KERNEL node (mnf_generic.hip).  A 4-byte hipMemsetAsync in front of a kernel
// that counts into the word was fine eagerly but, recorded in a hipGraph, faulted after ~100 replays of the MNF-LeNet
// training step (memory access fault; the same step with these resets as kernels replays cleanly).
int zero_word_async(void* word, hipStream_t stream);

// MNF_DETERMINISTIC=1 in the environment (read once): the RNVP and MNFLinear gradient launches leave their parameter
// sums as one block per row part (plain stores into an extension of the caller's workspace, which the *_workspace_bytes
// queries then include) and det_reduce_async adds the blocks up in a fixed order; the workgroups' waves add into LDS
// one after the other.  Off: float atomics, sums that differ in their last bits run to run.  mnf_deterministic().
bool deterministic();
// n floats := 0 / dst[i] += part[0][i] + part[1][i] + ... (rows in order, row r at part + r * stride), as kernel nodes
int zero_floats_async(float* p, int64_t n, hipStream_t stream);
int det_reduce_async(const float* part, int n_rows, int64_t stride, int64_t count, float* dst, hipStream_t stream);
// The fp32 fix-up passes under MNF_DETERMINISTIC: the matrix-core gradient launches hand back the tiles / row groups
// whose operands left the split range as a LIST filled through an atomic counter -- the same ids every run, in any order.
// det_sort_ids_async sorts ids[0 .. min(*count, capacity)) (distinct values in [0, n_items)) ascending, in place, as a
// kernel node; the fix-up pass then runs as ONE workgroup, whose additions to a parameter's sum follow the list: the same
// sums bit for bit every run (slow, but these rows are the exception).  MNF_ERR_UNSUPPORTED beyond 524,288 possible ids
// (8.4 M rows of 16-row tiles): the caller then runs the pass as it does without the switch.
int det_sort_ids_async(int32_t* ids, const int32_t* count, int capacity, int64_t n_items, hipStream_t stream);
// the waves of a workgroup add into LDS: atomically, or -- det -- wave 0, then wave 1, ... with plain read-modify-writes
// (the lanes of one wave must name distinct addresses).  add(op) calls op(float* p, float v) for each of the wave's sums.
template <int WAVES, typename F>
__device__ __forceinline__ void lds_wave_add(bool det, int wave, F&& add) {
  if (!det) {
    add([](float* p, float v) { atomicAdd(p, v); });
    return;
  }
#pragma unroll 1
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) add([](float* p, float v) { *p += v; });
    __syncthreads();
  }
}

// the generic RNVP gradient kernel (mnf_backward.hip); list != nullptr: only the row groups list[1 .. list[0]] (the
// fix-up pass of mnf_rnvp_bwd_mfma)
int rnvp_bwd_generic_launch(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                            float* grad_z, float* grad_flat, const float* flat, int64_t rows, int dim, int n_hidden,
                            const int* hidden, const int32_t* list, int rows_per_group, hipStream_t stream);

}  // namespace mnf
