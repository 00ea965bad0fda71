// Host-side declarations shared between the translation units of libmnf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mnf_device.h"

namespace mnf {

extern thread_local int g_last_hip_error;
int check_launch();
int64_t fill_net(NetDesc& nd, int n_sizes, const int* sizes, int64_t base);
bool hidden_ok(int n_hidden, const int* hidden);

// Specialised launchers: return MNF_ERR_UNSUPPORTED when the shape has no MFMA kernel, in
// which case the caller falls through to the generic kernel.
int ahf_mfma_launch(const float* x, float* y, float* log_det, float* ysq, int accumulate,
                    const float* image, int64_t rows, int dim, int parity, int inverse, int n_hidden,
                    const int* hidden, int has_scale, int has_shift, hipStream_t stream);
int nsf_mfma_launch(const float* x, float* y, float* log_det, int accumulate, const float* image,
                    int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden,
                    const int* hidden, hipStream_t stream);
// mask == nullptr: the mask is generated in-kernel from `seed`
int rnvp_mfma_launch(const float* z, const float* mask, float* x, float* log_det, int accumulate,
                     const float* image, int64_t rows, int dim, int n_hidden, const int* hidden,
                     uint64_t seed, hipStream_t stream);

}  // namespace mnf
