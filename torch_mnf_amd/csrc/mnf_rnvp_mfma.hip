// RNVP specialised kernel (placeholder until the tiled MFMA GEMM kernel lands: every shape
// reports "unsupported", so mnf_rnvp runs the generic kernel).
#include "mnf_host.h"

namespace mnf {
int rnvp_mfma_launch(const float*, const float*, float*, float*, int, const float*, int64_t, int, int,
                     const int*, hipStream_t) {
  return MNF_ERR_UNSUPPORTED;
}
}  // namespace mnf

extern "C" {
int64_t mnf_rnvp_image_floats(int, int, const int*) { return 0; }
int mnf_rnvp_image_index(int, int, const int*, int32_t*) { return MNF_ERR_UNSUPPORTED; }
}
