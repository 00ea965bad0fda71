// NSF_CL specialised kernel (placeholder until the MFMA spline kernel lands: every shape
// reports "unsupported", so mnf_nsf_cl runs the generic kernel).
#include "mnf_host.h"

namespace mnf {
int nsf_mfma_launch(const float*, float*, float*, int, const float*, int64_t, int, int, float, int, int,
                    const int*, hipStream_t) {
  return MNF_ERR_UNSUPPORTED;
}
}  // namespace mnf

extern "C" {
int64_t mnf_nsf_cl_image_floats(int, int, int, const int*) { return 0; }
int mnf_nsf_cl_image_index(int, int, int, const int*, int32_t*) { return MNF_ERR_UNSUPPORTED; }
}
