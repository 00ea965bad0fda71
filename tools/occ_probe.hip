#include "/root/repo/torch_mnf_amd/csrc/mnf_nsf_mfma.hip"
#include <cstdio>
namespace mnf {
thread_local int g_last_hip_error = 0;
int check_launch() { return hipGetLastError() == hipSuccess ? 0 : -3; }
int64_t fill_net(NetDesc& nd, int n_sizes, const int* sizes, int64_t base) { return 0; }
bool hidden_ok(int, const int*) { return true; }
}
int main() {
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mnf::nsf_mfma_kernel<16, 8, 8, true>, 256, 0);
  hipFuncAttributes a; hipFuncGetAttributes(&a, (const void*)mnf::nsf_mfma_kernel<16, 8, 8, true>);
  printf("nsf<16,8,8,inv>: occupancy API %d blocks/CU; numRegs %d sharedSizeBytes %zu localSizeBytes %zu maxThreadsPerBlock %d\n", occ, a.numRegs, a.sharedSizeBytes, a.localSizeBytes, a.maxThreadsPerBlock);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: sharedMemPerMultiprocessor %zu regsPerMultiprocessor %d maxThreadsPerMultiProcessor %d sharedMemPerBlock %zu\n", p.sharedMemPerMultiprocessor, p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor, p.sharedMemPerBlock);
  return 0;
}
