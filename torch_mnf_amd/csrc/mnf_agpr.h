// Rows resident in the ACCUMULATOR half of the register file, in registers the kernel assigns itself (shared by
// mnf_rnvp_resident.hip and mnf_ahf_bwd_split.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mnf_device.h"

namespace mnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the resident rows live in the ACCUMULATOR half of the register file, in registers this file assigns itself:
// group g of the row tile is a[4 g : 4 g + 3].  hipcc allocates at most 256 ordinary VGPRs and treats everything
// above as spills (a plain `f32x4 zr[50]` put 800 dwords per lane in scratch), and values it allocates itself -- also
// "a"-constrained asm operands -- may be copied or spilled by the register allocator at any point, e.g. at the
// loop back-edge, WHILE the asynchronous load that fills them is still in flight.  So the compiler never sees these
// registers as values: reserve_agprs() marks them used (clobbers) so that they are part of the kernel's register
// allocation, the file is compiled with -mllvm -amdgpu-spill-vgpr-to-agpr=0 (the one way hipcc would otherwise touch
// the accumulator half here: as spill space for VGPRs -- a clobber does not keep it from picking the same registers;
// MFMA results are in VGPRs, -amdgpu-mfma-vgpr-form), and every access is an asm statement with the register number as
// an immediate.  All of them are
// volatile: they keep their program order among themselves.
// The loads are asm, so the compiler's s_waitcnt insertion does not know them: row_wait<N> is the explicit wait
// (vector-memory operations complete in issue order; the compiler's own counted waits stay correct with extra
// operations in flight -- they only become stricter).
constexpr int kResAgprs = 208;   // a0 .. a207: the rows (G <= 52)
constexpr int kResMaskAgpr = 208;  // a208 .. a239: the rows' mask words (one per 32 dims)
#define MNF_A4(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
__device__ __forceinline__ void reserve_agprs() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MNF_A4(1), MNF_A4(2), MNF_A4(3),
               MNF_A4(4), MNF_A4(5), MNF_A4(6), MNF_A4(7), MNF_A4(8), MNF_A4(9), MNF_A4(10), MNF_A4(11), MNF_A4(12),
               MNF_A4(13), MNF_A4(14), MNF_A4(15), MNF_A4(16), MNF_A4(17), MNF_A4(18), MNF_A4(19), MNF_A4(20),
               MNF_A4(21), MNF_A4(22), MNF_A4(23));
}
// a92 .. a255: the TOP of the accumulator file (mnf_ahf_bwd_split.hip: its own values overflow the 256 vector
// registers and the compiler places them from a0 upwards; check_agpr.py is told how far up it may go)
constexpr int kTopAgprBase = 92;
__device__ __forceinline__ void reserve_agprs_top() {
  asm volatile("" ::: "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", MNF_A4(10), MNF_A4(11), MNF_A4(12),
               MNF_A4(13), MNF_A4(14), MNF_A4(15), MNF_A4(16), MNF_A4(17), MNF_A4(18), MNF_A4(19), MNF_A4(20),
               MNF_A4(21), MNF_A4(22), MNF_A4(23), MNF_A4(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
#undef MNF_A4
template <int GRP>
__device__ __forceinline__ void row_load(const float* p) {  // a[4 GRP : 4 GRP + 3] <- 16 bytes at p + 64 GRP
  asm volatile("global_load_dwordx4 a[%1:%2], %0, off offset:%3" ::"v"(p), "n"(4 * GRP), "n"(4 * GRP + 3), "n"(64 * GRP)
               : "memory");
}
template <int GRP, int BYTE_OFFSET, bool NT = false>
__device__ __forceinline__ void row_load_at(const float* p) {  // a[4 GRP : 4 GRP + 3] <- 16 bytes at p + BYTE_OFFSET
  if constexpr (NT)  // streamed once: no reason to keep the line in the caches
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off offset:%3 nt" ::"v"(p), "n"(4 * GRP), "n"(4 * GRP + 3),
                 "n"(BYTE_OFFSET)
                 : "memory");
  else
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off offset:%3" ::"v"(p), "n"(4 * GRP), "n"(4 * GRP + 3),
                 "n"(BYTE_OFFSET)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void row_wait() {  // at most N vector-memory operations still in flight
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int GRP>
__device__ __forceinline__ f32x4 row_read() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
               "v_accvgpr_read_b32 %3, a[%7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
               : "n"(4 * GRP), "n"(4 * GRP + 1), "n"(4 * GRP + 2), "n"(4 * GRP + 3));
  return v;
}
template <int REG>
__device__ __forceinline__ uint32_t agpr_get() {
  uint32_t v;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(REG));
  return v;
}
template <int REG>
__device__ __forceinline__ void agpr_put(uint32_t v) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(v), "n"(REG));
}
template <int GRP>
__device__ __forceinline__ void row_write(const f32x4& v) {
  asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%5], %1\n\tv_accvgpr_write_b32 a[%6], %2\n\t"
               "v_accvgpr_write_b32 a[%7], %3" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]),
               "n"(4 * GRP), "n"(4 * GRP + 1), "n"(4 * GRP + 2), "n"(4 * GRP + 3));
}

}  // namespace mnf
