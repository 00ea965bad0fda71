// Microbenchmark: achieved HBM bandwidth of the register-resident RNVP kernels' access pattern, as a function of how
// the per-tile accesses are grouped in time.  A wave owns 16 rows x T 16-float tiles (lane = (row j, quarter q): 16
// bytes per lane per tile, i.e. 16 rows x 64 bytes per instruction, row stride 4 d bytes).  Per pass it "processes"
// the tiles one after the other (a delay of SLEEP x 64 cycles each), storing the tile to x and re-loading its
// registers from the next 64 rows' z -- either right away (B = 1, what the kernels do) or in batches of B tiles.
// EXTRA (round 4, verdict item 4b): a pass covers 96 rows -- 64 in registers as above plus 32 parked in LDS (100 KB,
// filled by LDS-DMA, stored out of LDS, tile by tile like the registers) -- against the SAME operand stream per pass
// (ring cut to 56 KB): does amortising the 613 KB over 96 rows buy the time the linear model predicts?
// build: hipcc -O3 --offload-arch=gfx950 tools/hbm_pattern.hip -o tools/bin/hbm_pattern ; run: tools/bin/hbm_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;
// DMA > 0: every processing step also copies DMA 1 KB pieces per wave of an L2-resident image into LDS (the operand
// ring of the real kernels: 613 KB per 64-row pass per workgroup ~ 2 pieces per wave per step), waiting until at most
// 16 vector-memory operations are in flight
template <int DMA, int WAIT, bool PLAIN, int RING = 128>
__device__ __forceinline__ void dma_step(const uint32_t* image, uint32_t* lds, int& piece, int lane, int wave) {
#pragma unroll
  for (int i = 0; i < DMA; ++i) {
    const int p = (piece++ % 72) * 8 + wave;  // 72 x 8 KB = 576 KB image, 128 KB ring in LDS
    if constexpr (PLAIN) {  // the same bytes as an ordinary load into registers (never read: timing only)
      asm volatile("global_load_dwordx4 a[0:3], %0, off" ::"v"(image + (int64_t)p * 256 + lane * 4) : "memory", "a0", "a1", "a2", "a3");
    } else {
      __builtin_amdgcn_global_load_lds(image + (int64_t)p * 256 + lane * 4, (lds_ptr)(lds + (p % RING) * 256), 16, 0, 0);
    }
  }
  if (DMA > 0 && WAIT < 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
}

template <int T, int B, int SLEEP, int DMA = 0, int WAIT = 16, bool PLAIN = false, int LAYOUT = 0, int NT = 0>
__global__ void __launch_bounds__(512) pattern(const float* __restrict__ z, float* __restrict__ x, int64_t rows, int d, int gemm1_sleep,
                                               const uint32_t* __restrict__ image) {
  extern __shared__ uint32_t lds[];
  int piece = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave >> 1, side = wave & 1;
  const int j = lane & 15, q = lane >> 4;
  const int n_pass = (int)(rows / 64);
  f32x4 r[T];
  // LAYOUT 1: tiles 2p and 2p + 1 are the two 16-byte halves of the lane's 32 contiguous bytes (a row's four lanes
  // cover one whole 128-byte line per pair of tiles)
  auto toff = [](int m) { return LAYOUT ? 32 * (m >> 1) + 4 * (m & 1) : 16 * m; };
  int pass = blockIdx.x;
  if (pass >= n_pass) return;
  {
    const float* zp = z + ((int64_t)pass * 64 + pair * 16 + j) * d + (LAYOUT ? 8 * q : 4 * q) + side * (16 * T);
#pragma unroll
    for (int m = 0; m < T; ++m) r[m] = *reinterpret_cast<const f32x4*>(zp + toff(m));
  }
  for (; pass < n_pass; pass += gridDim.x) {
    const int next = pass + gridDim.x < n_pass ? pass + gridDim.x : blockIdx.x;
    float* xp = x + ((int64_t)pass * 64 + pair * 16 + j) * d + (LAYOUT ? 8 * q : 4 * q) + side * (16 * T);
    const float* zp = z + ((int64_t)next * 64 + pair * 16 + j) * d + (LAYOUT ? 8 * q : 4 * q) + side * (16 * T);
    for (int s = 0; s < gemm1_sleep; ++s) {  // the GEMM-1 phase: no row traffic
      dma_step<DMA, WAIT, PLAIN>(image, lds, piece, lane, wave);
      __builtin_amdgcn_s_sleep(SLEEP);
    }
    if constexpr (B == 0) {  // operand stream only: no row traffic at all
      for (int m = 0; m < T; ++m) {
        dma_step<DMA, WAIT, PLAIN>(image, lds, piece, lane, wave);
        __builtin_amdgcn_s_sleep(SLEEP);
      }
    } else
#pragma unroll
    for (int m0 = 0; m0 < T; m0 += B) {
#pragma unroll
      for (int m = m0; m < m0 + B && m < T; ++m) {
        dma_step<DMA, WAIT, PLAIN>(image, lds, piece, lane, wave);
        __builtin_amdgcn_s_sleep(SLEEP);
        r[m] = r[m] * 1.0001f + 1.f;
        asm volatile("" : "+v"(r[m]));
      }
#pragma unroll
      for (int m = m0; m < m0 + B && m < T; ++m) {
        if (NT & 1) __builtin_nontemporal_store(r[m], reinterpret_cast<f32x4*>(xp + toff(m)));
        else *reinterpret_cast<f32x4*>(xp + toff(m)) = r[m];
      }
#pragma unroll
      for (int m = m0; m < m0 + B && m < T; ++m)
        r[m] = (NT & 2) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zp + toff(m))) : *reinterpret_cast<const f32x4*>(zp + toff(m));
    }
  }
}

// (the LDS-DMA builtin does not exist in the host pass: named directly in a __global__ template it silently costs the
//  kernel its host stub)
__device__ __forceinline__ void lds_dma16(const float* g, uint32_t* l) {
  __builtin_amdgcn_global_load_lds(g, (lds_ptr)l, 16, 0, 0);
}
// 96 rows per pass: the 64 register rows of `pattern` (B = 1) + 32 rows in LDS behind a 56 KB operand ring.
// Step m of a pass: every wave its register tile (store, reload from the next pass), four of the eight waves one
// 1 KB unit of the LDS rows (16 rows x 16 dims: read from LDS, store, refill from the next pass by LDS-DMA).
template <int T, int SLEEP, int DMA, int WAIT>
__global__ void __launch_bounds__(512) pattern96(const float* __restrict__ z, float* __restrict__ x, int64_t rows, int d, int gemm1_sleep,
                                                 const uint32_t* __restrict__ image) {
  extern __shared__ uint32_t lds[];
  constexpr int RING = 56;
  uint32_t* const rows_lds = lds + RING * 256;  // 100 units of 256 words
  int piece = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave >> 1, side = wave & 1;
  const int j = lane & 15, q = lane >> 4;
  const int n_pass = (int)(rows / 96);
  f32x4 r[T];
  int pass = blockIdx.x;
  if (pass >= n_pass) return;
  auto unit_ptr = [&](int ps, int u) {  // unit u = 50 (row group) + tile: rows 64 + 16 rg + j, dims 16 tile + 4 q
    return ((int64_t)ps * 96 + 64 + (u / 50) * 16 + j) * d + (u % 50) * 16 + 4 * q;
  };
  {
    const float* zp = z + ((int64_t)pass * 96 + pair * 16 + j) * d + 4 * q + side * (16 * T);
#pragma unroll
    for (int m = 0; m < T; ++m) r[m] = *reinterpret_cast<const f32x4*>(zp + 16 * m);
    for (int u = wave; u < 100; u += 8)
      lds_dma16(z + unit_ptr(pass, u), rows_lds + u * 256);
  }
  for (; pass < n_pass; pass += gridDim.x) {
    const int next = pass + gridDim.x < n_pass ? pass + gridDim.x : blockIdx.x;
    float* xp = x + ((int64_t)pass * 96 + pair * 16 + j) * d + 4 * q + side * (16 * T);
    const float* zp = z + ((int64_t)next * 96 + pair * 16 + j) * d + 4 * q + side * (16 * T);
    for (int s = 0; s < gemm1_sleep; ++s) {  // the GEMM-1 phase: no row traffic
      dma_step<DMA, WAIT, false, RING>(image, lds, piece, lane, wave);
      __builtin_amdgcn_s_sleep(SLEEP);
    }
#pragma unroll
    for (int m = 0; m < T; ++m) {
      dma_step<DMA, WAIT, false, RING>(image, lds, piece, lane, wave);
      __builtin_amdgcn_s_sleep(SLEEP);
      r[m] = r[m] * 1.0001f + 1.f;
      asm volatile("" : "+v"(r[m]));
      *reinterpret_cast<f32x4*>(xp + 16 * m) = r[m];
      r[m] = *reinterpret_cast<const f32x4*>(zp + 16 * m);
      if ((wave & 1) == (m & 1)) {
        const int u = 4 * m + (wave >> 1);
        f32x4 v = *reinterpret_cast<const f32x4*>(rows_lds + u * 256 + lane * 4);
        v = v * 1.0001f + 1.f;
        *reinterpret_cast<f32x4*>(x + unit_ptr(pass, u)) = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the read has left LDS before the refill lands)
        lds_dma16(z + unit_ptr(next, u), rows_lds + u * 256);
      }
    }
    __syncthreads();  // (the real kernel's per-pass barrier: the next pass's GEMM 1 reads every LDS row)
  }
}

static uint32_t* g_image = nullptr;
template <int SLEEP, int DMA, int WAIT = 32>
void run96(const float* z, float* x, int64_t rows, int d, int g1) {
  constexpr int T = 25;
  const size_t lds_bytes = (56 + 100) * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pattern96<T, SLEEP, DMA, WAIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) pattern96<T, SLEEP, DMA, WAIT><<<256, 512, lds_bytes>>>(z, x, rows, d, g1, g_image);
  CK(hipEventRecord(a));
  const int N = 10;
  for (int i = 0; i < N; ++i) pattern96<T, SLEEP, DMA, WAIT><<<256, 512, lds_bytes>>>(z, x, rows, d, g1, g_image);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / N, bytes = 8.0 * (rows / 96 * 96) * d;
  printf("96 rows/pass  sleep %3d x64 cycles/tile  gemm1 %2d  dma %d KB/step/wave wait vmcnt(%d) : %7.1f us  %5.2f TB/s  (operand stream %.0f KB per pass)\n",
         SLEEP, g1, DMA, WAIT, us, bytes / us / 1e6, 8.0 * DMA * (g1 + T));
}
template <int B, int SLEEP, int DMA = 0, int WAIT = 16, bool PLAIN = false, int LAYOUT = 0, int NT = 0>
void run(const float* z, float* x, int64_t rows, int d, int g1) {
  constexpr int T = LAYOUT ? 24 : 25;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pattern<T, B, SLEEP, DMA, WAIT, PLAIN, LAYOUT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) pattern<T, B, SLEEP, DMA, WAIT, PLAIN, LAYOUT, NT><<<256, 512, 128 * 1024>>>(z, x, rows, d, g1, g_image);
  CK(hipEventRecord(a));
  const int N = 10;
  for (int i = 0; i < N; ++i) pattern<T, B, SLEEP, DMA, WAIT, PLAIN, LAYOUT, NT><<<256, 512, 128 * 1024>>>(z, x, rows, d, g1, g_image);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / N, bytes = 8.0 * rows * d;
  printf("batch %2d  sleep %3d x64 cycles/tile  gemm1 %2d  dma %d KB/step/wave wait vmcnt(%d) %s nt=%d : %7.1f us  %5.2f TB/s\n", B, SLEEP, g1, DMA, WAIT, LAYOUT ? "lds-dma, line-paired tiles (24 of 25)" : PLAIN ? "plain loads" : "lds-dma", NT, us, bytes / us / 1e6);
}

int main(int argc, char** argv) {
  const int64_t rows = 256000; const int d = 800;
  float *z, *x;
  CK(hipMalloc(&z, rows * d * 4)); CK(hipMalloc(&x, rows * d * 4));
  CK(hipMemset(z, 0, rows * d * 4)); CK(hipMemset(x, 0, rows * d * 4));
  CK(hipMalloc(&g_image, 1 << 20)); CK(hipMemset(g_image, 0, 1 << 20));
  if (argc > 1 && argv[1][0] == '9') {  // item 4b only: the 64-row pattern at the kernel's settings, then 96 rows per pass
    run<1, 22, 2, 32>(z, x, rows, d, 12); run<1, 0, 2, 32>(z, x, rows, d, 12); run<1, 22, 1, 32>(z, x, rows, d, 12);
    // 96 rows: 1.5 x the "compute" per pass (the same 37 steps, 33 instead of 22 x 64 cycles each), the same 592 KB
    // operand stream per pass (2 KB per step per wave); then without compute, with half / no operand stream, and with
    // the 64-row pass's compute (an arithmetic that got no slower: the lower bound of what 96 rows can cost)
    run96<33, 2>(z, x, rows, d, 12); run96<0, 2>(z, x, rows, d, 12); run96<33, 1>(z, x, rows, d, 12); run96<33, 0>(z, x, rows, d, 12);
    run96<22, 2>(z, x, rows, d, 12);
    return 0;
  }
  // no delay at all: the pattern's own ceiling
  run<1, 0>(z, x, rows, d, 0); run<5, 0>(z, x, rows, d, 0); run<25, 0>(z, x, rows, d, 0);
  // ~ 350 us of "compute" per launch = 22.4 us per pass = ~ 53k cycles: 12 GEMM-1 units + 25 tiles -> 1400 cycles each
  run<1, 22>(z, x, rows, d, 12); run<5, 22>(z, x, rows, d, 12); run<25, 22>(z, x, rows, d, 12);
  run<1, 12>(z, x, rows, d, 12); run<5, 12>(z, x, rows, d, 12); run<25, 12>(z, x, rows, d, 12);
  // the same with the operand stream through the same L1
  run<1, 22, 2>(z, x, rows, d, 12); run<5, 22, 2>(z, x, rows, d, 12); run<25, 22, 2>(z, x, rows, d, 12);
  run<1, 22, 1>(z, x, rows, d, 12); run<1, 22, 4>(z, x, rows, d, 12);
  // is it the wave's in-order wait or the L1 itself?  the same stream with later / no waits
  run<1, 22, 2, 32>(z, x, rows, d, 12); run<1, 22, 2, 48>(z, x, rows, d, 12); run<1, 22, 2, 63>(z, x, rows, d, 12);
  run<1, 22, 2, 4>(z, x, rows, d, 12);
  run<1, 0, 2, 32>(z, x, rows, d, 12); run<0, 0, 2, 32>(z, x, rows, d, 12); run<0, 22, 2, 32>(z, x, rows, d, 12); run<0, 0, 4, 32>(z, x, rows, d, 12);
  run<1, 22, 2, 32, true>(z, x, rows, d, 12); run<1, 22, 1, 32, true>(z, x, rows, d, 12); run<1, 22, 4, 32, true>(z, x, rows, d, 12);
  // tiles paired into whole 128-byte lines (needs the K order of the operand image permuted to match)
  run<1, 0, 0, 16, false, 1>(z, x, rows, d, 0); run<2, 0, 0, 16, false, 1>(z, x, rows, d, 0);
  run<1, 22, 0, 16, false, 1>(z, x, rows, d, 12); run<2, 22, 0, 16, false, 1>(z, x, rows, d, 12);
  run<1, 22, 2, 32, false, 1>(z, x, rows, d, 12); run<2, 22, 2, 32, false, 1>(z, x, rows, d, 12); run<8, 22, 2, 32, false, 1>(z, x, rows, d, 12);
  // non-temporal hints (1 stores, 2 loads, 3 both), standard and line-paired layout, with the operand stream
  run<1, 22, 2, 32, false, 0, 1>(z, x, rows, d, 12); run<1, 22, 2, 32, false, 0, 2>(z, x, rows, d, 12); run<1, 22, 2, 32, false, 0, 3>(z, x, rows, d, 12);
  run<2, 22, 2, 32, false, 1, 1>(z, x, rows, d, 12); run<2, 22, 2, 32, false, 1, 2>(z, x, rows, d, 12); run<2, 22, 2, 32, false, 1, 3>(z, x, rows, d, 12);
  // standard layout, two tiles (the two 64-byte halves of a line) back to back, with / without the hint on the stores
  run<2, 22, 2, 32, false, 0, 0>(z, x, rows, d, 12); run<2, 22, 2, 32, false, 0, 1>(z, x, rows, d, 12); run<2, 22, 2, 32, false, 0, 3>(z, x, rows, d, 12);
  run<4, 22, 2, 32, false, 0, 1>(z, x, rows, d, 12);
  return 0;
}
