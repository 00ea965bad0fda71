// Which launch/loop shape moves 2^20 rows of 256 B (read + write + log_det RMW) fastest with the
// MFMA operand access shape (16 rows x 64 B per instruction)?  Persistent loops vs one tile per
// wave vs deeper prefetch vs dynamic tile claiming vs non-temporal hints.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f32x4 ld4(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT> __device__ __forceinline__ void st4(float* p, f32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = v;
}

// MODE 0 persistent static stride; 1 persistent + next tile prefetched before the stores; 2 two tiles per
// iteration; 3 dynamic claiming (atomic counter); 4 = 0 with nt loads+stores; 5 = 1 with nt; 6 = 1 with nt loads only
template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) copy_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ ld, int n_tiles, int* counter) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int stride = gridDim.x * WAVES;
  constexpr bool NTL = MODE == 4 || MODE == 5 || MODE == 6, NTS = MODE == 4 || MODE == 5;
  auto src = [&](int t) { return x + ((size_t)t * 16 + j) * 64 + 4 * q; };
  auto dst = [&](int t) { return y + ((size_t)t * 16 + j) * 64 + 4 * q; };
  if (MODE == 0 || MODE == 4) {
    for (int tile = blockIdx.x * WAVES + wave; tile < n_tiles; tile += stride) {
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = ld4<NTL>(src(tile) + 16 * g);
#pragma unroll
      for (int g = 0; g < 4; ++g) st4<NTS>(dst(tile) + 16 * g, v[g]);
      if (q == 0) ld[(size_t)tile * 16 + j] += v[0][0];
    }
  } else if (MODE == 1 || MODE == 5 || MODE == 6) {
    int tile = blockIdx.x * WAVES + wave;
    f32x4 v[4], n[4];
    if (tile < n_tiles) {
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = ld4<NTL>(src(tile) + 16 * g);
    }
    for (; tile < n_tiles; tile += stride) {
      const int nt = tile + stride < n_tiles ? tile + stride : n_tiles - 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) n[g] = ld4<NTL>(src(nt) + 16 * g);
#pragma unroll
      for (int g = 0; g < 4; ++g) st4<NTS>(dst(tile) + 16 * g, v[g]);
      if (q == 0) ld[(size_t)tile * 16 + j] += v[0][0];
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = n[g];
    }
  } else if (MODE == 2) {
    for (int tile = 2 * (blockIdx.x * WAVES + wave); tile < n_tiles; tile += 2 * stride) {
      f32x4 v[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) v[g] = ld4<false>(src(tile + (g >> 2)) + 16 * (g & 3));
#pragma unroll
      for (int g = 0; g < 8; ++g) st4<false>(dst(tile + (g >> 2)) + 16 * (g & 3), v[g]);
      if (q < 2) ld[(size_t)(tile + q) * 16 + j] += v[4 * q][0];
    }
  } else if (MODE == 3) {
    for (;;) {
      int tile = 0;
      if (lane == 0) tile = atomicAdd(counter, 1);
      tile = __builtin_amdgcn_readfirstlane(tile);
      if (tile >= n_tiles) break;
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = ld4<false>(src(tile) + 16 * g);
#pragma unroll
      for (int g = 0; g < 4; ++g) st4<false>(dst(tile) + 16 * g, v[g]);
      if (q == 0) ld[(size_t)tile * 16 + j] += v[0][0];
    }
  }
}

template <int MODE, int WAVES>
float run(const float* x, float* y, float* ld, int n_tiles, int blocks, int* counter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10;
  for (int i = 0; i < 2; ++i) { hipMemsetAsync(counter, 0, 4); hipLaunchKernelGGL((copy_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, x, y, ld, n_tiles, counter); }
  float total = 0;
  for (int i = 0; i < iters; ++i) {
    hipMemsetAsync(counter, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL((copy_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, x, y, ld, n_tiles, counter);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total += ms;
  }
  return total * 1000.f / iters;
}

int main() {
  const size_t rows = 1 << 20;
  float *x, *y, *ld; int* counter;
  hipMalloc(&x, rows * 256); hipMalloc(&y, rows * 256); hipMalloc(&ld, rows * 4); hipMalloc(&counter, 4);
  hipMemset(x, 0, rows * 256); hipMemset(ld, 0, rows * 4);
  const int n_tiles = rows / 16;
  const char* names[] = {"persistent", "persist+prefetch", "2 tiles/iter", "dynamic", "nt ld+st", "prefetch+nt", "prefetch+nt-ld"};
  for (int round = 0; round < 2; ++round)
    for (int wpc : {8, 16, 24, 32}) {  // waves per CU
      printf("waves/CU %2d (8-wave blocks):", wpc);
      const int blocks = 256 * wpc / 8;
      printf(" %s %.1f |", names[0], run<0, 8>(x, y, ld, n_tiles, blocks, counter));
      printf(" %s %.1f |", names[1], run<1, 8>(x, y, ld, n_tiles, blocks, counter));
      printf(" %s %.1f |", names[2], run<2, 8>(x, y, ld, n_tiles, blocks, counter));
      printf(" %s %.1f |", names[3], run<3, 8>(x, y, ld, n_tiles, blocks, counter));
      printf(" %s %.1f |", names[4], run<4, 8>(x, y, ld, n_tiles, blocks, counter));
      printf(" %s %.1f |", names[5], run<5, 8>(x, y, ld, n_tiles, blocks, counter));
      printf(" %s %.1f\n", names[6], run<6, 8>(x, y, ld, n_tiles, blocks, counter));
    }
  printf("one tile per wave (4-wave blocks): %.1f\n", run<0, 4>(x, y, ld, n_tiles, n_tiles / 4, counter));
  printf("one tile per wave (8-wave blocks): %.1f\n", run<0, 8>(x, y, ld, n_tiles, n_tiles / 8, counter));
  return 0;
}
