// Masked / gated RNVP coupling (flows/rnvp.py:25-39), rows resident in registers, TWO waves per 16-row tile (gfx950).
//
// mnf_rnvp_resident.hip keeps a wave's 16 rows -- d / 4 = 200 registers per lane at d = 800 -- in the register file so
// that every z is read once; that takes the whole 512-entry file, i.e. one wave per SIMD, and a lone wave issues a
// vector instruction only every 4 cycles and has nobody to hide its stalls (profiles/r2: the compute side alone costs
// 344 us of the 553-590).  Here a PAIR of waves shares a tile: side 0 holds the first GA 16-dim groups of the 16
// rows, side 1 the rest (about 100 registers each), so a workgroup is eight waves at two per SIMD:
//
//   GEMM 1   each side accumulates y over ITS dims (the K axis splits);
//   y        side 0 hands its partial y to side 1 through LDS, side 1 adds its own and hands the sum back: both
//            hold the full y (and the pair's range verdict rides on the second barrier);
//   GEMM 2   each side computes shift / scale and the gate epilogue for ITS output dims -- exactly the dims whose
//            z it holds -- stores x and re-loads the registers with the same dims of the pair's next 16 rows;
//   log_det  side 1's row sums go to side 0 through LDS.
//
// The rows live in the accumulator half of the wave's 256 registers, managed by hand as in the resident kernel
// (mnf_agpr.h: as ordinary values hipcc kept two copies of the loop-carried array and spilled).  The operand image (the same one as the other split kernels) is streamed L2 -> LDS by LDS-DMA in
// 32 KB slots -- 16 KB of side 0's K-steps / tiles and 16 KB of side 1's -- four buffers, three slots ahead, one
// barrier per slot; the waits for the pieces are counted at compile time per side (PairPlan), as in the resident
// kernel.  64 rows per pass, like there: the same operand traffic per row.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <utility>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_split.h"
#include "mnf_agpr.h"

namespace mnf {

typedef __attribute__((address_space(3))) void* lds_void_ptr_p;

#ifndef MNF_PAIR_ABL
#define MNF_PAIR_ABL 0  // timing only, results wrong: 1 stores to one line, 2 row re-loads from one line, 3 both
#endif
constexpr int kPairAbl = MNF_PAIR_ABL;
#ifndef MNF_PAIR_NT
#define MNF_PAIR_NT 0  // 1: x stored with the non-temporal hint, 2: the row re-loads carry it, 3: both
#endif
constexpr int kPairNt = MNF_PAIR_NT;
constexpr int kPairWaves = 8;      // four pairs: 64 rows per pass
constexpr int kPairBufs = 4;       // LDS operand buffers: a slot is requested kPairBufs - 1 slots before it is used
constexpr int kPairColdWords = 64; // one bit per 128-row super-group of a workgroup that has to be redone in fp32

template <typename F, int... I>
__device__ __forceinline__ void pair_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void pair_static_for(F&& f) {
  pair_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// The split of the row between the two sides and each side's stream of vector-memory operations per 64-row pass
// (known at compile time: the waits for the LDS-DMA pieces are counts of operations still in flight).
template <int HN, int G>
struct PairPlan {
  using S = RnvpSplitShape<HN>;
  static constexpr int D = kPairBufs - 1;
  static constexpr int NKS1 = (G + 1) / 2;
  static constexpr int GA = 2 * (G / 4);  // side 0: groups [0, GA) (whole K-steps); side 1: [GA, G)
  static constexpr int groups(int side) { return side ? G - GA : GA; }
  static constexpr int ksteps(int side) { return side ? NKS1 - GA / 2 : GA / 2; }
  static constexpr int first_ks(int side) { return side ? GA / 2 : 0; }
  static constexpr int first_tile(int side) { return side ? GA : 0; }
  static constexpr int max2(int a, int b) { return a > b ? a : b; }
  static constexpr int SLOTS1 = (max2(ksteps(0), ksteps(1)) + 1) / 2;  // two K-steps per side per slot
  static constexpr int SLOTS2 = (max2(groups(0), groups(1)) + 1) / 2;  // two tiles per side per slot
  static constexpr int NC = SLOTS1 + SLOTS2;
  static_assert(S::KS1_WORDS == S::TILE2_WORDS, "a K-step and a tile have the same operand size (YT = 2 NKS2)");
  static constexpr int UNIT_WORDS = S::KS1_WORDS;            // one K-step / one tile
  static constexpr int SIDE_WORDS = 2 * UNIT_WORDS;          // a side's part of a slot
  static constexpr int SLOT_WORDS = 2 * SIDE_WORDS;
  static constexpr int UNIT_PIECES = UNIT_WORDS / 256;       // 1 KB LDS-DMA pieces per unit
  static_assert(UNIT_PIECES % kPairWaves == 0, "every wave copies the same number of pieces of a unit");
  // units (K-steps or tiles) of `side` in slot c (c taken modulo NC)
  static constexpr int units(int side, int c) {
    c %= NC;
    const int have = c < SLOTS1 ? ksteps(side) - 2 * c : groups(side) - 2 * (c - SLOTS1);
    return have < 0 ? 0 : have > 2 ? 2 : have;
  }
  static constexpr int pieces(int c) { return (units(0, c) + units(1, c)) * (UNIT_PIECES / kPairWaves); }  // per wave
  // operations issued by `side` from the start of a pass up to: the end of the request block of slot c ...
  static constexpr int ops_top(int side, int c) {
    int n = 0;
    for (int k = 0; k <= c; ++k) {
      n += pieces(k + D);
      if (k < c && k >= SLOTS1) n += 2 * units(side, k);  // the tiles of earlier GEMM-2 slots: store + load each
    }
    return n;
  }
  // ... and the end of slot c
  static constexpr int ops_end(int side, int c) { return ops_top(side, c) + (c >= SLOTS1 ? 2 * units(side, c) : 0); }
  static constexpr int ops_pass(int side) { return ops_end(side, NC - 1); }
  // in flight behind this wave's pieces of slot e + 1 at the end of slot e (requested at the top of slot e + 1 - D)
  static constexpr int dma_wait(int side, int e) {
    const int top = e + 1 - D;
    const int n = top >= 0 ? ops_end(side, e) - ops_top(side, top) : ops_end(side, e) + ops_pass(side) - ops_top(side, top + NC);
    return n < 63 ? n : 63;
  }
  // operations issued up to and including the re-load of local tile lt (GEMM-2 slot SLOTS1 + lt / 2: store, load)
  static constexpr int ops_after_load(int side, int lt) { return ops_top(side, SLOTS1 + lt / 2) + 2 * (lt % 2 + 1); }
  // in flight behind that load (issued by the previous pass) when GEMM-1 slot c of this pass wants the tile
  static constexpr int row_wait(int side, int lt, int c) {
    const int n = ops_pass(side) - ops_after_load(side, lt) + ops_top(side, c);
    return n < 63 ? n : 63;
  }
  static constexpr size_t lds_bytes() {
    return sizeof(uint32_t) * ((size_t)kPairBufs * SLOT_WORDS + S::plain_words(16 * G) + 2 * 16 * G + (kPairWaves / 2) * 1024 +
                               kPairColdWords);
  }
};

// The fp32 recomputation of a flagged 128-row super-group: plain fmaf chains in k order over the weights of the fp32
// operand image (addressed through its layout, mnf_rnvp_mfma.hip build_index), y and the row sums in LDS.  Slow
// (tens of microseconds per super-group) and rare; it must be lean in registers: a call to the streaming kernels'
// fp32 MFMA body would raise the whole kernel's register allocation beyond two waves per SIMD.
template <int HN, int G, bool SAMPLE>
__device__ __forceinline__ void rnvp_pair_cold(float* lds, int sg, const float* __restrict__ z, float* __restrict__ x,
                                               float* __restrict__ log_det, const float* __restrict__ image, int64_t rows,
                                               int accumulate, uint64_t seed, const float* zprm) {
  using F = RnvpShape<HN>;
  constexpr int d = 16 * G;
  const float* img1 = image;
  const float* img2 = image + F::part1_floats(d);
  const float* bias_y = img2 + F::part2_floats(d);
  auto Wn = [&](int u, int col) -> float {  // net.0.weight[u][col]
    const int m = u >> 4, rem = u & 15, i = (rem & 3) * 4 + (rem >> 2);
    const int g = col >> 4, kq = (col & 15) >> 2, e = col & 3;
    return img1[(int64_t)(4 * g + e) * 256 + (kq * 16 + i) * 4 + m];
  };
  auto bn = [&](int u) -> float {
    const int m = u >> 4, rem = u & 15, i = (rem & 3) * 4 + (rem >> 2);
    return bias_y[m * 16 + i];
  };
  auto Wts = [&](int which, int jdim, int unit) -> float {  // t.weight / s.weight [jdim][unit]
    const int m = jdim >> 4, i = jdim & 15, c = unit >> 2, kq = unit & 3, n = 2 * c + which;
    return img2[(int64_t)m * F::TILE2_FLOATS + (n >> 2) * 256 + (kq * 16 + i) * 4 + (n & 3)];
  };
  auto bts = [&](int which, int jdim) -> float {
    return img2[(int64_t)(jdim >> 4) * F::TILE2_FLOATS + F::G2 * 256 + which * 16 + (jdim & 15)];
  };
  float* y = lds;             // [128][HN]
  float* ld_rows = y + 128 * HN;  // [128]
  const int64_t row0 = (int64_t)sg * 128;
  const int n_rows = (int)(rows - row0 < 128 ? rows - row0 : 128);
  auto zval = [&](int64_t row, int k) -> float {
    const float v = z[row * d + k];
    return SAMPLE ? fmaf(v, zprm[d + k], zprm[k]) : v;
  };
  __syncthreads();
  for (int r = threadIdx.x; r < 128; r += blockDim.x) ld_rows[r] = 0.f;
  for (int idx = threadIdx.x; idx < n_rows * HN; idx += blockDim.x) {
    const int r = idx / HN, u = idx - r * HN;
    float acc = bn(u);
    for (int k = 0; k < d; ++k) acc = fmaf(Wn(u, k), rnvp_mask_bit(seed, row0 + r, k) * zval(row0 + r, k), acc);
    y[idx] = acc;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n_rows * d; idx += blockDim.x) {
    const int r = idx / d, jd = idx - r * d;
    float shift = bts(0, jd), scale = bts(1, jd);
    for (int u = 0; u < HN; ++u) {
      shift = fmaf(Wts(0, jd, u), y[r * HN + u], shift);
      scale = fmaf(Wts(1, jd, u), y[r * HN + u], scale);
    }
    const float m = rnvp_mask_bit(seed, row0 + r, jd), zz = zval(row0 + r, jd), gate = sigmoidf(scale);
    x[(row0 + r) * d + jd] = ((1.f - m) * zz * gate + (1.f - gate) * shift) + m * zz;  // rnvp.py:37
    atomicAdd(&ld_rows[r], (1.f - m) * logf(gate));                                       // :36
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n_rows; r += blockDim.x) {
    float* p = log_det + row0 + r;
    *p = accumulate ? *p + ld_rows[r] : ld_rows[r];
  }
  __syncthreads();
}

template <int HN, int G, bool SAMPLE>
__global__ void __launch_bounds__(kPairWaves * 64, 2)
rnvp_pair_kernel(const float* __restrict__ z, float* __restrict__ x, float* __restrict__ log_det,
                 const uint32_t* __restrict__ simage, const float* __restrict__ image, int64_t rows, int accumulate,
                 uint64_t seed, const float* __restrict__ q0_mean, const float* __restrict__ q0_log_var) {
  using S = RnvpSplitShape<HN>;
  using P = PairPlan<HN, G>;
  constexpr int d = 16 * G, YT = S::YT, NKS2 = S::NKS2, NB = kPairBufs, D = P::D, NC = P::NC;
  static_assert(NC % NB == 0, "the operand ring keeps its buffer assignment from one pass to the next");
  static_assert(128 * HN + 128 <= NB * P::SLOT_WORDS, "the fp32 path's y and row sums fit the operand buffers");
  constexpr int MAXG = P::groups(1) > P::groups(0) ? P::groups(1) : P::groups(0);
  constexpr int kMaskAgpr = 4 * MAXG;  // a[4 lt : 4 lt + 3]: local tile lt of the pair's rows; a[kMaskAgpr + lk]: mask words
  static_assert(kMaskAgpr + (MAXG + 1) / 2 <= 117 && 64 * (G - 1) < 4096, "reserved registers / immediates");
  reserve_agprs_117();

  // LDS: [NB operand buffers][(bt | bs) per tile, then bn][mean | std of the prologue][pair exchange 4 x 4 KB][flags]
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  float* const bias_lds = reinterpret_cast<float*>(lds_dyn + NB * P::SLOT_WORDS);
  float* const zprm_lds = bias_lds + S::plain_words(d);
  float* const xch_lds = zprm_lds + 2 * d;
  uint32_t* const cold_flags = reinterpret_cast<uint32_t*>(xch_lds + (kPairWaves / 2) * 1024);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar)
  const int pair = wave >> 1, side = wave & 1;
  const int j = lane & 15, q = lane >> 4;

  if (threadIdx.x < kPairColdWords) cold_flags[threadIdx.x] = 0u;
  {
    const float* bias_src = reinterpret_cast<const float*>(simage + S::split_words(d));
    for (int i = threadIdx.x; i < (int)S::plain_words(d); i += kPairWaves * 64) bias_lds[i] = bias_src[i];
  }
  if (SAMPLE) {
    for (int i = threadIdx.x; i < d; i += kPairWaves * 64) {
      zprm_lds[i] = q0_mean[i];
      zprm_lds[d + i] = sqrtf(expf(q0_log_var[i]));  // mnf_linear.py:60
    }
  }
  const float* zprm = SAMPLE ? zprm_lds : nullptr;
  const float wmax = __builtin_bit_cast(float, simage[S::split_words(d) + S::plain_words(d)]);
  const bool split_ok = __builtin_amdgcn_readfirstlane((int)(wmax <= kSplitWeightLimit)) != 0;
  __syncthreads();

  // byte offsets of the LDS regions as opaque registers (every access = one of these + an instruction immediate)
  uint32_t bias_off = NB * P::SLOT_WORDS * 4 + q * 16, buf_off = lane * 16 + side * (P::SIDE_WORDS * 4);
  uint32_t zprm_off = (NB * P::SLOT_WORDS + (uint32_t)S::plain_words(d)) * 4 + q * 16;
  uint32_t xch_off = (NB * P::SLOT_WORDS + (uint32_t)S::plain_words(d) + 2 * d) * 4 + pair * 4096 + lane * 16;
  asm volatile("" : "+v"(bias_off), "+v"(buf_off), "+v"(zprm_off), "+v"(xch_off));
  uint32_t buf_off_hi = buf_off + 2 * P::SLOT_WORDS * 4;
  asm volatile("" : "+v"(buf_off_hi));
  auto buf_base = [&](int c) -> uint32_t {  // this lane's 16 bytes in this side's part of the buffer of slot c
    const int u = c % NB;
    return u < 2 ? buf_off + u * (P::SLOT_WORDS * 4) : buf_off_hi + (u - 2) * (P::SLOT_WORDS * 4);
  };
  auto lds_f4 = [&](uint32_t byte_off) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds_dyn) + byte_off);
  };
  auto lds_h8 = [&](uint32_t byte_off) -> f16x8 {
    return *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(lds_dyn) + byte_off);
  };
  auto lds_put4 = [&](uint32_t byte_off, const f32x4& v) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds_dyn) + byte_off) = v;
  };

  // ---- operand ring: slot c holds [side 0: two K-steps / tiles][side 1: two K-steps / tiles]
  const uint32_t lane_off = lane * 16;
  uint32_t img_off = 0;  // always 0, but opaque and re-declared per pass (keeps the slot addresses out of registers)
  auto request = [&](auto cc) {
    constexpr int c = decltype(cc)::value % NC;
    uint32_t* dst = lds_dyn + (c % NB) * P::SLOT_WORDS;
    pair_static_for<2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int n_units = P::units(s, c);
      if constexpr (n_units > 0) {
        constexpr int64_t word0 = c < P::SLOTS1
                                      ? (int64_t)(P::first_ks(s) + 2 * c) * S::KS1_WORDS
                                      : S::part1_words(d) + (int64_t)(P::first_tile(s) + 2 * (c - P::SLOTS1)) * S::TILE2_WORDS;
        // scalar base + this lane's 16 bytes: the address costs no vector registers
        const char* src = reinterpret_cast<const char*>(simage + word0) + img_off + wave * 1024;
#pragma unroll
        for (int i = 0; i < n_units * (P::UNIT_PIECES / kPairWaves); ++i) {
          const int piece = i * kPairWaves + wave;  // wave-uniform
          const char* piece_src = src + i * (kPairWaves * 1024);
          asm volatile("" : "+s"(piece_src));  // (opaque: stays a scalar base, the lane's offset stays 32 bits wide)
          __builtin_amdgcn_global_load_lds(piece_src + lane_off, (lds_void_ptr_p)(dst + s * P::SIDE_WORDS + piece * 256), 16,
                                           0, 0);
        }
      }
    });
    asm volatile("" ::: "memory");  // (keeps the epilogue's stores behind the pieces: the counts assume it)
  };

  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};
  auto mask_bits = [&](uint32_t wq, auto gc) -> i32x4 {  // (asm: see mnf_rnvp_resident.hip)
    constexpr int o = 16 * (decltype(gc)::value & 1);
    i32x4 m;
    asm("v_bfe_i32 %0, %4, %5, 1\n\tv_bfe_i32 %1, %4, %6, 1\n\tv_bfe_i32 %2, %4, %7, 1\n\tv_bfe_i32 %3, %4, %8, 1"
        : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3])
        : "v"(wq), "n"(o), "n"(o + 1), "n"(o + 2), "n"(o + 3));
    return m;
  };
  auto and_bits = [](const f32x4& v, const i32x4& m) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, v) & m);
  };
  auto bfi = [](int32_t m, float a, float b) -> float {  // m ? a : b, bitwise
    float r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
  };

  const int n_sg = (int)((rows + 127) / 128);  // 128-row super-groups: two 64-row passes each

  // one 64-row pass of one side.  Returns false (workgroup-uniform) when the pass has to be redone in fp32.
  auto run_pass = [&](auto side_c, int64_t row, const float* zn, float* xr) -> bool {
    constexpr int SD = decltype(side_c)::value;
    constexpr int NKS = P::ksteps(SD), KS0 = P::first_ks(SD), NT = P::groups(SD), T0 = P::first_tile(SD);
    float ld_prev = 0.f;
    if (SD == 0 && accumulate && q == 0) ld_prev = log_det[row];
    // ---- GEMM 1 over this side's K-steps
    f32x4 ym[YT], yc[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) {
      ym[m] = SD == 0 ? lds_f4(bias_off + (G * 32 + m * 16) * 4) : zero4;  // (the bias once per pair)
      yc[m] = zero4;
    }
    float mx = 0.f;
    const uint32_t row_hash =
        mix32((uint32_t)row * 0x9e3779b1u + (uint32_t)((uint64_t)row >> 32) + (uint32_t)(seed >> 32));
    pair_static_for<P::SLOTS1>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      request(std::integral_constant<int, c + D>{});
      pair_static_for<2>([&](auto kc) {
        constexpr int lk = 2 * c + decltype(kc)::value;  // local K-step
        if constexpr (lk < NKS) {
          constexpr int ks = KS0 + lk, g0 = 2 * ks, g1 = 2 * ks + 1 < G ? 2 * ks + 1 : 2 * ks;
          constexpr int l0 = g0 - T0, l1 = g1 - T0;
          row_wait<P::row_wait(SD, l1, c)>();  // local tiles l0, l1 have landed
          f32x4 v0 = row_read<l0>(), v1 = row_read<l1>();
          if constexpr (SAMPLE) {  // the sample_z prologue, once per row: z = q0_mean + q0_std * eps
            v0 = v0 * lds_f4(zprm_off + (d + 16 * g0) * 4) + lds_f4(zprm_off + 16 * g0 * 4);
            row_write<l0>(v0);
            if constexpr (g1 != g0) {
              v1 = v1 * lds_f4(zprm_off + (d + 16 * g1) * 4) + lds_f4(zprm_off + 16 * g1 * 4);
              row_write<l1>(v1);
            }
          }
          const uint32_t mwk = mix32(row_hash ^ ((uint32_t)ks * 0x85ebca77u + (uint32_t)seed)) >> (4 * q);  // rnvp_mask_word
          agpr_put<kMaskAgpr + lk>(mwk);
          u32x2 h0, l0h, h1 = zero2, l1h = zero2;
          split_tile(and_bits(v0, mask_bits(mwk, std::integral_constant<int, g0>{})), h0, l0h, mx);
          if constexpr (g1 != g0) split_tile(and_bits(v1, mask_bits(mwk, std::integral_constant<int, g1>{})), h1, l1h, mx);
          asm volatile("" : "+v"(mx));  // (the running maximum is formed here, not from kept copies after the GEMM)
          const f16x8 bh = pair_operand(h0, h1), bl = pair_operand(l0h, l1h);
          const uint32_t a_off = buf_base(c) + decltype(kc)::value * (P::UNIT_WORDS * 4);  // operand o at + 1024 o
#pragma unroll
          for (int m = 0; m < YT; ++m)
            split_mac(lds_h8(a_off + 1024 * (2 * m)), lds_h8(a_off + 1024 * (2 * m + 1)), bh, bl, ym[m], yc[m]);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::dma_wait(SD, c)) : "memory");  // this wave's pieces of slot c + 1
      __syncthreads();
    });
    // ---- y: side 0 -> side 1 (adds its own) -> side 0; the range verdict of the whole 64-row pass on the way
    f32x4 y[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) y[m] = yc[m] * kSplitInvScale + ym[m];
    if (SD == 0) {
#pragma unroll
      for (int m = 0; m < YT; ++m) lds_put4(xch_off + m * 1024, y[m]);
    }
    __syncthreads();
    u32x2 yh[YT], yl[YT];
    if (SD == 1) {
#pragma unroll
      for (int m = 0; m < YT; ++m) {
        y[m] += lds_f4(xch_off + m * 1024);
        lds_put4(xch_off + m * 1024, y[m]);
      }
#pragma unroll
      for (int m = 0; m < YT; ++m) split_tile(y[m], yh[m], yl[m], mx);
    }
    if (__syncthreads_or(!(mx <= kSplitLimit) && kPairAbl == 0 ? 1 : 0)) return false;  // nothing has been stored yet
    if (SD == 0) {
      float unused = 0.f;
#pragma unroll
      for (int m = 0; m < YT; ++m) split_tile(lds_f4(xch_off + m * 1024), yh[m], yl[m], unused);
    }
    f16x8 ybh[NKS2], ybl[NKS2];
#pragma unroll
    for (int ks = 0; ks < NKS2; ++ks) {
      ybh[ks] = pair_operand(yh[2 * ks], 2 * ks + 1 < YT ? yh[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
      ybl[ks] = pair_operand(yl[2 * ks], 2 * ks + 1 < YT ? yl[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
    }
    // ---- GEMM 2 + gate over this side's tiles; the tile's registers are re-loaded with the next pass's rows
    float ld2 = 0.f;  // sum of log2(1 + e^-s) over the gated elements
    pair_static_for<P::SLOTS2>([&](auto uc) {
      constexpr int c = P::SLOTS1 + decltype(uc)::value;
      request(std::integral_constant<int, c + D>{});
      pair_static_for<2>([&](auto tc_) {
        constexpr int lt = 2 * decltype(uc)::value + decltype(tc_)::value;  // local tile
        if constexpr (lt < NT) {
          constexpr int m = T0 + lt;
          const uint32_t t_off = buf_base(c) + decltype(tc_)::value * (P::UNIT_WORDS * 4);
          f32x4 tm = lds_f4(bias_off + m * 128), sm = lds_f4(bias_off + m * 128 + 64), tc = zero4, sc = zero4;
#pragma unroll
          for (int ks = 0; ks < NKS2; ++ks) {
            split_mac(lds_h8(t_off + 1024 * (2 * ks)), lds_h8(t_off + 1024 * (2 * ks + 1)), ybh[ks], ybl[ks], tm, tc);
            split_mac(lds_h8(t_off + 1024 * (2 * (NKS2 + ks))), lds_h8(t_off + 1024 * (2 * (NKS2 + ks) + 1)), ybh[ks],
                      ybl[ks], sm, sc);
          }
          const f32x4 t4 = tc * kSplitInvScale + tm;
          const f32x4 s4 = sc * kSplitInvScale + sm;
          // binary mask: x = (1 - gate) t + (m ? z : gate z);  log_det -= (1 - m) ln(1 + e^-s)   (rnvp.py:36-37)
          const i32x4 mb = mask_bits(agpr_get<kMaskAgpr + (m >> 1) - KS0>(), std::integral_constant<int, m>{});
          const f32x4 zv = row_read<lt>();
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float zz = zv[r];
            const float den = 1.f + __builtin_amdgcn_exp2f(s4[r] * -1.44269504088896341f);
            const float gate = __builtin_amdgcn_rcpf(den);
            o[r] = __builtin_fmaf(-gate, t4[r], t4[r]) + bfi(mb[r], zz, zz * gate);
            ld2 += __builtin_amdgcn_logf(bfi(mb[r], 1.f, den));  // (mask before the log: no asm reads a transcendental)
          }
          if constexpr ((kPairNt & 1) != 0)
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(xr + ((kPairAbl & 1) ? 0 : 16 * m)));
          else
            *reinterpret_cast<f32x4*>(xr + ((kPairAbl & 1) ? 0 : 16 * m)) = o;
          row_load_at<lt, ((kPairAbl & 2) ? 0 : 64 * m), (kPairNt & 2) != 0>(zn);  // the same dims of the pair's next 16 rows
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c == NC - 1) {  // side 1's row sums to side 0, under the last slot's barrier
        const float part = sum_over_q(-0.693147180559945309f * ld2);
        if (SD == 1 && q == 0) *reinterpret_cast<float*>(reinterpret_cast<char*>(lds_dyn) + xch_off) = part;
        ld2 = part;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::dma_wait(SD, c)) : "memory");
      __syncthreads();
    });
    if (SD == 0 && q == 0)
      log_det[row] = ld_prev + ld2 + *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds_dyn) + xch_off);
    return true;
  };

  // every row of this side's part of a 16-row tile, asynchronously (the caller waits)
  auto load_rows = [&](const float* zq) {
    if (side == 0)
      pair_static_for<P::groups(0)>([&](auto gc) { row_load_at<decltype(gc)::value, 64 * (P::first_tile(0) + decltype(gc)::value)>(zq); });
    else
      pair_static_for<P::groups(1)>([&](auto gc) { row_load_at<decltype(gc)::value, 64 * (P::first_tile(1) + decltype(gc)::value)>(zq); });
  };
  // (re)start the operand ring: slots 0 .. D - 1 requested, slot 0 landed and published
  auto start_ring = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // nobody reads the buffers any more
    pair_static_for<D>([&](auto cc) { request(cc); });
  };

  // passes of this workgroup: super-group sg = blockIdx.x + it * gridDim.x, halves 0 and 1
  const int n_iter = n_sg > (int)blockIdx.x ? (n_sg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  auto pass_hot = [&](int pass) -> bool {  // workgroup-uniform: a whole 128-row super-group of this workgroup
    const int sg = (int)blockIdx.x + (pass >> 1) * (int)gridDim.x;
    return pass < 2 * n_iter && (int64_t)(sg + 1) * 128 <= rows;
  };
  auto pass_row = [&](int pass) -> int64_t {  // this lane's row of that pass
    const int sg = (int)blockIdx.x + (pass >> 1) * (int)gridDim.x;
    return (int64_t)sg * 128 + (pass & 1) * 64 + pair * 16 + j;
  };
  if (split_ok && n_iter > 0 && pass_hot(0)) {
    start_ring();
    load_rows(z + pass_row(0) * d + 4 * q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counted waits assume the steady state: more in flight)
    __syncthreads();
  }
  for (int pass = 0; pass < 2 * n_iter; ++pass) {
    const int it = pass >> 1;
    const int64_t row = pass_row(pass);
    const bool next_hot = pass_hot(pass + 1);
    // the next pass's rows (in-place prefetch); past the last hot pass every lane re-reads 16 bytes of row 0
    const float* zn = z + (next_hot ? pass_row(pass + 1) : 0) * d + 4 * q;
    const bool flagged =  // (its first half failed; readfirstlane: the value is workgroup-uniform, say so)
        ((__builtin_amdgcn_readfirstlane((int)cold_flags[it >> 5]) >> (it & 31)) & 1) != 0;
    if (!split_ok || !pass_hot(pass) || flagged) {
      // fp32 body after the loop: weights out of range, a short last super-group, or a failed first half.  The rows
      // in the registers belong to this pass: fetch the next one's; the operand ring has not moved.
      __syncthreads();  // (the flag read above against the write below)
      if (threadIdx.x == 0) cold_flags[it >> 5] |= 1u << (it & 31);
      if (split_ok && next_hot) {
        load_rows(zn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      continue;
    }
    asm volatile("" : "+s"(img_off));
    float* xr = x + row * d + 4 * q;
    const bool ok_lane = side == 0 ? run_pass(std::integral_constant<int, 0>{}, row, zn, xr)
                                   : run_pass(std::integral_constant<int, 1>{}, row, zn, xr);
    const bool ok = __builtin_amdgcn_readfirstlane((int)ok_lane) != 0;  // (workgroup-uniform by construction)
    if (!ok) {  // stopped after GEMM 1, nothing stored: flag it, restart the ring, fetch the next pass's rows
      if (threadIdx.x == 0) cold_flags[it >> 5] |= 1u << (it & 31);
      start_ring();
      if (next_hot) load_rows(zn);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // the flagged super-groups, on the fp32 MFMA body (128 rows each; it reads its rows from memory itself)
  __syncthreads();
  for (int it = 0; it < n_iter; ++it)
    if ((__builtin_amdgcn_readfirstlane((int)cold_flags[it >> 5]) >> (it & 31)) & 1)
      rnvp_pair_cold<HN, G, SAMPLE>(reinterpret_cast<float*>(lds_dyn), (int)blockIdx.x + it * (int)gridDim.x, z, x, log_det,
                                    image, rows, accumulate, seed, zprm);
}

// ---------------------------------------------------------------- host
template <int HN, int G, bool SAMPLE>
static int launch_pair(const float* z, float* x, float* log_det, int accumulate, const uint32_t* simage,
                       const float* image, int64_t rows, uint64_t seed, const float* q0_mean, const float* q0_log_var,
                       hipStream_t stream) {
  constexpr size_t lds_bytes = PairPlan<HN, G>::lds_bytes();
  static_assert(lds_bytes <= 160 * 1024, "operand ring + biases + exchange must fit the CU's LDS");
  static DeviceMemo memo;
  const int cus = memo.get([](int dev) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(rnvp_pair_kernel<HN, G, SAMPLE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess
               ? device_cus(dev)
               : -1;
  });
  if (cus <= 0) return MNF_ERR_UNSUPPORTED;
  const int64_t n_sg = (rows + 127) / 128;
  const int64_t blocks = n_sg < cus ? n_sg : cus;  // one persistent workgroup per CU
  if ((n_sg + blocks - 1) / (blocks > 0 ? blocks : 1) > 32 * kPairColdWords) return MNF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((rnvp_pair_kernel<HN, G, SAMPLE>), dim3((unsigned)blocks), dim3(kPairWaves * 64), lds_bytes, stream,
                     z, x, log_det, simage, image, rows, accumulate, seed, q0_mean, q0_log_var);
  return check_launch();
}

#define MNF_RNVP_PAIR_SHAPES(X) X(50, 50)

// MNF_ERR_UNSUPPORTED: no pair kernel for this shape (or switched off) -- the caller goes on to the next kernel
int rnvp_pair_launch(const float* z, float* x, float* log_det, int accumulate, const void* split_image,
                     const float* image, int64_t rows, int dim, int hn_pad, uint64_t seed, const float* q0_mean,
                     const float* q0_log_var, int vec, hipStream_t stream) {
  // Opt-in (MNF_RNVP_PAIR=1; read per call: tests and A/B runs flip it).  Two waves per SIMD do not pay here: both
  // register-resident kernels sit on the same limit -- the operand ring's L2 -> LDS stream and the rows' HBM
  // misses share the CU's vector L1 and slow each other down (tools/hbm_pattern.hip, profiles/r2/c5_l1_path.txt)
  // -- and this one measures 586 us against the one-wave kernel's 578 (256,000 rows), so that one stays the default.
  const char* env = getenv("MNF_RNVP_PAIR");
  if (!(env && env[0] == '1') || !split_image || !image || !log_det) return MNF_ERR_UNSUPPORTED;
  if ((dim & 15) || !vec) return MNF_ERR_UNSUPPORTED;
  const uint32_t* simage = static_cast<const uint32_t*>(split_image);
#define X(HN, GG)                                                                                                 \
  if (hn_pad == HN && dim == 16 * GG)                                                                             \
    return q0_mean ? launch_pair<HN, GG, true>(z, x, log_det, accumulate, simage, image, rows, seed, q0_mean,     \
                                               q0_log_var, stream)                                                \
                   : launch_pair<HN, GG, false>(z, x, log_det, accumulate, simage, image, rows, seed, q0_mean,    \
                                                q0_log_var, stream);
  MNF_RNVP_PAIR_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf
