// Masked / gated RNVP coupling (flows/rnvp.py:25-39) with the rows RESIDENT IN REGISTERS, gfx950.
//
//   y      = Wn (m * z) + bn                 GEMM 1, K = d          (needs every z of the row)
//   shift  = Wt y + bt ; scale = Ws y + bs   GEMM 2, K = h
//   x      = (1 - gate) shift + (m ? z : gate z) ;  log_det = sum_j (1 - m_j) log gate_j     (needs z again)
//
// z is needed twice with all of GEMM 1 in between.  The streaming kernels (mnf_rnvp_mfma.hip) read it from
// memory twice: 12 d + 8 bytes per row against 8 d + 8 algorithmic (PMC: 1.70 x).  Here a wave keeps its 16
// rows -- d / 4 registers per lane, 200 at d = 800 -- in the register file from the one load to the gate
// epilogue, so every z is read ONCE.  That takes the whole 512-entry file: one wave per SIMD, one 4-wave
// workgroup per CU (`__launch_bounds__(256, 1)`).  The loops over the row are fully unrolled (G = d / 16 is a
// template parameter), so every register index is static.
//
// Latency is hidden inside the wave instead of by other waves: as soon as the epilogue has consumed the 16 dims
// of group g (x stored), the registers of that group are re-loaded with the same dims of the wave's NEXT 16 rows;
// those loads are in flight under the rest of the epilogue and land long before GEMM 1 of the next row tile reads
// them.  HBM therefore sees one load and one store per 64 B of row, interleaved, all through the epilogue.
//
// Both GEMMs run in split arithmetic (mnf_split.h) on v_mfma_f32_16x16x32_f16 from the same operand image as the
// streaming split kernel; the image (600 KB at d = 800, h = 50) is streamed L2 -> registers -> LDS in 32 KB chunks
// shared by the four waves (64 rows per pass), double buffered, one barrier per chunk.  A 64-row group whose
// operands leave the f16 range (verdict after GEMM 1, before anything is stored) is recomputed by the fp32 body.
// The mask is the in-kernel counter-based one (`seed`): its 32-bit words are hashed once per row tile and kept
// (d / 32 registers) for the epilogue.  Explicit float masks stay on the streaming kernels.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <utility>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_split.h"
#include "mnf_agpr.h"

namespace mnf {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// experiment switches (tools/rnvp_variants.sh builds A/B libraries with them; the defaults are the product)
#ifndef MNF_RES_KC
#define MNF_RES_KC 4
#endif
#ifndef MNF_RES_MC
#define MNF_RES_MC 4
#endif
#ifndef MNF_RES_ABL
#define MNF_RES_ABL 0  // timing only, results wrong: 1 no x stores, 2 no row re-loads, 3 no MFMAs, 4 no gate math,
                       // 5 no operand requests, 6 = 1 + 2, 7 = 1 + 2 + 5, 8 stores all to the row's first 64 B,
                       // 9 row re-loads all from the row's first 64 B, 10 = 8 + 9, 11 no chunk waits / barriers
#endif
#ifndef MNF_RES_SCHED
#define MNF_RES_SCHED 1  // sched_barrier after every n-th K-step / tile (0: none)
#endif
constexpr int kResWaves = 4;        // one wave per SIMD
constexpr int kResKC = MNF_RES_KC;  // GEMM-1 K-steps (32 dims each) per operand chunk
constexpr int kResMC = MNF_RES_MC;  // GEMM-2 output tiles (16 dims each) per operand chunk
constexpr int kResAbl = MNF_RES_ABL;
constexpr int kResColdWords = 64;  // one bit per row group of a workgroup that has to be redone in fp32 (2,048 groups)

constexpr int kResBufs = 4;     // LDS operand buffers: a chunk is requested kResBufs - 1 chunks before it is used
template <int HN>
struct ResShape {
  using S = RnvpSplitShape<HN>;
  static constexpr int CHUNK_WORDS =
      kResKC * S::KS1_WORDS > kResMC * S::TILE2_WORDS ? kResKC * S::KS1_WORDS : kResMC * S::TILE2_WORDS;
  static constexpr int F32_WORDS = 2 * RnvpShape<HN, kResWaves>::CHUNK_FLOATS;  // the fp32 body's double buffer
  static_assert(kResBufs * CHUNK_WORDS >= F32_WORDS, "the fp32 body's window fits the operand buffers");
  // LDS: the operand buffers, the fp32 biases ((bt, bs) per tile, then bn), mean / std of the sample_z prologue,
  // the flags of the groups to redo in fp32
  static constexpr size_t lds_bytes(int d) {
    return sizeof(uint32_t) * (kResBufs * (size_t)CHUNK_WORDS + S::plain_words(d) + 2 * (size_t)d + kResColdWords);
  }
};

template <int HN, bool RAG>
__device__ __attribute__((noinline)) void rnvp_resident_f32_cold(float* lds, int grp, const float* z, float* x,
                                                                float* log_det, const float* image, int64_t rows, int d,
                                                                int accumulate, uint64_t seed, const float* zprm, int dm,
                                                                bool vec) {
  using F = RnvpShape<HN, kResWaves>;
  rnvp_group_f32<HN, true, RAG, kResWaves>(*reinterpret_cast<float(*)[2][F::CHUNK_FLOATS]>(lds), grp, z, nullptr, x,
                                           log_det, image, rows, d, accumulate, seed, zprm, dm, vec);
}

// ---- compile-time loops: every register index and every instruction offset below is a constant
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- the wave's stream of vector-memory operations per row group, known at compile time (the waits are counted):
//   GEMM-1 slot k :  [k % KC == 0: the LDS-DMA pieces of chunk c + D]
//   GEMM-2 slot s :  [s % MC == 0: the pieces of chunk c + D]  [s >= 2: store x of tile s - 2, load of its next row]
// (the log_det update at the end of a group is not counted: the counts are lower bounds of what is in flight)
template <int HN, int G>
struct ResPlan {
  using S = RnvpSplitShape<HN>;
  static constexpr int KC = kResKC, MC = kResMC, D = kResBufs - 1;
  static constexpr int NKS1 = (G + 1) / 2;
  static constexpr int NC1 = (NKS1 + KC - 1) / KC, NC2 = (G + MC - 1) / MC, NC = NC1 + NC2;
  // 1 KB LDS-DMA pieces per wave that fetch chunk c (the last chunk of either GEMM may be short)
  static constexpr int chunk_pieces(int c) {
    c %= NC;
    const int words = c < NC1 ? ((NKS1 - c * KC) < KC ? (NKS1 - c * KC) : KC) * S::KS1_WORDS
                              : ((G - (c - NC1) * MC) < MC ? (G - (c - NC1) * MC) : MC) * S::TILE2_WORDS;
    return words / 256 / kResWaves;
  }
  // operations issued from the start of a group up to: the end of the request block of GEMM-1 slot k ...
  static constexpr int ops_g1(int k) {
    int n = 0;
    for (int kk = 0; kk <= k && kk < NKS1; kk += KC) n += chunk_pieces(kk / KC + D);
    return n;
  }
  // ... the end of the request block of GEMM-2 slot s (before its epilogue's store) ...
  static constexpr int ops_g2(int s) {
    int n = ops_g1(NKS1);
    for (int ss = 0; ss <= s && ss < G; ss += MC) n += chunk_pieces(NC1 + ss / MC + D);
    return n + 2 * (s - 2 > 0 ? s - 2 : 0);  // epilogues of slots 2 .. s - 1
  }
  static constexpr int ops_group() { return ops_g2(G + 1) + 2; }
  // ... and after the row load of tile m (slot m + 2)
  static constexpr int ops_after_rowload(int m) { return ops_g2(m + 2) + 2; }
  // operations issued by the end of chunk e (e < NC1: GEMM 1; else GEMM 2, its last slot's epilogue included)
  static constexpr int ops_chunk_end(int e) {
    if (e < NC1) return ops_g1(((e + 1) * KC < NKS1 ? (e + 1) * KC : NKS1) - 1);
    const int s = ((e - NC1 + 1) * MC < G ? (e - NC1 + 1) * MC : G) - 1;
    return ops_g2(s) + (s >= 2 ? 2 : 0);
  }
  // operations issued by the end of the request block at the top of chunk c
  static constexpr int ops_chunk_top(int c) { return c < NC1 ? ops_g1(c * KC) : ops_g2((c - NC1) * MC); }
  // in flight behind the pieces of chunk e + 1 at the end of chunk e (they were requested at the top of chunk
  // e + 1 - D, possibly in the previous group)
  static constexpr int dma_wait_count(int e) {
    const int top = e + 1 - D;
    const int n = top >= 0 ? ops_chunk_end(e) - ops_chunk_top(top) : ops_chunk_end(e) + ops_group() - ops_chunk_top(top + NC);
    return n < 63 ? n : 63;
  }
  // in flight behind the row load of group g (issued by the previous row group) when GEMM-1 slot k wants it
  static constexpr int row_wait_count(int g, int k) {
    const int n = ops_group() - ops_after_rowload(g) + ops_g1(k);
    return n < 63 ? n : 63;
  }
};

template <int HN, int G, bool SAMPLE>
__global__ void __launch_bounds__(kResWaves * 64, 1)
rnvp_resident_kernel(const float* __restrict__ z, float* __restrict__ x, float* __restrict__ log_det,
                     const uint32_t* __restrict__ simage, const float* __restrict__ image, int64_t rows, int accumulate,
                     uint64_t seed, const float* __restrict__ q0_mean, const float* __restrict__ q0_log_var,
                     float* __restrict__ y_out) {
  using S = RnvpSplitShape<HN>;
  using R = ResShape<HN>;
  constexpr int d = 16 * G;
  constexpr int YT = S::YT, NKS2 = S::NKS2, KC = kResKC, MC = kResMC;
  constexpr int NKS1 = (G + 1) / 2;
  constexpr int NC1 = (NKS1 + KC - 1) / KC, NC2 = (G + MC - 1) / MC, NC = NC1 + NC2;
  static_assert(NC % 2 == 0, "the operand ring keeps its buffer / register-set parity from one row group to the next");
  static_assert(64 * (G - 1) < 4096, "row offsets are instruction immediates");
  static_assert(4 * G <= kResAgprs && (G + 1) / 2 <= 32, "resident rows and mask words fit the reserved registers");
  reserve_agprs();
  constexpr int GROUP_ROWS = 16 * kResWaves;
  constexpr int NB = kResBufs, D = NB - 1;
  static_assert(NC % NB == 0, "the operand ring keeps its buffer assignment from one row group to the next");
  constexpr int OPS1 = 2 * YT, OPS2 = 4 * NKS2;  // 1 KB A operands per GEMM-1 K-step / per GEMM-2 tile
  using P = ResPlan<HN, G>;
  static_assert(S::KS1_WORDS % (256 * kResWaves) == 0 && S::TILE2_WORDS % (256 * kResWaves) == 0,
                "every wave copies the same number of 1 KB pieces of a chunk");

  // LDS: [NB operand buffers][(bt | bs) per tile, then bn][mean | std of the sample_z prologue][cold-group flags]
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  float* const bias_lds = reinterpret_cast<float*>(lds_dyn + NB * R::CHUNK_WORDS);
  float* const zprm_lds = bias_lds + S::plain_words(d);
  uint32_t* const cold_flags = reinterpret_cast<uint32_t*>(zprm_lds + 2 * d);  // kResColdWords words

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;

  // once per workgroup: biases (and the prologue's mean / std) into LDS
  if (threadIdx.x < kResColdWords) cold_flags[threadIdx.x] = 0u;
  {
    const float* bias_src = reinterpret_cast<const float*>(simage + S::split_words(d));
    for (int i = threadIdx.x; i < (int)S::plain_words(d); i += kResWaves * 64) bias_lds[i] = bias_src[i];
  }
  // SAMPLE: `z` holds eps; the layer runs on q0_mean + q0_std * eps (MNFLinear.sample_z, mnf_linear.py:58-64)
  if (SAMPLE) {
    for (int i = threadIdx.x; i < d; i += kResWaves * 64) {
      zprm_lds[i] = q0_mean[i];
      zprm_lds[d + i] = sqrtf(expf(q0_log_var[i]));  // mnf_linear.py:60
    }
  }
  const float* zprm = SAMPLE ? zprm_lds : nullptr;
  const float wmax = __builtin_bit_cast(float, simage[S::split_words(d) + S::plain_words(d)]);
  const bool split_ok = wmax <= kSplitWeightLimit;  // else every group takes the fp32 body
  __syncthreads();
  // byte offsets of the LDS regions as opaque registers: every LDS access below is then one of these + an
  // instruction immediate (left constant, hipcc materialises one address register per access for the regions above
  // 64 KB -- the limit of the immediate -- and hoists all of them out of the group loop)
  uint32_t bias_off = NB * R::CHUNK_WORDS * 4 + q * 16, buf_off = lane * 16;
  uint32_t zprm_off = (NB * R::CHUNK_WORDS + (uint32_t)S::plain_words(d)) * 4 + q * 16;
  asm volatile("" : "+v"(bias_off), "+v"(buf_off), "+v"(zprm_off));
  // (a second base for the upper buffers: the immediate of an LDS instruction ends at 64 KB)
  uint32_t buf_off_hi = buf_off + 2 * R::CHUNK_WORDS * 4;
  asm volatile("" : "+v"(buf_off_hi));
  auto buf_base = [&](int c) -> uint32_t {  // byte offset of this lane's 16 bytes in the buffer of chunk c
    const int u = c % NB;
    return u < 2 ? buf_off + u * (R::CHUNK_WORDS * 4) : buf_off_hi + (u - 2) * (R::CHUNK_WORDS * 4);
  };
  auto lds_f4 = [&](uint32_t byte_off) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds_dyn) + byte_off);
  };
  auto lds_h8 = [&](uint32_t byte_off) -> f16x8 {
    return *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(lds_dyn) + byte_off);
  };

  // ---- operand ring.  Chunk c (mod NC) of the image: GEMM-1 K-steps for c < NC1, GEMM-2 tiles after.  The chunks
  // do not depend on the rows, so the ring runs on across row groups.  At the top of chunk c the wave requests its
  // pieces of chunk c + D by LDS-DMA (global_load_lds_dwordx4: L2 -> LDS, no staging registers) into buffer
  // (c + D) % NB -- last read during chunk c - 1, which every wave has left --; at the end of chunk c it waits until
  // its pieces of chunk c + 1 have landed and meets the others at the chunk's one barrier.  Vector-memory operations
  // complete in issue order and the wait is a count of operations still in flight, so it is exact only if every
  // operation of the loop is accounted for (dma_wait_count): a wait that is stricter than necessary makes the wave
  // stand until row loads and stores issued long after the pieces have completed too -- a couple of microseconds each
  // under load -- and with it the three other waves at the barrier.
  uint32_t img_off = 0;  // always 0, but opaque and re-declared at the top of every row group: see there
  auto request = [&](auto cc) {
    constexpr int c = decltype(cc)::value % NC;
    constexpr int64_t word0 = c < NC1 ? (int64_t)c * KC * S::KS1_WORDS
                                      : S::part1_words(d) + (int64_t)(c - NC1) * MC * S::TILE2_WORDS;
    const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(simage + word0) + img_off) + lane;
    uint32_t* dst = lds_dyn + (c % NB) * R::CHUNK_WORDS;
    if (kResAbl != 5 && kResAbl != 7) {
#pragma unroll
      for (int i = 0; i < P::chunk_pieces(c); ++i) {
        const int piece = i * kResWaves + wave;  // wave-uniform
        __builtin_amdgcn_global_load_lds(src + piece * 64, (lds_void_ptr)(dst + piece * 256), 16, 0, 0);
      }
    }
    asm volatile("" ::: "memory");  // (keeps the stores of the epilogue behind the pieces: the counts assume it)
  };
  auto landed = [&](auto ec) {  // end of chunk e: this wave's pieces of chunk e + 1 are in LDS
    constexpr int e = decltype(ec)::value;
    if (kResAbl != 11 && kResAbl != 13) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::dma_wait_count(e)) : "memory");
  };
  // (inline asm, not __syncthreads(): that one's release fence is `s_waitcnt vmcnt(0)` -- every chunk would wait for
  // all of the wave's outstanding row loads and stores, and the counted waits of landed() would count for nothing.
  // What the barrier orders is LDS: this wave's operand reads of the chunk, done, before anyone's DMA refills it.)
  auto chunk_barrier = [] {
    if (kResAbl != 11 && kResAbl != 13) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  auto mfma3 = [&](const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x4& mn, f32x4& cr) {
    if (kResAbl == 3) {
      mn += __builtin_bit_cast(f32x4, ah) * __builtin_bit_cast(f32x4, bh);
      cr += __builtin_bit_cast(f32x4, al) * __builtin_bit_cast(f32x4, bl);
    } else {
      mnf::split_mac(ah, al, bh, bl, mn, cr);
    }
  };
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};
  // the mask of element r of group g as all-ones / all-zeros (wq = the row's mask word g >> 1, pre-shifted by 4 q).
  // asm on purpose: left to itself hipcc turns  sext(bit) & v  into compare + select through an SGPR pair per
  // element, and a fully unrolled row then spills SGPRs through v_writelane.
  auto mask_bits = [&](uint32_t wq, auto gc) -> i32x4 {
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
  auto fence = [](int slot) {  // MNF_RES_SCHED = n: the scheduler may mix n consecutive slots (0: everything)
    if (MNF_RES_SCHED && (slot + 1) % MNF_RES_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
  };
  // resident rows: a[4 g : 4 g + 3] of lane (j, q) holds dims 16 g + 4 q .. + 3 of row j of the wave's 16 rows
  // their mask words (32 dims each, pre-shifted by 4 q): a[208 + k]
  bool primed = false;  // rows resident and the operand ring running (false at the start and after a flagged group)
  const int n_groups = (int)((rows + GROUP_ROWS - 1) / GROUP_ROWS);

  for (int grp = blockIdx.x, it = 0; grp < n_groups; grp += gridDim.x, ++it) {
    const int64_t row = (int64_t)grp * GROUP_ROWS + wave * 16 + j;  // (every row of a hot-path group exists)
    const int64_t rowc = row < rows ? row : rows - 1;
    float* xr = x + rowc * d + 4 * q;
    // Groups for the fp32 body -- weights outside the f16 range, the one group that may be short of 64 rows, and
    // (below) operands outside the f16 range -- are only FLAGGED here and redone after the loop: a call inside this
    // loop would have the 200 resident registers live across it.
    if (!split_ok || (int64_t)(grp + 1) * GROUP_ROWS > rows) {
      if (threadIdx.x == 0) cold_flags[it >> 5] |= 1u << (it & 31);
      primed = false;
      continue;
    }
    // opaque per group: the operand addresses of a pass are formed where they are used (scalar base + 32-bit
    // thread offset) instead of being hoisted out of the group loop into 300 registers
    asm volatile("" : "+s"(img_off));
    // rows of the NEXT group, loaded in place while this group's epilogue frees the registers; past the last group
    // every lane re-reads 16 bytes of row 0 instead (a cache hit, no branch in the unrolled epilogue)
    const bool has_next = grp + (int)gridDim.x < n_groups;
    const float* zn;
    {
      const int64_t rn = (int64_t)(grp + (int)gridDim.x) * GROUP_ROWS + wave * 16 + j;
      zn = z + (has_next ? (rn < rows ? rn : rows - 1) : 0) * d + 4 * q;
    }
    if (!primed) {  // the only exposed loads: operand chunks 0 .. D - 1, then the group's rows
      __syncthreads();  // (the previous group may still be reading the buffers)
      static_for<D>([&](auto cc) { request(cc); });
      const float* zq = z + rowc * d + 4 * q;
      static_for<G>([&](auto gc) { row_load<decltype(gc)::value>(zq); });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // everything has landed: the counted waits below assume
      __syncthreads();                                   // the steady state, where MORE is in flight behind a load
      primed = true;
    }
    // log_det of the rows, read early: its wait must not be the one that drains the loop's loads at the end
    // (Round 4: as a C++ load under `if (accumulate && q == 0)` its use at the group's end made hipcc wait vmcnt(0) there
    //  -- behind a load under a branch its wait counts are not exact --, i.e. for the NEXT group's rows and this
    //  group's stores: the launches that add to log_det took 590 us against 550.  Now an asm load into a240, issued
    //  by every lane without a branch, read back at the end behind a counted wait: vector-memory operations complete
    //  in order and ops_group() of them are issued behind it -- with more than 63, the counter's range, no wait at all.)
    {
      const float* ldp = accumulate ? log_det + row : z;
      asm volatile("global_load_dword a240, %0, off" ::"v"(ldp) : "memory", "a240");
    }

    // ---- GEMM 1: y^T += Wn[:, 32 dims] (m z)^T, two 16-dim groups per K-step.  Software pipeline, one K-step per
    // slot (one wave per SIMD: nothing else hides a latency): slot k issues the MFMAs of K-step k - 1 (operands read
    // one slot earlier), then the LDS reads of K-step k's operands into the same registers, then -- under those MFMAs
    // and reads -- turns the rows of K-step k into B operands (mask, split).
    f32x4 ym[YT], yc[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) {
      ym[m] = lds_f4(bias_off + (G * 32 + m * 16) * 4);
      yc[m] = zero4;
    }
    float mx = 0.f;
    const uint32_t row_hash =
        mix32((uint32_t)rowc * 0x9e3779b1u + (uint32_t)((uint64_t)rowc >> 32) + (uint32_t)(seed >> 32));
    f16x8 a1[OPS1], b1h, b1l;
    static_for<NKS1 + 1>([&](auto kc_) {
      constexpr int k = decltype(kc_)::value;
      if constexpr (k >= 1) {  // M(k - 1)
#pragma unroll
        for (int m = 0; m < YT; ++m) mfma3(a1[2 * m], a1[2 * m + 1], b1h, b1l, ym[m], yc[m]);
      }
      if constexpr (k < NKS1) {  // R(k): into the registers the MFMAs above have just been issued from
        constexpr int c = k / KC, kk = k % KC;
        if constexpr (kk == 0) request(std::integral_constant<int, c + D>{});
        const uint32_t a_off = buf_base(c) + kk * (S::KS1_WORDS * 4);  // operand o at + 1024 o
#pragma unroll
        for (int o = 0; o < OPS1; ++o) a1[o] = lds_h8(a_off + 1024 * o);
      }
      if constexpr (k < NKS1) {  // P(k)
        constexpr int c = k / KC;
        constexpr int g0 = 2 * k, g1 = 2 * k + 1 < G ? 2 * k + 1 : 2 * k;
        // rows: groups g0, g1 have landed once at most this many vector-memory operations are in flight (everything
        // the previous group's epilogue issued after the load of group g1, and this group's requests so far)
        if (kResAbl != 12 && kResAbl != 13) row_wait<P::row_wait_count(g1, k)>();
        f32x4 v0 = row_read<g0>(), v1 = row_read<g1>();
        if constexpr (SAMPLE) {  // the sample_z prologue, once per row: z = q0_mean + q0_std * eps, kept for the epilogue
          v0 = v0 * lds_f4(zprm_off + (d + 16 * g0) * 4) + lds_f4(zprm_off + 16 * g0 * 4);
          row_write<g0>(v0);
          if constexpr (g1 != g0) {
            v1 = v1 * lds_f4(zprm_off + (d + 16 * g1) * 4) + lds_f4(zprm_off + 16 * g1 * 4);
            row_write<g1>(v1);
          }
        }
        // mask word of dims 32 k .. 32 k + 31 (rnvp_mask_word), kept for the epilogue
        const uint32_t mwk = mix32(row_hash ^ ((uint32_t)k * 0x85ebca77u + (uint32_t)seed)) >> (4 * q);
        agpr_put<kResMaskAgpr + k>(mwk);
        u32x2 h0, l0, h1 = zero2, l1 = zero2;
        split_tile(and_bits(v0, mask_bits(mwk, std::integral_constant<int, g0>{})), h0, l0, mx);
        if constexpr (g1 != g0) split_tile(and_bits(v1, mask_bits(mwk, std::integral_constant<int, g1>{})), h1, l1, mx);
        b1h = pair_operand(h0, h1);
        b1l = pair_operand(l0, l1);
        fence(k);
        if constexpr (k % KC == KC - 1 || k == NKS1 - 1) {  // end of chunk c
          landed(std::integral_constant<int, c>{});
          if constexpr (k < NKS1 - 1) chunk_barrier();  // (the last chunk's barrier is the verdict below)
        }
      }
    });
    // y complete: operands of GEMM 2, and the range verdict for the whole 64-row group
    u32x2 yh[YT], yl[YT];
#pragma unroll
    for (int m = 0; m < YT; ++m) split_tile(yc[m] * kSplitInvScale + ym[m], yh[m], yl[m], mx);
    if (__syncthreads_or(!(mx <= kSplitLimit) && kResAbl == 0 ? 1 : 0)) {  // nothing has been stored yet
      if (threadIdx.x == 0) cold_flags[it >> 5] |= 1u << (it & 31);
      primed = false;
      continue;
    }
    // training: y is kept for the gradient pass (mnf_rnvp_bwd_mfma's launch A then skips its own GEMM-1 sweep over z).
    // (YT stores the wait counts below do not know about: the counts are lower bounds, the waits get stricter.)
    if (y_out) {
#pragma unroll
      for (int m = 0; m < YT; ++m)
        *reinterpret_cast<f32x4*>(y_out + row * (16 * YT) + 16 * m + 4 * q) = yc[m] * kSplitInvScale + ym[m];
    }
    f16x8 ybh[NKS2], ybl[NKS2];
#pragma unroll
    for (int ks = 0; ks < NKS2; ++ks) {
      ybh[ks] = pair_operand(yh[2 * ks], 2 * ks + 1 < YT ? yh[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
      ybl[ks] = pair_operand(yl[2 * ks], 2 * ks + 1 < YT ? yl[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
    }

    // ---- GEMM 2 + gate, 16 output dims per tile, three-stage software pipeline: slot m issues the MFMAs of tile m - 1,
    // then the LDS reads of tile m's operands (into the registers those MFMAs were issued from), then -- under both --
    // the gate epilogue of tile m - 2: x stored, and the tile's row registers re-loaded with the same dims of the NEXT
    // group's rows.
    float ld2 = 0.f;  // sum of log2(1 + e^-s) over the gated elements
    f16x8 a2[OPS2];
    f32x4 tm[2], tc[2], sm[2], sc[2], tb, sb;  // (tb, sb: the biases of the tile whose operands are in a2)
    static_for<G + 2>([&](auto mc_) {
      constexpr int s = decltype(mc_)::value;
      if constexpr (s >= 1 && s - 1 < G) {  // M(s - 1)
        constexpr int u = (s - 1) & 1;
        tm[u] = tb;
        sm[u] = sb;
        tc[u] = zero4;
        sc[u] = zero4;
#pragma unroll
        for (int ks = 0; ks < NKS2; ++ks) {
          mfma3(a2[2 * ks], a2[2 * ks + 1], ybh[ks], ybl[ks], tm[u], tc[u]);
          mfma3(a2[2 * (NKS2 + ks)], a2[2 * (NKS2 + ks) + 1], ybh[ks], ybl[ks], sm[u], sc[u]);
        }
      }
      if constexpr (s < G) {  // R(s): into the registers the MFMAs above have just been issued from
        constexpr int c = NC1 + s / MC, mi = s % MC;
        if constexpr (mi == 0) request(std::integral_constant<int, c + D>{});
        const uint32_t t_off = buf_base(c) + mi * (S::TILE2_WORDS * 4);  // operand o at + 1024 o
#pragma unroll
        for (int o = 0; o < OPS2; ++o) a2[o] = lds_h8(t_off + 1024 * o);
        tb = lds_f4(bias_off + s * 128);
        sb = lds_f4(bias_off + s * 128 + 64);
      }
      if constexpr (s >= 2) {  // E(s - 2)
        constexpr int m = s - 2, u = m & 1;
        const f32x4 t4 = tc[u] * kSplitInvScale + tm[u];
        const f32x4 s4 = sc[u] * kSplitInvScale + sm[u];
        // binary mask: x = (1 - gate) t + (m ? z : gate z);  log_det -= (1 - m) ln(1 + e^-s)   (rnvp.py:36-37; the
        // shift term reaches the kept elements too)
        const i32x4 mb = mask_bits(agpr_get<kResMaskAgpr + (m >> 1)>(), std::integral_constant<int, m>{});
        const f32x4 zv = row_read<m>();
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zz = zv[r];
          if (kResAbl == 4) {
            o[r] = zz + t4[r] + s4[r];
            continue;
          }
          const float den = 1.f + __builtin_amdgcn_exp2f(s4[r] * -1.44269504088896341f);
          const float gate = __builtin_amdgcn_rcpf(den);
          o[r] = __builtin_fmaf(-gate, t4[r], t4[r]) + bfi(mb[r], zz, zz * gate);
          // kept elements contribute log2(1) = 0.  (The mask goes on BEFORE the logarithm: an asm statement must not
          // read the result of a transcendental instruction directly -- gfx950 needs a wait state there that hipcc
          // only inserts for instructions it has selected itself; log2 of the masked value feeds a plain add.)
          ld2 += __builtin_amdgcn_logf(bfi(mb[r], 1.f, den));
        }
        if (kResAbl == 8 || kResAbl == 10) *reinterpret_cast<f32x4*>(xr) = o;
        else if ((kResAbl != 1 && kResAbl != 6 && kResAbl != 7) || o[0] == 1.2345e30f) *reinterpret_cast<f32x4*>(xr + 16 * m) = o;
        if (kResAbl == 9 || kResAbl == 10) row_load_at<m, 0>(zn);
        else if (kResAbl != 2 && kResAbl != 6 && kResAbl != 7) row_load<m>(zn);  // the same dims of the next group's row
      }
      fence(s);
      if constexpr (s < G && (s % MC == MC - 1 || s == G - 1)) {  // end of chunk c (its last tile's reads are issued)
        constexpr int c = NC1 + s / MC;
        landed(std::integral_constant<int, c>{});
        chunk_barrier();
      }
    });
    {  // (log_det is never null here: a conditional use would let the compiler sink the 4 G adds of ld2 down to it)
      const float ld = sum_over_q(-0.693147180559945309f * ld2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::ops_group() < 63 ? P::ops_group() : 63) : "memory");
      const float ld_prev = accumulate ? __builtin_bit_cast(float, agpr_get<240>()) : 0.f;
      if (q == 0) log_det[row] = ld_prev + ld;
    }
  }
  // the flagged groups, on the fp32 MFMA body (reads its rows from memory itself)
  __syncthreads();
  for (int grp = blockIdx.x, it = 0; grp < n_groups; grp += gridDim.x, ++it)
    if ((cold_flags[it >> 5] >> (it & 31)) & 1u) {
      rnvp_resident_f32_cold<HN, false>(reinterpret_cast<float*>(lds_dyn), grp, z, x, log_det, image, rows, d,
                                        accumulate, seed, zprm, d, true);
      if (y_out) {  // no y from the fp32 body: NaN rows make launch A flag the group for its own fp32 fix-up
        for (int i = threadIdx.x; i < GROUP_ROWS * 16 * YT; i += blockDim.x) {
          const int64_t r = (int64_t)grp * GROUP_ROWS + i / (16 * YT);
          if (r < rows) y_out[r * (16 * YT) + i % (16 * YT)] = __builtin_nanf("");
        }
      }
    }
}

// ---------------------------------------------------------------- host
template <int HN, int G, bool SAMPLE>
static int launch_resident(const float* z, float* x, float* log_det, int accumulate, const uint32_t* simage,
                           const float* image, int64_t rows, uint64_t seed, const float* q0_mean,
                           const float* q0_log_var, float* y_out, hipStream_t stream) {
  constexpr size_t lds_bytes = ResShape<HN>::lds_bytes(16 * G);
  static_assert(lds_bytes <= 160 * 1024, "operand window + biases must fit the CU's LDS");
  static DeviceMemo memo;  // per device: CU count once the dynamic-LDS attribute is set there, -1 if it cannot be
  const int cus = memo.get([](int dev) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(rnvp_resident_kernel<HN, G, SAMPLE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess
               ? device_cus(dev)
               : -1;
  });
  if (cus <= 0) return MNF_ERR_UNSUPPORTED;
  const int64_t n_groups = (rows + 16 * kResWaves - 1) / (16 * kResWaves);
  const int64_t blocks = n_groups < cus ? n_groups : cus;  // one persistent workgroup per CU
  if ((n_groups + blocks - 1) / (blocks > 0 ? blocks : 1) > 32 * kResColdWords) return MNF_ERR_UNSUPPORTED;
  tag_kernel("rnvp_resident");
  hipLaunchKernelGGL((rnvp_resident_kernel<HN, G, SAMPLE>), dim3((unsigned)blocks), dim3(kResWaves * 64), lds_bytes,
                     stream, z, x, log_det, simage, image, rows, accumulate, seed, q0_mean, q0_log_var, y_out);
  return check_launch();
}

// (hidden width the layer runs at, padded dim / 16) pairs with an instantiated kernel: MNFLinear(800, 50)'s flow_q
// (BASELINE configs[4]) and MNFFeedForward's 784-wide first layer
#define MNF_RNVP_RESIDENT_SHAPES(X) X(50, 50)

// MNF_ERR_UNSUPPORTED: no resident kernel for this shape -- the caller runs the streaming split kernel
int rnvp_resident_launch(const float* z, float* x, float* log_det, int accumulate, const void* split_image,
                         const float* image, int64_t rows, int dim, int hn_pad, uint64_t seed, const float* q0_mean,
                         const float* q0_log_var, int vec, hipStream_t stream, float* y_out) {
  // MNF_RNVP_RESIDENT=0 (read per call: tests and A/B runs flip it): the streaming kernels only
  const char* env = getenv("MNF_RNVP_RESIDENT");
  if ((env && env[0] == '0') || !split_image || !image || !log_det) return MNF_ERR_UNSUPPORTED;
  if ((dim & 15) || !vec) return MNF_ERR_UNSUPPORTED;  // whole 16-dim groups, 16-byte aligned rows
  const uint32_t* simage = static_cast<const uint32_t*>(split_image);
#define X(HN, GG)                                                                                                  \
  if (hn_pad == HN && dim == 16 * GG)                                                                              \
    return q0_mean ? launch_resident<HN, GG, true>(z, x, log_det, accumulate, simage, image, rows, seed, q0_mean,   \
                                                   q0_log_var, y_out, stream)                                      \
                   : launch_resident<HN, GG, false>(z, x, log_det, accumulate, simage, image, rows, seed, q0_mean,  \
                                                    q0_log_var, y_out, stream);
  MNF_RNVP_RESIDENT_SHAPES(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

}  // namespace mnf
