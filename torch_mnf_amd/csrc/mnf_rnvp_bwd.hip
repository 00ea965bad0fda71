// Gradients of the masked / gated RNVP coupling (flows/rnvp.py:25-39 under loss.backward()) on the f16 matrix pipe in
// split (hi + lo) fp32 arithmetic (mnf_split.h), gfx950.  This is what MNFLinear / MNFConv2d train through
// (layers/mnf_linear.py:58-64,84; tests/test_mnf_mnist.py:14-56).
//
//   forward   k = m z;  y = Wn k + bn;  t = Wt y + bt;  s = Ws y + bs;  gate = sigmoid(s)
//             x = (1-m) z gate + (1-gate) t + m z;      log_det = sum_c (1-m_c) log gate_c
//   backward  g_t = G (1-gate);   g_s = (G ((1-m) z - t) gate + g_ld (1-m)) (1-gate)           [G = grad_x]
//             g_y = Wt^T g_t + Ws^T g_s;     g_k = Wn^T g_y;     grad_z = G ((1-m) gate + m) + m g_k
//             dWt = g_t^T y, dbt = sum g_t (same for s);  dWn = g_y^T k, dbn = sum g_y            [sums over rows]
//
// Two things pull in opposite directions.  g_y is a sum over ALL d dims of a row, so it wants a wave to own rows and
// sweep the dims (like the forward kernels).  The weight gradients are sums over ALL rows with 3 d h = 120,000
// accumulators at d = 800, h = 50 (480 KB: no workgroup can hold them), so they want a workgroup to own a SLAB of dims
// and sweep the rows.  Hence two launches with a small per-row hand-over between them:
//
//   A  (row-parallel, the forward kernel's streaming scheme)   y = GEMM 1 over z;  second sweep over z and G: s, t, gate,
//      g_t, g_s and g_y += [Wt^T | Ws^T] [g_t; g_s] (one K = 32 step per 16 dims).  Writes, per 16-row tile, y and g_y
//      as ready-made split MFMA operands in BOTH orientations (units along the lane's registers for the K = units
//      products of B, rows along them for B's sums over rows; the second orientation is one MFMA against the identity
//      per tile, exact) -- 1 KB per row -- and dbn.  No row data is written.
//   B  (a workgroup owns 32 dims and a range of rows)           per 32 rows: s, t again from y (K = 64 units), gate, g_t,
//      g_s, g_k = Wn^T g_y, grad_z (the one write of row data), and dWt, dWs, dWn, dbt, dbs as K = 32-ROW products
//      into 24 accumulator tiles that stay in registers over the whole row range.
//      Everything here is computed TRANSPOSED (rows on the MFMA M axis): an accumulator then holds four ROWS of one dim
//      per lane, which is exactly the operand layout of a sum over rows, so no tile is ever transposed in B; rows are
//      read and written as 8-byte pieces (dims 2 j, 2 j + 1 of the slab: the slab's two 16-column tiles are its even
//      and its odd dims), 4 rows x 128 B per instruction.
//
// HBM traffic per row: z three times (A twice, the second mostly from cache; B once), G twice, grad_z once: <= 6 x 4 d
// bytes against the 3 x 4 d a gradient pass must move.  Storing g_t / g_s instead would cost 8 x 4 d.
//
// Cotangents of a mean over 256,000 rows are ~4e-6, below f16's normal range: G and g_ld are multiplied by a power of
// two from mnf_affine_half_grad_scale on the way in (exact) and the results scaled back on the way out.  Range guard: A
// tracks max|operand| per 128-row group (z, y, g_t, g_s, g_y); a group that reaches the split limit -- or any group when
// a weight exceeds the weight limit -- is flagged, skipped by B, and redone by the generic fp32 kernel on the flagged
// groups behind B (mnf_rnvp_bwd's kernel with a flag list: it returns at once for every other group).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "mnf_device.h"
#include "mnf_host.h"
#include "mnf_rnvp_common.h"
#include "mnf_split.h"

#include <type_traits>

namespace mnf {


constexpr int kBwdGroupRows = 16 * kRnvpWaves;  // rows per flag (A's 8-wave group)
constexpr int kBwdBWaves = 4;                   // waves of a B workgroup
// MNF_DETERMINISTIC (mnf_host.h): launch A's workgroups (at most kDetBlocksA) leave 64 floats each, the slab launches'
// row parts (at most kDetParts) a block of the layer's parameter count each, behind the tiles of the workspace
constexpr int kDetBlocksA = 1024, kDetParts = 64;
__host__ __device__ inline int64_t rnvp_bwd_params(int dm, int hn) { return 3 * (int64_t)dm * hn + hn + 2 * (int64_t)dm; }
#ifndef MNF_RNVP_BWD_ABL
#define MNF_RNVP_BWD_ABL 0  // timing experiments only (results are wrong): bit 0 no hand-over loads in B, bit 1 no row
#endif                      // loads, bit 2 no row-sum MFMAs, bit 3 no grad_z stores, bit 4 no K = units MFMAs
constexpr int kBwdAbl = MNF_RNVP_BWD_ABL;

template <int HN>
struct RnvpBwdShape {
  using S = RnvpSplitShape<HN>;
  static constexpr int YT = S::YT, NKS2 = S::NKS2;
  // backward-only operand image: [A3: per 16-dim group, YT (hi, lo) operands of [Wt^T | Ws^T]]
  //                              [B2: per 32-dim slab: (dim tile E/O) x (t, s) x NKS2 x (hi, lo)   B operands, K = units]
  //                              [B4: per slab: (E/O) x NKS2 x (hi, lo)                            Wn as B operand    ]
  //                      plain:  per slab: bt_E[16] bt_O[16] bs_E[16] bs_O[16]
  static constexpr int A3_TILE_WORDS = YT * 512;
  static constexpr int B2_SLAB_WORDS = 2 * 2 * NKS2 * 512;
  static constexpr int B4_SLAB_WORDS = 2 * NKS2 * 512;
  static constexpr int B_SLAB_PLAIN = 64;
  static constexpr int64_t n_slabs(int dm) { return (dm + 31) / 32; }
  static constexpr int64_t a3_words(int d16) { return (int64_t)(d16 / 16) * A3_TILE_WORDS; }
  static constexpr int64_t b2_words(int dm) { return n_slabs(dm) * B2_SLAB_WORDS; }
  static constexpr int64_t b4_words(int dm) { return n_slabs(dm) * B4_SLAB_WORDS; }
  static constexpr int64_t split_words(int dm, int d16) { return a3_words(d16) + b2_words(dm) + b4_words(dm); }
  static constexpr int64_t plain_words(int dm) { return n_slabs(dm) * B_SLAB_PLAIN; }
  // hand-over per 16-row tile: y ("a") and g_y ("b"), see HandoverShape
  using H = HandoverShape<YT>;
  static constexpr int TILE_WORDS = H::TILE_WORDS;
  static constexpr int Y_OP = H::A_OP, G_OP = H::B_OP, Y_TR = H::A_TR, G_TR = H::B_TR;
  // A's LDS window: a GEMM-1 chunk (KC K-steps) or MT second-sweep tiles (forward GEMM-2 tile + its A3 operands each)
  // followed by their (bt, bs) biases (32 words per tile; the window is padded to whole 1 KB pieces)
  static constexpr int MT = 2;  // (three tiles per chunk: 967 us against 817, and the extra row registers spill)
  static constexpr int TILE_OPS_WORDS = S::TILE2_WORDS + A3_TILE_WORDS;
  static constexpr int SWEEP2_WORDS = (MT * TILE_OPS_WORDS + MT * 32 + 255) / 256 * 256;
  static constexpr int CHUNK_WORDS = S::KC * S::KS1_WORDS > SWEEP2_WORDS ? S::KC * S::KS1_WORDS : SWEEP2_WORDS;
  static constexpr int STAGE_U4 = (CHUNK_WORDS / 4 + kRnvpWaves * 64 - 1) / (kRnvpWaves * 64);
};

// ================================================================================================ kernel A
typedef __attribute__((address_space(3))) void* lds_void_ptr_a;
constexpr int kBwdDist = 2;              // launch A requests its operands this many chunks ahead ...
constexpr int kBwdRing = kBwdDist + 1;   // ... into a ring of this many LDS buffers (chunks c .. c + kBwdDist)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
#else
typedef int BufRsrc;  // (hipcc's host pass over this file: the type and its builtins exist on the device side only)
#endif

template <int HN, bool SEEDED, bool RAG>
__device__ __forceinline__ void rnvp_bwd_group_a(uint32_t* lds0, const BufRsrc& s_rsrc, const BufRsrc& b_rsrc, int grp,
                                                 const float* __restrict__ z,
                                                 const float* __restrict__ mask, const float* __restrict__ gx,
                                                 const float* __restrict__ gld, const uint32_t* __restrict__ simage,
                                                 const uint32_t* __restrict__ bimage, uint32_t* __restrict__ side,
                                                 int32_t* __restrict__ flags, int32_t* __restrict__ list, float gscale,
                                                 bool weights_ok,
                                                 f32x4 (&bn_acc)[RnvpSplitShape<HN>::YT], int64_t rows, int d,
                                                 uint64_t seed, int dm_ragged, bool vec, const float* __restrict__ y_in) {
  using S = RnvpSplitShape<HN>;
  using B = RnvpBwdShape<HN>;
  const int dm = RAG ? dm_ragged : d;
  constexpr int YT = S::YT, NKS2 = S::NKS2, KC = S::KC;
  // (the wave index as a SCALAR: which image a piece comes from is then a scalar select of the buffer resource, not a
  //  waterfall loop per piece)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int G = d / 16;
  const int n_ks1 = (G + 1) / 2;
  constexpr int MT = B::MT;
  const int nc1 = (n_ks1 + KC - 1) / KC, nc = nc1 + (G + MT - 1) / MT;  // second sweep: MT 16-dim tiles per chunk
  const float* bias2 = reinterpret_cast<const float*>(simage + S::split_words(d));
  const float* bias_y = bias2 + (int64_t)G * 32;

  const int64_t row = (int64_t)grp * kBwdGroupRows + wave * 16 + j;
  const bool live = row < rows;
  const int64_t rowc = live ? row : rows - 1;
  const float* zr = z + rowc * dm + 4 * q;
  const float* mr = SEEDED ? nullptr : mask + rowc * dm + 4 * q;
  // (no grad_x: its loads read z instead and count for nothing -- every load of the second sweep is issued without a
  //  branch: behind a load under a guard hipcc's wait counts fall back to vmcnt(0), which would empty the two-chunk
  //  ring of rows and operand pieces at every chunk)
  const float* gr = (gx ? gx : z) + rowc * dm + 4 * q;
  const float g_on = gx ? gscale : 0.f;
  const float glr = (gld && live) ? gld[rowc] * gscale : 0.f;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};
  auto mask4 = [&](int g) -> f32x4 {
    if (g < 0) return zero4;
    if (!SEEDED) return row_load4<RAG>(mr, 16 * g, 4 * q, dm, vec);
    const int dd = 16 * g + 4 * q;
    const uint32_t w = rnvp_mask_word(seed, rowc, dd >> 5) >> (dd & 31);
    return f32x4{(float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u)};
  };
  auto z4 = [&](int g) -> f32x4 { return row_load4<RAG>(zr, 16 * (g < 0 ? 0 : g), 4 * q, dm, vec); };
  auto g4 = [&](int g) -> f32x4 { return row_load4<RAG>(gr, 16 * (g < 0 ? 0 : g), 4 * q, dm, vec); };

  // Operands of chunk c -- a GEMM-1 chunk is contiguous in the forward image; a second-sweep chunk is the forward GEMM-2
  // tile followed by the tile's A3 operands from the backward image -- travel L2 -> LDS by LDS-DMA (buffer_load ... lds:
  // no staging registers) into a ring of kBwdRing buffers, requested TWO chunks ahead: at the top of chunk c the wave
  // asks for its pieces of chunk c + 2, at the end of chunk c it waits for its pieces of chunk c + 1 and meets the
  // others at the chunk's one barrier.  (Round 3 staged chunk c + 1 through registers during chunk c and wrote it to LDS
  // at the chunk's end: with one workgroup per CU two thirds of the wave cycles went to waiting there.)
  constexpr int N_PIECES = B::CHUNK_WORDS / 256, N_DMA = (N_PIECES + kRnvpWaves - 1) / kRnvpWaves;
  [[maybe_unused]] constexpr int T2_PIECES = S::TILE2_WORDS / 256, TILE_PIECES = B::TILE_OPS_WORDS / 256;
  static_assert(B::CHUNK_WORDS % 256 == 0 && S::TILE2_WORDS % 256 == 0 && B::TILE_OPS_WORDS % 256 == 0, "whole 1 KB pieces");
  auto request_operands = [&](int c) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource builtins do not exist in hipcc's host pass over this file)
    const int cc = c < nc ? c : nc - 1;  // (past the end: the last chunk once more, into the buffer it already fills)
    uint32_t* dst = lds0 + (cc % kBwdRing) * B::CHUNK_WORDS;
#pragma unroll
    for (int i = 0; i < N_DMA; ++i) {
      // wave-uniform; a piece past the chunk's operands repeats the last one (same bytes to the same place)
      const int piece = min(i * kRnvpWaves + wave, (cc < nc1 ? KC * S::KS1_WORDS / 256 : MT * TILE_PIECES) - 1);
      int64_t word;
      bool from_b = false;
      if (cc < nc1) {
        word = (int64_t)cc * KC * S::KS1_WORDS + piece * 256;  // (a short last chunk reads on into part 2: unused)
      } else {
        // piece -> (tile of the chunk, piece of the tile); a tile past the end (odd G) repeats the last one: not computed
        const int mi = piece / TILE_PIECES, pt = piece - mi * TILE_PIECES;
        const int m = min(MT * (cc - nc1) + mi, G - 1);
        from_b = pt >= T2_PIECES;
        word = from_b ? (int64_t)m * B::A3_TILE_WORDS + (pt - T2_PIECES) * 256
                      : S::part1_words(d) + (int64_t)m * S::TILE2_WORDS + pt * 256;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(from_b ? b_rsrc : s_rsrc, (lds_void_ptr_a)(dst + piece * 256), 16, lane * 16,
                                               (int)(word * 4), 0, 0);
    }
    if (cc >= nc1) {  // the chunk's (bt, bs) biases: 64 words behind the operands, 4 bytes per lane (every wave: same bytes)
      const int64_t word = S::split_words(d) + (int64_t)MT * (cc - nc1) * 32;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void_ptr_a)(dst + MT * B::TILE_OPS_WORDS), 4, lane * 4,
                                               (int)(word * 4), 0, 0);
    }
#endif
    asm volatile("" ::: "memory");
  };
  // the same for a chunk of the second sweep only (c >= nc1 known: no branch around the requests)
  auto request_operands2 = [&](int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int cc = c < nc ? c : nc - 1;
    uint32_t* dst = lds0 + (cc % kBwdRing) * B::CHUNK_WORDS;
#pragma unroll
    for (int i = 0; i < N_DMA; ++i) {
      const int piece = min(i * kRnvpWaves + wave, MT * TILE_PIECES - 1);
      const int mi = piece / TILE_PIECES, pt = piece - mi * TILE_PIECES;
      const int m = min(MT * (cc - nc1) + mi, G - 1);
      const bool from_b = pt >= T2_PIECES;
      const int64_t word = from_b ? (int64_t)m * B::A3_TILE_WORDS + (pt - T2_PIECES) * 256
                                  : S::part1_words(d) + (int64_t)m * S::TILE2_WORDS + pt * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(from_b ? b_rsrc : s_rsrc, (lds_void_ptr_a)(dst + piece * 256), 16, lane * 16,
                                               (int)(word * 4), 0, 0);
    }
    const int64_t bword = S::split_words(d) + (int64_t)MT * (cc - nc1) * 32;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void_ptr_a)(dst + MT * B::TILE_OPS_WORDS), 4, lane * 4,
                                             (int)(bword * 4), 0, 0);
#endif
    asm volatile("" ::: "memory");
  };
  // end of a chunk: this wave's pieces of the NEXT chunk are in LDS (vector-memory operations complete in issue order;
  // behind those pieces the wave has issued at least the N_DMA pieces of the chunk after it and two row / bias loads),
  // every LDS read of this chunk has returned, then the workgroup's barrier.  Inline asm: __syncthreads() drains every
  // outstanding vector-memory operation.
  auto chunk_barrier = [&](bool drain) {
    if (drain)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N_DMA * (kBwdDist - 1) + 2) : "memory");
  };
  auto row_group1 = [&](int c, int i) -> int {
    const int g = 2 * (c * KC) + i;
    return g < G ? g : -1;
  };

  // ---- sweep 1: y^T = Wn (m z)^T + bn, exactly as the forward kernel (rows requested D1 chunks ahead)
  constexpr int D1 = SEEDED ? 2 : 1;
  f32x4 z1[D1][2 * KC], m1[D1][2 * KC];
  auto request_rows1 = [&](int c, int u) {
#pragma unroll
    for (int i = 0; i < 2 * KC; ++i) {
      const int g = row_group1(c < nc1 ? c : nc1 - 1, i);
      z1[u][i] = z4(g);
      m1[u][i] = mask4(g);
    }
  };
  // y_in != nullptr: the forward pass kept y = Wn (m z) + bn (rows x 16 YT floats): sweep 1 -- a third of this launch,
  // one of its two reads of z -- is skipped and the operand pipeline starts at the first second-sweep tile
  const bool have_y = y_in != nullptr;
  const int c_first = have_y ? nc1 : 0;
  chunk_barrier(true);  // the previous group's last chunk is fully consumed (and its hand-over stores are out)
  if (!have_y) {
#pragma unroll
    for (int u = 0; u < D1; ++u) request_rows1(u, u);
  }
#pragma unroll
  for (int u = 0; u < kBwdDist; ++u) request_operands(c_first + u);
  chunk_barrier(true);  // (the first chunk's pieces: the one exposed operand latency of the group)
  f32x4 ym[YT], yc[YT];
#pragma unroll
  for (int m = 0; m < YT; ++m) {
    if (have_y)
      ym[m] = *reinterpret_cast<const f32x4*>(y_in + rowc * (16 * YT) + m * 16 + 4 * q);
    else
      ym[m] = *reinterpret_cast<const f32x4*>(bias_y + m * 16 + 4 * q);
    yc[m] = zero4;
  }
  float mx = weights_ok ? 0.f : __builtin_inff();
  if (have_y) {  // NaN rows: the forward pass recomputed their group in fp32 and kept no y -- this group's fix-up, then
                 // (explicitly: the fmaxf of the range check drops NaNs)
#pragma unroll
    for (int m = 0; m < YT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (live && !(ym[m][r] == ym[m][r])) mx = __builtin_inff();
  }
  for (int c0 = 0; c0 < (have_y ? 0 : nc1); c0 += D1) {
#pragma unroll
    for (int u = 0; u < D1; ++u) {
      const int c = c0 + u;
      if (c < nc1) {
        f16x8 bh[KC], bl[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          u32x2 h0, l0, h1, l1;
          split_tile(m1[u][2 * kk] * z1[u][2 * kk], h0, l0, mx);
          split_tile(m1[u][2 * kk + 1] * z1[u][2 * kk + 1], h1, l1, mx);
          bh[kk] = pair_operand(h0, h1);
          bl[kk] = pair_operand(l0, l1);
        }
        request_operands(c + kBwdDist);  // (c + 1 == nc1: the first second-sweep tile)
        request_rows1(c + D1, u);
        const uint32_t* buf = lds0 + (c % kBwdRing) * B::CHUNK_WORDS;
        const f16x8* A8 = reinterpret_cast<const f16x8*>(buf) + lane;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          if (c * KC + kk < n_ks1) {
#pragma unroll
            for (int m = 0; m < YT; ++m)
              split_mac(A8[64 * (2 * (kk * YT + m))], A8[64 * (2 * (kk * YT + m) + 1)], bh[kk], bl[kk], ym[m], yc[m]);
          }
        }
        chunk_barrier(false);
      }
    }
  }
  // ---- y complete: GEMM-2 operands
  // rows of a chunk's MT tiles, requested two chunks ahead into the register set of the chunk's parity (the in-kernel
  // mask is hashed where it is used; (bt, bs) come out of the LDS window)
  f32x4 z2[2][MT], g2[2][MT], m2[2][MT];
  auto tile_of = [&](int c, int mi) { return min(MT * ((c < nc ? c : nc - 1) - nc1) + mi, G - 1); };
  auto request_rows2 = [&](int c, int u) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      z2[u][mi] = z4(tile_of(c, mi));
      g2[u][mi] = g4(tile_of(c, mi));
      if (!SEEDED) m2[u][mi] = mask4(tile_of(c, mi));
    }
  };
#pragma unroll
  for (int u = 0; u < 2; ++u) request_rows2(nc1 + u, u);  // (register set = the chunk's parity counted from nc1)
  u32x2 yh[YT], yl[YT];
#pragma unroll
  for (int m = 0; m < YT; ++m) split_tile(yc[m] * kSplitInvScale + ym[m], yh[m], yl[m], mx);

  // ---- sweep 2: per 16 dims  s, t -> gate -> g_t, g_s -> g_y += [Wt^T | Ws^T] [g_t; g_s]; MT tiles per chunk (one
  // barrier per 32 dims, and two independent tiles for the scheduler to overlap)
  f32x4 gm[YT], gc[YT];
#pragma unroll
  for (int m = 0; m < YT; ++m) gm[m] = gc[m] = zero4;
  // (pairs of chunks without a guard around their requests -- see g_on above --, then the odd last one)
  auto chunk2 = [&](int c, auto u_c) {
    constexpr int u = decltype(u_c)::value;
    request_operands2(c + kBwdDist);
    const uint32_t* buf = lds0 + (c % kBwdRing) * B::CHUNK_WORDS;
    const float* bias = reinterpret_cast<const float*>(buf + MT * B::TILE_OPS_WORDS);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = MT * (c - nc1) + mi;
      // (one tile at a time: left free, the scheduler interleaves the chunk's tiles and spills ~90 registers)
      __builtin_amdgcn_sched_barrier(0);
      if (m < G) {  // (wave-uniform: an odd number of tiles leaves the last chunk's second tile empty)
        const f32x4 bt = *reinterpret_cast<const f32x4*>(bias + mi * 32 + 4 * q);
        const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + mi * 32 + 16 + 4 * q);
        const f16x8* T8 = reinterpret_cast<const f16x8*>(buf + mi * B::TILE_OPS_WORDS) + lane;
        const f16x8* A3 = reinterpret_cast<const f16x8*>(buf + mi * B::TILE_OPS_WORDS + S::TILE2_WORDS) + lane;
        f32x4 tm = zero4, tc = zero4, sm = zero4, sc = zero4;
#pragma unroll
        for (int ks = 0; ks < NKS2; ++ks) {
          const f16x8 bh = pair_operand(yh[2 * ks], 2 * ks + 1 < YT ? yh[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
          const f16x8 bl = pair_operand(yl[2 * ks], 2 * ks + 1 < YT ? yl[2 * ks + 1 < YT ? 2 * ks + 1 : 0] : zero2);
          split_mac(T8[64 * (2 * ks)], T8[64 * (2 * ks + 1)], bh, bl, tm, tc);
          split_mac(T8[64 * (2 * (NKS2 + ks))], T8[64 * (2 * (NKS2 + ks) + 1)], bh, bl, sm, sc);
        }
        const f32x4 t4 = tc * kSplitInvScale + tm + bt;
        const f32x4 s4 = sc * kSplitInvScale + sm + bs;
        const f32x4 mk = SEEDED ? mask4(m) : m2[u][mi];
        f32x4 gt, gs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zz = z2[u][mi][r], mm = mk[r], nm = 1.f - mm;
          mx = __builtin_fmaxf(mx, __builtin_fabsf(mm * zz));  // (launch B-n splits m z: sweep 1's guard when it is skipped)
          const float GG = live ? g2[u][mi][r] * g_on : 0.f;
          const float gate = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(s4[r] * -1.44269504088896341f));
          const float omg = 1.f - gate;
          gt[r] = GG * omg;
          gs[r] = (GG * (nm * zz - t4[r]) * gate + glr * nm) * omg;
        }
        u32x2 th, tl, sh, sl;
        split_tile(gt, th, tl, mx);
        split_tile(gs, sh, sl, mx);
        const f16x8 bh = pair_operand(th, sh), bl = pair_operand(tl, sl);
#pragma unroll
        for (int m2i = 0; m2i < YT; ++m2i)
          split_mac(A3[64 * (2 * m2i)], A3[64 * (2 * m2i + 1)], bh, bl, gm[m2i], gc[m2i]);
      }
    }
    request_rows2(c + 2, u);
    chunk_barrier(false);
  };
  int c2 = nc1;
  for (; c2 + 1 < nc; c2 += 2) {
    chunk2(c2, std::integral_constant<int, 0>{});
    chunk2(c2 + 1, std::integral_constant<int, 1>{});
  }
  if (c2 < nc) chunk2(c2, std::integral_constant<int, 0>{});
  // ---- g_y complete: the group's verdict, then the hand-over
  f32x4 gy[YT];
  u32x2 gh[YT], gl2[YT];
#pragma unroll
  for (int m = 0; m < YT; ++m) {
    gy[m] = gc[m] * kSplitInvScale + gm[m];
    split_tile(gy[m], gh[m], gl2[m], mx);
  }
  const bool bad = __syncthreads_or(!(mx <= kSplitLimit) ? 1 : 0) != 0;
  if (threadIdx.x == 0) {
    flags[grp] = bad ? 1 : 0;
    if (bad) list[1 + atomicAdd(list, 1)] = grp;  // (list[0] zeroed by the launcher)
  }
  if (bad) return;
  const int64_t tile = (int64_t)grp * kRnvpWaves + wave;
  if (tile * 16 >= rows) return;  // wave-uniform: a tile past the end (the group's last rows)
#pragma unroll
  for (int m = 0; m < YT; ++m) bn_acc[m] += gy[m];  // (dead rows carry g_y = 0)
  store_handover<YT>(side + tile * B::TILE_WORDS, yh, yl, gh, gl2, lane, j, q);
}

template <int HN, bool SEEDED, bool RAG>
__global__ void __launch_bounds__(kRnvpWaves * 64, 2)
rnvp_bwd_a_kernel(const float* __restrict__ z, const float* __restrict__ mask, const float* __restrict__ gx,
                  const float* __restrict__ gld, const uint32_t* __restrict__ simage, const uint32_t* __restrict__ bimage,
                  uint32_t* __restrict__ side, int32_t* __restrict__ flags, int32_t* __restrict__ list,
                  const float* __restrict__ gscale_dev, float* __restrict__ grad_flat, int64_t rows, int d, int dm_ragged,
                  int hn, uint64_t seed, int vec_ok, int64_t bimage_tail, const float* __restrict__ y_in,
                  float* __restrict__ det_bn) {
  using S = RnvpSplitShape<HN>;
  using B [[maybe_unused]] = RnvpBwdShape<HN>;  // (used by the device pass only)
  extern __shared__ __attribute__((aligned(16))) uint32_t a_lds[];  // kBwdRing x RnvpBwdShape::CHUNK_WORDS (101 KB at 64 units)
  const int dm = RAG ? dm_ragged : d;
#if defined(__HIP_DEVICE_COMPILE__)
  const BufRsrc s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(simage), 0, (int)((S::split_words(d) + S::plain_words(d) + kSplitTailWords) * 4), 0x00020000);
  const BufRsrc b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(bimage), 0, (int)((bimage_tail + kSplitTailWords) * 4), 0x00020000);
#else
  const BufRsrc s_rsrc = 0, b_rsrc = 0;
#endif
  const float gscale = gscale_dev[0];
  const float wmax_f = __builtin_bit_cast(float, simage[S::split_words(d) + S::plain_words(d)]);
  const float wmax_b = __builtin_bit_cast(float, bimage[bimage_tail]);
  const bool weights_ok = wmax_f <= kSplitWeightLimit && wmax_b <= kSplitWeightLimit;
  f32x4 bn_acc[S::YT];
#pragma unroll
  for (int m = 0; m < S::YT; ++m) bn_acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_groups = (int)((rows + kBwdGroupRows - 1) / kBwdGroupRows);
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x)
    rnvp_bwd_group_a<HN, SEEDED, RAG>(a_lds, s_rsrc, b_rsrc, grp, z, mask, gx, gld, simage, bimage, side, flags, list, gscale,
                                      weights_ok, bn_acc, rows, d, seed, dm, vec_ok != 0, y_in);
  // dbn: sum over the wave's rows (the 16 lanes j of a q), over the workgroup's waves in LDS, then ONE atomic per unit
  // per workgroup (atomics on a few dozen addresses serialise at the memory side: one per unit per WAVE cost 0.16 ms)
  if (grad_flat && !(kBwdAbl & 64)) {
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    float* bsum = reinterpret_cast<float*>(a_lds);
    __syncthreads();  // the operand window is no longer read
    if (threadIdx.x < 16 * S::YT) bsum[threadIdx.x] = 0.f;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < S::YT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = bn_acc[m][r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        bn_acc[m][r] = v;
      }
    // (det_bn: the waves add in turn and the workgroup's sums go out as its own block of 64, see mnf_host.h)
    const bool det = det_bn != nullptr;
    lds_wave_add<kRnvpWaves>(det, (int)(threadIdx.x >> 6), [&](auto op) {
#pragma unroll
      for (int m = 0; m < S::YT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j == 0) op(bsum + 16 * m + 4 * q + r, bn_acc[m][r]);
    });
    __syncthreads();
    if ((int)threadIdx.x < hn) {
      const float v = bsum[threadIdx.x] * (1.f / gscale);
      if (det)
        det_bn[(int64_t)blockIdx.x * 64 + threadIdx.x] = v;
      else if (v != 0.f)
        atomicAdd(grad_flat + (int64_t)hn * dm + threadIdx.x, v);
    }
  }
}

// ================================================================================================ kernels B
// the mask words of a tile's 16 rows for one 32-dim slab: lane l hashes row (l & 15) once, the four rows a lane needs
// (4 q + r) are fetched from the lanes that hold them -- one hash and four cross-lane reads instead of four hashes
__device__ __forceinline__ void tile_mask_words(uint64_t seed, int64_t tbase, int n_live, int slab, int lane, int q,
                                                uint32_t (&w)[4]) {
  const int jr = lane & 15;
  const uint32_t h = rnvp_mask_word(seed, tbase + (jr < n_live ? jr : 0), slab);
#pragma unroll
  for (int r = 0; r < 4; ++r) w[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (4 * q + r), (int)h);
}

// ------------------------------------------------------------------------------------------------ B-ts, shared hand-over
// The kernel above re-reads launch A's hand-over (1 KB per row) once per 32-dim slab -- 4 x the bytes of the slab's own
// row data -- and holds both row tiles' hand-over operands in registers (64 of them) next to 64 accumulators: it spills
// ~70 registers at two waves per SIMD and waits on each load in turn.  Here a workgroup of EIGHT waves owns FOUR adjacent
// slabs and walks its row pairs together: wave (slab s, tile t) computes tile 2 p + t of pair p for slab s.  The pair's
// hand-over (y and g_y with units on K, y with rows on K: 12 KB per tile at 64 units) and its 32 g_ld values are copied
// ONCE per workgroup into a double-buffered LDS window by LDS-DMA (global_load_lds: no staging registers) while the
// previous pair is being computed, and every wave reads its MFMA operands out of it as it needs them: a quarter of the
// hand-over traffic, no hand-over registers, and the next pair's rows prefetched into the registers that frees.  A wave
// owns ONE 16-row tile, so the sums over rows run straight off the tile's accumulators (as K = 32 products whose two K
// halves carry the head and the residual of y: see row_sums).  LDS: 4 slabs x 24 KB of operands + 1 KB of biases + 2 x 2 x 12 KB of hand-over = 145 KB: one workgroup
// per CU, two waves per SIMD.  One barrier per pair.
typedef __attribute__((address_space(3))) void* lds_void_ptr_b;
#ifndef MNF_RNVP_TS_ABL
#define MNF_RNVP_TS_ABL 0  // timing experiments only (results are wrong): bit 0 no arithmetic (loads, pieces, stores and
#endif                     // barriers only), bit 1 no row loads / pieces / stores (arithmetic on whatever is there)
constexpr int kTsAbl = MNF_RNVP_TS_ABL;
constexpr int kTsSlabs = 4;              // slabs per workgroup
constexpr int kTsWaves = 2 * kTsSlabs;   // (slab, tile of the pair)

template <int HN>
struct RnvpTsShape {
  using B = RnvpBwdShape<HN>;
  using H = typename B::H;
  static constexpr int YT = B::YT;
  static constexpr int W_WORDS = B::B2_SLAB_WORDS + B::B4_SLAB_WORDS;  // one slab's split operands
  static constexpr int HT_WORDS = 2 * H::OP_WORDS + H::TR_WORDS;       // A_OP, B_OP, A_TR: the head of a tile's hand-over
  static_assert(H::A_OP == 0 && H::B_OP == H::OP_WORDS && H::A_TR == 2 * H::OP_WORDS, "the three parts are contiguous");
  static_assert(HT_WORDS % 256 == 0, "whole 1 KB pieces");
  static constexpr int HT_PIECES = HT_WORDS / 256;
  static constexpr int N_DMA = (2 * HT_PIECES + kTsWaves - 1) / kTsWaves;  // pieces per wave and pair
  static constexpr int BIAS_OFF = kTsSlabs * W_WORDS;
  static constexpr int H_OFF = BIAS_OFF + kTsSlabs * B::B_SLAB_PLAIN;
  static constexpr int GL_OFF = H_OFF + 2 * 2 * HT_WORDS;  // g_ld of the pair's rows: [buffer][wave][64 words]
  static constexpr int LDS_WORDS = GL_OFF + 2 * kTsWaves * 64;
  static constexpr int UP = 16 * YT + 1;                 // padded [dim] row of the flush area
  static constexpr int RED_SLAB = 2 * 32 * UP + 64;      // Wt, Ws blocks and the two bias rows of one slab
  static_assert(kTsSlabs * RED_SLAB <= BIAS_OFF, "the flush area fits the operand area");
  static_assert(LDS_WORDS * 4 <= 160 * 1024, "fits the CU's LDS");
};

template <int HN, bool SEEDED, bool RAG>
__global__ void __launch_bounds__(kTsWaves * 64, 2)
rnvp_bwd_ts_shared_kernel(const float* __restrict__ z, const float* __restrict__ mask, const float* __restrict__ gx,
                          const float* __restrict__ gld, float* __restrict__ grad_z, float* __restrict__ grad_flat,
                          const uint32_t* __restrict__ bimage, const uint32_t* __restrict__ side,
                          const int32_t* __restrict__ flags, const float* __restrict__ gscale_dev, int64_t rows, int dm,
                          int d16, int hn, uint64_t seed, int n_slabs, int row_parts, int vec2,
                          float* __restrict__ det_part) {
  using S = RnvpSplitShape<HN>;
  using B = RnvpBwdShape<HN>;
  using T = RnvpTsShape<HN>;
  constexpr int YT = S::YT, NKS2 = S::NKS2;
  extern __shared__ __attribute__((aligned(16))) uint32_t ts_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sl = wave & (kTsSlabs - 1), tt = wave / kTsSlabs;  // (waves w and w + 4 share a SIMD: same slab, other tile)
  const int j = lane & 15, q = lane >> 4;
  const float gscale = gscale_dev[0], inv_gscale = 1.f / gscale;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t n_tiles = (rows + 15) / 16, n_pairs = (n_tiles + 1) / 2;
  const int64_t per_part = (n_pairs + row_parts - 1) / row_parts;
  const float* gsrc = gx ? gx : z;  // (no cotangent for x: read z and multiply by zero)
  const float gx_scale = gx ? gscale : 0.f;
  const float gl_scale = gld ? gscale : 0.f;
  [[maybe_unused]] const float* lsrc = gld ? gld : z;
  const int n_groups = (n_slabs + kTsSlabs - 1) / kTsSlabs;
  const SlabItems items(n_groups, row_parts);
  for (int item = items.first; item < items.n_items; item += items.step) {
    const int sg = items.slab(item), part = items.part(item);
    const int slab = sg * kTsSlabs + sl;
    const bool slab_ok = slab < n_slabs;  // (wave-uniform: the last group may hold fewer than four slabs)
    const int64_t p0 = (int64_t)part * per_part, p_end = min(n_pairs, (int64_t)(part + 1) * per_part);
    if (p0 >= p_end) continue;  // (workgroup-uniform)
    __syncthreads();  // the previous item's flush is over
#pragma unroll 1
    for (int s = 0; s < kTsSlabs; ++s) {
      const int sb = sg * kTsSlabs + s;
      if (sb >= n_slabs) break;
      const uint4* b2 = reinterpret_cast<const uint4*>(bimage + B::a3_words(d16) + (int64_t)sb * B::B2_SLAB_WORDS);
      const uint4* b4 =
          reinterpret_cast<const uint4*>(bimage + B::a3_words(d16) + B::b2_words(dm) + (int64_t)sb * B::B4_SLAB_WORDS);
      const uint32_t* pl = bimage + B::split_words(dm, d16) + (int64_t)sb * B::B_SLAB_PLAIN;
      uint4* w = reinterpret_cast<uint4*>(ts_lds + s * T::W_WORDS);
      for (int i = threadIdx.x; i < B::B2_SLAB_WORDS / 4; i += blockDim.x) w[i] = b2[i];
      for (int i = threadIdx.x; i < B::B4_SLAB_WORDS / 4; i += blockDim.x) w[B::B2_SLAB_WORDS / 4 + i] = b4[i];
      if ((int)threadIdx.x < B::B_SLAB_PLAIN) ts_lds[T::BIAS_OFF + s * B::B_SLAB_PLAIN + threadIdx.x] = pl[threadIdx.x];
    }
    // (visible to every wave behind the pair loop's first barrier)
    const int dim0 = 32 * (slab_ok ? slab : n_slabs - 1) + 2 * j;  // the lane's even dim; + 1: its odd dim
    const bool in0 = dim0 < dm, in1 = dim0 + 1 < dm;
    const uint32_t lane_off = (uint32_t)(4 * q) * (uint32_t)dm + (uint32_t)dim0;
    int w_lane = lane;  // opaque, refreshed per pair: keeps the (pair-independent) operand reads inside the loop
    const f16x8* W8 = reinterpret_cast<const f16x8*>(ts_lds + sl * T::W_WORDS);
    auto w2 = [&](int dt, int net, int ks, int part_) { return W8[w_lane + 64 * (2 * ((dt * 2 + net) * NKS2 + ks) + part_)]; };
    auto w4 = [&](int dt, int ks, int part_) { return W8[w_lane + 64 * (2 * (4 * NKS2 + dt * NKS2 + ks) + part_)]; };

    f32x4 aWt[2][YT], aWs[2][YT];
    float abt[2] = {0.f, 0.f}, abs_[2] = {0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int m = 0; m < YT; ++m) aWt[dt][m] = aWs[dt][m] = zero4;

    // a tile's rows as LOADED (nothing here waits for a load: the values are first touched by compute(), one pair later)
    struct RowsIn {
      f32x2 zz[4], GG[4], mm[4];
      uint32_t mbits;  // SEEDED: bit 2 r + dt = the mask of (row 4 q + r, dim dim0 + dt)
      bool active;
      int n_live;
      int64_t tbase;
    };
    auto load_rows = [&](int64_t p, RowsIn& in) {
      const int64_t tile = 2 * p + tt;
      const bool has = tile < n_tiles;
      in.tbase = (has ? tile : 2 * p) * 16;  // wave-uniform
      in.n_live = has ? (int)min((int64_t)16, rows - in.tbase) : 0;
      in.active = has && slab_ok && flags[(p * 32) / kBwdGroupRows] == 0;  // the generic kernel redoes flagged groups
      // (an idle wave loads all the same, from its clamped slab: no branch around the loads, so that the compiler's
      // count of the operations in flight stays exact)
      const float* zt = z + in.tbase * dm;
      const float* gt_ = gsrc + in.tbase * dm;
      const float* mt = SEEDED ? nullptr : mask + in.tbase * dm;
      in.mbits = 0u;
      if (SEEDED) {
        uint32_t mw[4];
        tile_mask_words(seed, in.tbase, in.n_live, slab, lane, q, mw);
#pragma unroll
        for (int r = 0; r < 4; ++r) in.mbits |= ((mw[r] >> (2 * j)) & 3u) << (2 * r);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * q + r;
        const bool live = rr < in.n_live;
        // a row past the end reads the tile's first row instead (its cotangents are zeroed, nothing of it is stored);
        // byte offsets in 32 bits: [uniform tile base] + offset addressing, no 64-bit address pair per access
        const uint32_t ob = (live ? lane_off + (uint32_t)r * (uint32_t)dm : (uint32_t)dim0) * 4u;
        auto at = [&](const float* base) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + ob); };
        f32x2 zv = {0.f, 0.f}, gv = {0.f, 0.f}, mv = {0.f, 0.f};
        if (kTsAbl & 2) {
          zv = gv = f32x2{0.25f * lane, 1.f};
        } else if (!RAG) {
          zv = *reinterpret_cast<const f32x2*>(at(zt));
          gv = *reinterpret_cast<const f32x2*>(at(gt_));
          if (!SEEDED) mv = *reinterpret_cast<const f32x2*>(at(mt));
        } else if (vec2) {
          if (in0) {
            zv = *reinterpret_cast<const f32x2*>(at(zt));
            gv = *reinterpret_cast<const f32x2*>(at(gt_));
            if (!SEEDED) mv = *reinterpret_cast<const f32x2*>(at(mt));
          }
        } else {
          if (in0) {
            zv[0] = at(zt)[0];
            gv[0] = at(gt_)[0];
            if (!SEEDED) mv[0] = at(mt)[0];
          }
          if (in1) {
            zv[1] = at(zt)[1];
            gv[1] = at(gt_)[1];
            if (!SEEDED) mv[1] = at(mt)[1];
          }
        }
        in.zz[r] = zv;
        in.GG[r] = gv;
        in.mm[r] = mv;
      }
    };
    // the pair's hand-over -> LDS buffer `buf`: 2 x HT_PIECES pieces of 1 KB, dealt to the eight waves (no branches: a
    // piece past the end, or of a second tile that does not exist, repeats a valid one), and g_ld of the pair's rows
    // (buffer_load ... lds, not global_load_lds: the compiler files the latter under "flat, may touch LDS and memory" and
    // from then on waits for EVERY outstanding load -- s_waitcnt vmcnt(0) -- at the first use of any loaded register)
    auto request_handover = [&](int64_t p, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource builtins do not exist in hipcc's host pass over this file)
      if (kTsAbl & 2) return;
      const int has1 = 2 * p + 1 < n_tiles ? 1 : 0;
      const __amdgpu_buffer_rsrc_t pair_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint32_t*>(side + (2 * p) * B::TILE_WORDS), 0, 2 * B::TILE_WORDS * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < T::N_DMA; ++i) {
        const int piece = min(i * kTsWaves + wave, 2 * T::HT_PIECES - 1);  // wave-uniform
        const int t = piece / T::HT_PIECES, k = piece - t * T::HT_PIECES;
        uint32_t* dst = ts_lds + T::H_OFF + (buf * 2 + t) * T::HT_WORDS + k * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pair_rsrc, (lds_void_ptr_b)dst, 16, lane * 16,
                                             (t & has1) * (B::TILE_WORDS * 4) + k * 1024, 0, 0);
      }
      {  // g_ld of the pair's rows (past the end: the last row's value, multiplied by zero later)
        const int64_t left = rows - p * 32;  // >= 1
        const int l = left > lane ? lane : (int)left - 1;
        const __amdgpu_buffer_rsrc_t gl_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lsrc + p * 32), 0, 64 * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gl_rsrc, (lds_void_ptr_b)(ts_lds + T::GL_OFF + (buf * kTsWaves + wave) * 64), 4,
                                             l * 4, 0, 0, 0);
      }
#endif
      asm volatile("" ::: "memory");  // (the pair's stores stay behind the pieces: landed_barrier() counts on it)
    };
    // Vector-memory operations complete in issue order: with `stores` operations known to have been issued behind this
    // wave's pieces, all but the youngest `stores` being complete means the pieces are in LDS (the stores stay in flight).
    // Inline asm, not __syncthreads(): that one's release fence drains every outstanding store (s_waitcnt vmcnt(0)).
    auto landed_barrier = [&](int stores) {
      if (stores == 4)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // returns the number of vector-memory instructions it issued for certain (4 or 0)
    auto compute = [&](RowsIn& in, int buf) -> int {
      if (!in.active) return 0;
      asm volatile("" : "+v"(w_lane));
      uint32_t h_off = (uint32_t)(T::H_OFF + (buf * 2 + tt) * T::HT_WORDS) * 4u + (uint32_t)lane * 16u;
      asm volatile("" : "+v"(h_off));
      const char* hb = reinterpret_cast<const char*>(ts_lds) + h_off;             // + operand * 1 KB
      const char* hb_tr = reinterpret_cast<const char*>(ts_lds) + h_off - lane * 8u + 2 * T::H::OP_WORDS * 4;  // 8 B per lane
      auto y_op = [&](int ks, int part_) { return *reinterpret_cast<const f16x8*>(hb + (2 * ks + part_) * 1024); };
      auto g_op = [&](int ks, int part_) {
        return *reinterpret_cast<const f16x8*>(hb + T::H::OP_WORDS * 4 + (2 * ks + part_) * 1024);
      };
      auto y_tr = [&](int m, int part_) { return *reinterpret_cast<const u32x2*>(hb_tr + (2 * m + part_) * 512); };
      const float* bias = reinterpret_cast<const float*>(ts_lds + T::BIAS_OFF + sl * B::B_SLAB_PLAIN);
      const f32x4 gl4 = *reinterpret_cast<const f32x4*>(ts_lds + T::GL_OFF + (buf * kTsWaves + wave) * 64 + tt * 16 + 4 * q);
      // Three blocks: U (K = units products: t, s from y and g_k from g_y; 36 MFMAs and their operand reads), V (the gate
      // arithmetic) and R (the sums over the tile's rows, 48 MFMAs).  What bounds this kernel is INSTRUCTION ISSUE: the two
      // waves of a SIMD share its issue slots, and at ~600 instructions per wave and pair the arithmetic alone took 737 us
      // of a 956 us launch (software-pipelining U / V / R across the dim tiles in pinned pieces of ~25 vector instructions
      // and 3 .. 6 MFMAs changed nothing: 951 us).  So V is written on f32x2 values -- the lane's two dims of a row, as
      // they are loaded -- and goes out as packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two
      // elements per issue slot) wherever the operation has a packed form.
      struct Units {
        f32x4 t4, s4, gk;
      };
      auto units = [&](int dt) -> Units {  // t^T, s^T [row][dim] = y [row][unit] W^T [unit][dim];  g_k^T = g_y Wn
        f32x4 tm = zero4, tc = zero4, sm = zero4, sc = zero4, km = zero4, kc = zero4;
#pragma unroll
        for (int ks = 0; ks < NKS2; ++ks) {
          const f16x8 yh = y_op(ks, 0), yl = y_op(ks, 1);
          split_mac(yh, yl, w2(dt, 0, ks, 0), w2(dt, 0, ks, 1), tm, tc);
          split_mac(yh, yl, w2(dt, 1, ks, 0), w2(dt, 1, ks, 1), sm, sc);
          split_mac(g_op(ks, 0), g_op(ks, 1), w4(dt, ks, 0), w4(dt, ks, 1), km, kc);
        }
        Units u;
        u.t4 = tc * kSplitInvScale + tm + bias[dt * 16 + j];
        u.s4 = sc * kSplitInvScale + sm + bias[32 + dt * 16 + j];
        u.gk = kc * kSplitInvScale + km;
        return u;
      };
      // R(dt):  D [unit][dim] += A [unit][row] B [row][dim] over the tile's 16 rows, three partial products (yh th, yl th,
      // yh tl) into one accumulator.  NOT as K = 16 products: v_mfma_f32_16x16x16_f16 occupies the matrix pipe four times
      // as long as v_mfma_f32_16x16x32_f16 on gfx950 (48 of them per wave and pair were 3/4 of this kernel's arithmetic
      // time).  A K = 32 product sums its two K halves, so with A = [yh | yl] (one ds_read2: the two parts of a unit tile
      // side by side) B = [th | th] gives yh th + yl th in ONE instruction and B = [tl | 0] adds yh tl: two K = 32
      // MFMAs per (unit tile, net) instead of three K = 16 ones, a sixth of the pipe time.
      auto row_sums = [&](int dt, const u32x2& th, const u32x2& tl, const u32x2& sh, const u32x2& sl_) {
        const u32x2 zero2 = u32x2{0u, 0u};
        const f16x8 t_hh = pair_operand(th, th), t_l0 = pair_operand(tl, zero2);
        const f16x8 s_hh = pair_operand(sh, sh), s_l0 = pair_operand(sl_, zero2);
#pragma unroll
        for (int m = 0; m < YT; ++m) {
          const f16x8 y_hl = pair_operand(y_tr(m, 0), y_tr(m, 1));
          aWt[dt][m] = mfma_h(y_hl, t_hh, aWt[dt][m]);
          aWs[dt][m] = mfma_h(y_hl, s_hh, aWs[dt][m]);
          aWt[dt][m] = mfma_h(y_hl, t_l0, aWt[dt][m]);
          aWs[dt][m] = mfma_h(y_hl, s_l0, aWs[dt][m]);
        }
      };
      if (!(kTsAbl & 1)) {
        const Units u0 = units(0), u1 = units(1);
        f32x4 gt[2], gs[2];
        const f32x2 one2 = f32x2{1.f, 1.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float keep = 4 * q + r < in.n_live ? 1.f : 0.f;
          f32x2 m2;
          if (SEEDED)
            m2 = f32x2{(float)((in.mbits >> (2 * r)) & 1u), (float)((in.mbits >> (2 * r + 1)) & 1u)};
          else
            m2 = in.mm[r];
          const f32x2 t2 = f32x2{u0.t4[r], u1.t4[r]}, s2 = f32x2{u0.s4[r], u1.s4[r]}, k2 = f32x2{u0.gk[r], u1.gk[r]};
          const f32x2 nm = one2 - m2;
          const f32x2 G2 = in.GG[r] * (gx_scale * keep);
          const f32x2 sx = s2 * -1.44269504088896341f;
          const f32x2 ex = f32x2{__builtin_amdgcn_exp2f(sx[0]), __builtin_amdgcn_exp2f(sx[1])} + one2;
          const f32x2 gate_ = f32x2{__builtin_amdgcn_rcpf(ex[0]), __builtin_amdgcn_rcpf(ex[1])};
          const f32x2 omg = one2 - gate_;
          const f32x2 gt2 = G2 * omg;
          const f32x2 gs2 = (G2 * (nm * in.zz[r] - t2) * gate_ + nm * (gl4[r] * (gl_scale * keep))) * omg;
          in.zz[r] = (G2 * (nm * gate_ + m2) + m2 * k2) * inv_gscale;  // grad_z takes z's register
          gt[0][r] = gt2[0];
          gt[1][r] = gt2[1];
          gs[0][r] = gs2[0];
          gs[1][r] = gs2[1];
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          abt[dt] += (gt[dt][0] + gt[dt][1]) + (gt[dt][2] + gt[dt][3]);
          abs_[dt] += (gs[dt][0] + gs[dt][1]) + (gs[dt][2] + gs[dt][3]);
          u32x2 th, tl, sh, sl_;  // B operands of the sums over the tile's 16 rows
          split_plain(gt[dt], th, tl);
          split_plain(gs[dt], sh, sl_);
          row_sums(dt, th, tl, sh, sl_);
        }
      }
      if (kTsAbl & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) in.zz[r] = in.zz[r] * gl4[r] + in.GG[r] * (float)in.mbits;
      }
      float* ot = grad_z + in.tbase * dm;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * q + r < in.n_live && (!(kTsAbl & 2) || in.zz[r][0] == 1.2345e30f)) {
          const uint32_t off = lane_off + (uint32_t)r * (uint32_t)dm;
          if (!RAG) {
            *reinterpret_cast<f32x2*>(ot + off) = in.zz[r];
          } else if (vec2) {  // (dm even: in0 implies in1)
            if (in0) *reinterpret_cast<f32x2*>(ot + off) = in.zz[r];
          } else {
            if (in0) ot[off] = in.zz[r][0];
            if (in1) ot[off + 1] = in.zz[r][1];
          }
        }
      }
      return (in.n_live == 16 && (!RAG || vec2)) ? 4 : 0;
    };

    {
      // two register sets, alternating roles (no copies: a copy would wait for the loads it moves); the last pair
      // requests itself once more into the idle buffer rather than branching around the requests
      RowsIn ra, rb;
      load_rows(p0, ra);
      request_handover(p0, 0);
      int stores = 0;
      int64_t p = p0;
      while (true) {
        landed_barrier(stores);  // every wave's pieces of pair p are in LDS, pair p - 1 (the other buffer) is consumed
        {
          const int64_t pn = min(p + 1, p_end - 1);
          load_rows(pn, rb);
          request_handover(pn, 1);
        }
        stores = compute(ra, 0);
        if (++p >= p_end) break;
        landed_barrier(stores);
        {
          const int64_t pn = min(p + 1, p_end - 1);
          load_rows(pn, ra);
          request_handover(pn, 0);
        }
        stores = compute(rb, 1);
        if (++p >= p_end) break;
      }
    }
    if (!grad_flat) continue;
    // flush: as in the kernel above, per slab -- the workgroup's waves add their tiles up in LDS (the operand area is free
    // now) as [slab][tensor][dim of the slab][unit], then the group's CONTIGUOUS blocks of Wt / Ws go to grad_flat
    float* red = reinterpret_cast<float*>(ts_lds);
    __syncthreads();
    for (int i = threadIdx.x; i < kTsSlabs * T::RED_SLAB; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    if (slab_ok) {
      float* rs = red + sl * T::RED_SLAB;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int m = 0; m < YT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(rs + (2 * j + dt) * T::UP + 16 * m + 4 * q + r, aWt[dt][m][r]);
            atomicAdd(rs + 32 * T::UP + (2 * j + dt) * T::UP + 16 * m + 4 * q + r, aWs[dt][m][r]);
          }
        float vt = abt[dt], vs = abs_[dt];
        vt += __shfl_xor(vt, 16, 64);
        vt += __shfl_xor(vt, 32, 64);
        vs += __shfl_xor(vs, 16, 64);
        vs += __shfl_xor(vs, 32, 64);
        if (q == 0) {
          atomicAdd(rs + 2 * 32 * T::UP + 2 * j + dt, vt);
          atomicAdd(rs + 2 * 32 * T::UP + 32 + 2 * j + dt, vs);
        }
      }
    }
    __syncthreads();
    const int64_t bn = (int64_t)hn * dm, wt = bn + hn, btf = wt + (int64_t)dm * hn, ws = btf + dm,
                  bsf = ws + (int64_t)dm * hn;
    const int dim_g = 32 * kTsSlabs * sg;                       // first dim of the group
    const int n_dims = min(32 * kTsSlabs, dm - dim_g);          // dims of this group that exist
    // (two waves per slab added into the zeroed area: a + b in either order is the same number.  det_part: the item's
    // sums as plain stores into the row part's block, added up in a fixed order by det_reduce_async)
    float* const out = det_part ? det_part + (int64_t)part * rnvp_bwd_params(dm, hn) : grad_flat;
    auto emit = [&](int64_t at, float v) {
      if (det_part)
        out[at] = v;
      else
        atomicAdd(out + at, v);
    };
    for (int e = threadIdx.x; e < n_dims * hn; e += blockDim.x) {
      const int dl = e / hn, unit = e - dl * hn;
      const float* rs = red + (dl >> 5) * T::RED_SLAB + (dl & 31) * T::UP + unit;
      emit(wt + (int64_t)dim_g * hn + e, rs[0] * inv_gscale);
      emit(ws + (int64_t)dim_g * hn + e, rs[32 * T::UP] * inv_gscale);
    }
    if ((int)threadIdx.x < n_dims) {
      const float* rs = red + (threadIdx.x >> 5) * T::RED_SLAB + 2 * 32 * T::UP + (threadIdx.x & 31);
      emit(btf + dim_g + threadIdx.x, rs[0] * inv_gscale);
      emit(bsf + dim_g + threadIdx.x, rs[32] * inv_gscale);
    }
  }
}

// B-n: dWn [unit][dim] += sum over rows of g_y [row][unit] (m z) [row][dim] -- needs z, the mask and launch A's g_y only
// (no gate arithmetic, 8 accumulator tiles): its own launch, four waves per SIMD, so that B-ts keeps 16 accumulator
// tiles instead of 24 and fits two waves per SIMD.
template <int HN, bool SEEDED, bool RAG>
__global__ void __launch_bounds__(kBwdBWaves * 64, 4)
rnvp_bwd_n_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ grad_flat,
                  const uint32_t* __restrict__ side, const int32_t* __restrict__ flags,
                  const float* __restrict__ gscale_dev, int64_t rows, int dm, int hn, uint64_t seed, int n_slabs,
                  int row_parts, int vec2, float* __restrict__ det_part) {
  using S = RnvpSplitShape<HN>;
  using B = RnvpBwdShape<HN>;
  constexpr int YT = S::YT;
  __shared__ float red[16 * YT * 33];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const float inv_gscale = 1.f / gscale_dev[0];
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x2 zero2 = u32x2{0u, 0u};
  const int64_t n_tiles = (rows + 15) / 16, n_pairs = (n_tiles + 1) / 2;
  const int64_t per_part = (n_pairs + row_parts - 1) / row_parts;
  const SlabItems items(n_slabs, row_parts);
  for (int item = items.first; item < items.n_items; item += items.step) {
    const int slab = items.slab(item), part = items.part(item);
    const int dim0 = 32 * slab + 2 * j;
    const bool in0 = dim0 < dm, in1 = dim0 + 1 < dm;
    const uint32_t lane_off = (uint32_t)(4 * q) * (uint32_t)dm + (uint32_t)dim0;
    f32x4 aWn[2][YT];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int m = 0; m < YT; ++m) aWn[dt][m] = zero4;
    const int64_t p_end = min(n_pairs, (int64_t)(part + 1) * per_part);
    for (int64_t p = (int64_t)part * per_part + wave; p < p_end; p += kBwdBWaves) {
      if (flags[(p * 32) / kBwdGroupRows]) continue;
      const bool has1 = 2 * p + 1 < n_tiles;
      u32x2 kh[2][2], kl[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const bool has = tt == 0 || has1;
        const int64_t tbase = (has ? 2 * p + tt : 2 * p) * 16;
        const int n_live = has ? (int)min((int64_t)16, rows - tbase) : 0;
        const float* zt = z + tbase * dm;
        const float* mt = SEEDED ? nullptr : mask + tbase * dm;
        uint32_t mw[4];
        if (SEEDED) tile_mask_words(seed, tbase, n_live, slab, lane, q, mw);
        f32x4 k0, k1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool live = 4 * q + r < n_live;
          const uint32_t off = live ? lane_off + (uint32_t)r * (uint32_t)dm : (uint32_t)dim0;
          f32x2 zv = {0.f, 0.f}, mv = {0.f, 0.f};
          if (!RAG) {
            zv = *reinterpret_cast<const f32x2*>(zt + off);
            if (!SEEDED) mv = *reinterpret_cast<const f32x2*>(mt + off);
          } else if (vec2) {
            if (in0) {
              zv = *reinterpret_cast<const f32x2*>(zt + off);
              if (!SEEDED) mv = *reinterpret_cast<const f32x2*>(mt + off);
            }
          } else {
            if (in0) {
              zv[0] = zt[off];
              if (!SEEDED) mv[0] = mt[off];
            }
            if (in1) {
              zv[1] = zt[off + 1];
              if (!SEEDED) mv[1] = mt[off + 1];
            }
          }
          if (SEEDED) {
            const uint32_t w = mw[r] >> (2 * j);
            mv = f32x2{(float)(w & 1u), (float)((w >> 1) & 1u)};
          }
          const float keep = live ? 1.f : 0.f;  // (g_y of a row past the end is zero anyway; keep k finite)
          k0[r] = mv[0] * zv[0] * keep;
          k1[r] = mv[1] * zv[1] * keep;
        }
        split_plain(k0, kh[tt][0], kl[tt][0]);
        split_plain(k1, kh[tt][1], kl[tt][1]);
        if (!has) kh[tt][0] = kl[tt][0] = kh[tt][1] = kl[tt][1] = zero2;
      }
      const uint32_t* s0 = side + (2 * p) * B::TILE_WORDS + B::G_TR + lane * 2;
      const uint32_t* s1 = side + (2 * p + (has1 ? 1 : 0)) * B::TILE_WORDS + B::G_TR + lane * 2;
#pragma unroll
      for (int m = 0; m < YT; ++m) {
        const u32x2 g0h = *reinterpret_cast<const u32x2*>(s0 + (2 * m) * 128);
        const u32x2 g0l = *reinterpret_cast<const u32x2*>(s0 + (2 * m + 1) * 128);
        const u32x2 g1h = *reinterpret_cast<const u32x2*>(s1 + (2 * m) * 128);
        const u32x2 g1l = *reinterpret_cast<const u32x2*>(s1 + (2 * m + 1) * 128);
        const f16x8 gh8 = pair_operand(g0h, g1h), gl8 = pair_operand(g0l, g1l);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f16x8 kH = pair_operand(kh[0][dt], kh[1][dt]), kL = pair_operand(kl[0][dt], kl[1][dt]);
          aWn[dt][m] = mfma_h(gh8, kH, aWn[dt][m]);
          aWn[dt][m] = mfma_h(gh8, kL, aWn[dt][m]);
          aWn[dt][m] = mfma_h(gl8, kH, aWn[dt][m]);
        }
      }
    }
    // flush through LDS as in B-ts: [unit][dim of the slab], then 128 contiguous bytes of a Wn row per half wave
    if ((kBwdAbl & 32) && aWn[0][0][0] != 1.2345e30f) continue;
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * YT * 33; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    lds_wave_add<kBwdBWaves>(det_part != nullptr, wave, [&](auto op) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int m = 0; m < YT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) op(red + (16 * m + 4 * q + r) * 33 + 2 * j + dt, aWn[dt][m][r]);
    });
    __syncthreads();
    const int n_dims = min(32, dm - 32 * slab);
    float* const out = det_part ? det_part + (int64_t)part * rnvp_bwd_params(dm, hn) : grad_flat;
    for (int e = threadIdx.x; e < hn * 32; e += blockDim.x) {
      const int unit = e >> 5, dl = e & 31;
      if (dl < n_dims) {
        const float v = red[unit * 33 + dl] * inv_gscale;
        if (det_part)
          out[(int64_t)unit * dm + 32 * slab + dl] = v;
        else
          atomicAdd(out + (int64_t)unit * dm + 32 * slab + dl, v);
      }
    }
  }
}

// ================================================================================================ host
// index table of the backward-only image (2 entries per split word, then 1 per plain word; see build_split_index)
template <int HN>
static void build_bwd_index(int dm, int32_t* idx, int hn) {
  using S = RnvpSplitShape<HN>;
  using B = RnvpBwdShape<HN>;
  constexpr int YT = S::YT, NKS2 = S::NKS2;
  const int d16 = rnvp_padded_dim(dm), G = d16 / 16, n_slabs = (int)B::n_slabs(dm);
  const int64_t wn = 0, bn = wn + (int64_t)hn * dm, wt = bn + hn, bt = wt + (int64_t)dm * hn, ws = bt + dm,
                bs = ws + (int64_t)dm * hn;
  const int64_t n_split = B::split_words(dm, d16), n_entries = 2 * n_split + B::plain_words(dm);
  for (int64_t i = 0; i < n_entries; ++i) idx[i] = -1;
  auto put = [&](int64_t base_words, int op, int lane, int e, int64_t src) {
    for (int part = 0; part < 2; ++part)
      idx[2 * base_words + (((int64_t)(2 * op + part) * 64 + lane) * 4 + (e >> 1)) * 2 + (e & 1)] =
          (int32_t)src | (part ? kSplitLoBit : 0);
  };
  // A3: tile m, unit tile u: A[unit 16 u + i][slot 8 kq + e]; slots 0..3 of a quad = t dims 16 m + 4 kq + e, 4..7 = s dims
  for (int m = 0; m < G; ++m)
    for (int u = 0; u < YT; ++u)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4, unit = 16 * u + i;
        if (unit >= hn) continue;
        for (int e = 0; e < 8; ++e) {
          const int dim = 16 * m + 4 * kq + (e & 3);
          if (dim < dm) put(0, m * YT + u, lane, e, ((e >> 2) ? ws : wt) + (int64_t)dim * hn + unit);
        }
      }
  // slot of a K = 32 step over units: slot 8 kq + e <-> unit 16 (2 ks + (e >> 2)) + 4 kq + (e & 3)
  auto unit_of = [&](int ks, int kq, int e) { return 16 * (2 * ks + (e >> 2)) + 4 * kq + (e & 3); };
  const int64_t b2 = B::a3_words(d16), b4 = b2 + B::b2_words(dm);
  for (int sl = 0; sl < n_slabs; ++sl)
    for (int dt = 0; dt < 2; ++dt)
      for (int lane = 0; lane < 64; ++lane) {
        const int c = lane & 15, kq = lane >> 4, dim = 32 * sl + 2 * c + dt;
        if (dim >= dm) continue;
        for (int ks = 0; ks < NKS2; ++ks)
          for (int e = 0; e < 8; ++e) {
            const int unit = unit_of(ks, kq, e);
            if (unit >= hn || 2 * ks + (e >> 2) >= YT) continue;
            for (int net = 0; net < 2; ++net)  // B[unit][dim] = W[dim][unit]
              put(b2 + (int64_t)sl * B::B2_SLAB_WORDS, (dt * 2 + net) * NKS2 + ks, lane, e,
                  (net ? ws : wt) + (int64_t)dim * hn + unit);
            put(b4 + (int64_t)sl * B::B4_SLAB_WORDS, dt * NKS2 + ks, lane, e, wn + (int64_t)unit * dm + dim);
          }
      }
  int32_t* pl = idx + 2 * n_split;
  for (int sl = 0; sl < n_slabs; ++sl)
    for (int dt = 0; dt < 2; ++dt)
      for (int c = 0; c < 16; ++c) {
        const int dim = 32 * sl + 2 * c + dt;
        const bool real = dim < dm;
        pl[(int64_t)sl * B::B_SLAB_PLAIN + dt * 16 + c] = real ? (int32_t)(bt + dim) : -1;
        pl[(int64_t)sl * B::B_SLAB_PLAIN + 32 + dt * 16 + c] = real ? (int32_t)(bs + dim) : kPackBigBias;
      }
}

// workspace: [list: count, then up to n_groups flagged groups][flags: n_groups], padded to 256 B; then the hand-over
static int64_t bwd_header_bytes(int64_t rows) {
  const int64_t n_groups = (rows + kBwdGroupRows - 1) / kBwdGroupRows;
  return (((1 + 2 * n_groups) * 4 + 255) & ~(int64_t)255);
}
template <int HN>
static int64_t bwd_tiles_end(int64_t rows) {
  return (bwd_header_bytes(rows) + ((rows + 15) / 16) * RnvpBwdShape<HN>::TILE_WORDS * 4 + 255) & ~(int64_t)255;
}
template <int HN>
static int64_t bwd_workspace_bytes(int64_t rows, int dm, int hn) {
  const int64_t det = deterministic() ? ((int64_t)kDetBlocksA * 64 + kDetParts * rnvp_bwd_params(dm, hn)) * 4 : 0;
  return bwd_tiles_end<HN>(rows) + det;
}

// phases: bit 0 launch A, bit 1 B-ts, bit 2 B-n (bit 3, the fp32 fix-up, is the caller's)
template <int HN, bool SEEDED, bool RAG>
static int launch_bwd(const float* z, const float* mask, uint64_t seed, const float* gx, const float* gld, float* grad_z,
                      float* grad_flat, const uint32_t* simage, const uint32_t* bimage, const float* gscale, void* work,
                      int64_t rows, int dm, int hn, int vec4, int vec2, int phases, const float* y_in, hipStream_t stream) {
  using B = RnvpBwdShape<HN>;
  const int d16 = rnvp_padded_dim(dm);
  const int64_t n_groups = (rows + kBwdGroupRows - 1) / kBwdGroupRows;
  int32_t* list = static_cast<int32_t*>(work);
  int32_t* flags = list + 1 + n_groups;
  uint32_t* side = reinterpret_cast<uint32_t*>(static_cast<char*>(work) + bwd_header_bytes(rows));
  static DeviceMemo memo_a;
  constexpr int a_lds_bytes = kBwdRing * B::CHUNK_WORDS * 4;
  void (*const a_kernel)(const float*, const float*, const float*, const float*, const uint32_t*, const uint32_t*, uint32_t*,
                         int32_t*, int32_t*, const float*, float*, int64_t, int, int, int, uint64_t, int, int64_t,
                         const float*, float*) = rnvp_bwd_a_kernel<HN, SEEDED, RAG>;
  const int resident_a = memo_a.get([a_kernel](int dev) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            a_lds_bytes) != hipSuccess)
      return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, a_kernel, kRnvpWaves * 64, a_lds_bytes) != hipSuccess || per_cu < 1)
      per_cu = 1;
    return per_cu * device_cus(dev);
  });
  if (resident_a < 0) return MNF_ERR_UNSUPPORTED;
  int64_t blocks_a = n_groups < resident_a ? n_groups : resident_a;
  const bool det = deterministic() && grad_flat != nullptr;
  float* const det_bn = det ? reinterpret_cast<float*>(static_cast<char*>(work) + bwd_tiles_end<HN>(rows)) : nullptr;
  float* const det_part = det ? det_bn + (int64_t)kDetBlocksA * 64 : nullptr;
  const int64_t n_params = rnvp_bwd_params(dm, hn);
  const int max_l = det ? kDetParts / 8 : 32;
  if (det && blocks_a > kDetBlocksA) blocks_a = kDetBlocksA;
  const int64_t tail = B::split_words(dm, d16) + B::plain_words(dm);
  if (phases & 1) {
  if (int rc = zero_word_async(list, stream)) return rc;
  tag_kernel("rnvp_bwd_mfma");
  hipLaunchKernelGGL((rnvp_bwd_a_kernel<HN, SEEDED, RAG>), dim3((unsigned)blocks_a), dim3(kRnvpWaves * 64), a_lds_bytes, stream, z,
                     mask, gx, gld, simage, bimage, side, flags, list, gscale, grad_flat, rows, d16, dm, hn, seed, vec4, tail,
                     y_in, det_bn);
  if (int rc = check_launch()) return rc;
  if (det)
    if (int rc = det_reduce_async(det_bn, (int)blocks_a, 64, hn, grad_flat + (int64_t)hn * dm, stream)) return rc;
  }
  // B: (row part, slab) work items over a persistent grid.  Row parts come in multiples of 8 (one XCD each, see
  // BwdItems); their number per XCD is chosen so that the XCD's items fill whole rounds of its resident workgroups.
  const int n_slabs = (int)B::n_slabs(dm);
  const int64_t n_pairs = ((rows + 15) / 16 + 1) / 2;
  auto plan = [&](int resident, int& row_parts, int& grid) {
    plan_slab_launch(n_pairs, kBwdBWaves, n_slabs, resident, row_parts, grid, max_l);
  };
  static DeviceMemo memo_n;
  int row_parts, grid;
  if (phases & 2) {
    // the shared-hand-over kernel: four slabs per workgroup, one workgroup per CU (145 KB of LDS at 64 units)
    using T = RnvpTsShape<HN>;
    static DeviceMemo memo_s;
    void (*const ts_kernel)(const float*, const float*, const float*, const float*, float*, float*, const uint32_t*,
                            const uint32_t*, const int32_t*, const float*, int64_t, int, int, int, uint64_t, int, int, int,
                            float*) = rnvp_bwd_ts_shared_kernel<HN, SEEDED, RAG>;  // (named out here: a kernel first named inside a lambda gets no host stub)
    const int resident_s = memo_s.get([ts_kernel](int dev) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(ts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS_WORDS * 4) != hipSuccess)
        return -1;
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ts_kernel, kTsWaves * 64, T::LDS_WORDS * 4) != hipSuccess ||
          per_cu < 1)
        per_cu = 1;
      return per_cu * device_cus(dev);
    });
    if (resident_s < 0) return MNF_ERR_LAUNCH;  // (the LDS request was refused: not an MI355X)
    const int n_slab_groups = (n_slabs + kTsSlabs - 1) / kTsSlabs;
    plan_slab_launch(n_pairs, 1, n_slab_groups, resident_s, row_parts, grid, max_l);
    if (det)  // (a row part without pairs writes nothing)
      if (int rc = zero_floats_async(det_part, row_parts * n_params, stream)) return rc;
    hipLaunchKernelGGL((rnvp_bwd_ts_shared_kernel<HN, SEEDED, RAG>), dim3((unsigned)grid), dim3(kTsWaves * 64),
                       T::LDS_WORDS * 4, stream, z, mask, gx, gld, grad_z, grad_flat, bimage, side, flags, gscale, rows,
                       dm, d16, hn, seed, n_slabs, row_parts, vec2, det_part);
    if (int rc = check_launch()) return rc;
    if (det) {
      const int64_t wt = (int64_t)hn * dm + hn;
      if (int rc = det_reduce_async(det_part + wt, row_parts, n_params, n_params - wt, grad_flat + wt, stream)) return rc;
    }
  }
  if (!grad_flat || !(phases & 4)) return MNF_OK;
  const int resident_n = memo_n.get(
      [](int dev) { return resident_by_occupancy(rnvp_bwd_n_kernel<HN, SEEDED, RAG>, kBwdBWaves * 64, dev, 4); });
  plan(resident_n, row_parts, grid);
  if (det)
    if (int rc = zero_floats_async(det_part, row_parts * n_params, stream)) return rc;
  hipLaunchKernelGGL((rnvp_bwd_n_kernel<HN, SEEDED, RAG>), dim3((unsigned)grid), dim3(kBwdBWaves * 64), 0, stream, z,
                     mask, grad_flat, side, flags, gscale, rows, dm, hn, seed, n_slabs, row_parts, vec2, det_part);
  if (int rc = check_launch()) return rc;
  return det ? det_reduce_async(det_part, row_parts, n_params, (int64_t)hn * dm, grad_flat, stream) : MNF_OK;
}

}  // namespace mnf

extern "C" {

using namespace mnf;

int64_t mnf_rnvp_bwd_mfma_workspace_bytes(int64_t rows, int dim, int n_hidden, const int* hidden) {
  if (rows < 0 || !rnvp_shape_ok(dim, n_hidden, hidden)) return 0;
#define X(HN) if (rnvp_padded_hidden(n_hidden, hidden) == HN) return bwd_workspace_bytes<HN>(rows < 1 ? 1 : rows, dim, hidden[0]);
  MNF_RNVP_HIDDEN(X)
#undef X
  return 0;
}

int mnf_rnvp_bwd_mfma_layout(int dim, int n_hidden, const int* hidden, int64_t* n_split_words, int64_t* n_plain_words) {
  if (!n_split_words || !n_plain_words) return MNF_ERR_INVALID_ARG;
  if (!rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
#define X(HN)                                                                       \
  if (rnvp_padded_hidden(n_hidden, hidden) == HN) {                                 \
    *n_split_words = RnvpBwdShape<HN>::split_words(dim, rnvp_padded_dim(dim));      \
    *n_plain_words = RnvpBwdShape<HN>::plain_words(dim);                            \
    return MNF_OK;                                                                  \
  }
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_rnvp_bwd_mfma_index(int dim, int n_hidden, const int* hidden, int32_t* idx_host) {
  if (!idx_host) return MNF_ERR_INVALID_ARG;
  if (!rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
#define X(HN)                                         \
  if (rnvp_padded_hidden(n_hidden, hidden) == HN) {   \
    build_bwd_index<HN>(dim, idx_host, hidden[0]);    \
    return MNF_OK;                                    \
  }
  MNF_RNVP_HIDDEN(X)
#undef X
  return MNF_ERR_UNSUPPORTED;
}

int mnf_rnvp_bwd_mfma_phases(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                             float* grad_z, float* grad_flat, const float* flat, const void* split_image,
                             const void* bwd_image, const float* grad_scale_dev, void* workspace, int64_t workspace_bytes,
                             int64_t rows, int dim, int n_hidden, const int* hidden, int phases, const float* y,
                             void* stream) {
  if (!z || !grad_z || !flat || !split_image || !bwd_image || !grad_scale_dev || !workspace || rows < 0 || dim < 1 ||
      n_hidden < 1 || !hidden_ok(n_hidden, hidden))
    return MNF_ERR_INVALID_ARG;
  if (!rnvp_shape_ok(dim, n_hidden, hidden)) return MNF_ERR_UNSUPPORTED;
  if (rows == 0) return MNF_OK;
  if (workspace_bytes < mnf_rnvp_bwd_mfma_workspace_bytes(rows, dim, n_hidden, hidden)) return MNF_ERR_INVALID_ARG;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(mask) |
                         reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(grad_z);
  if ((reinterpret_cast<uintptr_t>(split_image) | reinterpret_cast<uintptr_t>(bwd_image) |
       reinterpret_cast<uintptr_t>(workspace)) & 15)
    return MNF_ERR_UNSUPPORTED;
  const int d16 = rnvp_padded_dim(dim);
  const bool ragged = d16 != dim || (dim & 31) != 0 || (ptrs & 15) != 0;
  const int vec4 = (ptrs & 15) == 0 && (dim & 3) == 0, vec2 = (ptrs & 7) == 0 && (dim & 1) == 0;
  const int hn_pad = rnvp_padded_hidden(n_hidden, hidden);
  const uint32_t* si = static_cast<const uint32_t*>(split_image);
  const uint32_t* bi = static_cast<const uint32_t*>(bwd_image);
  const float* gs = grad_scale_dev;
  hipStream_t st = (hipStream_t)stream;
  const float* y_in = y;
  int rc = MNF_ERR_UNSUPPORTED;
#define X(HN)                                                                                                            \
  if (hn_pad == HN)                                                                                                      \
    rc = mask ? (ragged ? launch_bwd<HN, false, true>(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, si, bi, gs,     \
                                                      workspace, rows, dim, hidden[0], vec4, vec2, phases, y_in, st)                   \
                        : launch_bwd<HN, false, false>(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, si, bi, gs,    \
                                                       workspace, rows, dim, hidden[0], vec4, vec2, phases, y_in, st))                 \
              : (ragged ? launch_bwd<HN, true, true>(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, si, bi, gs,      \
                                                     workspace, rows, dim, hidden[0], vec4, vec2, phases, y_in, st)                    \
                        : launch_bwd<HN, true, false>(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, si, bi, gs,     \
                                                      workspace, rows, dim, hidden[0], vec4, vec2, phases, y_in, st));
  MNF_RNVP_HIDDEN(X)
#undef X
  if (rc != MNF_OK || !(phases & 8)) return rc;
  // groups outside the split range: the generic fp32 kernel, on the flagged groups only
  return rnvp_bwd_generic_launch(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, flat, rows, dim, n_hidden, hidden,
                                 static_cast<const int32_t*>(workspace), kBwdGroupRows, st);  // (the list heads the workspace)
}

int mnf_rnvp_bwd_mfma(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                      float* grad_z, float* grad_flat, const float* flat, const void* split_image, const void* bwd_image,
                      const float* grad_scale_dev, void* workspace, int64_t workspace_bytes, int64_t rows, int dim,
                      int n_hidden, const int* hidden, void* stream) {
  return mnf_rnvp_bwd_mfma_phases(z, mask, seed, grad_x, grad_ld, grad_z, grad_flat, flat, split_image, bwd_image,
                                  grad_scale_dev, workspace, workspace_bytes, rows, dim, n_hidden, hidden, 15, nullptr, stream);
}

}  // extern "C"
