// Tiling of the AffineHalfFlow gradient kernels (mnf_ahf_bwd_mfma.hip, mnf_ahf_bwd_split.hip): operand counts of the
// fp32 image, the order of the weight-gradient tiles and the flush tables both kernels use.
#pragma once
#include "mnf_ahf_shape.h"

namespace mnf {

constexpr int kBwdWaves = 4;                    // waves per workgroup of the split kernel and of the fp32 kernel at d <= 64
constexpr int kTilePitch = 20;                     // floats per row of an LDS scratch tile
constexpr int kTileFloats = 16 * kTilePitch;

template <int H, int HID>
struct BwdShape {
  static_assert(H % 16 == 0 && HID % 4 == 0, "unsupported shape");
  static constexpr int G = H / 16;
  static constexpr int NT = (2 * HID + 15) / 16;
  static constexpr int tile_nets(int m) {  // bit 0 = s, bit 1 = t
    int nets = 0;
    for (int i = 0; i < 16; ++i)
      if (16 * m + i < 2 * HID) nets |= 1 << ((16 * m + i) / HID);
    return nets;
  }
  static constexpr bool needs(int m, int mt) { return (tile_nets(m) & tile_nets(mt)) != 0; }
  static constexpr int pairs() {
    int n = 0;
    for (int m = 0; m < NT; ++m)
      for (int mt = 0; mt < NT; ++mt) n += needs(m, mt) ? 1 : 0;
    return n;
  }
  static constexpr int net_tiles() {  // (hidden tile, net) incidences
    int n = 0;
    for (int m = 0; m < NT; ++m) n += (tile_nets(m) & 1) + ((tile_nets(m) >> 1) & 1);
    return n;
  }
  static constexpr int PAIRS = pairs(), NET_TILES = net_tiles();
  // fp32 kernel at d = 256: the layer-1 and output-layer weights sit in the image ONCE; the backward products read
  // the forward blocks transposed (their lane slots are permuted so that both reads are bank-conflict free)
  static constexpr bool SHARED_T = H >= 128;
  // MFMA operand counts (one op = 64 floats), in image order
  static constexpr int N_F1 = NT * G * 4, N_FH = PAIRS * 4, N_F4 = NET_TILES * G * 4;
  static constexpr int N_B4 = SHARED_T ? 0 : NET_TILES * G * 4, N_BH = PAIRS * 4, N_B1 = SHARED_T ? 0 : G * NT * 4;
  static constexpr int F4_GROUP0 = (N_F1 + 2 * N_FH) / 4;  // first 4-op group of the output-layer block
  static constexpr int f4_rank(int nn, int g, int mt) {    // position of block (net, dim group, hidden tile) in it
    int n = 0;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < G; ++b)
        for (int c = 0; c < NT; ++c) {
          if (a == nn && b == g && c == mt) return n;
          if ((tile_nets(c) >> a) & 1) ++n;
        }
    return -1;
  }
  static constexpr int N_OPS = N_F1 + 2 * N_FH + N_F4 + N_B4 + 2 * N_BH + N_B1;
  static constexpr int A_FLOATS = ((N_OPS + 3) / 4) * 256;
  static constexpr int BIAS_TILES = 3 * NT + 2 * G;
  static constexpr int IMAGE_FLOATS = A_FLOATS + BIAS_TILES * 16;
  // weight-gradient tiles: layer 1 (NT x G), hidden x 2 (PAIRS each), output (NET_TILES x G)
  static constexpr int DW_TILES = NT * G + 2 * PAIRS + NET_TILES * G;
  static constexpr int DB_TILES = 3 * NT + 2 * G;
  // LDS scratch tiles per wave (fp32 kernel): h1..h3 (3 NT), deltas of the layer in flight (max(NT, G): the output
  // layer's go through one net at a time; layer 1's input x0 is read back from global memory, rows on K)
  static constexpr int D_TILES = NT > G ? NT : (G > 4 ? 4 : G);  // (at most 4 output-delta tiles at a time)
  static constexpr int SCRATCH_TILES = 3 * NT + D_TILES;
  // index table: image gather, then dW flush ([tile][lane][reg]), then db flush ([tile][unit])
  static constexpr int INDEX_INTS = IMAGE_FLOATS + DW_TILES * 256 + DB_TILES * 16;
  // fp32 kernel: as many waves (<= kBwdWaves) as find room for their scratch tiles beside the operand images in the
  // CU's 160 KB of LDS (4 everywhere today: d = 128 keeps 152 KB busy)
  static constexpr int waves_that_fit() {
    int w = kBwdWaves;
    while (w > 1 && (IMAGE_FLOATS + w * SCRATCH_TILES * kTileFloats) * 4 > 160 * 1024) --w;
    return w;
  }
  static constexpr int WAVES = waves_that_fit();
  static constexpr int LDS_FLOATS = IMAGE_FLOATS + WAVES * SCRATCH_TILES * kTileFloats;
};

}  // namespace mnf
