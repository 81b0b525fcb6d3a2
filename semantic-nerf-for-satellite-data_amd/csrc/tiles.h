// Workgroup -> tile maps shared by every tiled kernel (XCD-aware: speed only, never correctness).
#pragma once
#include "common.h"

namespace snerf {

// Workgroup -> tile map: blocks b and b+8 share an XCD (round-robin dispatch), and the J-tiles of
// one I-tile re-read the same A rows, so give each XCD group runs of consecutive J-tiles of the
// same I-tile: those re-reads then hit that XCD's L2 instead of HBM. Speed only, never correctness.
// `rev`: every group walks its chunk backwards -- a launch that consumes the tensor the previous launch has just written
// (layer n + 1 after layer n, dX of layer n after dX of layer n + 1) then starts with the rows written last, which the
// memory-side cache (256 MB) still holds.
__device__ __forceinline__ void tile_of_block(int b, int tiles_i, int tiles_j, int& ti, int& tj, bool rev = false) {
  const int n = tiles_i * tiles_j;
  const int xcd = b & 7, q = b >> 3;
  const int per = n >> 3;  // tiles per XCD group (exact part)
  if (b < (per << 3)) {
    const int lin = xcd * per + (rev ? per - 1 - q : q);  // contiguous chunk of the (ti-major) tile order per XCD group
    ti = lin / tiles_j;
    tj = lin - ti * tiles_j;
  } else {  // remainder tiles (n % 8): identity order
    ti = b / tiles_j;
    tj = b - ti * tiles_j;
  }
}

// Split-K launches (dW): grid = (tiles, 1, splits), dispatched x-fastest, so block f = x + tiles*z lands on XCD f % 8.
// All tiles of one split read the same rows of dZ and X; remap so that one XCD group runs ALL tiles of a split
// back to back (each operand block is then fetched from HBM once per split and shared through that L2).
__device__ __forceinline__ void split_tile_of_block(int x, int z, int tiles, int splits, int& tile, int& split) {
  const int s8 = splits & ~7;
  tile = x; split = z;
  if (z < s8) {
    const int f = x + tiles * z;
    const int xcd = f & 7, idx = f >> 3;
    split = (idx / tiles) * 8 + xcd;
    tile = idx - (idx / tiles) * tiles;
  }
}

}  // namespace snerf
