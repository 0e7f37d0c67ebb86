// svr_bin.h — the tile-range and pair-emission pieces of binning, shared by the setup kernel (which
// emits the (bin, record) pairs of the triangles it sets up) and the binning kernels (k_bin.hip).
#pragma once
#include "svr_device.h"

namespace svr {

constexpr int SMALL_MAX_TILES = 16;  // above this a triangle's tiles are walked by a whole wave

struct EdgeSet {
  double A0, A1, A2, B0, B1, B2, C0, C1, C2;
};

// can any pixel centre of [x0,x1]x[y0,y1] be inside? (max of each edge function over the box)
__device__ __forceinline__ bool box_overlaps(const EdgeSet& e, int x0, int y0, int x1, int y1) {
  double fx0 = (double)x0, fx1 = (double)x1, fy0 = (double)y0, fy1 = (double)y1;
  double m0 = fma(e.A0, e.A0 >= 0.0 ? fx1 : fx0, fma(e.B0, e.B0 >= 0.0 ? fy1 : fy0, e.C0));
  double m1 = fma(e.A1, e.A1 >= 0.0 ? fx1 : fx0, fma(e.B1, e.B1 >= 0.0 ? fy1 : fy0, e.C1));
  double m2 = fma(e.A2, e.A2 >= 0.0 ? fx1 : fx0, fma(e.B2, e.B2 >= 0.0 ? fy1 : fy0, e.C2));
  return m0 >= 0.0 && m1 >= 0.0 && m2 >= 0.0;
}

// pixel bbox -> tile range of the scissor-anchored tile grid
struct TileRange {
  int tx0, ty0, ntx, nt;
};
__device__ __forceinline__ TileRange tile_range(const FrameParams& P, int minx, int miny, int maxx, int maxy, bool valid) {
  TileRange t;
  t.tx0 = (minx - (int)P.sx) >> TILE_SHIFT;
  t.ty0 = (miny - (int)P.sy) >> TILE_SHIFT;
  t.ntx = ((maxx - (int)P.sx) >> TILE_SHIFT) - t.tx0 + 1;
  int nty = ((maxy - (int)P.sy) >> TILE_SHIFT) - t.ty0 + 1;
  t.nt = valid ? t.ntx * nty : 0;
  return t;
}

// Small triangles (<= 16 tiles), called by every lane of the setup kernel's wave: each lane tests its
// own tiles (registers only), a wave prefix sum and ONE atomic reserve the wave's span of the pair
// list, and the lanes store their (bin, record) pairs there.  Nothing here waits on memory more than
// once, where a loop of per-tile atomics was a chain of round trips as long as the largest triangle.
__device__ __forceinline__ void emit_small_pairs(const FrameParams& P, bool small, const TileRange& tr, int minx, int miny,
                                                 int maxx, int maxy, uint32_t binbase, const EdgeSet& e, uint32_t rec) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mask = 0;
  if (small) {
    if (tr.nt <= 4) {
      mask = (1u << tr.nt) - 1u;
    } else {
      for (int j = 0; j < tr.nt; j++) {
        int ty = tr.ty0 + j / tr.ntx, tx = tr.tx0 + j % tr.ntx;
        int x0 = max(minx, (int)P.sx + tx * TILE), x1 = min(maxx, (int)P.sx + tx * TILE + TILE - 1);
        int y0 = max(miny, (int)P.sy + ty * TILE), y1 = min(maxy, (int)P.sy + ty * TILE + TILE - 1);
        if (box_overlaps(e, x0, y0, x1, y1)) mask |= 1u << j;
      }
    }
  }
  uint32_t cnt = (uint32_t)__popc(mask), inc = cnt;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t u = __shfl_up(inc, off);
    if ((int)lane >= off) inc += u;
  }
  uint32_t total = __shfl(inc, 63), base = 0;
  if (total == 0) return;  // wave-uniform
  if (lane == 0) base = atomicAdd(&P.counters->n_pairs, total);
  base = __shfl(base, 0);
  if (base + total > P.bin_cap) {  // the pass is void; the host grows the lists from n_pairs and replays
    if (lane == 0) atomicOr(&P.counters->overflow, 4u);
    return;
  }
  uint32_t pos = base + inc - cnt;
  while (mask) {
    int j = __ffs((int)mask) - 1;
    mask &= mask - 1u;
    uint32_t ty = (uint32_t)(tr.ty0 + j / tr.ntx), tx = (uint32_t)(tr.tx0 + j % tr.ntx);
    P.pairs[pos++] = make_uint2(binbase + ty * P.tiles_x + tx, rec);
  }
}

}  // namespace svr
