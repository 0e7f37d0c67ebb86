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
  int l1;
  local_tile_rows(P, miny, maxy, t.ty0, l1);  // the pass's own tile rows (all of them unless the rows are interleaved)
  t.ntx = ((maxx - (int)P.sx) >> TILE_SHIFT) - t.tx0 + 1;
  int nty = l1 - t.ty0 + 1;
  t.nt = (valid && nty > 0) ? t.ntx * nty : 0;
  return t;
}

// Small triangles (<= 16 tiles), called by every lane of the setup kernel's WORKGROUP (barriers inside): each lane tests its
// own tiles (registers only), a wave prefix sum and ONE atomic reserve the wave's span of the pair
// list, and the lanes store their (bin, record) pairs there.  Nothing here waits on memory more than
// once, where a loop of per-tile atomics was a chain of round trips as long as the largest triangle.
__device__ __forceinline__ void emit_small_pairs(const FrameParams& P, bool small, const TileRange& tr, int minx, int miny,
                                                 int maxx, int maxy, uint32_t binbase, const EdgeSet& e, uint32_t rec,
                                                 uint32_t* s_tot) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mask = 0;
  if (small) {
    if (tr.nt <= 4) {
      mask = (1u << tr.nt) - 1u;
    } else {
      int tx = tr.tx0, ty = tr.ty0;  // running tile coordinates: no integer division per tile
      for (int j = 0; j < tr.nt; j++) {
        int x0 = max(minx, (int)P.sx + tx * TILE), x1 = min(maxx, (int)P.sx + tx * TILE + TILE - 1);
        int y0 = max(miny, tile_row_y(P, ty)), y1 = min(maxy, tile_row_y(P, ty) + TILE - 1);
        if (box_overlaps(e, x0, y0, x1, y1)) mask |= 1u << j;
        if (++tx == tr.tx0 + tr.ntx) {
          tx = tr.tx0;
          ty++;
        }
      }
    }
  }
  uint32_t cnt = (uint32_t)__popc(mask), inc = cnt;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t u = __shfl_up(inc, off);
    if ((int)lane >= off) inc += u;
  }
  // one atomic per WORKGROUP reserves the span (s_tot: 5 words of LDS handed in by the kernel): the
  // counter is one address for the whole grid, and same-address device atomics retire one at a time
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t total = __shfl(inc, 63), base = 0;
  if (lane == 63) s_tot[wave] = total;
  __syncthreads();
  uint32_t before = 0, block_total = 0;
  for (uint32_t w = 0; w < 4; w++) {
    uint32_t t = s_tot[w];
    if (w < wave) before += t;
    block_total += t;
  }
  if (block_total == 0) return;  // block-uniform
  if (threadIdx.x == 0) s_tot[4] = atomicAdd(&P.counters->n_pairs, block_total);
  __syncthreads();
  base = s_tot[4];
  if (base + block_total > P.bin_cap) {  // the pass is void; the host grows the lists from n_pairs and replays
    if (threadIdx.x == 0) atomicOr(&P.counters->overflow, 4u);
    return;
  }
  base += before;
  uint32_t pos = base + inc - cnt;
  uint32_t row_bin = binbase + (uint32_t)tr.ty0 * P.tiles_x + (uint32_t)tr.tx0;
  int tx = 0;
  for (; mask; mask >>= 1) {
    if (mask & 1u) P.pairs[pos++] = make_uint2(row_bin + (uint32_t)tx, rec);
    if (++tx == tr.ntx) {
      tx = 0;
      row_bin += P.tiles_x;
    }
  }
}

}  // namespace svr
