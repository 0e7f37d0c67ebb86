// k_bin.hip — sort-middle binning of set-up triangles into 32x32-pixel screen tiles.
//
// The rasteriser hardware behind vkCmdDrawIndexed walks each triangle's pixels itself; here the
// frame is cut into tiles (one workgroup each, k_tile.hip) and every triangle is appended to the
// bin of every tile it can touch.  count -> exclusive scan -> fill, so bins are contiguous spans of
// one buffer and no per-tile capacity exists.  Bin order is arbitrary (atomics): the tile kernel
// is order-independent (depth+sequence-key resolve for opaque, key-ordered peeling for blended).
// Two bin sets share the arrays: [0,n_tiles) opaque, [n_tiles,2*n_tiles) transparent.
//
// One lane per record; triangles touching more than 16 tiles are handed to the whole wave, which
// walks their tile range 64 tiles at a time.  Tiles inside the bbox that no edge function can
// reach are skipped (conservative corner test), identically in the count and fill passes.
#include "svr_launch.h"

namespace svr {

constexpr int SMALL_MAX_TILES = 16;

struct EdgeSet {
  double A0, A1, A2, B0, B1, B2, C0, C1, C2;
};

__device__ __forceinline__ EdgeSet load_edges(const TriRec* rec) {
  EdgeSet e;
  const double2* p = reinterpret_cast<const double2*>(rec);
  double2 a = p[2], b = p[3], c = p[4], d = p[5];
  e.A0 = a.x; e.A1 = a.y; e.A2 = b.x; e.B0 = b.y; e.B1 = c.x; e.B2 = c.y; e.C0 = d.x; e.C1 = d.y;
  e.C2 = rec->C[2];
  return e;
}

// can any pixel centre of [x0,x1]x[y0,y1] be inside? (max of each edge function over the box)
__device__ __forceinline__ bool box_overlaps(const EdgeSet& e, int x0, int y0, int x1, int y1) {
  double fx0 = (double)x0, fx1 = (double)x1, fy0 = (double)y0, fy1 = (double)y1;
  double m0 = fma(e.A0, e.A0 >= 0.0 ? fx1 : fx0, fma(e.B0, e.B0 >= 0.0 ? fy1 : fy0, e.C0));
  double m1 = fma(e.A1, e.A1 >= 0.0 ? fx1 : fx0, fma(e.B1, e.B1 >= 0.0 ? fy1 : fy0, e.C1));
  double m2 = fma(e.A2, e.A2 >= 0.0 ? fx1 : fx0, fma(e.B2, e.B2 >= 0.0 ? fy1 : fy0, e.C2));
  return m0 >= 0.0 && m1 >= 0.0 && m2 >= 0.0;
}

template <bool FILL>
__device__ __forceinline__ void emit(const FrameParams& P, uint32_t bin, uint32_t rec) {
  if (!FILL) {
    atomicAdd(&P.tile_count[bin], 1u);
  } else {
    uint32_t slot = atomicAdd(&P.tile_cursor[bin], 1u);
    uint32_t pos = P.tile_offset[bin] + slot;
    if (pos < P.bin_cap) P.bins[pos] = rec;
  }
}

template <bool FILL>
__global__ __launch_bounds__(256) void bin_kernel(FrameParams P) {
  uint32_t ovf = P.counters->overflow;
  if (ovf & 3u) return;            // geometry overflowed: the pass is void, the host retries
  if (FILL && ovf) return;
  uint32_t n_rec = P.n_tris + min(P.counters->n_extra, P.extra_cap);
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = r < n_rec;
  int minx = 1, miny = 0, maxx = 0, maxy = 0;
  uint32_t flags = 0;
  if (valid) {
    uint4 h = *reinterpret_cast<const uint4*>(P.recs + r);
    minx = (int)(int16_t)(h.x & 0xffffu);
    miny = (int)(int16_t)(h.x >> 16);
    maxx = (int)(int16_t)(h.y & 0xffffu);
    maxy = (int)(int16_t)(h.y >> 16);
    flags = h.w;
    valid = minx <= maxx;
  }
  int tx0 = (minx - (int)P.sx) >> TILE_SHIFT, tx1 = (maxx - (int)P.sx) >> TILE_SHIFT;
  int ty0 = (miny - (int)P.sy) >> TILE_SHIFT, ty1 = (maxy - (int)P.sy) >> TILE_SHIFT;
  int ntx = tx1 - tx0 + 1, nty = ty1 - ty0 + 1;
  int nt = valid ? ntx * nty : 0;
  uint32_t binbase = (flags & F_TRANSPARENT) ? P.n_tiles : 0u;

  // Small triangles (<= 16 tiles): step j emits every lane's j-th tile.  Neighbouring triangles of a
  // mesh land in the same few tiles, so the lanes of a wave that target the same bin are merged into
  // ONE atomic (device-scope atomics run at the memory side and serialise per line: unmerged, the
  // 16 counters of a 64-byte line made this kernel atomic-bound).
  uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  bool small = valid && nt <= SMALL_MAX_TILES;
  EdgeSet e;
  if (small && nt > 4) e = load_edges(P.recs + r);
  for (int j = 0; __ballot(small && j < nt); j++) {
    bool has = small && j < nt;
    uint32_t bin = 0;
    if (has) {
      int ty = ty0 + j / ntx, tx = tx0 + j % ntx;
      if (nt > 4) {
        int x0 = max(minx, (int)P.sx + tx * TILE), x1 = min(maxx, (int)P.sx + tx * TILE + TILE - 1);
        int y0 = max(miny, (int)P.sy + ty * TILE), y1 = min(maxy, (int)P.sy + ty * TILE + TILE - 1);
        has = box_overlaps(e, x0, y0, x1, y1);
      }
      bin = binbase + (uint32_t)ty * P.tiles_x + (uint32_t)tx;
    }
    unsigned long long todo = __ballot(has);
    uint32_t leader = lane, rank = 0, cnt = 0;
    while (todo) {
      int l = __ffsll((long long)todo) - 1;
      uint32_t b = __shfl(bin, l);
      unsigned long long same = __ballot(has && bin == b);
      if (has && bin == b) {
        leader = (uint32_t)l;
        rank = (uint32_t)__popcll(same & below);
        cnt = (uint32_t)__popcll(same);
      }
      todo &= ~same;
    }
    if (!FILL) {
      if (has && leader == lane) atomicAdd(&P.tile_count[bin], cnt);
    } else {
      uint32_t base = 0;
      if (has && leader == lane) base = atomicAdd(&P.tile_cursor[bin], cnt);
      base = __shfl(base, (int)leader);
      if (has) {
        uint32_t pos = P.tile_offset[bin] + base + rank;
        if (pos < P.bin_cap) P.bins[pos] = r;
      }
    }
  }
  // large triangles: the wave takes them one at a time, 64 tiles per step
  unsigned long long big = __ballot(valid && nt > SMALL_MAX_TILES);
  while (big) {
    int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    uint32_t rr = __shfl(r, src);
    int bminx = __shfl(minx, src), bminy = __shfl(miny, src), bmaxx = __shfl(maxx, src), bmaxy = __shfl(maxy, src);
    int btx0 = __shfl(tx0, src), bty0 = __shfl(ty0, src), bntx = __shfl(ntx, src), bnt = __shfl(nt, src);
    uint32_t bbase = __shfl(binbase, src);
    EdgeSet e = load_edges(P.recs + rr);  // same address in every lane: one broadcast load
    for (int t = (int)lane; t < bnt; t += 64) {
      int ty = bty0 + t / bntx, tx = btx0 + t % bntx;
      int x0 = max(bminx, (int)P.sx + tx * TILE), x1 = min(bmaxx, (int)P.sx + tx * TILE + TILE - 1);
      int y0 = max(bminy, (int)P.sy + ty * TILE), y1 = min(bmaxy, (int)P.sy + ty * TILE + TILE - 1);
      if (box_overlaps(e, x0, y0, x1, y1)) emit<FILL>(P, bbase + (uint32_t)ty * P.tiles_x + (uint32_t)tx, rr);
    }
  }
}

// exclusive scan of tile_count[0 .. 2*n_tiles) by one 1024-thread workgroup
__global__ __launch_bounds__(1024) void scan_kernel(FrameParams P) {
  __shared__ uint32_t wave_tot[16];
  uint32_t n = 2u * P.n_tiles;
  uint32_t per = (n + 1023u) / 1024u;
  uint32_t b = threadIdx.x * per, e = min(b + per, n);
  uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; i++) sum += P.tile_count[i];
  // inclusive scan inside the wave by shuffles, then over the 16 wave totals
  uint32_t inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t v = __shfl_up(inc, off);
    if ((int)lane >= off) inc += v;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  uint32_t wave_base = 0, total = 0;
  for (uint32_t w = 0; w < 16; w++) {
    uint32_t t = wave_tot[w];
    if (w < wv) wave_base += t;
    total += t;
  }
  uint32_t run = wave_base + inc - sum;
  for (uint32_t i = b; i < e; i++) {
    P.tile_offset[i] = run;
    run += P.tile_count[i];
  }
  if (threadIdx.x == 0) {
    P.counters->total_entries = total;
    if (total > P.bin_cap) atomicOr(&P.counters->overflow, 4u);
  }
  // Tile launch order for the tile kernel: heaviest class first (longest-processing-time-first), so
  // the few tiles with hundreds of triangles start at once and the light ones fill in behind them.
  // Class = bit length of (opaque + 2*transparent) entries; order inside a class is arbitrary.
  __shared__ uint32_t cls_count[33], cls_base[33];
  __syncthreads();
  if (threadIdx.x < 33) cls_count[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < P.n_tiles; t += 1024u) {
    uint32_t wgt = P.tile_count[t] + 2u * P.tile_count[P.n_tiles + t];
    atomicAdd(&cls_count[32 - __clz(wgt)], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run2 = 0;
    for (int c = 32; c >= 0; c--) {
      cls_base[c] = run2;
      run2 += cls_count[c];
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < P.n_tiles; t += 1024u) {
    uint32_t wgt = P.tile_count[t] + 2u * P.tile_count[P.n_tiles + t];
    uint32_t pos = atomicAdd(&cls_base[32 - __clz(wgt)], 1u);
    P.tile_order[pos] = t;
  }
}

static inline uint32_t bin_blocks(const FrameParams& P) { return (P.n_tris + P.extra_cap + 255u) / 256u; }

void launch_bin_count(const FrameParams& P, hipStream_t s) {
  hipLaunchKernelGGL(bin_kernel<false>, dim3(bin_blocks(P)), dim3(256), 0, s, P);
}
void launch_bin_scan(const FrameParams& P, hipStream_t s) {
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, P);
}
void launch_bin_fill(const FrameParams& P, hipStream_t s) {
  hipLaunchKernelGGL(bin_kernel<true>, dim3(bin_blocks(P)), dim3(256), 0, s, P);
}

}  // namespace svr
