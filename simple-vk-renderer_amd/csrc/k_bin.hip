// k_bin.hip — sort-middle binning of set-up triangles into 32x32-pixel screen tiles.
//
// The rasteriser hardware behind vkCmdDrawIndexed walks each triangle's pixels itself; here the
// frame is cut into tiles (one workgroup each, k_tile.hip) and every triangle is appended to the
// bin of every tile it can touch.  count -> exclusive scan -> fill, so bins are contiguous spans of
// one buffer and no per-tile capacity exists.  Bin order is arbitrary (atomics): the tile kernel
// resolves visibility order-independently and sorts the transparent bins itself.
// Two bin sets share the arrays: [0,n_tiles) opaque, [n_tiles,2*n_tiles) transparent.
//
// The setup kernel itself (k_geometry.hip, svr_bin.h emit_small_pairs) decides the tiles of every
// triangle of <= 16 tiles while its bbox and edge functions are in registers, and appends (bin, record)
// pairs to one list.  count_kernel takes a slot in its bin for every pair (and walks what setup could
// not take — the clipper's records and the queued big triangles — wave per record, appending their
// pairs with the slot already taken); scan turns counts into offsets; fill scatters.
#include "svr_bin.h"
#include "svr_launch.h"

namespace svr {

__device__ __forceinline__ EdgeSet load_edges(const TriRec* rec) {
  EdgeSet e;
  const double2* p = reinterpret_cast<const double2*>(rec);
  double2 a = p[2], b = p[3], c = p[4], d = p[5];
  e.A0 = a.x; e.A1 = a.y; e.A2 = b.x; e.B0 = b.y; e.B1 = c.x; e.B2 = c.y; e.C0 = d.x; e.C1 = d.y;
  e.C2 = rec->C[2];
  return e;
}

struct RecBox {
  int minx, miny, maxx, maxy;
  uint32_t flags;
  bool valid;
};
__device__ __forceinline__ RecBox load_box(const TriRec* rec) {
  uint4 h = *reinterpret_cast<const uint4*>(rec);
  RecBox b;
  b.minx = (int)(int16_t)(h.x & 0xffffu);
  b.miny = (int)(int16_t)(h.x >> 16);
  b.maxx = (int)(int16_t)(h.y & 0xffffu);
  b.maxy = (int)(int16_t)(h.y >> 16);
  b.flags = h.w;
  b.valid = b.minx <= b.maxx;
  return b;
}

// The "rest": records the lane-per-triangle path of the setup kernel does not take — the clipper's
// pieces and the queued triangles over 16 tiles.  One WAVE per record (all loads wave-uniform: one
// broadcast each), lanes take the tiles of its bbox 64 at a time; each hit takes its slot from the
// tile counter and is appended to the pair list with the slot already known.
__device__ __forceinline__ void bin_rest(const FrameParams& P, uint32_t first_wave, uint32_t n_waves) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t n_extra = min(P.counters->n_extra, P.extra_cap), n_big = min(P.counters->n_big, P.n_tris);
  uint32_t n_items = n_extra + n_big;
  for (uint32_t it = first_wave; it < n_items; it += n_waves) {
    uint32_t r = it < n_extra ? P.n_tris + it : P.big_queue[it - n_extra];
    const TriRec* rec = P.recs + r;
    RecBox b = load_box(rec);
    if (!b.valid) continue;  // wave-uniform
    TileRange tr = tile_range(P, b.minx, b.miny, b.maxx, b.maxy, true);
    uint32_t binbase = (b.flags & F_TRANSPARENT) ? P.n_tiles : 0u;
    EdgeSet e = load_edges(rec);
    for (int t0 = 0; t0 < tr.nt; t0 += 64) {
      int t = t0 + (int)lane;
      bool hit = t < tr.nt;
      uint32_t bin = 0;
      if (hit) {
        int ty = tr.ty0 + t / tr.ntx, tx = tr.tx0 + t % tr.ntx;
        if (tr.nt > 4) {
          int x0 = max(b.minx, (int)P.sx + tx * TILE), x1 = min(b.maxx, (int)P.sx + tx * TILE + TILE - 1);
          int y0 = max(b.miny, (int)P.sy + ty * TILE), y1 = min(b.maxy, (int)P.sy + ty * TILE + TILE - 1);
          hit = box_overlaps(e, x0, y0, x1, y1);
        }
        bin = binbase + (uint32_t)ty * P.tiles_x + (uint32_t)tx;
      }
      unsigned long long hits = __ballot(hit);
      if (!hits) continue;
      uint32_t slot = 0, base = 0;
      if (hit) slot = atomicAdd(&P.tile_count[bin], 1u);
      if (lane == 0) base = atomicAdd(&P.counters->n_pairs, (uint32_t)__popcll(hits));
      base = __shfl(base, 0);
      if (base + (uint32_t)__popcll(hits) > P.bin_cap) {
        if (lane == 0) atomicOr(&P.counters->overflow, 4u);
        continue;
      }
      if (hit) {
        uint32_t pos = base + (uint32_t)__popcll(hits & below);
        P.pairs[pos] = make_uint2(bin, r);
        P.pair_slot[pos] = slot;
      }
    }
  }
}

// Blocks [0, rest_blocks): the rest (above).  The others: one lane per pair the setup kernel emitted,
// grid-stride; the lanes of a wave that target the same bin share ONE atomic (device-scope atomics run
// at the memory side and serialise per line: unmerged, neighbouring triangles made binning
// atomic-bound), and its return value is the pair's position in its bin — kept, so the fill is a plain
// scatter with no second round of atomics.
__global__ __launch_bounds__(256) void count_kernel(FrameParams P, uint32_t rest_blocks) {
  if (P.counters->overflow) return;  // a queue overflowed: the pass is void, the host grows it and replays
  if (blockIdx.x < rest_blocks) {
    bin_rest(P, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (rest_blocks * blockDim.x) >> 6);
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t n = P.counters->n_pairs_setup, rounded = (n + 63u) & ~63u;
  const uint32_t stride = (gridDim.x - rest_blocks) * blockDim.x;
  for (uint32_t i = (blockIdx.x - rest_blocks) * blockDim.x + threadIdx.x; i < rounded; i += stride) {
    bool has = i < n;
    uint32_t bin = has ? P.pairs[i].x : 0u;
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
    uint32_t base = 0;
    if (has && leader == lane) base = atomicAdd(&P.tile_count[bin], cnt);
    base = __shfl(base, (int)leader);
    if (has) P.pair_slot[i] = base + rank;
  }
}

// bins[offset[bin] + slot] = record, for every pair
__global__ __launch_bounds__(256) void fill_kernel(FrameParams P) {
  if (P.counters->overflow) return;
  const uint32_t n = min(P.counters->n_pairs, P.bin_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint2 p = P.pairs[i];
    uint32_t pos = P.tile_offset[p.x] + P.pair_slot[i];
    if (pos < P.bin_cap) P.bins[pos] = p.y;
  }
}

// Exclusive scan of tile_count[0 .. 2*n_tiles) by one 1024-thread workgroup, plus the tile launch
// order for the tile kernel: heaviest class first (longest-processing-time-first), so the few tiles
// with hundreds of triangles start at once and the light ones fill in behind them.  Class = bit
// length of (opaque + 2*transparent) entries; order inside a class is arbitrary.
// Every thread scans 16 consecutive counters (four 16-byte loads) and the block scans the 1024 partial
// sums: 16K counters (a 4K frame has 16,320) take one sweep, larger grids more with a running carry.
constexpr int SCAN_PER = 16;
__global__ __launch_bounds__(1024) void scan_kernel(FrameParams P) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t cls_count[33], cls_base[33];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t n = 2u * P.n_tiles;  // a 16-byte load may run 2 counters into tile_cursor (same allocation)
  const unsigned long long below = (1ull << lane) - 1ull;
  if (tid < 33) cls_count[tid] = 0;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += 1024u * SCAN_PER) {
    uint32_t first = base + tid * SCAN_PER;
    uint32_t v[SCAN_PER];
#pragma unroll
    for (int q = 0; q < SCAN_PER / 4; q++) {
      uint4 w = first + 4u * q < n ? reinterpret_cast<const uint4*>(P.tile_count + first)[q] : make_uint4(0, 0, 0, 0);
      v[4 * q + 0] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) {
      if (first + (uint32_t)k >= n) v[k] = 0;
      mine += v[k];
    }
    uint32_t inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t u = __shfl_up(inc, off);
      if ((int)lane >= off) inc += u;
    }
    __syncthreads();  // wave_tot of the previous sweep consumed
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (uint32_t w = 0; w < 16; w++) {
      uint32_t t = wave_tot[w];
      if (w < wv) wave_base += t;
      total += t;
    }
    uint32_t run = carry + wave_base + inc - mine;
    uint32_t o[SCAN_PER];
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) {
      o[k] = run;
      run += v[k];
    }
#pragma unroll
    for (int q = 0; q < SCAN_PER / 4; q++) {
      if (first + 4u * q + 3u < n) {
        reinterpret_cast<uint4*>(P.tile_offset + first)[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
      } else {
        for (int k = 4 * q; k < 4 * q + 4; k++)
          if (first + (uint32_t)k < n) P.tile_offset[first + (uint32_t)k] = o[k];
      }
    }
    carry += total;
  }
  if (tid == 0) {
    P.counters->total_entries = carry;
    if (carry > P.bin_cap) atomicOr(&P.counters->overflow, 4u);
  }
  // class histogram and placement; the lanes of a wave that share a class share one LDS atomic.
  // Eight tiles per thread and sweep, their counters loaded up front (one memory round trip).
  constexpr int CLS_PER = 8;
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t base = 0; base < P.n_tiles; base += 1024u * CLS_PER) {
      uint32_t cls[CLS_PER];
#pragma unroll
      for (int k = 0; k < CLS_PER; k++) {
        uint32_t t = base + (uint32_t)k * 1024u + tid;
        cls[k] = t < P.n_tiles ? P.tile_count[t] + 2u * P.tile_count[P.n_tiles + t] : 0u;
      }
#pragma unroll
      for (int k = 0; k < CLS_PER; k++) {
        uint32_t t = base + (uint32_t)k * 1024u + tid;
        bool has = t < P.n_tiles;
        uint32_t mine = 32u - (uint32_t)__clz(cls[k]);
        unsigned long long todo = __ballot(has);
        while (todo) {
          int l = __ffsll((long long)todo) - 1;
          uint32_t c = __shfl(mine, l);
          unsigned long long same = __ballot(has && mine == c);
          uint32_t got = 0;
          if ((int)lane == l) got = atomicAdd(pass == 0 ? &cls_count[c] : &cls_base[c], (uint32_t)__popcll(same));
          got = __shfl(got, l);
          if (pass == 1 && has && mine == c) P.tile_order[got + (uint32_t)__popcll(same & below)] = t;
          todo &= ~same;
        }
      }
    }
    __syncthreads();
    if (pass == 0) {
      if (tid == 0) {
        uint32_t run = 0;
        for (int c = 32; c >= 0; c--) {
          cls_base[c] = run;
          run += cls_count[c];
        }
      }
      __syncthreads();
    }
  }
}

void launch_bin_count(const FrameParams& P, hipStream_t s) {
  const uint32_t rest_blocks = 128;
  hipLaunchKernelGGL(count_kernel, dim3(rest_blocks + 1024u), dim3(256), 0, s, P, rest_blocks);
}
void launch_bin_scan(const FrameParams& P, hipStream_t s) {
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, P);
}
void launch_bin_fill(const FrameParams& P, hipStream_t s) {
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s, P);
}

}  // namespace svr
