// k_bin.hip — sort-middle binning of set-up triangles into 32x32-pixel screen tiles.
//
// The rasteriser hardware behind vkCmdDrawIndexed walks each triangle's pixels itself; here the
// frame is cut into tiles (one workgroup each, k_tile.hip) and every triangle is appended to the
// bin of every tile it can touch.  count -> span allocation -> fill, so bins are contiguous spans of
// one buffer and no per-tile capacity exists.  Bin order is arbitrary (atomics): the tile kernel
// resolves visibility order-independently and sorts the transparent bins itself.
// Two bin sets share the arrays: [0,n_tiles) opaque, [n_tiles,2*n_tiles) transparent.
//
// The setup kernel itself (k_geometry.hip, svr_bin.h emit_small_pairs) decides the tiles of every
// triangle of <= 16 tiles while its bbox and edge functions are in registers, and appends (bin, record)
// pairs to one list.  count_kernel takes a slot in its bin for every pair; the same launch runs the clipper and
// walks what setup could not take — the clipper's pieces and the queued big triangles — wave per record,
// appending their pairs with the slot already taken; offsets_kernel gives every bin a span; fill scatters.
#include <hip/hip_ext.h>

#include <algorithm>

#include "svr_bin.h"
#include "svr_clip.h"
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

// One record, wave-wide: the lanes take the tiles of its bbox 64 at a time (all loads wave-uniform: one broadcast
// each); each hit takes its slot from the tile counter and is appended to the pair list with the slot already known.
// These pairs grow from the END of the list downwards (n_pairs_rest) while the setup kernel's lie at its head
// (n_pairs, final when this launch starts): the two parts never need a snapshot of each other's length.
__device__ __forceinline__ void bin_one(const FrameParams& P, uint32_t r, int minx, int miny, int maxx, int maxy, uint32_t binbase,
                                        const EdgeSet& e, uint32_t n_setup) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  TileRange tr = tile_range(P, minx, miny, maxx, maxy, true);
  for (int t0 = 0; t0 < tr.nt; t0 += 64) {
    int t = t0 + (int)lane;
    bool hit = t < tr.nt;
    uint32_t bin = 0;
    if (hit) {
      int ty = tr.ty0 + t / tr.ntx, tx = tr.tx0 + t % tr.ntx;
      if (tr.nt > 4) {
        int x0 = max(minx, (int)P.sx + tx * TILE), x1 = min(maxx, (int)P.sx + tx * TILE + TILE - 1);
        int y0 = max(miny, tile_row_y(P, ty)), y1 = min(maxy, tile_row_y(P, ty) + TILE - 1);
        hit = box_overlaps(e, x0, y0, x1, y1);
      }
      bin = binbase + (uint32_t)ty * P.tiles_x + (uint32_t)tx;
    }
    unsigned long long hits = __ballot(hit);
    if (!hits) continue;
    uint32_t slot = 0, base = 0;
    if (hit) slot = atomicAdd(&P.tile_count[bin], 1u);
    if (lane == 0) base = atomicAdd(&P.counters->n_pairs_rest, (uint32_t)__popcll(hits));
    base = __shfl(base, 0);
    if ((unsigned long long)n_setup + base + (uint32_t)__popcll(hits) > P.bin_cap) {
      if (lane == 0) atomicOr(&P.counters->overflow, 4u);
      continue;
    }
    if (hit) {
      uint32_t pos = P.bin_cap - 1u - (base + (uint32_t)__popcll(hits & below));
      P.pairs[pos] = make_uint2(bin, r);
      P.pair_slot[pos] = slot;
    }
  }
}

// The triangles over 16 tiles the setup kernel queued instead of deciding their tiles lane by lane: one WAVE per record.
__device__ __forceinline__ void bin_big(const FrameParams& P, uint32_t first_wave, uint32_t n_waves) {
  const uint32_t n_big = min(P.counters->n_big, P.n_tris), n_setup = min(P.counters->n_pairs, P.bin_cap);
  for (uint32_t it = first_wave; it < n_big; it += n_waves) {
    const uint32_t r = P.big_queue[it];
    const TriRec* rec = P.recs + r;
    RecBox b = load_box(rec);
    if (!b.valid) continue;  // wave-uniform
    bin_one(P, r, b.minx, b.miny, b.maxx, b.maxy, (b.flags & F_TRANSPARENT) ? P.n_tiles : 0u, load_edges(rec), n_setup);
  }
}

// The clipper, in the binning launch: the few triangles that cross the near / far planes or the guard band (a few
// hundred per frame) were queued by the setup kernel.  CLIP_LANES lanes of a wave take one each — Sutherland-Hodgman
// with the polygon in LDS (indexed by run-time values: as a local array it would live in scratch memory, and a kernel
// with a scratch frame pays ~2.5 us at every dispatch, tools/gapbench.hip) — and emit the fan piece by piece, in step;
// after every step the WHOLE wave bins the pieces just made, from their bounding boxes and edges in LDS (what
// setup_triangle hands out beside the record), so nothing waits for the records to come back from memory.  Until round 3
// this was a kernel of its own between setup and count, whose "rest" blocks then read the pieces back: one launch,
// one boundary and ~5 us of a small pass's chain more.
constexpr uint32_t CLIP_LANES = 8;
struct ClipLds {
  VOut poly[CLIP_LANES][12];
  VOut tmp[CLIP_LANES][12];
  TriGeom geom[CLIP_LANES];
};
__device__ __forceinline__ void clip_and_bin(const FrameParams& P, ClipLds& L, uint32_t block, uint32_t n_blocks) {
  const uint32_t lane = threadIdx.x;  // wave 0 of the block
  const uint32_t n = min(P.counters->n_clip, P.clip_cap), n_setup = min(P.counters->n_pairs, P.bin_cap);
  const float hw = (float)P.W * 0.5f, hh = (float)P.H * 0.5f;
  for (uint32_t q0 = block * CLIP_LANES; q0 < n; q0 += n_blocks * CLIP_LANES) {  // wave-uniform
    const uint32_t q = q0 + lane;
    const bool worker = lane < CLIP_LANES && q < n;
    VOut* poly = L.poly[lane & (CLIP_LANES - 1u)];
    int np = 0;
    uint32_t first = 0, want = 0, used = 0, seq = 0, draw = 0;
    if (worker) {
      ClipItem it = P.clip_queue[q];
      draw = it.draw;
      const DrawDesc& d = P.draws[draw];
      const uint32_t kind = (d.flags >> F_KIND_SHIFT) & 3u;
      seq = d.tri_base + it.tri;
      if (kind == PIPE_COLORED_TRIANGLE) {
        colored_triangle_vert(0, poly[0]);
        colored_triangle_vert(1, poly[1]);
        colored_triangle_vert(2, poly[2]);
      } else {
        for (int k = 0; k < 3; k++) shade_corner(d, kind, d.mvp, d.idx[3 * it.tri + k], poly[k]);
      }
      np = clip_polygon(poly, L.tmp[lane], 3);
      if (np >= 3) {
        // The fan's records are one contiguous block, and the parent's (invalid) main slot links to it:
        // the tile kernel's visibility pass keeps only (depth, key) per pixel and (key >> 2) - 1 names the main
        // slot, so shading finds a clipped parent's covering piece through this link.
        want = (uint32_t)(np - 2);
        first = atomicAdd(&P.counters->n_extra, want);
        if (first + want > P.extra_cap) {
          atomicOr(&P.counters->overflow, 2u);
          np = 0;
        }
      }
    }
    int steps = np;  // the longest fan of the wave's polygons
    for (int m = 1; m < (int)CLIP_LANES; m <<= 1) steps = max(steps, __shfl_xor(steps, m));
    steps = __shfl(steps, 0);
    for (int i = 1; i + 1 < steps; i++) {  // wave-uniform
      bool made = false;
      uint32_t rec = 0, transparent = 0;
      if (worker && i + 1 < np) {
        const DrawDesc& d = P.draws[draw];
        ScreenV s0 = to_screen(poly[0].clip, hw, hh), s1 = to_screen(poly[i].clip, hw, hh), s2 = to_screen(poly[i + 1].clip, hw, hh);
        if (s0.ok && s1.ok && s2.ok && poly[0].clip[3] > 0.0f && poly[i].clip[3] > 0.0f && poly[i + 1].clip[3] > 0.0f) {
          // the record is written where it goes (no staging in registers: sixteen more live uint4 put the kernel over its
          // register cap); a piece that turns out degenerate leaves a slot that the next one, or store_invalid below, overwrites
          rec = P.n_tris + first + used;
          uint4* dst = reinterpret_cast<uint4*>(P.recs + rec);
          if (setup_triangle(P, &poly[0], &poly[i], &poly[i + 1], s0, s1, s2, make_key(seq, d.flags, P.tex[d.tex], true), d.flags, P.tex[d.tex],
                             *reinterpret_cast<uint4(*)[16]>(dst), &L.geom[lane])) {
            made = true;
            used++;
            transparent = (d.flags & F_TRANSPARENT) ? 1u : 0u;
            if (P.instrument) atomicAdd(&P.counters->binned, 1ull);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the pieces' boxes and edges are in LDS: every lane may read them
      for (unsigned long long todo = __ballot(made); todo; todo &= todo - 1ull) {
        const int l = __ffsll((long long)todo) - 1;
        const TriGeom& gm = L.geom[l];
        EdgeSet e;
        e.A0 = gm.A[0]; e.A1 = gm.A[1]; e.A2 = gm.A[2]; e.B0 = gm.B[0]; e.B1 = gm.B[1]; e.B2 = gm.B[2];
        e.C0 = gm.C[0]; e.C1 = gm.C[1]; e.C2 = gm.C[2];
        bin_one(P, (uint32_t)__shfl((int)rec, l), gm.minx, gm.miny, gm.maxx, gm.maxy, __shfl((int)transparent, l) ? P.n_tiles : 0u, e, n_setup);
      }
      __builtin_amdgcn_wave_barrier();  // ... before the next step's pieces overwrite them
    }
    if (worker && np >= 3) {
      for (uint32_t k = used; k < want; k++) store_invalid(P.recs + P.n_tris + first + k);  // reserved, unused
      uint4 link;
      link.x = 1u;  // minx = 1 > maxx = 0: still an invalid record for binning
      link.y = 0u;
      link.z = P.n_tris + first;
      link.w = used;
      *reinterpret_cast<uint4*>(P.recs + seq) = link;
    }
  }
}

// Blocks [0, big_blocks): the queued big triangles (above).  [big_blocks, big_blocks + clip_blocks): the clipper (above;
// wave 0 of each).  The others: one lane per pair the setup kernel emitted,
// grid-stride; the lanes of a wave that target the same bin share ONE atomic (device-scope atomics run
// at the memory side and serialise per line: unmerged, neighbouring triangles made binning
// atomic-bound), and its return value is the pair's position in its bin — kept, so the fill is a plain
// scatter with no second round of atomics.
__global__ __launch_bounds__(256, 4) void count_kernel(FrameParams P, uint32_t big_blocks, uint32_t clip_blocks) {
  __shared__ ClipLds s_clip;
  if (P.counters->overflow) return;  // a queue overflowed: the pass is void, the host grows it and replays
  if (blockIdx.x < big_blocks) {
    bin_big(P, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (big_blocks * blockDim.x) >> 6);
    return;
  }
  const uint32_t rest_blocks = big_blocks + clip_blocks;
  if (blockIdx.x < rest_blocks) {
    if (threadIdx.x < 64u) clip_and_bin(P, s_clip, blockIdx.x - big_blocks, clip_blocks);
    return;
  }
  // Which lanes of the wave target the same bin is found through a small LDS hash table of the wave's own (128
  // entries for 64 pairs): a lane claims its bin's entry (compare-and-swap, linear probing), takes its rank from the
  // entry's counter, and the lane that got rank 0 does the bin's ONE device atomic with the final count.  (The first
  // version found the groups with a ballot per distinct bin: a wave of the 8K x16 frame's pairs — big triangles, every
  // pair of a triangle a different tile — went round that loop dozens of times: 169 us for 6.6 M pairs.)
  __shared__ uint32_t s_key[4][128], s_cnt[4][128], s_base[4][128];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t* hk = s_key[wv];
  uint32_t* hc = s_cnt[wv];
  uint32_t* hb = s_base[wv];
  hk[lane] = 0u; hk[lane + 64u] = 0u;
  hc[lane] = 0u; hc[lane + 64u] = 0u;
  __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave retire in order; this pins the compiler's order too)
  const uint32_t n = min(P.counters->n_pairs, P.bin_cap), rounded = (n + 63u) & ~63u;
  const uint32_t stride = (gridDim.x - rest_blocks) * blockDim.x;
  for (uint32_t i = (blockIdx.x - rest_blocks) * blockDim.x + threadIdx.x; i < rounded; i += stride) {
    const bool has = i < n;
    const uint32_t bin = has ? P.pairs[i].x : 0u;
    uint32_t h = 0, rank = 1;
    if (has) {
      h = (bin * 0x9E3779B1u) >> 25;
      for (;;) {
        const uint32_t k = atomicCAS(&hk[h], 0u, bin + 1u);
        if (k == 0u || k == bin + 1u) break;
        h = (h + 1u) & 127u;
      }
      rank = atomicAdd(&hc[h], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    if (has && rank == 0u) hb[h] = atomicAdd(&P.tile_count[bin], hc[h]);  // (hc[h] is final: every lane's add of this round is behind it)
    __builtin_amdgcn_wave_barrier();
    if (has) P.pair_slot[i] = hb[h] + rank;
    __builtin_amdgcn_wave_barrier();
    if (has && rank == 0u) {
      hk[h] = 0u;
      hc[h] = 0u;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Bin offsets without a scan: bins need not lie in tile order, only be disjoint spans, so every wave
// reserves the span of its 64 bins with one atomic on the running total (wave prefix sum inside).
// The same kernel builds the histogram of tile weight classes for the tile kernel's launch order:
// class = bit length of (opaque + 2*transparent) entries.
// (Classes by estimated cost — tile_cost, two per octave, so that the curtain tiles lead the launch instead of
// sharing a class with merely triangle-rich opaque tiles — measured slower: 0.2007 vs 0.1959 ms at 4K.)
__device__ __forceinline__ uint32_t weight_class(const FrameParams& P, uint32_t t) {
  return 32u - (uint32_t)__clz(P.tile_count[t] + 2u * P.tile_count[P.n_tiles + t]);
}

__global__ __launch_bounds__(256) void offsets_kernel(FrameParams P) {
  __shared__ uint32_t l_cnt[33];
  __shared__ uint32_t s_void;
  // Other blocks of this kernel raise the flag while it runs: the early exit must be one decision per
  // block (barriers follow), so one thread reads it and the block takes its word.
  if (threadIdx.x == 0) s_void = P.counters->overflow;
  __syncthreads();
  if (s_void) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = 2u * P.n_tiles;
  if (threadIdx.x < 33) l_cnt[threadIdx.x] = 0;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = i < n ? P.tile_count[i] : 0u, inc = v;
  uint32_t cls = i < P.n_tiles ? weight_class(P, i) : 0u;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t u = __shfl_up(inc, off);
    if ((int)lane >= off) inc += u;
  }
  uint32_t base = 0;
  if (lane == 63 && inc) {
    base = atomicAdd(&P.counters->total_entries, inc);
    if (base + inc > P.bin_cap) atomicOr(&P.counters->overflow, 4u);
  }
  base = __shfl(base, 63);
  if (i < n) P.tile_offset[i] = base + inc - v;
  uint32_t cost = i < P.n_tiles ? tile_cost(v, P.tile_count[P.n_tiles + i]) : 0u;  // fill_kernel's split rule needs the sum
  {  // ... and per tile row, for whoever cuts the frame into row bands of equal cost (svr_get_row_costs): the lanes
     // of a wave hold consecutive tiles, so every row's lanes are a run; the first lane of each run adds its run's sum
    const uint32_t row = i < P.n_tiles ? i / P.tiles_x : 0xffffffffu;
    uint32_t run = cost;
    for (int off = 1; off < 64; off <<= 1) {  // suffix sums within equal-row runs
      uint32_t u = __shfl_down(run, off);
      uint32_t r2 = __shfl_down(row, off);
      if (lane + (uint32_t)off < 64u && r2 == row) run += u;
    }
    const uint32_t prev = __shfl_up(row, 1);
    if (row != 0xffffffffu && (lane == 0 || prev != row) && run) atomicAdd(&P.row_cost[row], run);
  }
  for (int off = 32; off > 0; off >>= 1) cost += __shfl_down(cost, off);
  if (lane == 0 && cost) atomicAdd(&P.counters->cost_sum, cost);
  __syncthreads();
  if (i < P.n_tiles) atomicAdd(&l_cnt[cls], 1u);  // LDS
  __syncthreads();
  if (threadIdx.x < 33 && l_cnt[threadIdx.x]) atomicAdd(&P.cls_count[threadIdx.x], l_cnt[threadIdx.x]);
}

// Tile launch order (blocks that own tiles), then bins[offset[bin] + slot] = record for every pair.
// Order: heaviest class first (longest-processing-time-first), so the few tiles with hundreds of
// triangles start at once and the light ones fill in behind them; order inside a class is arbitrary.
// A block ranks its tiles per class in LDS and takes one span per class from the global cursors.
__global__ __launch_bounds__(256) void fill_kernel(FrameParams P) {
  __shared__ uint32_t l_cnt[33], l_base[33];
  __shared__ uint32_t s_void;
  if (threadIdx.x == 0) s_void = P.counters->overflow;  // raised concurrently by other blocks: one decision per block
  __syncthreads();
  if (s_void) return;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x * blockDim.x < P.n_tiles) {  // block-uniform
    if (threadIdx.x < 33) l_cnt[threadIdx.x] = 0;
    __syncthreads();
    bool has = t < P.n_tiles;
    uint32_t cls = has ? weight_class(P, t) : 0u, rank = 0;
    if (has) rank = atomicAdd(&l_cnt[cls], 1u);  // LDS
    __syncthreads();
    if (threadIdx.x < 33) {
      uint32_t c = threadIdx.x, before = 0;
      for (uint32_t h = c + 1; h < 33; h++) before += P.cls_count[h];  // tiles of heavier classes
      uint32_t mine = l_cnt[c];
      l_base[c] = before + (mine ? atomicAdd(&P.cls_count[40 + c], mine) : 0u);
    }
    __syncthreads();
    if (has) {
      uint32_t slot = l_base[cls] + rank;
      P.tile_order[slot] = t;
      const uint32_t n_op = P.tile_count[t], n_tr = P.tile_count[P.n_tiles + t];
      // a transparent bin too large for the tile kernel's LDS sort gets a span of the global sort arena
      // (next power of two: the bitonic network pads); failing here voids the pass before it draws
      uint32_t sort_base = 0;
      if (n_tr > SPLIT_SORT_MAX) {  // (= the tile kernel's SORT_CAP)
        uint32_t np = 1u << (32 - __clz(n_tr - 1u));
        sort_base = atomicAdd(&P.counters->sort_used, np);
        if (sort_base + np > P.sort_cap) atomicOr(&P.counters->overflow, 4u);
      }
      // heavy tile: four quarters (svr_device.h SPLIT_*).  Each writes the sorted list of its own part of the
      // transparent bin to its quarter of the tile's span of the sort arena; none touches the bin itself.
      bool split = !(P.tuning & TUNE_NO_SPLIT) && n_tr <= SPLIT_SORT_MAX &&
                   tile_cost(n_op, n_tr) > max(SPLIT_MIN_COST, P.counters->cost_sum / TILE_SLOTS);
      uint32_t sp = 0;
      if (split) {
        sp = atomicAdd(&P.counters->n_split, 1u);
        split = sp < SPLIT_MAX;
      }
      if (split && n_tr) {
        const uint32_t span = 4u * ((n_tr + 1u) >> 1);  // four lists of up to n_tr 32-bit words
        sort_base = atomicAdd(&P.counters->sort_used, span);
        if (sort_base + span > P.sort_cap) atomicOr(&P.counters->overflow, 4u);
      }
      const uint4 i0 = make_uint4(t, n_op, P.tile_offset[t], n_tr);
      const uint32_t off_tr = P.tile_offset[P.n_tiles + t];
      if (split)
        for (uint32_t j = 0; j < 4; j++) {
          P.tile_info[2u * (4u * sp + j)] = i0;
          P.tile_info[2u * (4u * sp + j) + 1u] = make_uint4(off_tr, sort_base, 8u * j, 1u);  // z: first row of the quarter
        }
      P.tile_info[2u * (SPLIT_EXTRA + slot)] = i0;
      P.tile_info[2u * (SPLIT_EXTRA + slot) + 1u] = make_uint4(off_tr, sort_base, 0u, split ? 1u : 0u);  // w: "rendered by its quarters"
    }
  }
  // the setup kernel's pairs at the head of the list, the count launch's own (big triangles, clipper pieces) at its tail
  const uint32_t n = min(P.counters->n_pairs, P.bin_cap), n_rest = min(P.counters->n_pairs_rest, P.bin_cap - n);
  for (uint32_t i = t; i < n + n_rest; i += gridDim.x * blockDim.x) {
    const uint32_t at = i < n ? i : P.bin_cap - 1u - (i - n);
    uint2 p = P.pairs[at];
    uint32_t pos = P.tile_offset[p.x] + P.pair_slot[at];
    if (pos < P.bin_cap) P.bins[pos] = p.y;
  }
}

void launch_bin_count(const FrameParams& P, hipStream_t s) {
  // (128 blocks for the queued big triangles also in the 8K x16 frame: with 1024 the kernel is 246 us instead of 168 —
  // what bounds it there is the device atomics on the bins' counters, and more waves only queue up behind them)
  const uint32_t big_blocks = 128, clip_blocks = 512;
  hipLaunchKernelGGL(count_kernel, dim3(big_blocks + clip_blocks + 1024u), dim3(256), 0, s, P, big_blocks, clip_blocks);
}
void launch_bin_scan(const FrameParams& P, hipStream_t s) {
  hipLaunchKernelGGL(offsets_kernel, dim3((2u * P.n_tiles + 255u) / 256u), dim3(256), 0, s, P);
}
void launch_bin_fill(const FrameParams& P, hipStream_t s, hipEvent_t done) {
  uint32_t blocks = std::max(1024u, (P.n_tiles + 255u) / 256u);
  hipExtLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, P);
}

}  // namespace svr
