"""Multi-GPU form of the draw path: one process per GPU, the frame's rows sharded across ranks, the
finished bands exchanged once per frame (torch.distributed backend "nccl" = RCCL over xGMI on ROCm;
"gloo" on CPU for tests).

The reference is a single-device, single-queue renderer (deviceIndex 0, src/vk_engine.cpp:1295), so
nothing here translates reference code; it is the screen-space decomposition of SURVEY.md §8e:
  - scene (meshes, textures, materials) replicated on every rank — Sponza-scale is ~150 MB of 288 GB;
  - rank r owns the rows [bounds[r], bounds[r+1]): svr_set_scissor clips geometry (whole wave chunks
    whose box cannot reach the band are dropped before a vertex is fetched), binning and the tile grid to
    the band; pixels of a band never depend on another band;
  - each rank renders straight into its rows of a full-frame _draw_image tensor, presents them
    (svr_copy_to_swapchain, identity extent: the scissor's rows) into its rows of a full-frame
    B8G8R8A8 swapchain tensor, and the bands are exchanged in place: every rank ends up with the whole
    presentable frame at 4 bytes per pixel, as SURVEY §8e sizes the exchange (present=False exchanges
    the RGBA16F _draw_image itself, 8 bytes per pixel);
  - two frames in flight (the reference keeps FRAME_OVERLAP = 3, src/vk_engine.h:77): the exchange of
    frame i runs on the collective's stream while frame i+1 renders.

Partition.  Equal bands are what one all_gather_into_tensor moves, but the frame's cost is not uniform
in y (configs[3]: the busiest eighth holds 1.2x the mean; configs[4] before its camera was fixed: three
ranks of eight had nothing to do).  BandPlan cuts the rows into bands of equal COST instead: every rank
reports its tile rows' cost of a recent frame (svr_get_row_costs: the per-tile cost model the tile
kernel's own split rule uses, summed per 32-row tile row), one all_reduce makes the whole profile known
everywhere, and every rank derives the same boundaries from it (integer arithmetic on the summed
profile).  Unequal bands travel as one batch of point-to-point sends and receives (grouped
ncclSend/ncclRecv under RCCL: every band goes straight into its rows of every peer's frame, no padding,
no un-permute pass); equal bands keep the single all-gather.  xGMI is point-to-point (7 links per GPU),
so either way a band crosses each link once.
"""
import math

import numpy as np


def band_rows(height, rank, world):
    """(first_row, n_rows_rendered, band_height): equal bands of ceil(H/world) rows; the last may be short."""
    band = int(math.ceil(height / world))
    y0 = min(rank * band, height)
    y1 = min(y0 + band, height)
    return y0, y1 - y0, band


def equal_bounds(height, world):
    band = int(math.ceil(height / world))
    return [min(r * band, height) for r in range(world)] + [height]


def balanced_bounds(row_cost, world, fixed_cost=0):
    """Boundaries b[0..world] (b[0] = 0, b[world] = H, non-decreasing) that cut the rows into `world` bands
    whose costs  fixed_cost + sum(row_cost[b[r]:b[r+1]])  have the smallest possible maximum.

    row_cost: non-negative integers, one per pixel row.  Exact (binary search on the bottleneck over integer
    prefix sums, greedy feasibility test), deterministic, and the same on every rank given the same profile.
    A band may come out empty (a rank with nothing to render) when fewer rows than ranks carry any cost."""
    c = np.asarray(row_cost, dtype=np.int64)
    h = int(c.size)
    pre = np.concatenate([[0], np.cumsum(c)])

    def cut(limit):
        """greedy: each band takes as many rows as fit under `limit`; returns bounds or None"""
        b = [0]
        for _ in range(world):
            # largest e with pre[e] - pre[b[-1]] <= limit
            e = int(np.searchsorted(pre, pre[b[-1]] + limit, side="right")) - 1
            b.append(min(max(e, b[-1]), h))
        return b if b[-1] >= h else None

    lo, hi = int(c.max()) if h else 0, int(pre[-1])
    while lo < hi:
        mid = (lo + hi) // 2
        if cut(mid) is None:
            lo = mid + 1
        else:
            hi = mid
    b = cut(lo)
    b[0], b[world] = 0, h
    return [int(x) for x in b]


class BandPlan:
    """The partition all ranks share, and how it is kept balanced."""

    def __init__(self, height, world, balanced=True, min_gain=0.2):
        self.height, self.world, self.balanced = height, world, balanced
        self.bounds = equal_bounds(height, world)
        self.updates = 0
        # Unequal bands cost the HOST more per frame than equal ones (a batch of point-to-point operations instead of
        # one all-gather: tens of microseconds from Python), so a cost-balanced cut is adopted only where it shortens
        # the heaviest band by at least this fraction of the equal cut's (configs[3] at 4K: 11 % -> stays equal;
        # configs[4]: 50 % -> balanced)
        self.min_gain = min_gain

    def rows_of(self, rank):
        return self.bounds[rank], self.bounds[rank + 1] - self.bounds[rank]

    def is_equal(self):
        return self.bounds == equal_bounds(self.height, self.world)

    @staticmethod
    def spread(costs, y0, rows, height):
        """tile-row costs of a band (svr_get_row_costs) -> cost per pixel row of the frame (zeros elsewhere);
        a tile row's cost is shared by the pixel rows it covers, in integers (x1024 keeps the remainders)"""
        out = np.zeros(height, dtype=np.int64)
        for t, c in enumerate(np.asarray(costs, dtype=np.int64)):
            a, b = y0 + 32 * t, min(y0 + 32 * t + 32, y0 + rows)
            if b > a:
                out[a:b] = (int(c) * 1024) // (b - a)
        return out

    @staticmethod
    def scale_to(mine, measured_ms):
        """the band's modelled row costs, rescaled so that they add up to the band's MEASURED GPU time (in
        nanoseconds): the model supplies the shape inside a band, the measurement the level — whatever the
        model leaves out (the geometry stage every rank repeats, fragments per triangle) lands in the scale"""
        total = int(mine.sum())
        if measured_ms is None or measured_ms <= 0 or total == 0:
            return mine
        ns = int(measured_ms * 1e6)
        return (mine * ns) // total

    def recut(self, profile):
        """new boundaries from the frame's whole per-row profile (the same integers on every rank)"""
        profile = np.asarray(profile, dtype=np.int64)
        if int(profile.sum()) == 0:
            return False
        # rows nobody reported (a rank whose first frames have not been validated yet) count as average rows
        covered = profile > 0
        if not covered.all():
            profile = np.where(covered, profile, max(1, int(profile[covered].mean())))
        new = balanced_bounds(profile, self.world)
        eq = equal_bounds(self.height, self.world)
        heaviest = lambda b: max(int(profile[x:y].sum()) for x, y in zip(b, b[1:]))
        if heaviest(new) > (1.0 - self.min_gain) * heaviest(eq):
            new = eq
        changed = new != self.bounds
        self.bounds = new
        self.updates += 1
        return changed

    def rebalance(self, torch, dist, renderer, device, measured_ms=None):
        """Collective (every rank calls it at the same frame): re-cut the frame from the ranks' latest row costs,
        each band's costs scaled to its measured GPU time when the caller has one (all ranks or none) — the tile stage's
        (SvrStats.tile_ms): geometry and binning are the same for every band and would only tilt the rows' costs.
        Returns True when the boundaries changed."""
        if not self.balanced or self.world == 1:
            return False
        costs, y0, rows = renderer.row_costs()
        mine = self.scale_to(self.spread(costs, y0, rows, self.height), measured_ms)
        t = torch.from_numpy(mine).to(device)
        dist.all_reduce(t)  # bands are disjoint: the sum IS the frame's profile
        return self.recut(t.cpu().numpy())


class ShardedFrame:
    """One frame slot: full-frame colour/depth tensors, this rank's band, and the scissor that makes the
    renderer fill exactly that band.  plan=None: the fixed equal bands of round 1."""

    def __init__(self, torch, renderer, rank, world, device, color_format, bind=True, present=True, plan=None):
        from . import abi
        self.torch, self.r, self.rank, self.world = torch, renderer, rank, world
        self.W, self.H = renderer.width, renderer.height
        self.plan = plan if plan is not None else BandPlan(self.H, world, balanced=False)
        band = int(math.ceil(self.H / world))
        hp = band * world  # padded height: with equal bands every rank's chunk of the all-gather has `band` rows
        self.band = band
        cdtype = torch.float16 if color_format == abi.COLOR_RGBA16F else torch.uint8
        self.color = torch.zeros((hp, self.W, 4), dtype=cdtype, device=device)
        self.depth = torch.zeros((hp, self.W), dtype=torch.float32, device=device)
        self.present = present
        # what travels: the swapchain image (uint8 BGRA) or the colour target itself
        self.swapchain = torch.zeros((hp, self.W, 4), dtype=torch.uint8, device=device) if present else None
        self.image_t = self.swapchain if present else self.color
        self.flat = self.image_t.view(-1)
        self.bound = bind
        self.work = None
        self._dist = None
        self._replays = 0  # renderer's replayed_passes as of the last finish()
        self._ops, self._ops_key = [], None
        self.y0, self.rows = self.plan.rows_of(rank)
        self.bounds = list(self.plan.bounds)  # the partition this slot's frame in flight was rendered with

    def begin(self):
        """Make this slot the render target; waits (on the stream) for the slot's previous exchange."""
        self._wait()
        self.bounds = list(self.plan.bounds)
        self.y0, self.rows = self.bounds[self.rank], self.bounds[self.rank + 1] - self.bounds[self.rank]
        if self.bound:
            self.r.bind_targets(self.color.data_ptr(), self.depth.data_ptr())
        if self.rows > 0:
            self.r.set_scissor(0, self.y0, self.W, self.rows)

    def _wait(self):
        if self.work is not None:
            for w in (self.work if isinstance(self.work, list) else [self.work]):
                w.wait()
            self.work = None

    def _exchange(self, dist, async_op):
        """every rank's rows -> every rank's frame, in place"""
        equal = self.bounds == equal_bounds(self.H, self.world)
        if equal:
            n = self.band * self.W * 4
            h = dist.all_gather_into_tensor(self.flat, self.flat[self.rank * n:(self.rank + 1) * n], async_op=async_op)
            return h if async_op else None
        if self.image_t.is_cuda and dist.get_backend() == "gloo":
            # rehearsals on a one-GPU box (SVR_BENCH_REHEARSE): gloo moves device tensors only through its collectives,
            # not point to point — the bands travel padded to the tallest one (a list all_gather) and are copied into place
            tallest = max(b - a for a, b in zip(self.bounds, self.bounds[1:]))
            pad = self.torch.zeros((tallest, self.W, 4), dtype=self.image_t.dtype, device=self.image_t.device)
            rows = self.bounds[self.rank + 1] - self.bounds[self.rank]
            pad[:rows].copy_(self.image_t[self.bounds[self.rank]:self.bounds[self.rank + 1]])
            parts = [self.torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(parts, pad)
            for peer in range(self.world):
                n = self.bounds[peer + 1] - self.bounds[peer]
                if peer != self.rank and n:
                    self.image_t[self.bounds[peer]:self.bounds[peer + 1]].copy_(parts[peer][:n])
            return None
        key = tuple(self.bounds)
        if self._ops_key != key:  # the operation list only changes with the partition: built once, reused every frame
            ops = []
            mine = self.image_t[self.bounds[self.rank]:self.bounds[self.rank + 1]]
            for peer in range(self.world):
                if peer == self.rank:
                    continue
                if mine.shape[0] > 0:
                    ops.append(dist.P2POp(dist.isend, mine, peer))
                theirs = self.image_t[self.bounds[peer]:self.bounds[peer + 1]]
                if theirs.shape[0] > 0:
                    ops.append(dist.P2POp(dist.irecv, theirs, peer))
            self._ops, self._ops_key = ops, key
        works = dist.batch_isend_irecv(self._ops) if self._ops else []
        if async_op:
            return works
        for w in works:
            w.wait()
        return None

    def gather(self, dist, async_op=True):
        """Exchange the finished bands; in place (a rank's band is its rows of the full frame)."""
        if self.present:
            if self.rows > 0:  # vkutil::copy_image of this rank's rows (the scissor is still the band)
                self.r.copy_to_swapchain(self.swapchain.data_ptr(), self.W, self.H, 0)
        elif not self.bound:  # test path (CPU oracle owns its targets): copy the band out first
            col = self.r.read_color()
            t = self.torch.from_numpy(col.view(np.float16) if col.dtype == np.uint16 else col)
            self.color[self.y0:self.y0 + self.rows].copy_(t[self.y0:self.y0 + self.rows])
        if self.world == 1:
            return
        self._dist = dist
        self.work = self._exchange(dist, async_op)

    def _replayed_passes(self):
        try:
            return int(self.r.get_stats().replayed_passes)  # fences the renderer
        except AttributeError:  # the CPU oracle (tests) has no queues to overflow
            return 0

    def finish(self):
        """Wait for the slot's exchange.  The exchange is the caller's own stream work, which the renderer's
        overflow replay (svr_api.hip, operation log) does not know about: if a pass of this frame was
        replayed after the band had been sent, send the band again."""
        self._wait()
        if self.world > 1 and self.bound and self._dist is not None:
            now = self._replayed_passes()  # a fence, like this whole call
            flag = self.torch.tensor([1 if now != self._replays else 0], dtype=self.torch.int32, device=self.color.device)
            self._dist.all_reduce(flag, op=self._dist.ReduceOp.MAX)  # the re-send is a collective: all ranks or none
            self._replays = now
            if int(flag.item()):
                if self.present and self.rows > 0:  # the replayed rows have to be presented again first
                    self.r.bind_targets(self.color.data_ptr(), self.depth.data_ptr())
                    self.r.set_scissor(0, self.y0, self.W, self.rows)
                    self.r.copy_to_swapchain(self.swapchain.data_ptr(), self.W, self.H, 0)
                    self.r.sync()
                self._exchange(self._dist, False)

    def image(self):
        """The gathered frame without the padding rows (B8G8R8A8 swapchain bytes, or the colour target)."""
        return self.image_t[:self.H]
