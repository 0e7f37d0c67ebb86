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

The other partition SURVEY 8e names — interleaved tile rows, rank r owning the 32-row tile rows t with
t % world == r (svr_set_row_interleave) — is balanced by construction (deep and shallow regions are dealt out
evenly) at the price of every rank running the whole scene's vertex stage.  Its rows travel as one in-place
all-gather per `world` consecutive tile rows (coalesced into one launch under RCCL).  BandPlan.pick chooses
between the two from measured times, collectively.

A presented image is a finished one (src/vk_engine.cpp:1226, 1332): a pass that overflows the renderer's queues is
void and replayed later, the present behind it too, but the exchange is this module's own stream work.  Every
present therefore reports into a status word (svr_set_present_status) that is gathered behind the rows; a slot
whose gathered words are not all 0 is fenced and exchanged again before it is reused or handed out — on every rank
alike, because every rank reads the same words.
"""
import math

import numpy as np


def band_rows(height, rank, world):
    """(first_row, n_rows_rendered, band_height): equal bands of ceil(H/world) rows; the last may be short."""
    band = int(math.ceil(height / world))
    y0 = min(rank * band, height)
    y1 = min(y0 + band, height)
    return y0, y1 - y0, band


def interleaved_rows(height, rank, world):
    """[(first_row, n_rows)] of the 32-row tile rows t with t % world == rank"""
    return [(t * 32, min(32, height - t * 32)) for t in range(rank, (height + 31) // 32, world)]


def padded_height(height, world):
    """rows of the buffers that travel: whole equal bands AND whole groups of `world` tile rows"""
    band = int(math.ceil(height / world))
    groups = int(math.ceil(((height + 31) // 32) / world))
    return max(band * world, groups * world * 32)


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
        self.partition = "bands"  # or "interleaved"
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

    def pick(self, torch, dist, device, bands_ms, interleaved_ms):
        """Collective: every rank hands in what a frame cost it under either partition; the frame is as slow as its
        slowest rank, so the partition with the smaller maximum is taken on every rank alike.  Returns its name."""
        t = torch.tensor([float(bands_ms), float(interleaved_ms)], dtype=torch.float64, device=device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        b, i = (float(v) for v in t.tolist())
        self.partition = "interleaved" if i < b else "bands"
        return self.partition

    def rebalance(self, torch, dist, renderer, device, measured_ms=None):
        """Collective (every rank calls it at the same frame): re-cut the frame from the ranks' latest row costs,
        each band's costs scaled to its measured GPU time when the caller has one (all ranks or none) — the tile stage's
        (SvrStats.tile_ms): geometry and binning are the same for every band and would only tilt the rows' costs.
        Returns True when the boundaries changed."""
        if not self.balanced or self.world == 1 or self.partition == "interleaved":  # (balanced by construction)
            return False
        costs, y0, rows = renderer.row_costs()
        mine = self.scale_to(self.spread(costs, y0, rows, self.height), measured_ms)
        t = torch.from_numpy(mine).to(device)
        dist.all_reduce(t)  # bands are disjoint: the sum IS the frame's profile
        return self.recut(t.cpu().numpy())


class ShardedFrame:
    """One frame slot: full-frame colour/depth tensors, this rank's band, and the scissor that makes the
    renderer fill exactly that band.  plan=None: the fixed equal bands of round 1."""

    def __init__(self, torch, renderer, rank, world, device, color_format, bind=True, present=True, plan=None, verify="frame"):
        from . import abi
        self.torch, self.r, self.rank, self.world = torch, renderer, rank, world
        self.W, self.H = renderer.width, renderer.height
        self.plan = plan if plan is not None else BandPlan(self.H, world, balanced=False)
        band = int(math.ceil(self.H / world))
        hp = padded_height(self.H, world)  # with equal bands every rank's chunk of the all-gather has `band` rows
        self.band = band
        cdtype = torch.float16 if color_format == abi.COLOR_RGBA16F else torch.uint8
        self.color = torch.zeros((hp, self.W, 4), dtype=cdtype, device=device)
        self.depth = torch.zeros((hp, self.W), dtype=torch.float32, device=device)
        self.present = present
        # what travels: the swapchain image (uint8 BGRA) or the colour target itself
        self.swapchain = torch.zeros((hp, self.W, 4), dtype=torch.uint8, device=device) if present else None
        self.image_t = self.swapchain if present else self.color
        self.flat = self.image_t.view(-1)
        # (addresses asked for once: tensor.data_ptr() per frame is host time of a 60-us frame)
        self._color_ptr, self._depth_ptr = self.color.data_ptr(), self.depth.data_ptr()
        self._swap_ptr = self.swapchain.data_ptr() if present else 0
        self.bound = bind
        self.work = None
        self._dist = None
        self._ops, self._ops_key = [], None
        self._il_parts, self._eq_parts = None, None  # cached views of the image for the two all-gather forms
        self._equal = equal_bounds(self.H, world)
        self.y0, self.rows = self.plan.rows_of(rank)
        self.bounds = list(self.plan.bounds)  # the partition this slot's frame in flight was rendered with
        self.partition = self.plan.partition
        # every rank's present status of the frame in flight (svr_set_present_status; own word: [rank]) and its
        # host copy, made behind the exchange on a side stream so that reading it never stalls the render stream
        self.status = torch.zeros(world, dtype=torch.int32, device=device)
        self.on_gpu = self.status.is_cuda
        self.h_status = torch.zeros(world, dtype=torch.int32).pin_memory() if self.on_gpu else self.status
        self.side = torch.cuda.Stream(device=device) if self.on_gpu else None
        self.status_ready = torch.cuda.Event() if self.on_gpu else None
        self.exchanged_again = 0  # frames this slot had to exchange a second time
        # verify = "frame": every exchange carries the ranks' status words and every slot is checked before it is reused
        # or handed out (what libsvr_dist.so does, where the second collective costs a few microseconds of enqueue).
        # verify = "fence": nothing per frame — from Python a second collective and its host copy cost 20-30 us against
        # 60-us frames; finish() (a fence) then compares the renderer's replayed_passes with the last fence's, all-reduces
        # the verdict and presents + exchanges the slot again.  Frames between a void pass and the next fence may have
        # carried stale rows: for callers that hand out frames only at fences (bench.py).
        assert verify in ("frame", "fence")
        self.verify = verify
        self._replays = 0  # verify = "fence": the renderer's replayed_passes as of the last finish()

    def begin(self):
        """Make this slot the render target; waits (on the stream) for the slot's previous exchange."""
        self._wait()
        self._repair()  # the frame this slot held is final before the slot is used again
        self.bounds = list(self.plan.bounds)
        self.partition = self.plan.partition
        if self.bound:
            self.r.bind_targets(self._color_ptr, self._depth_ptr)
        if self.partition == "interleaved":
            self.y0, self.rows = 0, sum(n for _, n in interleaved_rows(self.H, self.rank, self.world))
            self._rows_state((0, 0, self.W, self.H), (self.world, self.rank))
            return
        self.y0, self.rows = self.bounds[self.rank], self.bounds[self.rank + 1] - self.bounds[self.rank]
        self._rows_state((0, self.y0, self.W, self.rows) if self.rows > 0 else None, (1, 0))

    def _rows_state(self, scissor, interleave):
        """scissor + row interleave of the renderer, set only when they change (every call is host time of a frame)"""
        last = getattr(self.r, "_sharded_rows", (None, None))
        if interleave != last[1]:
            self.r.set_row_interleave(*interleave)
        if scissor is not None and scissor != last[0]:
            self.r.set_scissor(*scissor)
        self.r._sharded_rows = (scissor if scissor is not None else last[0], interleave)

    def _wait(self):
        if self.work is not None:
            for w in (self.work if isinstance(self.work, list) else [self.work]):
                w.wait()
            self.work = None

    def _exchange(self, dist, async_op):
        """every rank's rows -> every rank's frame, in place; every rank's status word behind them"""
        works = self._exchange_rows(dist, async_op)
        if self.present and self.verify == "frame":
            h = dist.all_gather_into_tensor(self.status, self.status[self.rank:self.rank + 1], async_op=async_op)
            if async_op:
                works = (works if isinstance(works, list) else ([works] if works is not None else [])) + [h]
            if self.on_gpu:  # host copy of the words on the side stream, behind the exchange
                self.side.wait_stream(self.torch.cuda.current_stream(self.status.device))
                with self.torch.cuda.stream(self.side):
                    for w in (works or []):
                        w.wait()
                    self.h_status.copy_(self.status, non_blocking=True)
                    self.status_ready.record()
        return works if async_op else None

    def _exchange_rows(self, dist, async_op):
        if self.partition == "interleaved":
            # tile row t belongs to rank t % world: every group of `world` consecutive tile rows is one in-place
            # all-gather of 32-row chunks (the buffers are padded to whole groups); RCCL: coalesced into one launch
            parts = self._il_parts
            if parts is None:  # the views never change: made once (eighteen tensor slices per frame were 30-40 us of host time)
                chunk = 32 * self.W * self.image_t.shape[2] * self.image_t.element_size()
                flat = self.image_t.view(self.torch.uint8).view(-1) if self.image_t.dtype != self.torch.uint8 else self.flat
                groups = int(math.ceil(((self.H + 31) // 32) / self.world))
                parts = self._il_parts = [(flat[g * self.world * chunk:(g + 1) * self.world * chunk],
                                           flat[(g * self.world + self.rank) * chunk:(g * self.world + self.rank + 1) * chunk]) for g in range(groups)]
            if self.on_gpu and dist.get_backend() == "nccl":
                # (no `device=`: the manager documents that argument for backends WITHOUT a coalesced all-gather; RCCL has
                # one, and the handle waited on below should be that operation's own)
                with dist._coalescing_manager(async_ops=async_op) as cm:
                    for out, mine in parts:
                        dist.all_gather_into_tensor(out, mine)
                if async_op and not cm.works:
                    raise RuntimeError("the coalesced all-gather of the interleaved rows returned no handle to wait on")
                return [cm] if async_op else None
            works = [dist.all_gather_into_tensor(out, mine, async_op=async_op) for out, mine in parts]
            return works if async_op else None
        equal = self.bounds == self._equal
        if equal:
            if self._eq_parts is None:
                n = self.band * self.W * 4
                self._eq_parts = (self.flat[:self.world * n], self.flat[self.rank * n:(self.rank + 1) * n])
            h = dist.all_gather_into_tensor(self._eq_parts[0], self._eq_parts[1], async_op=async_op)
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
            if self.rows > 0:  # vkutil::copy_image of this rank's rows (scissor / row set are still the frame's)
                if self.verify == "frame":
                    self.r.set_present_status(self.status[self.rank:].data_ptr())
                self.r.copy_to_swapchain(self._swap_ptr, self.W, self.H, 0)
                if self.verify == "frame":
                    self.r.set_present_status(0)
        elif not self.bound:  # test path (CPU oracle owns its targets): copy the band out first
            col = self.r.read_color()
            t = self.torch.from_numpy(col.view(np.float16) if col.dtype == np.uint16 else col)
            runs = interleaved_rows(self.H, self.rank, self.world) if self.partition == "interleaved" else [(self.y0, self.rows)]
            for y, n in runs:
                self.color[y:y + n].copy_(t[y:y + n])
        if self.world == 1:
            return
        self._dist = dist
        self.work = self._exchange(dist, async_op)

    def _repair(self):
        """The frame this slot last exchanged, made final: if any rank's present of it was void (1) or rerun by the
        replay, possibly under the exchange (2), fence the renderer — the owner's replay runs there — and exchange the
        slot again.  Collective without a message of its own: every rank reads the same gathered words."""
        if self.world == 1 or not self.present or self._dist is None or self.verify != "frame":
            return
        for attempt in range(5):
            if self.on_gpu:
                self.status_ready.synchronize()
            words = self.h_status.tolist()
            if not any(words):
                return
            if attempt == 4:
                raise RuntimeError("a rank's present stayed void after four exchanges")
            self.r.sync()
            if words[self.rank]:
                self.status[self.rank] = 0
            self._exchange(self._dist, False)
            self.exchanged_again += 1
            if self.on_gpu:
                self.torch.cuda.current_stream(self.status.device).synchronize()

    def finish(self):
        """Wait for the slot's exchange and make its frame final (see _repair)."""
        self._wait()
        self._repair()
        if self.verify == "fence" and self.world > 1 and self.bound and self._dist is not None:
            try:
                now = int(self.r.get_stats().replayed_passes)  # fences the renderer: a replay that was due has run
            except AttributeError:  # the CPU oracle (tests) has no queues to overflow
                now = 0
            flag = self.torch.tensor([1 if now != self._replays else 0], dtype=self.torch.int32, device=self.color.device)
            self._dist.all_reduce(flag, op=self._dist.ReduceOp.MAX)  # the re-send is a collective: all ranks or none
            self._replays = now
            if int(flag.item()):
                if self.present and self.rows > 0:  # the replayed rows: presented again (the replay's own present may have
                    self.r.bind_targets(self.color.data_ptr(), self.depth.data_ptr())  # gone to the other slot's image)
                    self.r._sharded_rows = (None, None)
                    self._rows_state((0, 0, self.W, self.H) if self.partition == "interleaved" else (0, self.y0, self.W, self.rows),
                                     (self.world, self.rank) if self.partition == "interleaved" else (1, 0))
                    self.r.copy_to_swapchain(self.swapchain.data_ptr(), self.W, self.H, 0)
                    self.r.sync()
                self._exchange(self._dist, False)
                self.exchanged_again += 1

    def image(self):
        """The gathered frame without the padding rows (B8G8R8A8 swapchain bytes, or the colour target)."""
        return self.image_t[:self.H]
