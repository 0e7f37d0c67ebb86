"""Multi-GPU form of the draw path: one process per GPU, the frame's rows sharded across ranks, the
finished bands exchanged with ONE all-gather per frame (torch.distributed backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The reference is a single-device, single-queue renderer (deviceIndex 0, src/vk_engine.cpp:1295), so
nothing here translates reference code; it is the screen-space decomposition of SURVEY.md §8e:
  - scene (meshes, textures, materials) replicated on every rank — Sponza-scale is ~150 MB of 288 GB;
  - rank r owns rows [r*band, (r+1)*band) with band = ceil(H / world): svr_set_scissor clips geometry,
    binning and the tile grid to the band, pixels of a band never depend on another band;
  - each rank renders straight into its rows of a full-frame _draw_image tensor, presents them
    (svr_copy_to_swapchain, identity extent: the scissor's rows) into its slice of a full-frame
    B8G8R8A8 swapchain tensor, then all_gather_into_tensor(swapchain, my_band) in place: every rank
    ends up with the whole presentable frame at 4 bytes per pixel, as SURVEY §8e sizes the exchange
    (present=False gathers the RGBA16F _draw_image itself, 8 bytes per pixel);
  - two frames in flight (the reference keeps FRAME_OVERLAP = 3, src/vk_engine.h:77): the gather of
    frame i runs on the collective's stream while frame i+1 renders.
xGMI is point-to-point (7 links per GPU): an all-gather of equal bands is one hop per peer and the
per-link load is band_bytes, so bands are kept equal and contiguous — the exact precondition of
all_gather_into_tensor — instead of interleaved tile rows that would need a second un-permute pass.
"""
import math

import numpy as np


def band_rows(height, rank, world):
    """(first_row, n_rows_rendered, band_height): equal bands of ceil(H/world) rows; the last may be short."""
    band = int(math.ceil(height / world))
    y0 = min(rank * band, height)
    y1 = min(y0 + band, height)
    return y0, y1 - y0, band


class ShardedFrame:
    """One frame slot: full-frame colour/depth tensors padded to world*band rows, this rank's band view,
    and the scissor that makes the renderer fill exactly that band."""

    def __init__(self, torch, renderer, rank, world, device, color_format, bind=True, present=True):
        from . import abi
        self.torch, self.r, self.rank, self.world = torch, renderer, rank, world
        self.W, self.H = renderer.width, renderer.height
        self.y0, self.rows, self.band = band_rows(self.H, rank, world)
        hp = self.band * world  # padded height: every rank's chunk of the gather has band rows
        cdtype = torch.float16 if color_format == abi.COLOR_RGBA16F else torch.uint8
        self.color = torch.zeros((hp, self.W, 4), dtype=cdtype, device=device)
        self.depth = torch.zeros((hp, self.W), dtype=torch.float32, device=device)
        self.present = present
        # what travels: the swapchain image (uint8 BGRA) or the colour target itself
        self.swapchain = torch.zeros((hp, self.W, 4), dtype=torch.uint8, device=device) if present else None
        self.flat = (self.swapchain if present else self.color).view(-1)
        n = self.band * self.W * 4
        self.my_band = self.flat[rank * n:(rank + 1) * n]
        self.bound = bind
        self.work = None
        self._dist = None
        self._replays = 0  # renderer's replayed_passes as of the last finish()

    def begin(self):
        """Make this slot the render target; waits (on the stream) for the slot's previous gather."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.bound:
            self.r.bind_targets(self.color.data_ptr(), self.depth.data_ptr())
        if self.rows > 0:
            self.r.set_scissor(0, self.y0, self.W, self.rows)

    def gather(self, dist, async_op=True):
        """Exchange the finished bands; in place (input is the rank's slice of the output)."""
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
        h = dist.all_gather_into_tensor(self.flat, self.my_band, async_op=async_op)
        self.work = h if async_op else None

    def _replayed_passes(self):
        try:
            return int(self.r.get_stats().replayed_passes)  # fences the renderer
        except AttributeError:  # the CPU oracle (tests) has no queues to overflow
            return 0

    def finish(self):
        """Wait for the slot's gather.  The gather is the caller's own stream work, which the renderer's
        overflow replay (svr_api.hip, operation log) does not know about: if a pass of this frame was
        replayed after the band had been sent, send the band again."""
        if self.work is not None:
            self.work.wait()
            self.work = None
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
                self._dist.all_gather_into_tensor(self.flat, self.my_band)

    def image(self):
        """The gathered frame without the padding rows (B8G8R8A8 swapchain bytes, or the colour target)."""
        return (self.swapchain if self.present else self.color)[:self.H]
