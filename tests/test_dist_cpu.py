"""The N>1 path on CPU: gloo runs (world size 2 and 4) of the row-band sharding + in-place exchange
(simple-vk-renderer_amd/dist.py), with the CPU oracle standing in for the renderer (test only).
The gathered frame must equal the single-process full-frame render bit for bit: equal bands (one
all-gather), a height that does not divide by the world size, cost-balanced unequal bands (batched
point-to-point), a rank whose band is empty, and the collective that re-cuts the frame."""
import os
import socket
import sys

import numpy as np
import pytest

import __graft_entry__ as g
import svr_testlib as T

pkg = g.load_package()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_dir, present, bounds=None, rebalance=False, partition="bands"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(g.ROOT, "tests"))
        ora = T.load_oracle()
        D = g.load_package().dist
        r, scene, opaque, transparent = T.setup_sponza(ora, width, height, lod=8, tex_size=32)
        plan = D.BandPlan(height, world, balanced=rebalance, min_gain=0.0)
        if bounds is not None:
            plan.bounds = list(bounds)
        if partition == "pick":  # the collective choice: this rank found bands cheaper, rank 1 found them dearer — the slowest rank decides
            assert plan.pick(torch, dist, torch.device("cpu"), 1.0 if rank == 0 else 5.0, 3.0) == "interleaved"
            assert plan.pick(torch, dist, torch.device("cpu"), 1.0, 3.0 if rank == 0 else 0.5) == "bands"
            assert plan.pick(torch, dist, torch.device("cpu"), 4.0, 3.0 if rank == 0 else 0.5) == "interleaved"
        else:
            plan.partition = partition
        slots = [D.ShardedFrame(torch, r, rank, world, torch.device("cpu"), pkg.abi.COLOR_RGBA16F, bind=False, present=present, plan=plan)
                 for _ in range(2)]
        for f in range(3):  # three frames through two slots, asynchronous gathers
            s = slots[f % 2]
            if rebalance and f == 2:  # the oracle weighs every tile row alike (6 tile rows here: 4 + 2): the re-cut moves the boundary up
                before = list(plan.bounds)
                assert plan.rebalance(torch, dist, r, torch.device("cpu")) and plan.updates == 1
                assert plan.bounds != before and plan.bounds[0] == 0 and plan.bounds[-1] == height and plan.bounds[1] < before[1]
            s.begin()
            r.clear_color((1, 1, 1, 1))
            if s.rows:
                r.draw_geometry(scene, opaque, transparent)
            r.sync()
            s.gather(dist, async_op=True)
        for s in slots:
            s.finish()
        img = slots[0].image().numpy()
        img = img if present else img.view(np.uint16)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), img)
        r.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size,present", [((96, 54), True), ((80, 45), True), ((80, 45), False)])
def test_band_allgather_world2(tmp_path, oracle, size, present):
    """present: the B8G8R8A8 swapchain image travels (bench default); else the RGBA16F target."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = size
    mp.spawn(_worker, args=(2, _free_port(), w, h, str(tmp_path), present), nprocs=2, join=True)
    full = T.render_sponza(oracle, w, h, lod=8, tex_size=32)
    ref = full["rgba8"][..., [2, 1, 0, 3]] if present else full["color"]
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"rank {rank}: gathered frame differs from the full-frame render"


@pytest.mark.parametrize("world,bounds", [(2, (0, 13, 54)), (4, (0, 7, 7, 40, 54)), (4, (0, 0, 30, 31, 54)), (3, (0, 54, 54, 54))])
def test_unequal_bands_travel_point_to_point(tmp_path, oracle, world, bounds):
    """Cost-balanced partitions give unequal bands, possibly empty ones: every rank still ends up with the frame."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = 96, 54
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path), True, bounds), nprocs=world, join=True)
    full = T.render_sponza(oracle, w, h, lod=8, tex_size=32)
    ref = full["rgba8"][..., [2, 1, 0, 3]]
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), ref), f"rank {rank} with bounds {bounds}"


@pytest.mark.parametrize("world,size,present,partition", [(2, (96, 54), True, "interleaved"), (3, (80, 100), True, "interleaved"),
                                                          (4, (64, 70), True, "interleaved"), (2, (80, 100), False, "interleaved"),
                                                          (2, (96, 135), True, "pick")])
def test_interleaved_tile_rows_travel_as_grouped_allgathers(tmp_path, oracle, world, size, present, partition):
    """rank r renders the tile rows t with t % world == r; every group of `world` tile rows is one in-place all-gather.
    54 rows = two tile rows for two ranks; 100 rows = four for three (the last 4 rows tall); 70 rows = three for four (a
    rank with nothing); the RGBA16F target instead of the swapchain image; BandPlan.pick's collective choice."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = size
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path), present, None, False, partition), nprocs=world, join=True)
    full = T.render_sponza(oracle, w, h, lod=8, tex_size=32)
    ref = full["rgba8"][..., [2, 1, 0, 3]] if present else full["color"]
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), ref), f"rank {rank} of {world}, {partition}"


def test_rebalance_is_a_collective_that_keeps_the_frame(tmp_path, oracle):
    """Skewed bands to begin with, re-cut between frames from the ranks' reported row costs (all_reduce of the profile)."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = 64, 135
    mp.spawn(_worker, args=(2, _free_port(), w, h, str(tmp_path), True, (0, 100, 135), True), nprocs=2, join=True)
    full = T.render_sponza(oracle, w, h, lod=8, tex_size=32)
    ref = full["rgba8"][..., [2, 1, 0, 3]]
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), ref)


def test_balanced_bounds_minimise_the_heaviest_band():
    D = pkg.dist
    rng = np.random.default_rng(5)
    for trial in range(200):
        h = int(rng.integers(1, 14))
        world = int(rng.integers(1, 6))
        c = rng.integers(0, 50, h) * (rng.random(h) < 0.7)
        b = D.balanced_bounds(c, world)
        assert b[0] == 0 and b[-1] == h and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        got = max(int(c[x:y].sum()) for x, y in zip(b, b[1:]))
        # brute force over all cuts
        import itertools
        best = min(max(int(c[x:y].sum()) for x, y in zip((0,) + cut, cut + (h,)))
                   for cut in itertools.combinations_with_replacement(range(h + 1), world - 1))
        assert got == best, (c.tolist(), world, b)
    # a 4K-like profile: empty top, heavy middle
    prof = np.concatenate([np.zeros(700, np.int64), np.full(800, 900), np.full(660, 150)])
    b = D.balanced_bounds(prof, 8)
    costs = [int(prof[x:y].sum()) for x, y in zip(b, b[1:])]
    assert max(costs) <= 1.02 * (prof.sum() / 8) + 900
    # the plan keeps the equal bands (one all-gather) unless the balanced cut shortens the heaviest band enough
    mild = np.concatenate([np.full(1080, 100), np.full(1080, 115)]).astype(np.int64)
    plan = D.BandPlan(2160, 8, min_gain=0.2)
    assert not plan.recut(mild) and plan.bounds == D.equal_bounds(2160, 8)
    steep = np.where(prof > 0, prof, 1)  # (a zero means "nobody reported this row" to recut: real rows always cost something)
    assert plan.recut(steep) and plan.bounds == D.balanced_bounds(steep, 8) != D.equal_bounds(2160, 8)
    assert plan.recut(mild) and plan.bounds == D.equal_bounds(2160, 8)   # and goes back when the frame evens out
    # spread(): tile-row costs become per-row costs, clipped to the band
    rows = D.BandPlan.spread([64, 32], 10, 40, 100)
    assert rows[:10].sum() == 0 and rows[50:].sum() == 0 and rows[10] == 64 * 1024 // 32 and rows[42] == 32 * 1024 // 8


def test_band_rows_partition():
    D = pkg.dist
    for h in (1, 31, 32, 33, 90, 2160, 4320):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += [y for y0, n in D.interleaved_rows(h, r, world) for y in range(y0, y0 + n)]
            assert sorted(seen) == list(range(h))
            hp = D.padded_height(h, world)
            assert hp >= h and hp % world == 0 and hp >= -(-((h + 31) // 32) // world) * world * 32
    for h in (1, 7, 54, 55, 2160, 4320, 1080):
        for world in (1, 2, 3, 4, 8):
            rows = [D.band_rows(h, r, world) for r in range(world)]
            band = rows[0][2]
            assert all(b == band for _, _, b in rows) and band * world >= h
            covered = []
            for y0, n, _ in rows:
                covered += list(range(y0, y0 + n))
            assert covered == list(range(h))  # every row exactly once, in order
