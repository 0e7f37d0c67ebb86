"""The N>1 path on CPU: world_size-2 gloo run of the row-band sharding + in-place all-gather
(simple-vk-renderer_amd/dist.py), with the CPU oracle standing in for the renderer (test only).
The gathered frame must equal the single-process full-frame render bit for bit, also when the
height does not divide by the world size."""
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


def _worker(rank, world, port, width, height, out_dir, present):
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
        slots = [D.ShardedFrame(torch, r, rank, world, torch.device("cpu"), pkg.abi.COLOR_RGBA16F, bind=False, present=present)
                 for _ in range(2)]
        for f in range(3):  # three frames through two slots, asynchronous gathers
            s = slots[f % 2]
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


def test_band_rows_partition():
    D = pkg.dist
    for h in (1, 7, 54, 55, 2160, 4320, 1080):
        for world in (1, 2, 3, 4, 8):
            rows = [D.band_rows(h, r, world) for r in range(world)]
            band = rows[0][2]
            assert all(b == band for _, _, b in rows) and band * world >= h
            covered = []
            for y0, n, _ in rows:
                covered += list(range(y0, y0 + n))
            assert covered == list(range(h))  # every row exactly once, in order
