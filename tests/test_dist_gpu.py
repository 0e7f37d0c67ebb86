"""The N>1 path with the HIP library: two processes on the one GPU of the test box (gloo stands in for
RCCL, which refuses two ranks on one device), caller-bound targets, the present step and the in-place
all-gather of the swapchain image through two frame slots.  The gathered frame must be the single-process
frame's swapchain image, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

import __graft_entry__ as g
import svr_testlib as T

pkg = g.load_package()
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(g.ROOT, "tests"))
        P = g.load_package()
        hip = P.load_product_library()
        dev = torch.device("cuda", 0)
        r, scene, opaque, transparent = T.setup_sponza(hip, width, height, lod=8, tex_size=32)
        r.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        slots = [P.dist.ShardedFrame(torch, r, rank, world, dev, P.abi.COLOR_RGBA16F) for _ in range(2)]
        for f in range(5):  # five frames through two slots, nothing fenced in between
            s = slots[f % 2]
            s.begin()
            r.clear_color((1, 1, 1, 1))
            if s.rows:
                r.draw_geometry(scene, opaque, transparent)
            s.gather(dist, async_op=True)
        for s in slots:
            s.finish()
        r.sync()
        torch.cuda.synchronize(dev)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), slots[0].image().cpu().numpy())
        r.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [(192, 108), (160, 91)])
def test_two_ranks_present_and_gather(tmp_path, hip, size):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = size
    mp.spawn(_worker, args=(2, _free_port(), w, h, str(tmp_path)), nprocs=2, join=True)
    r, scene, opaque, transparent = T.setup_sponza(hip, w, h, lod=8, tex_size=32)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)
    ref = r.read_swapchain(w, h, pkg.abi.SWAPCHAIN_B8G8R8A8)
    r.close()
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"rank {rank}: gathered swapchain image differs from the single-process frame"
