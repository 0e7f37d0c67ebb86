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


def _worker(rank, world, port, width, height, out_dir, partition="bands", queue_caps=None, verify="frame"):
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
        plan = P.dist.BandPlan(height, world, balanced=False)
        plan.partition = partition
        if queue_caps is not None:
            r.set_option(P.abi.OPT_QUEUE_CAPS, queue_caps)  # the first passes overflow, are void, and replayed later
            r.set_option(P.abi.OPT_TUNING, 16)              # ... found at a fence, not in passing: the present behind them is void
        slots = [P.dist.ShardedFrame(torch, r, rank, world, dev, P.abi.COLOR_RGBA16F, plan=plan, verify=verify) for _ in range(2)]
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
        np.save(os.path.join(out_dir, f"again{rank}.npy"), np.array([sum(s.exchanged_again for s in slots), r.get_stats().replayed_passes]))
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


def _reference(hip, w, h):
    r, scene, opaque, transparent = T.setup_sponza(hip, w, h, lod=8, tex_size=32)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)
    ref = r.read_swapchain(w, h, pkg.abi.SWAPCHAIN_B8G8R8A8)
    r.close()
    return ref


@pytest.mark.parametrize("world,size", [(2, (192, 108)), (3, (160, 91))])
def test_interleaved_tile_rows_present_and_gather(tmp_path, hip, world, size):
    """rank r renders tile rows t % world == r into its rows of the full frame, presents them in place, and every group
    of `world` tile rows is gathered in place (108 rows: four tile rows, the last 12 rows tall; 91 rows: three for three)"""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = size
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path), "interleaved"), nprocs=world, join=True)
    ref = _reference(hip, w, h)
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), ref), f"rank {rank} of {world}"


@pytest.mark.parametrize("partition,verify", [("bands", "frame"), ("interleaved", "frame"), ("bands", "fence")])
def test_a_replayed_pass_is_exchanged_again(tmp_path, hip, partition, verify):
    """queues start tiny: the first passes overflow, their presents are void and say so in the status words that travel
    behind the rows; every rank then fences and exchanges the slot again before it reuses it (dist.py _repair).
    verify = "fence": nothing travels per frame; finish() compares replayed_passes and repairs collectively"""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = 192, 108
    mp.spawn(_worker, args=(2, _free_port(), w, h, str(tmp_path), partition, 16, verify), nprocs=2, join=True)
    ref = _reference(hip, w, h)
    again = [np.load(tmp_path / f"again{rank}.npy") for rank in range(2)]
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), ref), f"rank {rank}"
    assert again[0][0] == again[1][0] >= 1, again          # the repair is collective
    assert max(a[1] for a in again) >= 1, again            # and there was something to repair


def _nccl_single(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sys.path.insert(0, os.path.join(g.ROOT, "tests"))
        P = g.load_package()
        hip = P.load_product_library()
        w, h = 192, 108
        r, scene, opaque, transparent = T.setup_sponza(hip, w, h, lod=8, tex_size=32)
        r.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        ref = r.read_swapchain(w, h, P.abi.SWAPCHAIN_B8G8R8A8)
        for partition in ("bands", "interleaved"):
            plan = P.dist.BandPlan(h, 1, balanced=False)
            plan.partition = partition
            s = P.dist.ShardedFrame(torch, r, 0, 1, dev, P.abi.COLOR_RGBA16F, plan=plan)
            for _ in range(3):
                s.begin()
                r.clear_color((1, 1, 1, 1))
                r.draw_geometry(scene, opaque, transparent)
                s.gather(dist)                 # world 1: presents, no exchange ...
                s._dist = dist
                s.work = s._exchange(dist, True)  # ... so the exchange is called by hand: the RCCL code path, one rank
                s.finish()
            torch.cuda.synchronize(dev)
            assert np.array_equal(s.image().cpu().numpy(), ref), partition
            assert s.h_status.tolist() == [0]
            assert plan.pick(torch, dist, dev, 2.0, 1.0) == "interleaved" and plan.pick(torch, dist, dev, 1.0, 2.0) == "bands"
        r.close()
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_the_rccl_code_path_with_one_rank(tmp_path, hip):
    """torch.distributed backend "nccl" (= RCCL) refuses two ranks on one device, so the calls bench.py --gpus N makes —
    the in-place all_gather_into_tensor of equal bands, the coalesced group of per-tile-row-group all-gathers of the
    interleaved partition, the status words' all-gather and its side-stream copy, the pick's all-reduce — run here with
    a world of one: the API use is exercised on hardware even though no peer exists"""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_nccl_single, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()
