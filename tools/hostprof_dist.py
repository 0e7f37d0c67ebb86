"""Developer tool: host time per frame of the sharded frame's loop as bench.py --gpus N drives it (dist.ShardedFrame:
begin, clear, draw, present, exchange), with the collective stubbed out and a GPU load too small to matter — what a
rank of eight has to enqueue in 60 us.

    python tools/hostprof_dist.py [--partition interleaved|bands] [--frames 4000]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


class _Work:
    def wait(self):
        pass


class _StubDist:
    """the calls ShardedFrame makes of torch.distributed, doing nothing"""
    class ReduceOp:
        MAX = 0

    def get_backend(self):
        return "stub"

    def all_gather_into_tensor(self, out, inp, async_op=False):
        return _Work() if async_op else None

    def all_reduce(self, t, op=None):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partition", default="interleaved")
    ap.add_argument("--frames", type=int, default=4000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    import torch
    pkg = g.load_package()
    S, A, D = pkg.scenes, pkg.abi, pkg.dist
    hip = pkg.load_product_library()
    dev = torch.device("cuda", 0)
    W, H = 3840, 2160
    sc = S.sponza_like(lod=8, tex_size=32)  # a GPU load that does not matter; the 338 RenderObjects are all there
    r = hip.create(W, H, A.COLOR_RGBA16F)
    r.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    plan = D.BandPlan(H, args.world, balanced=False)
    plan.partition = args.partition
    slots = [D.ShardedFrame(torch, r, 3, args.world, dev, A.COLOR_RGBA16F, present=True, plan=plan, verify="fence") for _ in range(2)]
    handles = sc.upload(r)
    opaque, transparent = sc.render_objects(handles)
    scene = S.scene_data_struct(*S.config3_camera(), W, H)
    dist = _StubDist()
    state = {"i": 0}

    def frame():
        s = slots[state["i"] & 1]
        state["i"] += 1
        s.begin()
        r.clear_color((1.0, 1.0, 1.0, 1.0))
        if s.rows:
            r.draw_geometry(scene, opaque, transparent)
        s.gather(dist)

    for _ in range(200):
        frame()
    r.sync()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        frame()
    dt = (time.perf_counter() - t0) / args.frames
    r.sync()
    print(f"{args.partition}, rank 3 of {args.world}: {dt * 1e6:.1f} us of host time per frame ({len(opaque) + len(transparent)} objects)")
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            frame()
        pr.disable()
        r.sync()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    r.close()


if __name__ == "__main__":
    main()
