"""Developer tool: what one GPU of an N-GPU run has to do, measured on one GPU.

Rank r of N renders the rows [r*H/N, (r+1)*H/N) (dist.py); this times every band of N = 1, 2, 4, 8 in
turn on the one GPU present (scissor = the band, present of the band included, no gather) and prints
the slowest band per N: the frame rate an N-GPU run cannot exceed however fast the gather is.
--interleaved does the same for the other partition (rank r = the tile rows t with t % N == r).

    python tools/bands.py [--frames 200] [--balanced | --interleaved] [--stages]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--instances", type=int, default=1)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--balanced", action="store_true", help="cut the bands by cost (dist.balanced_bounds over svr_get_row_costs of the full frame) instead of equally")
    ap.add_argument("--iterations", type=int, default=4, help="--balanced: re-cuts (each from the bands' modelled rows scaled to their measured time)")
    ap.add_argument("--wall", action="store_true", help="--balanced: scale by the band's wall-clock frame time instead of its GPU time")
    ap.add_argument("--gpu", action="store_true", help="--balanced: scale by the band's whole GPU time instead of its tile stage's (geometry + binning are the same for every band: spread over its rows they make a short, dense band look dearer per row than it is, and the cut swings)")
    ap.add_argument("--stages", action="store_true", help="also print per-stage kernel times (adds events to the stream)")
    ap.add_argument("--interleaved", action="store_true", help="the other partition of SURVEY 8e: rank r renders the tile rows t with t %% N == r (svr_set_row_interleave)")
    ap.add_argument("--ns", default="1,2,4,8")
    args = ap.parse_args()
    import torch
    pkg = g.load_package()
    hip = pkg.load_product_library()
    S, A = pkg.scenes, pkg.abi
    sc = S.sponza_like(lod=1, tex_size=1024)
    W, H = args.width, args.height
    r = hip.create(W, H)
    handles = sc.upload(r)
    inst = S.config5_instances() if args.instances == 16 else None
    opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
    pos, pitch, yaw = S.config5_camera() if args.instances == 16 else S.config3_camera()
    scene = S.scene_data_struct(pos, pitch, yaw, W, H)
    swap = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")

    def frame(y0, rows):
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.copy_to_swapchain(swap.data_ptr(), W, H, A.SWAPCHAIN_B8G8R8A8)

    def time_band(y0, rows, n, rk):
        if args.interleaved:
            r.set_scissor(0, 0, W, H)
            r.set_row_interleave(n, rk)
        else:
            r.set_row_interleave(1, 0)
            r.set_scissor(0, y0, W, rows)
        r.set_option(A.OPT_KERNEL_TIMING, 0)
        for _ in range(10):
            frame(y0, rows)
        r.sync()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            frame(y0, rows)
        host_ms = (time.perf_counter() - t0) / args.frames * 1e3  # the host's share: it must stay below the GPU's
        r.sync()
        ms = (time.perf_counter() - t0) / args.frames * 1e3
        stage, gpu_ms, tile_ms = "", None, None
        if args.stages or args.balanced:
            r.set_option(A.OPT_KERNEL_TIMING, 2)
            for _ in range(20):
                frame(y0, rows)
            r.sync()
            st = r.get_stats()
            gpu_ms = st.geometry_ms + st.binning_ms + st.tile_ms
            tile_ms = st.tile_ms
            stage = f" (geometry {st.geometry_ms:.3f} binning {st.binning_ms:.3f} tile {st.tile_ms:.3f})"
        costs, y0c, rowsc = r.row_costs()
        what = f"tile rows {rk} mod {n}" if args.interleaved else f"rows {y0}..{y0 + rows}"
        print(f"  N={n} band {rk}: {what}: {ms:.4f} ms/frame (host enqueue {host_ms:.4f}){stage}", flush=True)
        if args.interleaved:
            return ms, (gpu_ms if args.gpu else tile_ms), np.zeros(H, dtype=np.int64)
        return ms, (gpu_ms if args.gpu else tile_ms), pkg.dist.BandPlan.spread(costs, y0c, rowsc, H)

    base = None
    for n in [int(v) for v in args.ns.split(",")]:
        band = (H + n - 1) // n
        plan = pkg.dist.BandPlan(H, n, balanced=args.balanced, min_gain=0.0)
        for it in range(args.iterations if args.balanced and n > 1 else 1):
            bounds = plan.bounds
            per, profile = [], np.zeros(H, dtype=np.int64)
            for rk in range(n):
                y0, rows = bounds[rk], bounds[rk + 1] - bounds[rk]
                if rows == 0:
                    per.append(0.0)
                    continue
                ms, gpu_ms, mine = time_band(y0, rows, n, rk)
                per.append(ms)
                # what the ranks' all_reduce would assemble: every band's modelled rows scaled to its measured time
                profile += pkg.dist.BandPlan.scale_to(mine, ms if args.wall else gpu_ms)
            worst, mean = max(per), float(np.mean(per))
            if base is None:
                base = worst
            print(f"N={n}{' interleaved' if args.interleaved else ''}{' iteration ' + str(it) + ' rows ' + str(bounds) if args.balanced else ''}: slowest band {worst:.4f} ms, "
                  f"mean {mean:.4f} ms (slowest / mean {worst / mean:.3f}) -> at most {base / worst:.2f}x of N=1 "
                  f"({base / worst / n * 100:.0f} % efficiency before the gather)", flush=True)
            if args.balanced:
                plan.recut(profile)


if __name__ == "__main__":
    main()
