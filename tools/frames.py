"""Developer tool: render N frames of a BASELINE config on the GPU (a short, quiet target for
rocprofv3) and optionally print the per-tile bin histogram.

    python tools/frames.py --frames 20 --hist
    rocprofv3 --pmc SQ_WAVES ... -- python tools/frames.py --frames 5
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
    ap.add_argument("--lod", type=int, default=1)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--instances", type=int, default=1)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--wgtimes", action="store_true", help="with a -DSVR_DEBUG_WG_TIMES build (--lib): where the tile kernel's time goes between workgroups")
    ap.add_argument("--hist", action="store_true")
    ap.add_argument("--slots", type=int, default=5, help="--wgtimes: tile workgroups a CU holds (5 since round 3, 4 before)")
    ap.add_argument("--flatten", type=int, default=0, help="SVR_OPT_DEVICE_FLATTEN: 0 auto, 1 device, 2 host")
    ap.add_argument("--timing", type=int, default=2, help="SVR_OPT_KERNEL_TIMING during the frames (2 = events around every stage, perturbs the pipeline; 0 for traces)")
    ap.add_argument("--tuning", type=int, default=0, help="SVR_OPT_TUNING mask (2 = stages serialised)")
    ap.add_argument("--lib", default="", help="another build of libsvr_hip.so (build_ab/...) instead of the product's")
    ap.add_argument("--ab", default="", help="comma list of SVR_OPT_TUNING masks to time interleaved, e.g. 0,1")
    args = ap.parse_args()
    pkg = g.load_package()
    S, A = pkg.scenes, pkg.abi
    hip = A.SvrLib(os.path.abspath(args.lib)) if args.lib else pkg.load_product_library()
    sc = S.sponza_like(lod=args.lod, tex_size=args.tex_size)
    r = hip.create(args.width, args.height)
    handles = sc.upload(r)
    inst = S.config5_instances() if args.instances == 16 else None
    opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
    pos, pitch, yaw = S.config5_camera() if args.instances == 16 else S.config3_camera()
    scene = S.scene_data_struct(pos, pitch, yaw, args.width, args.height)
    r.set_option(A.OPT_DEVICE_FLATTEN, args.flatten)
    r.set_option(A.OPT_TUNING, args.tuning)
    r.set_option(A.OPT_KERNEL_TIMING, args.timing)
    for _ in range(3):
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
    r.sync()
    r.set_option(A.OPT_KERNEL_TIMING, args.timing)
    t0 = time.perf_counter()
    for _ in range(args.frames):
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
    r.sync()
    dt = (time.perf_counter() - t0) / args.frames
    st = r.get_stats()
    print(f"{args.width}x{args.height} x{args.instances}: {dt * 1e3:.3f} ms/frame wall; geometry {st.geometry_ms:.3f} "
          f"binning {st.binning_ms:.3f} tile {st.tile_ms:.3f} ms; host record {st.mesh_draw_time:.3f} ms; "
          f"bin entries {st.bin_entries}")
    if args.ab:
        masks = [int(m) for m in args.ab.split(",")]
        res = {m: [] for m in masks}
        for rnd in range(8):
            for m in masks:
                r.set_option(A.OPT_TUNING, m)
                r.set_option(A.OPT_KERNEL_TIMING, (2 if rnd < 2 else 1) if rnd < 4 else 0)  # all stages | tile pair | no events
                r.sync()
                tw = time.perf_counter()
                for _ in range(20):
                    r.clear_color((1, 1, 1, 1))
                    r.draw_geometry(scene, opaque, transparent)
                r.sync()
                tw = (time.perf_counter() - tw) / 20 * 1e3
                s2 = r.get_stats()
                res[m].append((s2.geometry_ms, s2.binning_ms, s2.tile_ms, tw if rnd >= 4 else float("nan"),
                               tw if 2 <= rnd < 4 else float("nan"), tw if rnd < 2 else float("nan")))
        for m in masks:
            a = np.array(res[m])
            print(f"  tuning {m}: geometry/binning/tile median {np.nanmedian(a[:2, :3], axis=0).round(4).tolist()} ms; "
                  f"wall ms/frame: {np.nanmedian(a[:, 3]):.4f} (no events) {np.nanmedian(a[:, 4]):.4f} (tile event pair) "
                  f"{np.nanmedian(a[:, 5]):.4f} (all stage events)")
        r.set_option(A.OPT_TUNING, 0)
    host = []
    for _ in range(10):  # GPU idle at every call: mesh_draw_time is then pure host record cost
        r.clear_color((1, 1, 1, 1))
        host.append(r.draw_geometry(scene, opaque, transparent).mesh_draw_time)
        r.sync()
    print(f"  host record time with an idle GPU: median {float(np.median(host)):.3f} ms")
    if args.wgtimes:
        r.set_option(A.OPT_TILE_CYCLES, 1)
        for _ in range(4):
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, opaque, transparent)
        r.sync()
        w = r.read_tile_cycles()
        r.set_option(A.OPT_TILE_CYCLES, 0)
        start, end, hw, xcc = (w[:, k].astype(np.int64) for k in range(4))
        ok = end != 0
        start, end, hw, xcc = start[ok], end[ok], hw[ok], xcc[ok] & 15
        t0 = start.min()
        start, end = (start - t0) / 100.0, (end - t0) / 100.0  # us (100 MHz wall clock)
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)  # xcc, se, sh, cu
        span = end.max()
        busy = (end - start).sum()
        ncu = len(np.unique(cu))
        print(f"  tile kernel by workgroup: {start.size} workgroups on {ncu} CUs; first start .. last end {span:.1f} us; "
              f"sum of residence {busy / 1e3:.1f} ms = {busy / (args.slots * ncu):.1f} us per slot at {args.slots} per CU")
        # resident workgroups per CU over time
        grid = np.arange(0.0, span, 1.0)
        resident = np.zeros(grid.size)
        for a, b in zip(start, end):
            resident[int(a):int(np.ceil(b))] += 1
        resident /= ncu
        marks = [int(x) for x in np.linspace(0, grid.size - 1, 24)]
        print("  mean resident workgroups per CU at", [f"{grid[m]:.0f}us:{resident[m]:.2f}" for m in marks])
        S = float(args.slots)
        full = np.nonzero(resident > S - 0.1)[0]
        if full.size:
            print(f"  chip full (>{S - 0.1:.1f} per CU) from {grid[full[0]]:.0f} to {grid[full[-1]]:.0f} us; lost slot-time before {np.sum(S - resident[:full[0]]) / S:.1f} us-equivalents, "
                  f"after {np.sum(S - resident[full[-1]:]) / S:.1f}, in between {np.sum(S - resident[full[0]:full[-1]]) / S:.1f}")
        print(f"  time-average of resident workgroups per CU: {resident.mean():.2f} of {args.slots}")
        # hand-over on a CU: from a workgroup's end to the next start on that CU (greedy matching in time order)
        gaps = []
        for c in np.unique(cu):
            m = cu == c
            ev = sorted([(t, 1) for t in end[m]] + [(t, 0) for t in start[m]])
            free = []
            for t, is_end in ev:
                if is_end:
                    free.append(t)
                elif free:
                    gaps.append(t - free.pop(0))
        gaps = np.array(gaps)
        if gaps.size:
            print(f"  hand-over of a CU slot (a workgroup ends -> the next starts there): {gaps.size} hand-overs, median {np.median(gaps):.2f} us, "
                  f"mean {gaps.mean():.2f}, p90 {np.percentile(gaps, 90):.2f}, sum {gaps.sum() / 1e3:.2f} ms = {gaps.sum() / (args.slots * ncu):.1f} us per slot")
        dur = end - start
        print(f"  workgroup residence: median {np.median(dur):.1f} us, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}; the last 5 % of workgroups start after "
              f"{np.percentile(start, 95):.0f} us")
    if args.hist:
        op, tr = r.read_bins()
        for name, c in (("opaque", op), ("transparent", tr)):
            qs = np.percentile(c, [0, 25, 50, 75, 90, 99, 100]).astype(int).tolist()
            print(f"  {name}: tiles {c.size} sum {int(c.sum())} mean {c.mean():.1f} nonempty {(c > 0).sum()} "
                  f"percentiles(0,25,50,75,90,99,100) {qs}")
        tx = (args.width + 31) // 32
        rows = op.reshape(-1, tx).sum(axis=1)
        print("  opaque entries per tile row:", rows.tolist())
        # one frame of the production kernel with phase stamps: shader-clock cycles per tile and phase
        r.set_option(A.OPT_TILE_CYCLES, 1)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        cyc = r.read_tile_cycles().astype(np.float64)
        tot = cyc.sum(axis=1)
        print("  tile cycles: phase sums A/B/C/D (Mcycles):",
              [round(float(v) / 1e6, 1) for v in cyc.sum(axis=0)])
        print("  per-tile total cycles percentiles(0,50,90,99,100):",
              np.percentile(tot, [0, 50, 90, 99, 100]).astype(int).tolist(), "sum", int(tot.sum()))
        for lo, hi in ((0, 4), (4, 16), (16, 64), (64, 256), (256, 1 << 30)):
            m = (op >= lo) & (op < hi)
            if m.any():
                print(f"    tiles with {lo}<=n<{hi}: {int(m.sum())} tiles, mean cycles A {cyc[m, 0].mean():.0f} B {cyc[m, 1].mean():.0f} "
                      f"C {cyc[m, 2].mean():.0f} D {cyc[m, 3].mean():.0f}; mean n {op[m].mean():.1f}")
        top = np.argsort(-tot)[:12]
        print("    slowest tiles (tile, x, y, n_opaque, n_transparent, cycles A, B, C, D):")
        for t in top:
            print(f"      {int(t):6d} {int(t % tx):4d} {int(t // tx):4d} {int(op[t]):6d} {int(tr[t]):6d} "
                  f"{int(cyc[t, 0]):9d} {int(cyc[t, 1]):9d} {int(cyc[t, 2]):9d} {int(cyc[t, 3]):9d}")
        n = op.astype(np.float64)
        big = op >= 64
        if big.sum() > 4:
            coef = np.polyfit(n[big], cyc[big, 0], 1)
            print(f"    phase A fit on heavy tiles: {coef[0]:.1f} cycles/triangle + {coef[1]:.0f}")
    r.close()


if __name__ == "__main__":
    main()
