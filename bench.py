#!/usr/bin/env python3
"""bench.py — shaded fragments/s (and frames/s) of the draw_geometry replacement on MI355X.

Workload (BASELINE.json metric, configs[3]): the synthetic Sponza-style scene of SURVEY.md §8d
(262,144 triangles, 25 mipmapped 1024^2 textures, 2 Transparent materials) at 3840x2160, colour
target RGBA16F + D32 as in the reference (src/vk_engine.cpp:749,774).  A "step" is one frame:
draw_background's fill (svr_clear_color) + svr_draw_geometry, scene resident in HBM; the per-frame
RenderObject list crosses the ABI from host memory every frame exactly as the reference rebuilds it.

  python bench.py --gpus 1 --steps 300 --warmup 30
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, scene replicated, rank r renders the band of rows [r*H/N,(r+1)*H/N)
(svr_set_scissor) and the finished bands are exchanged with one RCCL all-gather per frame
(torch.distributed backend "nccl" == RCCL), two frames in flight so the gather of frame i overlaps
the rendering of frame i+1; total work is fixed -> "scaling": "strong".

Prints ONE JSON line on rank 0.  Extra keys: roofline (tile kernel, HBM bound), cpu_baseline (the
CPU oracle timed on this host), frames_per_s, rasterized_fragments_per_s, kernel_ms.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lod", type=int, default=1, help="tessellation divisor of the scene generator (1 = 262,144 triangles)")
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--instances", type=int, default=1, help="16 = BASELINE config 5's 4x4 instancing")
    ap.add_argument("--gltf", default="", help="render this .glb/.gltf instead of the synthetic scene (not the BASELINE workload)")
    ap.add_argument("--camera", default="", help="with --gltf: x,y,z,pitch,yaw")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-fp16", action="store_true", help="N > 1: all-gather the RGBA16F target instead of the swapchain image")
    ap.add_argument("--cpu-frames", type=int, default=3)
    return ap.parse_args()


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # a one-GPU box grants 16 cores; never oversubscribe


def cpu_baseline(args, pkg, shaded_per_frame, sc):
    """The oracle (a scalar C++ port of the path) on this host's cores, same workload, bounded."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import svr_testlib as T
    ora = T.load_oracle()
    cores = host_cores()
    S = pkg.scenes
    r = ora.create(args.width, args.height, pkg.abi.COLOR_RGBA16F)
    handles = sc.upload(r)
    inst = S.config5_instances() if args.instances == 16 else None
    opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
    pos, pitch, yaw = camera_of(args, S)
    scene = S.scene_data_struct(pos, pitch, yaw, args.width, args.height)
    ora.lib.svr_oracle_set_threads(r.h, cores)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)  # warm-up
    t0 = time.perf_counter()
    for _ in range(args.cpu_frames):
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
    dt = (time.perf_counter() - t0) / args.cpu_frames
    r.close()
    return {"value": shaded_per_frame / dt, "unit": "fragments/s", "cores": cores, "kind": "port",
            "frames_per_s": 1.0 / dt,
            "sample": f"{args.cpu_frames} full frames of the same workload ({args.width}x{args.height}) after 1 warm-up, "
                      f"geometry single-threaded, rasterisation row-band parallel over {cores} threads; "
                      "fragments counted as the GPU path counts them (each visible pixel once + transparent layers)"}


def load_scene(args, pkg):
    if args.gltf:
        import importlib
        return importlib.import_module(pkg.__name__ + ".gltf_io").load_gltf(args.gltf)
    return pkg.scenes.sponza_like(lod=args.lod, tex_size=args.tex_size)


def camera_of(args, S):
    if args.gltf and args.camera:
        v = [float(x) for x in args.camera.split(",")]
        return (v[0], v[1], v[2]), v[3], v[4]
    return S.config5_camera() if args.instances == 16 else S.config3_camera()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # SVR_BENCH_REHEARSE=1 (development only): every rank on device 0 and gloo instead of RCCL, to walk the
    # N > 1 code path (bands, present, in-place gather, two frame slots) on a one-GPU box; timings mean nothing
    rehearse = os.environ.get("SVR_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pkg = g.load_package()
    hip = pkg.load_product_library()
    S, A, D = pkg.scenes, pkg.abi, pkg.dist
    W, H = args.width, args.height

    sc = load_scene(args, pkg)
    r = hip.create(W, H, A.COLOR_RGBA16F, device=device_index)
    r.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    # N > 1: the presentable B8G8R8A8 image is what the ranks exchange (4 B/px; --gather-fp16 sends the target)
    present = world > 1 and not args.gather_fp16
    slots = [D.ShardedFrame(torch, r, rank, world, dev, A.COLOR_RGBA16F, present=present) for _ in range(2)]
    handles = sc.upload(r)
    inst = S.config5_instances() if args.instances == 16 else None
    opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
    pos, pitch, yaw = camera_of(args, S)
    scene = S.scene_data_struct(pos, pitch, yaw, W, H)
    state = {"i": 0}

    def frame():
        s = slots[state["i"] & 1]
        state["i"] += 1
        s.begin()                                   # _draw_image of this frame slot + the rank's scissor band
        r.clear_color((1.0, 1.0, 1.0, 1.0))         # draw_background's result
        if s.rows:
            r.draw_geometry(scene, opaque, transparent)
        s.gather(dist)                              # finished rows over xGMI, in place, asynchronous

    def fence():
        for s in slots:
            s.finish()
        r.sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    # one instrumented frame: fragment counts (not timed)
    r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
    frame()
    fence()
    st = r.get_stats()
    counts = torch.tensor([st.shaded_fragments, st.rasterized_fragments, st.binned_triangles, st.bin_entries],
                          dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(counts)
    shaded, rasterized, binned, entries = [int(v) for v in counts.tolist()]
    r.set_option(A.OPT_COUNT_FRAGMENTS, 0)

    for _ in range(args.warmup):
        frame()
    fence()
    r.set_option(A.OPT_KERNEL_TIMING, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    st = r.get_stats()
    r.set_option(A.OPT_KERNEL_TIMING, 0)

    if rank == 0:
        fps = args.steps / dt
        counts_scene = sc.counts()
        # algorithmic bytes of the tile kernel per launch: 5 B per shaded fragment
        # (one RGBA8 texel at matched LOD x1.25 for the second mip) + one final store of
        # RGBA16F (8 B) + D32 (4 B) per pixel of this rank's band
        frag_rank = shaded / world
        tile_bytes = 5.0 * frag_rank + 12.0 * W * slots[0].rows
        tile_s = st.tile_ms * 1e-3
        # HBM bytes per launch of that kernel from the PMC counters cannot be collected from inside this
        # process; the committed rocprofv3 --pmc summary of this very command (1 GPU, default size) is quoted
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_h_traffic.json")
        if world == 1 and not args.gltf and (W, H, args.instances, args.lod, args.tex_size) == (3840, 2160, 1, 1, 1024) and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic, traffic_source = tj["traffic_bytes_per_launch"], "profiles/r01_h_traffic.json: " + tj["correction"]
        achieved = tile_bytes / tile_s / 1e9 if tile_s > 0 else 0.0
        out = {
            "metric": "shaded fragments/s", "value": shaded * fps, "unit": "fragments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"glTF file {os.path.basename(args.gltf)}, {counts_scene['triangles']} triangles, {W}x{H}, RGBA16F+D32,"
                                    " mesh.vert/mesh.frag (NOT a BASELINE config)") if args.gltf else
                                   (f"sponza-style synthetic scene seed 0x53505A41, {counts_scene['triangles']} triangles"
                                    f" x{args.instances} instances, {W}x{H}, RGBA16F+D32, mesh.vert/mesh.frag"
                                    f" (BASELINE configs[{4 if args.instances == 16 else 3}])"),
                       "width": W, "height": H, "triangles": counts_scene["triangles"] * args.instances,
                       "draws": int(len(opaque) + len(transparent)),
                       "textures": f"{len(sc.textures)} images from the file" if args.gltf else f"25 x {args.tex_size}^2 RGBA8 mipmapped",
                       "parallelism": (f"row bands x{world} + all_gather of the " + ("B8G8R8A8 swapchain image" if present else "RGBA16F target"))
                                      if world > 1 else "single GPU"},
            "frames_per_s": fps,
            "shaded_fragments_per_frame": shaded, "rasterized_fragments_per_frame": rasterized,
            "rasterized_fragments_per_s": rasterized * fps,
            "binned_triangles": binned, "bin_entries": entries,
            "kernel_ms": {"tile": st.tile_ms, "passes": st.timed_passes},
            "roofline": {"bound": "hbm", "kernel": "tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": tile_bytes, "avg_launch_ms": st.tile_ms},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, pkg, shaded, sc)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
