#!/usr/bin/env python3
"""bench.py — shaded fragments/s (and frames/s) of the draw_geometry replacement on MI355X.

Workload (BASELINE.json metric, configs[3]): the synthetic Sponza-style scene of SURVEY.md §8d
(262,144 triangles, 25 mipmapped 1024^2 textures, 2 Transparent materials) at 3840x2160, colour
target RGBA16F + D32 as in the reference (src/vk_engine.cpp:749,774).  A "step" is one frame:
draw_background's fill (svr_clear_color) + svr_draw_geometry, scene resident in HBM; the per-frame
RenderObject list crosses the ABI from host memory every frame exactly as the reference rebuilds it.

  python bench.py --gpus 1 --steps 300 --warmup 30
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, scene replicated, rank r renders a band of rows (svr_set_scissor) cut so that
the bands cost alike (dist.BandPlan, re-cut every --rebalance frames from the ranks' tile-row costs), and the
finished bands are exchanged once per frame over RCCL (torch.distributed backend "nccl"): grouped
send/recv for unequal bands, one all-gather for equal ones; two frames in flight so the exchange of frame i
overlaps the rendering of frame i+1; total work is fixed -> "scaling": "strong".

Timing: --warmup untimed frames, then the block of EXACTLY --steps frames (fence + barrier on both sides, max
over ranks) is repeated >= --blocks times and >= 0.5 s; ms_per_step is the MEDIAN block, p10 / p90 beside it.

Prints ONE JSON line on rank 0.  Extra keys: roofline (tile kernel, HBM bound), cpu_baseline (the
CPU oracle timed on this host), frames_per_s, rasterized_fragments_per_s, kernel_ms.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and streams that
# share one run their work in order.  A rank uses the caller's stream, the library's stage-1 stream and what the
# collective library creates: give them room, so that stage 1 never shares a queue with the tile kernel it should overlap
# (set before the runtime starts; an operator's own setting wins).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    ap.add_argument("--equal-bands", action="store_true", help="N > 1: fixed bands of H/N rows (one all-gather) instead of cost-balanced ones")
    ap.add_argument("--min-gain", type=float, default=0.2, help="N > 1: a cost-balanced cut replaces the equal bands only if it shortens the heaviest band by this fraction (unequal bands cost the host a batch of send/recv per frame instead of one all-gather)")
    ap.add_argument("--rebalance", type=int, default=64, help="N > 1: frames between two re-cuts of the row bands (0 = never)")
    ap.add_argument("--gather-fp16", action="store_true", help="N > 1: all-gather the RGBA16F target instead of the swapchain image")
    ap.add_argument("--partition", default="auto", choices=["auto", "bands", "interleaved"],
                    help="N > 1: contiguous row bands, interleaved tile rows (t %% N == rank), or whichever a short trial of both finds faster")
    ap.add_argument("--cpu-frames", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=25, help="repetitions of the timed --steps block (the median block is reported)")
    ap.add_argument("--profile-tag", default="r03_h", help="profiles/<tag>_traffic.json and <tag>_valu.json of this build are quoted in the line")
    return ap.parse_args()


def host_cores():
    """(threads used for the all-core leg, cores this process may run on, model name from /proc/cpuinfo)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # a container's CPU quota (cgroup v2 cpu.max "quota period"): threads beyond it only take turns
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return max(1, n), os.cpu_count() or n, model


def cpu_baseline(args, pkg, shaded_per_frame, sc):
    """The oracle — a scalar C++ restatement of the path, NOT a binned rasteriser: the draws are set up side by side
    over the threads, rasterisation is split into 16-row bands over the threads, each walking the triangles that reach
    it (oracle/svr_oracle.cpp run_pass / raster_rows) — on this host's cores, same workload, bounded: --cpu-frames
    frames on every core this process may use, then one frame on one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import svr_testlib as T
    ora = T.load_oracle()
    cores, hw, model = host_cores()
    S = pkg.scenes
    r = ora.create(args.width, args.height, pkg.abi.COLOR_RGBA16F)
    handles = sc.upload(r)
    inst = S.config5_instances() if args.instances == 16 else None
    opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
    pos, pitch, yaw = camera_of(args, S)
    scene = S.scene_data_struct(pos, pitch, yaw, args.width, args.height)

    def frames(n):
        t0 = time.perf_counter()
        for _ in range(n):
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, opaque, transparent)
        return (time.perf_counter() - t0) / n

    ora.lib.svr_oracle_set_threads(r.h, cores)
    frames(1)  # warm-up
    dt_all = frames(args.cpu_frames)
    ora.lib.svr_oracle_set_threads(r.h, 1)
    dt_one = frames(1)
    r.close()
    return {"value": shaded_per_frame / dt_all, "unit": "fragments/s", "cores": cores, "kind": "port",
            "description": "scalar C++ oracle (-O2, no intrinsics, -ffp-contract=off), forward rasteriser without tiles: the draws set up "
                           "side by side over the threads, then 16-row bands over the threads, each walking the triangles that reach it in "
                           "submission order",
            "frames_per_s": 1.0 / dt_all,
            "single_thread": {"value": shaded_per_frame / dt_one, "frames_per_s": 1.0 / dt_one, "cores": 1},
            "hardware_concurrency": hw, "cpu_model": model,
            "sample": f"{args.cpu_frames} full frames of the same workload ({args.width}x{args.height}) on {cores} threads after 1 warm-up, "
                      "then 1 frame on 1 thread; fragments counted as the GPU path counts them (each visible pixel once + transparent layers)"}


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
    # cost-balanced row bands (dist.BandPlan): re-cut from the ranks' tile-row costs every --rebalance frames
    plan = D.BandPlan(H, world, balanced=not args.equal_bands, min_gain=args.min_gain)
    # verify="fence": frames are only handed out at fences here, and a second collective per frame would cost the host
    # 20-30 us against 60-us frames (dist.py ShardedFrame; libsvr_dist.so verifies every frame)
    slots = [D.ShardedFrame(torch, r, rank, world, dev, A.COLOR_RGBA16F, present=present, plan=plan, verify="fence") for _ in range(2)]
    handles = sc.upload(r)
    inst = S.config5_instances() if args.instances == 16 else None
    opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
    pos, pitch, yaw = camera_of(args, S)
    scene = S.scene_data_struct(pos, pitch, yaw, W, H)
    state = {"i": 0}

    def frame():
        if world > 1 and args.rebalance > 0 and state["i"] % args.rebalance == args.rebalance // 2:
            plan.rebalance(torch, dist, r, dev)     # a collective, part of the job: timed like everything else
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

    def note(msg):  # progress on stderr (rank 0): a multi-rank run that stalls says where
        if rank == 0 and world > 1:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"{world} ranks up, scene resident")
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

    note(f"instrumented frame done: {shaded} shaded fragments")
    for _ in range(args.warmup):
        frame()
    fence()
    note(f"warm-up done, rows {plan.bounds}")
    if world > 1 and args.partition != "bands":
        if args.partition == "interleaved":
            plan.partition = "interleaved"
        else:  # a short trial of either partition (wall clock of a few blocks of frames, fenced), then the collective pick
            def trial(name):
                plan.partition = name
                for _ in range(max(args.warmup, 4)):
                    frame()
                fence()
                t0 = time.perf_counter()
                for _ in range(4 * args.steps):
                    frame()
                fence()
                return (time.perf_counter() - t0) / (4 * args.steps) * 1e3
            tb, ti = trial("bands"), trial("interleaved")
            picked = plan.pick(torch, dist, dev, tb, ti)
            note(f"partition trial: bands {tb:.4f} ms, interleaved {ti:.4f} ms per frame on rank 0 -> {picked}")
        for _ in range(args.warmup):
            frame()
        fence()
    r.set_option(A.OPT_KERNEL_TIMING, 1)

    def block():
        """EXACTLY --steps frames between two fences (barrier + synchronize on both sides), max over ranks."""
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frame()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # SURVEY 8d protocol: the block is repeated (>= 25 times and >= 0.5 s in all, at most 400 times) and the MEDIAN
    # block is the reported one, with p10 / p90 beside it: one block of a few milliseconds is at the mercy of
    # whatever else the box does in that instant.  Every rank runs the same number of blocks (rank 0 decides).
    first = block()
    n_blocks = int(min(400, max(args.blocks, 0.5 / max(first, 1e-6))))
    nb = torch.tensor([n_blocks], dtype=torch.int64, device=dev)
    if world > 1:
        dist.broadcast(nb, 0)
    n_blocks = int(nb.item())
    note(f"first block {first * 1e3:.2f} ms, {n_blocks} blocks to go")
    times = sorted([first] + [block() for _ in range(n_blocks - 1)])
    note(f"timed blocks done, rows {plan.bounds}")
    pick = lambda q: times[min(len(times) - 1, max(0, int(round(q * (len(times) - 1)))))]
    dt, dt_p10, dt_p90 = pick(0.5), pick(0.1), pick(0.9)
    st = r.get_stats()
    r.set_option(A.OPT_KERNEL_TIMING, 0)

    if rank == 0:
        fps = args.steps / dt
        counts_scene = sc.counts()
        # algorithmic bytes of the tile kernel per launch: 5 B per shaded fragment
        # (one RGBA8 texel at matched LOD x1.25 for the second mip) + one final store of
        # RGBA16F (8 B) + D32 (4 B) per pixel of this rank's band
        frag_rank = shaded / world
        tile_bytes = 5.0 * frag_rank + 12.0 * W * (H / world)
        tile_s = st.tile_ms * 1e-3
        # HBM bytes per launch of that kernel from the PMC counters cannot be collected from inside this
        # process; the committed rocprofv3 --pmc summary of this very command (1 GPU, default size) is quoted
        traffic, traffic_source, valu = None, None, None
        default_workload = world == 1 and not args.gltf and (W, H, args.instances, args.lod, args.tex_size) == (3840, 2160, 1, 1, 1024)
        tpath = os.path.join(ROOT, "profiles", args.profile_tag + "_traffic.json")
        vpath = os.path.join(ROOT, "profiles", args.profile_tag + "_valu.json")
        if default_workload and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic, traffic_source = tj["traffic_bytes_per_launch"], f"profiles/{args.profile_tag}_traffic.json: " + tj["correction"]
        if default_workload and os.path.exists(vpath):
            with open(vpath) as f:
                valu = json.load(f)
        achieved = tile_bytes / tile_s / 1e9 if tile_s > 0 else 0.0
        out = {
            "metric": "shaded fragments/s", "value": shaded * fps, "unit": "fragments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_p10": dt_p10 / args.steps * 1e3, "ms_per_step_p90": dt_p90 / args.steps * 1e3, "timed_blocks": len(times),
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
                       "parallelism": ((f"interleaved tile rows (t % {world} == rank), in-place all-gathers per {world} tile rows" if plan.partition == "interleaved" else
                                        f"row bands x{world} ({'equal' if args.equal_bands else 'cost-balanced, rows ' + str(plan.bounds)})") + f" [--partition {args.partition}] + exchange of the "
                                       + ("B8G8R8A8 swapchain image" if present else "RGBA16F target")) if world > 1 else "single GPU"},
            "frames_per_s": fps,
            "shaded_fragments_per_frame": shaded, "rasterized_fragments_per_frame": rasterized,
            "rasterized_fragments_per_s": rasterized * fps,
            "rasterized_per_shaded": rasterized / max(shaded, 1),
            "rasterized_definition": "covered samples of every submitted triangle that reached the depth test, as a forward rasteriser walks them "
                                     "(counted by the untimed instrumented frame, equal to the oracle's count); timed passes whose bins are deep "
                                     "skip triangles hidden behind what their tile already holds (hierarchical depth test; not in this workload's 4K frame)",
            "binned_triangles": binned, "bin_entries": entries,
            "kernel_ms": {"tile": st.tile_ms, "passes": st.timed_passes},
            "roofline": {"bound": "hbm", "kernel": "tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": tile_bytes, "avg_launch_ms": st.tile_ms},
            # what actually bounds the dominant kernel: VALU issue, from the committed SQ counters of this build
            "roofline_valu": valu,
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
