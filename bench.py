#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Whitted render path.

Metric (BASELINE.json): Mrays/s (primary + shadow + reflection rays actually
traversed) at 1920x1080; frame ms; 1/2/4/8-GPU scaling.

Workload (config.workload = "C3"): the dragon-scale stand-in — 819 200-triangle
torus knot (data/dragon.obj is absent from the reference), 1920x1080, Phong +
hard shadows + reflections depth 4, default camera — synthetic, generated in
process. One "step" = one full frame with the scene, BVH and framebuffer resident in
HBM. Throughput setting (defaults): a launch renders 8 successive frames
(rt_render_device_frames), two launches are in flight on alternating HIP streams,
8x8-pixel primary tiles; every frame is rendered in full, K steps = K frames.
`--frames-per-launch 1 --frames-in-flight 1` is the strictly serial loop.

N GPUs: one process per GPU; the frame's 8-row strips are dealt round-robin to
the ranks, each renders its strips into a packed buffer, an RCCL all-gather
over xGMI assembles them and a de-interleave kernel writes the reference's
row-flipped frame (strong scaling: the frame is fixed).

JSON line (rank 0): the contract's keys plus
  roofline      dominant kernel k_trace_level: ALGORITHMIC bytes per launch
                 / average launch duration measured with HIP
                events on the launch stream during the timed steps; peak 8 TB/s.
  cpu_baseline  the CPU oracle (kind "port": the repo's restatement of the
                reference path, OpenMP rows) timed on this host on the same frame.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STRIP_ROWS = 8


def algorithmic_bytes(cnt: dict, features, pixels: int) -> dict:
    """32 B per box test (24 B AABB + 8 B child meta; the leaf
    phase's triangle-group boxes are box tests too), 36 B per
    triangle test (three fp32 positions), 16 B per sphere test; per shaded hit
    32 B material (+36 B normals with normalInterp, +24 B uvs +12 B texel with
    textures) and its 12 B direct-colour write; 2 x 32 B per queued reflection ray
    (written by one level, read by the next); 12 B per pixel for the frame write
    (k_fold, not part of the trace kernel's figure)."""
    trav = 32 * (cnt["box_tests"] + cnt.get("group_tests", 0)) + 36 * cnt["tri_tests"] + 16 * cnt["sphere_tests"]
    per_hit = 32 + 12 + (36 if features.enableNormalInterp else 0) + (36 if features.enableTextureMapping else 0)
    shade = per_hit * cnt["hits_shaded"]
    queue = 64 * cnt["rays_queued"]
    return {"trace_kernel": trav + shade + queue, "frame": trav + shade + queue + 12 * pixels}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="successive frames on alternating HIP streams, so that the tail of frame k (a few long "
                         "path trees) overlaps the start of the next launch (default 2); "
                         "1 = strictly one frame at a time")
    ap.add_argument("--tile-log", type=int, nargs=2, default=None, metavar=("W", "H"),
                    help="primary tile = 2^W x 2^H pixels per wave (default: 3 3 with several frames per launch and "
                         "launches in flight, 2 3 with frames in flight only, else the library's latency setting 1 3)")
    ap.add_argument("--exchange-batch", type=int, default=None,
                    help="several GPUs: successive frames that share one all-gather + one de-interleave launch "
                         "(default 8); every frame is still rendered and assembled")
    ap.add_argument("--frames-per-launch", type=int, default=None,
                    help="successive frames rendered by ONE launch (rt_render_device_frames; default 8): a launch of "
                         "several frames has no idle tail between them — what a rank's small share of a frame needs "
                         "at 8 ranks, and worth 5 %% on one GPU; every frame is rendered in full")
    ap.add_argument("--exchange-always", action="store_true",
                    help="one GPU: take the strips + RCCL exchange path with a one-rank group (what the N > 1 run "
                         "adds on top of the render: packed strips, all-gather, de-interleave), for measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-events", action="store_true",
                    help="do not bracket trace kernels with HIP events during the timed steps")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.frames_in_flight is None:
        args.frames_in_flight = 2
    if args.frames_per_launch is None:
        args.frames_per_launch = 8
    if args.exchange_batch is None:
        args.exchange_batch = 1 if world == 1 else 8
    # HIP runtime setting, read when the runtime starts: how many hardware queues the streams of this
    # process are dealt over (default 4) = how many of the frames in flight really run side by side
    # (tools/queue_bench.hip); + the exchange stream, torch's RCCL stream and the null stream
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, args.frames_in_flight + 4)))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # Rehearsal on a one-GPU box only (never a reported number): BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and lets gloo carry the exchange, to exercise the N > 1 code path.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.exchange_always:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    capi = importlib.import_module("computer-graphics-engine_amd.capi")
    scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
    sharding = importlib.import_module("computer-graphics-engine_amd.sharding")

    rt = capi.Library()
    rt.set_device(local_rank)
    ndev, devname = rt.device_info()

    cfg = scenes.CONFIGS[args.workload]
    W, H = cfg.width, cfg.height
    f, p = cfg.make_features(), cfg.make_params()
    data = scenes.get_scene(cfg.scene)
    t0 = time.time()
    scene = rt.scene(data)
    bvh = rt.bvh(scene, f)
    build_s = time.time() - t0
    cam = rt.camera(*scenes.default_camera_args(W, H))

    stream = torch.cuda.current_stream().cuda_stream
    if args.frames_per_launch > 1:
        # Throughput setting of the library's tuning knobs: 8x8-pixel primary tiles (64 rays per wave) do the least
        # instruction work per ray but end in the longest path-tree tails — which a launch of several frames and
        # the next launch in flight cover (6 980 Mrays/s with 4x8 tiles, 8 030 with 8x8; a lone frame prefers
        # the library default 2x8); 64-ray waves fill a longer leaf queue before a flush pays (+1 %).
        # Results are identical for every setting (tests/test_gpu_options.py).
        rt.set_option("tile_wlog", 3)
        rt.set_option("tile_hlog", 3)
        rt.set_option("leaf_flush", 32)
    elif args.frames_in_flight > 1:
        rt.set_option("tile_wlog", 2)   # frames in flight only: 4x8 tiles
        rt.set_option("tile_hlog", 3)
    if args.tile_log:
        rt.set_option("tile_wlog", args.tile_log[0])
        rt.set_option("tile_hlog", args.tile_log[1])
    renderer = sharding.StripRenderer(rt, bvh, cam, f, p, W, H, rank, world, strip_rows=STRIP_ROWS,
                                      frames_in_flight=args.frames_in_flight, exchange_always=args.exchange_always,
                                      exchange_batch=args.exchange_batch)

    # ---- untimed counting pass: rays and tests of THIS rank's strips ------------
    cnt = renderer.count_rays(stream)
    keys = list(cnt.keys())
    cvec = torch.tensor([cnt[k] for k in keys], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(cvec)
    total = dict(zip(keys, [int(v) for v in cvec.tolist()]))
    rays_per_frame = total["rays"]

    # ---- warmup ----------------------------------------------------------------
    fpl = max(1, args.frames_per_launch)
    if renderer.sharded and renderer.batch % fpl:
        fpl = 1

    def run_frames(n):
        """n frames: launches of fpl frames, the rest one by one (every frame rendered in full)"""
        for _ in range(n // fpl):
            renderer.step(stream, frames=fpl)
        for _ in range(n % fpl):
            renderer.step(stream)

    run_frames(args.warmup)
    renderer.finish()     # nothing of the warmup (a partial exchange batch) is left for the timed region

    # ---- timed steps -----------------------------------------------------------
    use_events = not args.no_profile_events
    rt.profile_enable(bvh, use_events)
    launch_streams = renderer.launch_streams(stream)
    if use_events:
        for s_ in launch_streams:
            rt.profile_read(bvh, s_, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run_frames(args.steps)
    renderer.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms, launches = 0.0, 0
    if use_events:
        for s_ in launch_streams:
            ms_, n_ = rt.profile_read(bvh, s_, reset=True)
            kernel_ms, launches = kernel_ms + ms_, launches + n_
    rt.profile_enable(bvh, False)
    tvec = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tvec, op=dist.ReduceOp.MAX)
    elapsed = float(tvec.item())

    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_per_frame * args.steps / elapsed / 1e6

    if rank == 0:
        ab = algorithmic_bytes(total, f, W * H)
        n_levels = (cfg.depth + 1) if f.enableRecursive else 1
        roofline = None
        if use_events and launches:
            # this rank's share of the algorithmic bytes / its own kernel time
            ab_rank = algorithmic_bytes(cnt, f, 0)["trace_kernel"]
            # launches per frame: 1 with the default fused recursion (the wave keeps its own
            # reflection rays), depth + 1 with rt_set_option("fused", 0)
            per_frame = launches / args.steps
            per_launch_bytes = ab_rank / per_frame
            avg_launch_s = kernel_ms / 1e3 / launches
            achieved = per_launch_bytes / avg_launch_s / 1e9
            traffic = None
            tj = ROOT / "profiles" / "hbm_traffic.json"
            if tj.exists():
                tr = json.load(open(tj))
                if tr.get("workload") == args.workload and world == 1:
                    traffic = tr.get("hbm_bytes_per_launch")
            roofline = {"bound": "hbm", "kernel": "k_trace_level", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "algorithmic_bytes_per_launch": int(per_launch_bytes),
                        "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches": launches,
                        "launches_per_frame": round(per_frame, 3),
                        # frames in flight: launches of successive frames overlap, so one launch does not have
                        # the chip to itself; all launches together move the timed region's algorithmic bytes
                        # in the timed region's wall time
                        "frames_in_flight": renderer.flight,
                        "machine_achieved": round(ab_rank * args.steps / elapsed / 1e9, 2),
                        "machine_frac": round(ab_rank * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4)}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, data, scenes)
        out = {
            "metric": "Mrays/s (primary+shadow+reflect) at 1920x1080",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": data.name, "triangles": data.n_triangles,
                       "resolution": [W, H], "depth": cfg.depth, "features": cfg.features,
                       "rays_per_frame": rays_per_frame, "rays_reference_equivalent": total["rays_ref_equiv"],
                       "rays_primary": total["rays_primary"], "rays_shadow": total["rays_shadow"],
                       "rays_reflect": total["rays_reflect"], "box_tests": total["box_tests"],
                       "tri_tests": total["tri_tests"], "group_tests": total.get("group_tests", 0), "node_fetches": total["node_fetches"],
                       "tri_fetches": total["tri_fetches"], "algorithmic_bytes_per_frame": ab["frame"],
                       "parallelism": f"strips{STRIP_ROWS}x{world}" + ("+allgather(overlapped" + (f",{renderer.batch}-frames-each)" if renderer.batch > 1 else ")")
                                                              if renderer.sharded else "")
                       + (f"+{renderer.flight}-frames-in-flight" if renderer.flight > 1 else "")
                       + (f"+{fpl}-frames-per-launch" if fpl > 1 else "")
                       + ("+REHEARSAL-gloo-one-device" if rehearsal else ""),
                       "device": devname, "bvh_build_s": round(build_s, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.exchange_always:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cfg, data, scenes):
    """The CPU oracle (restatement of the reference path; traces reflections twice
    like the reference) on the same frame, all host threads. Bounded sample: every
    `stride`-th 8-row strip of the frame so the run stays within ~10-30 s."""
    from oracle.cpu_oracle import Oracle

    orc = Oracle()
    f, p = cfg.make_features(), cfg.make_params()
    ob = orc.build(data, f)
    cam = orc.camera(*scenes.default_camera_args(cfg.width, cfg.height))
    threads = orc.max_threads()
    # calibrate on the central strip, then size the sample to ~15 s
    mid = (cfg.height // 2 // 8) * 8
    t0 = time.perf_counter()
    _, _, _, c0 = orc.render(ob, cam, f, p, cfg.width, cfg.height, rows=(mid, mid + 8))
    dt0 = max(time.perf_counter() - t0, 1e-3)
    n_strips = cfg.height // 8
    budget_strips = max(1, int(15.0 / dt0))
    stride = max(1, -(-n_strips // budget_strips))
    rays, secs, used = 0, 0.0, 0
    for s in range(0, n_strips, stride):
        t0 = time.perf_counter()
        _, _, _, c = orc.render(ob, cam, f, p, cfg.width, cfg.height, rows=(s * 8, s * 8 + 8))
        secs += time.perf_counter() - t0
        rays += c.rays
        used += 1
    return {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{used} of {n_strips} 8-row strips of the same {cfg.width}x{cfg.height} frame "
                      f"(every {stride}th), {rays} rays in {secs:.1f} s, OpenMP {threads} threads",
            "rays": int(rays), "seconds": round(secs, 2)}


if __name__ == "__main__":
    main()
