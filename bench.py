#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Whitted render path.

Metric (BASELINE.json): Mrays/s (primary + shadow + reflection rays actually
traversed) at 1920x1080; frame ms; 1/2/4/8-GPU scaling.

Workload (config.workload = "C3"): the dragon-scale stand-in — 819 200-triangle
torus knot (data/dragon.obj is absent from the reference), 1920x1080, Phong +
hard shadows + reflections depth 4 — synthetic, generated in process. One "step" =
one full frame with the scene, BVH and framebuffer resident in HBM. EVERY FRAME HAS
ITS OWN CAMERA POSE: the default trackball orbits the scene once over the timed
steps (the reference renders one camera per renderRayTracing call, or one camera
per thread in CLI mode, src/main.cpp:512-531); rays are counted per pose in an
untimed pass. Throughput setting (defaults): a launch renders 8 successive frames
(rt_render_device_frames), three launches are in flight on HIP streams of their own,
8x8-pixel primary tiles; every frame is rendered in full, K steps = K frames; when 8
does not divide K the frames go out as equal launches (20 steps = 3 launches of 7, 7
and 6 — never single frames). Before anything is timed every buffer is allocated and
every launch shape has run on every launch stream (StripRenderer.prepare), whatever
--warmup says. `frame_ms_serial` is the other extreme: one frame per launch, one
launch at a time, host-synchronised, the library's default 4x4-pixel tiles.

N GPUs: one process per GPU; the frame's 8-row strips are dealt round-robin to
the ranks, each renders its strips into a packed buffer, the strips are gathered on
rank 0 over RCCL / xGMI (one grouped send / receive per peer) and a de-interleave
kernel writes the reference's row-flipped frame there (strong scaling: the frame
is fixed).

JSON line (rank 0): the contract's keys plus
  roofline      dominant kernel k_trace_level: ALGORITHMIC bytes per launch
                (mean over the timed poses) / average duration of
                the launches of the main shape, measured with HIP events on the
                launch streams during the timed steps; peak 8 TB/s. `traffic`,
                `valu` (instruction count against the VALU issue peak) and
                `salu_insts` come from the PMC passes of the same launch shape
                (profiles/pmc_<workload>.json, tools/r2_profile.sh).
  cpu_baseline  the CPU oracle (kind "port": the repo's restatement of the
                reference path, OpenMP guided rows like src/render.cpp:277-281) on
                this host: whole frames on all cores, a sample on one core.
"""
import argparse
import importlib
import json
import math
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
                         "path trees) overlaps the start of the next launches (default 3); "
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
    ap.add_argument("--exchange", default="gather", choices=["gather", "allgather"],
                    help="several GPUs: gather the strips on rank 0 (grouped send / receive, one peer per xGMI link of "
                         "the root; default) or all-gather them to every rank")
    ap.add_argument("--no-window", action="store_true",
                    help="several ranks, gather form: send whole strips instead of the part inside rt_frame_rect")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="rt_set_option(NAME, VALUE) before the run (tuning experiments; results do not depend on them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-events", action="store_true",
                    help="do not bracket trace kernels with HIP events during the timed steps")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.frames_in_flight is None:
        args.frames_in_flight = 3
    if args.frames_per_launch is None:
        # a launch should hold about 8 frames' worth of work: a rank's share of a frame is 1 / world of one, and a
        # launch of less ends in the same ~0.5 ms tail of its longest path trees (tools/share_partition.py: one rank's
        # share at 8 ranks 0.047 ms per frame with 8 shares per launch, 0.027 with 32)
        args.frames_per_launch = 8 if world == 1 else min(32, 8 * world)
    if args.exchange_batch is None:
        args.exchange_batch = 1 if world == 1 else args.frames_per_launch
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
    if world > 1 or args.exchange_always:
        # shares of frames: the launches in flight must really run side by side (the one-launch tile order does not
        # queue up behind the running trace kernel the way the count / scan / scatter kernels do)
        rt.set_option("order_one_launch", 2)
    for kv in args.set:
        k_, v_ = kv.split("=")
        rt.set_option(k_, int(v_))
    renderer = sharding.StripRenderer(rt, bvh, cam, f, p, W, H, rank, world, strip_rows=STRIP_ROWS,
                                      frames_in_flight=args.frames_in_flight, exchange_always=args.exchange_always,
                                      exchange_batch=args.exchange_batch, exchange=args.exchange, window=not args.no_window)

    # ---- camera poses: every frame its own (the reference renders one camera per renderRayTracing call, or one
    #      camera per thread in CLI mode, src/main.cpp:512-531): the default trackball orbits the scene once over
    #      the timed steps (rotation.y += 360 deg * k / P, P = min(steps, 256) distinct poses)
    fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
    n_poses = max(1, min(args.steps, 256))

    def pose(k):
        ang = 2.0 * math.pi * (k % n_poses) / n_poses
        return rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + ang, rot0[2]))

    cams = [pose(k) for k in range(n_poses)]

    # ---- untimed counting pass: rays and tests of THIS rank's strips, per pose ----
    keys, per_pose = None, []
    for k in range(n_poses):
        c = renderer.count_rays(stream, cam=cams[k])
        keys = keys or list(c.keys())
        per_pose.append([c[x] for x in keys])
    per_pose = np.asarray(per_pose, dtype=np.int64)            # [pose, counter], this rank
    per_pose_all = torch.tensor(per_pose, dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(per_pose_all)
    per_pose_all = per_pose_all.cpu().numpy()                  # all ranks
    reps = np.bincount(np.arange(args.steps) % n_poses, minlength=n_poses).astype(np.int64)  # timed uses of each pose
    total = dict(zip(keys, [int(v) for v in (per_pose_all * reps[:, None]).sum(0)]))   # over the K timed frames
    mine = dict(zip(keys, [int(v) for v in (per_pose * reps[:, None]).sum(0)]))
    rays_total = total["rays"]

    # ---- steady state before anything is timed: every buffer, every stream, every launch shape -----------------
    fpl = max(1, args.frames_per_launch)
    if renderer.sharded and renderer.batch % fpl:
        fpl = 1

    def launch_sizes(n):
        """n frames as launches of fpl frames; when fpl does not divide n, as equal launches of at most fpl frames
        whose count is a multiple of the launches in flight (20 frames, 8 per launch, 2 in flight: 4 x 5 — never
        single frames, and every stream ends at about the same time)"""
        if n <= 0:
            return []
        if n % fpl == 0:
            return [fpl] * (n // fpl)
        m = -(-n // fpl)
        m = -(-m // renderer.flight) * renderer.flight
        m = min(m, n)
        return [n // m + (1 if i < n % m else 0) for i in range(m)]

    sizes = launch_sizes(args.steps)
    main_size = max(set(sizes), key=sizes.count)
    renderer.prepare(stream, sorted(set(sizes + launch_sizes(args.warmup))))

    def run_frames(first, n):
        """frames first .. first + n - 1 of the orbit"""
        k = first
        for m in launch_sizes(n):
            renderer.step(stream, frames=m, cams=[cams[q % n_poses] for q in range(k, k + m)])
            renderer.flush()   # several ranks: a launch's frames go to the root while the next launch renders
            k += m

    run_frames(-args.warmup, args.warmup)
    renderer.finish()     # nothing of the warmup (a partial exchange batch) is left for the timed region

    # ---- timed steps -----------------------------------------------------------
    use_events = not args.no_profile_events
    rt.profile_enable(bvh, use_events)
    launch_streams = renderer.launch_streams(stream)
    if use_events:
        for s_ in launch_streams:
            rt.profile_read(bvh, s_, reset=True)
    renderer.start_log()
    exch0 = renderer.exchanges["bytes"] if hasattr(renderer, "exchanges") else 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run_frames(0, args.steps)
    renderer.finish()
    exch_bytes = (renderer.exchanges["bytes"] - exch0) if hasattr(renderer, "exchanges") else 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    launch_ms = []   # (frames of the launch, trace-kernel milliseconds), every launch of the timed region
    if use_events:
        for j_, s_ in enumerate(launch_streams):
            ms_list = rt.profile_read_launches(bvh, s_, reset=True)
            # with rt_set_option("fused", 0) a frame is depth + 1 launches: sum them per step() call
            per = len(ms_list) // max(1, len(renderer.log[j_]))
            for q, fr in enumerate(renderer.log[j_]):
                launch_ms.append((fr, sum(ms_list[q * per:(q + 1) * per]), per))
    rt.profile_enable(bvh, False)
    tvec = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tvec, op=dist.ReduceOp.MAX)
    elapsed = float(tvec.item())

    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_total / elapsed / 1e6

    # ---- one frame at a time (N = 1): the library's latency setting, no launches in flight ------------------
    frame_ms_serial = None
    if world == 1 and not args.exchange_always:
        rt.set_option("tile_wlog", 2)   # the library's default (latency) setting
        rt.set_option("tile_hlog", 2)
        rt.set_option("leaf_flush", 16)
        lone = sharding.StripRenderer(rt, bvh, cam, f, p, W, H, 0, 1, strip_rows=STRIP_ROWS, frames_in_flight=1)
        lone.prepare(stream, (1,))
        n_serial = min(16, n_poses)
        ts = []
        for k in range(n_serial):
            kk = k * n_poses // n_serial
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lone.step(stream, cams=[cams[kk]])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        td = []   # the default camera pose (the one round 1's single-pose bench rendered)
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lone.step(stream, cams=[cam])
            torch.cuda.synchronize()
            td.append(time.perf_counter() - t0)
        frame_ms_serial = {"mean": round(float(np.mean(ts)) * 1e3, 4), "min": round(min(ts) * 1e3, 4),
                           "max": round(max(ts) * 1e3, 4), "frames": n_serial,
                           "default_pose": round(float(np.mean(td[1:])) * 1e3, 4),
                           "setting": "one frame per launch, one launch at a time, host-synchronised, 4x4-pixel tiles (the library default)"}

    if rank == 0:
        ab = algorithmic_bytes(total, f, W * H * args.steps)
        roofline = None
        full = [x for x in launch_ms if x[0] == main_size]
        if use_events and full:
            # launches of ONE shape only (fpl frames each); this rank's share of their algorithmic bytes over
            # their own kernel time. Bytes per launch: the mean over the timed poses x fpl frames.
            ab_rank = algorithmic_bytes(mine, f, 0)["trace_kernel"]
            per_launch_bytes = ab_rank / args.steps * main_size
            avg_launch_s = sum(x[1] for x in full) / len(full) / 1e3
            achieved = per_launch_bytes / avg_launch_s / 1e9
            roofline = {"bound": "hbm", "kernel": "k_trace_level", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": None, "algorithmic_bytes_per_launch": int(per_launch_bytes),
                        "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches": len(full),
                        "frames_per_launch": main_size, "kernel_launches_per_frame_launch": full[0][2],
                        "other_launches": [[x[0], round(x[1] * 1e3, 2)] for x in launch_ms if x[0] != main_size],
                        "launches_in_flight": renderer.flight,
                        "note": "HBM is the roofline SURVEY.md 8(d) prescribes, but not the ceiling that binds: the "
                                "kernel's bytes come from L1/L2/Infinity Cache (traffic << algorithmic); what binds is "
                                "instruction issue per wave and the latency of dependent fetches (DESIGN.md 3) — see valu"}
            # the launches in flight share the chip: a launch's own duration understates what the chip does by the
            # overlap factor (summed launch durations over the wall time of the timed region)
            overlap = sum(x[1] for x in launch_ms) / 1e3 / elapsed
            chip = ab_rank / elapsed / 1e9
            roofline["chip_wide"] = {"launch_overlap": round(overlap, 3), "achieved": round(chip, 2),
                                     "frac": round(chip / HBM_PEAK_GBS, 4),
                                     "note": "all timed launches' algorithmic bytes over the wall time of the timed region"}
            pj = ROOT / "profiles" / f"pmc_{args.workload}.json"
            if pj.exists() and world == 1:
                # PMC passes of the bench's launch shape (tools/r2_profile.sh: same tile shape, frames per launch as
                # recorded there, ONE launch in flight so that a launch's counters are its own); launches of another
                # frame count are priced per frame
                pm = json.load(open(pj))
                tiles = [int(args.tile_log[0]), int(args.tile_log[1])] if args.tile_log else ([3, 3] if fpl > 1 else [2, 2])
                if pm.get("tile_log") == tiles and pm.get("frames_per_launch"):
                    k = main_size / pm["frames_per_launch"]
                    if pm.get("hbm_bytes_per_launch"):
                        roofline["traffic"] = int(pm["hbm_bytes_per_launch"] * k)
                    v = pm.get("valu_insts_per_launch")
                    if v:
                        peak_issue = 256 * 4 * 2.4e9 / 2   # CUs x SIMDs x clock / 2 cycles per wave64 VALU instruction
                        roofline["valu"] = {"insts": int(v * k), "peak_issue_per_s": peak_issue,
                                            "frac": round(v * k / avg_launch_s / peak_issue, 4),
                                            "active_lane_frac": pm.get("valu_active_lane_frac"),
                                            "source": f"profiles/{pj.name}: rocprofv3 --pmc of {pm.get('command', '')}; "
                                                      f"{pm['frames_per_launch']}-frame launches, priced per frame"}
                        roofline["salu_insts"] = int((pm.get("salu_insts_per_launch") or 0) * k)
                        roofline["chip_wide"]["valu_frac"] = round(v / pm["frames_per_launch"] * args.steps / elapsed / peak_issue, 4)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(cfg, data, scenes)
        rays_bvh = total["rays"] - total.get("rays_culled", 0)
        per_frame = lambda v: int(round(v / args.steps))  # noqa: E731
        out = {
            "metric": "Mrays/s (primary+shadow+reflect) at 1920x1080",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frame_ms_serial": frame_ms_serial,
            "mrays_bvh_traversed": round(rays_bvh / elapsed / 1e6, 2),
            "config": {"workload": args.workload, "scene": data.name, "triangles": data.n_triangles,
                       "resolution": [W, H], "depth": cfg.depth, "features": cfg.features,
                       "camera": f"default trackball orbiting the scene: {n_poses} distinct poses over the timed steps, "
                                 "one per frame (counters below: mean per frame over the timed poses)",
                       "rays_per_frame": per_frame(total["rays"]),
                       "rays_bvh_traversed": per_frame(rays_bvh),
                       "rays_culled_before_traversal": per_frame(total.get("rays_culled", 0)),
                       "rays_exact_engine": round(total.get("rays_exact", 0) / args.steps, 2),
                       "rays_shadow_skipped": per_frame(total.get("rays_shadow_skipped", 0)),
                       "rays_reference_equivalent": per_frame(total["rays_ref_equiv"]),
                       "rays_primary": per_frame(total["rays_primary"]), "rays_shadow": per_frame(total["rays_shadow"]),
                       "rays_reflect": per_frame(total["rays_reflect"]), "box_tests": per_frame(total["box_tests"]),
                       "tri_tests": per_frame(total["tri_tests"]), "group_tests": per_frame(total.get("group_tests", 0)),
                       "node_fetches": per_frame(total["node_fetches"]),
                       "tri_fetches": per_frame(total["tri_fetches"]), "algorithmic_bytes_per_frame": per_frame(ab["frame"]),
                       "parallelism": f"strips{STRIP_ROWS}x{world}" + (f"+{args.exchange}(overlapped" + (f",{renderer.batch}-frames-each)" if renderer.batch > 1 else ")")
                                                              if renderer.sharded else "")
                       + (f"+{renderer.flight}-launches-in-flight" if renderer.flight > 1 else "")
                       + (f"+{fpl}-frames-per-launch" if fpl > 1 else "")
                       + ("+REHEARSAL-gloo-one-device" if rehearsal else ""),
                       "exchange": ({"form": args.exchange, "window": bool(getattr(renderer, "window", False)),
                                     "root_receives_bytes_per_frame": int(exch_bytes / max(args.steps, 1)),
                                     "whole_strips_would_be": int(W * H * 12 * (world - 1) / world)}
                                    if renderer.sharded and hasattr(renderer, "exchanges") else None),
                       "launch_frames": sizes if len(sizes) <= 8 else {str(z): sizes.count(z) for z in sorted(set(sizes))},
                       "device": devname, "bvh_build_s": round(build_s, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.exchange_always:
        dist.barrier()
        dist.destroy_process_group()


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, data, scenes):
    """The CPU oracle (restatement of the reference path; traces reflections twice like the reference, OpenMP
    `parallel for schedule(guided)` over the image rows like src/render.cpp:277-281) on the same frame, on this
    host: ALL cores — whole frames in one call each, so every thread has rows to take, repeated until ~10 s have
    passed — and ONE core on a bounded sample (every k-th 8-row strip, ~10 s)."""
    from oracle.cpu_oracle import Oracle

    orc = Oracle()
    f, p = cfg.make_features(), cfg.make_params()
    ob = orc.build(data, f)
    W, H = cfg.width, cfg.height
    cam = orc.camera(*scenes.default_camera_args(W, H))
    threads = orc.max_threads()
    # all cores: whole frames
    rays, secs, frames = 0, 0.0, 0
    while secs < 10.0 and frames < 64:
        t0 = time.perf_counter()
        _, _, _, c = orc.render(ob, cam, f, p, W, H, want_rgb=True, threads=threads)
        secs += time.perf_counter() - t0
        rays += c.rays
        frames += 1
    # one core: calibrate on the central strip, then every stride-th strip for ~10 s
    n_strips = H // 8
    mid = (n_strips // 2) * 8
    t0 = time.perf_counter()
    orc.render(ob, cam, f, p, W, H, rows=(mid, mid + 8), threads=1)
    dt0 = max(time.perf_counter() - t0, 1e-3)
    stride = max(1, -(-n_strips // max(1, int(10.0 / dt0))))
    rays1, secs1, used = 0, 0.0, 0
    for s_ in range(0, n_strips, stride):
        t0 = time.perf_counter()
        _, _, _, c = orc.render(ob, cam, f, p, W, H, rows=(s_ * 8, s_ * 8 + 8), threads=1)
        secs1 += time.perf_counter() - t0
        rays1 += c.rays
        used += 1
    return {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "value_1core": round(rays1 / secs1 / 1e6, 4), "cpu_model": cpu_model(),
            "sample": f"all cores: {frames} whole {W}x{H} frames, one call each (OpenMP guided rows, {threads} threads), "
                      f"{rays} reference-equivalent rays in {secs:.1f} s; 1 core: {used} of {n_strips} 8-row strips "
                      f"(every {stride}th), {rays1} rays in {secs1:.1f} s",
            "rays": int(rays), "seconds": round(secs, 2), "frame_ms_all_cores": round(secs / frames * 1e3, 1)}


if __name__ == "__main__":
    main()
