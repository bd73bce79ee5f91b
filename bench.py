#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Whitted render path.

Metric (BASELINE.json): Mrays/s (primary + shadow + reflection rays; one ray = one call-equivalent of
BvhInterface::intersect) at 1920x1080; frame ms; 1/2/4/8-GPU scaling.

Workload (config.workload = "C3", the configuration the metric is quoted on): the dragon-scale stand-in — 819 200-
triangle torus knot (data/dragon.obj is absent from the reference), 1920x1080, Phong + hard shadows + reflections
depth 4 — synthetic, generated in process. One "step" = one full frame with the scene, BVH and framebuffer resident
in HBM. EVERY FRAME HAS ITS OWN CAMERA POSE: the default trackball orbits the scene once over the timed steps (the
reference renders one camera per renderRayTracing call, or one camera per thread in CLI mode, src/main.cpp:512-531);
rays are counted per pose in an untimed pass. Throughput setting (defaults): a launch renders 8 successive frames
(rt_render_device_frames), three launches are in flight on HIP streams of their own, 8x8-pixel primary tiles; every
frame is rendered in full, K steps = K frames; when 8 does not divide K the frames go out as equal launches (20 steps
= 7 + 7 + 6 — never single frames). Before anything is timed every buffer is allocated and every launch shape has run
on every launch stream (StripRenderer.prepare), whatever --warmup says.

After the C3 region the same protocol times BASELINE.json's other GPU configs — `workloads`: C2, C4, C5 at N = 1; C4
(the config BASELINE assigns to 8 GPUs at 1080p) at N > 1 — with --workload-steps frames each, so that every config has
a figure observed by whoever runs this file. `frame_ms_serial` is the other extreme of the protocol: one frame per
launch, one launch at a time, host-synchronised, the library's own choice of tile (tile_auto: rows of 32 / 64 pixels
for point lights on triangles, 4x4 for area lights); `frame_ms_host_call` is the drop-in call
itself: rt_render into a host buffer (what renderRayTracing → Screen costs), PCIe included — never `value`.

N GPUs: one process per GPU (started by torch.distributed.run — or by this file itself: `python bench.py --gpus N`
without a launcher starts its N ranks as child processes BEFORE anything touches a GPU and relays rank 0's line and
exit code); the frame's 8-row strips are dealt round-robin to the ranks, each renders its strips into a packed buffer,
the strips are gathered on rank 0 over RCCL / xGMI (one grouped send / receive per peer; the first exchange is checked
against the all-gather form, `config.exchange.fallback`) and a de-interleave kernel writes the reference's
row-flipped frame there (strong scaling: the frame is fixed).

JSON line (rank 0): the contract's keys plus
  roofline      dominant kernel k_trace_level. THREE measured ceilings, each <= 1 by construction, from a pass with ONE
                launch in flight (a launch's duration and counters are its own): HBM (PMC traffic of the launch /
                its duration / 8 TB/s), L2 (TCC requests x 128 B / duration / 34.5 TB/s) and instruction issue (VALU
                wave-instructions / duration / (1024 SIMDs x 2.4 GHz / 2 cycles); all wave-instructions against one
                per SIMD and cycle). `bound` names the highest; achieved / peak / unit / frac are that ceiling's.
                Counters come from the committed PMC passes of the same launch shape (profiles/pmc_<workload>.json,
                tools/r3_profile.sh: separate --pmc runs, one launch in flight); durations are measured live with HIP
                events on the launch streams. `algorithmic` keeps SURVEY.md 8(d)'s figure (bytes the tests would
                fetch without any cache / launch duration): it exceeds what HBM delivers because those bytes come
                from L1 / L2 / Infinity Cache — a yardstick, not a fraction of a ceiling. `timed` prices the whole
                timed region (launches overlap there) with the same three measured ceilings.
  cpu_baseline  the CPU oracle (kind "port": the repo's restatement of the reference path, OpenMP guided rows like
                src/render.cpp:277-281) on this host: whole frames on all cores, a sample on one core.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; L2 34.5 TB/s aggregate; v_fma_f32 (wave64) 2 cycles per SIMD, 256 CUs x 4
# SIMDs at 2.4 GHz; a wave issues at most one instruction per ~4 cycles and a SIMD holds 4 of this kernel's waves
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
VALU_PEAK_PER_S = 256 * 4 * 2.4e9 / 2
ISSUE_PEAK_PER_S = 256 * 4 * 2.4e9
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--workload-steps", type=int, default=20,
                    help="timed frames of each of the OTHER configs (`workloads`: C2, C4, C5 at one GPU, C4 at several) "
                         "after the main region; 0 = skip them")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="successive frames on alternating HIP streams, so that the tail of frame k (a few long "
                         "path trees) overlaps the start of the next launches (default 3); "
                         "1 = strictly one frame at a time")
    ap.add_argument("--tile-log", type=int, nargs=2, default=None, metavar=("W", "H"),
                    help="primary tile = 2^W x 2^H pixels per wave (default: 3 3 with several frames per launch and "
                         "launches in flight, 2 3 with frames in flight only, else the library's latency setting)")
    ap.add_argument("--exchange-batch", type=int, default=None,
                    help="several GPUs: successive frames that share one exchange + one de-interleave launch "
                         "(default: the frames of a launch); every frame is still rendered and assembled")
    ap.add_argument("--frames-per-launch", type=int, default=None,
                    help="successive frames rendered by ONE launch (rt_render_device_frames; default 8): a launch of "
                         "several frames has no idle tail between them — what a rank's small share of a frame needs "
                         "at 8 ranks, and worth 5 %% on one GPU; every frame is rendered in full")
    ap.add_argument("--exchange-always", action="store_true",
                    help="one GPU: take the strips + RCCL exchange path with a one-rank group (what the N > 1 run "
                         "adds on top of the render: packed strips, the exchange, de-interleave), for measurement; "
                         "with --exchange gather the root is its own peer (grouped ncclSend / ncclRecv to itself)")
    ap.add_argument("--exchange", default="gather", choices=["gather", "allgather"],
                    help="several GPUs: gather the strips on rank 0 (grouped send / receive, one peer per xGMI link of "
                         "the root; default) or all-gather them to every rank")
    ap.add_argument("--no-window", action="store_true",
                    help="several ranks, gather form: send whole strips instead of the part inside rt_frame_rect")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="rt_set_option(NAME, VALUE) before the run (tuning experiments; results do not depend on them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial", action="store_true", help="skip frame_ms_serial / frame_ms_host_call")
    ap.add_argument("--no-profile-events", action="store_true",
                    help="do not bracket trace kernels with HIP events during the timed steps")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes
    (torch.distributed.run) before this process has made any GPU call, relay their output (rank 0 prints the JSON
    line) and return their exit code. Never re-executes a process that has touched the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs across processes on this host
    env["BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


class Settings:
    """The library's result-neutral tuning knobs as the bench sets them: a mode's defaults, then whatever --tile-log /
    --set ask for (so that a tuning run reports figures for the options it names), with a description of what is in
    force."""

    def __init__(self, rt, args, world):
        self.rt, self.args, self.world = rt, args, world

    def apply(self, mode: str, features: dict = None) -> str:
        a = self.args
        if mode == "throughput":
            # 8x8-pixel primary tiles (64 rays per wave) do the least instruction work per ray but end in the longest
            # path-tree tails — which a launch of several frames and the next launch in flight cover; a lone frame
            # prefers the library default. 64-ray waves fill a longer leaf queue before a flush pays.
            # Results are identical for every setting (tests/test_gpu_options.py).
            opts = {"tile_wlog": 3, "tile_hlog": 3, "leaf_flush": 32}
            # several ranks: a launch holds only 1 / N of its frames' rows, so the launch's tail — the longest path
            # trees of its tiles — weighs more. Point lights: rows of pixels mix cheap and expensive path trees in one
            # wave and end sooner (one rank's share of the 20-frame region, tools/share_region.py: 1.10 -> 0.89 ms at 8
            # ranks, 1.64 -> 1.46 at 4, 2.51 -> 2.41 at 2 with 16 x 4; at one rank 8 x 8 stays best: 4.21 against 4.31 -
            # 4.36). Area lights (64 samples per hit): fewer hits per wave (C4: 11.5 -> 8.2 ms at 8 ranks with 4 x 4,
            # 16.8 -> 14.6 at 4 with 8 x 4; 8 x 8 at one and two ranks). profiles/r03_notes.md §7.
            area = bool(features and features.get("enableSoftShadow"))
            if not area:
                if self.world >= 3:
                    opts.update(tile_wlog=6, tile_hlog=0)
                elif self.world == 2:
                    opts.update(tile_wlog=4, tile_hlog=2)
            elif self.world >= 6:
                opts.update(tile_wlog=2, tile_hlog=2, leaf_flush=24)
            elif self.world >= 3:
                opts.update(tile_wlog=3, tile_hlog=2, leaf_flush=24)
        elif mode == "flight":
            opts = {"tile_wlog": 2, "tile_hlog": 3, "leaf_flush": 16}
        else:  # "latency": the library defaults — the tile shape picked per frame type (tile_auto)
            opts = {"leaf_flush": 16, "tile_auto": 1}
        # shares of frames: the launches in flight must really run side by side (the one-launch tile order does not
        # queue up behind the running trace kernel the way the count / scan / scatter kernels do)
        opts["order_one_launch"] = 2 if (self.world > 1 or a.exchange_always) and mode != "latency" else 1
        if a.tile_log and mode != "latency":
            opts["tile_wlog"], opts["tile_hlog"] = a.tile_log
        for kv in a.set:
            k_, v_ = kv.split("=")
            opts[k_] = int(v_)
        if ("tile_wlog" in opts or "tile_hlog" in opts) and not any(kv.startswith("tile_auto=") for kv in a.set):
            opts.pop("tile_auto", None)   # an explicit shape is meant
        self.rt.set_option("tile_hlog", 0)   # any (wlog, hlog) pair is reachable from here (their sum is bounded)
        for k_, v_ in opts.items():
            if k_ != "tile_auto":
                self.rt.set_option(k_, v_)
        if "tile_auto" in opts:               # last: setting a shape switches it off
            self.rt.set_option("tile_auto", opts["tile_auto"])
        self.tiles = [opts.get("tile_wlog"), opts.get("tile_hlog")]
        return ", ".join(f"{k_}={v_}" for k_, v_ in sorted(opts.items()))


class Bench:
    """Everything the workloads share: the library, the process group, the arguments."""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if args.frames_in_flight is None:
            args.frames_in_flight = 3
        if args.frames_per_launch is None:
            # a launch should hold about 8 frames' worth of work: a rank's share of a frame is 1 / world of one, and a
            # launch of less ends in the same ~0.5 ms tail of its longest path trees (tools/share_partition.py)
            args.frames_per_launch = 8 if self.world == 1 else min(32, 8 * self.world)
        if args.exchange_batch is None:
            args.exchange_batch = 1 if self.world == 1 and not args.exchange_always else args.frames_per_launch
        # HIP runtime setting, read when the runtime starts: how many hardware queues the streams of this process are
        # dealt over (default 4) = how many of the launches in flight really run side by side (tools/queue_bench.hip);
        # + the exchange stream, torch's RCCL stream and the null stream
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, args.frames_in_flight + 4)))

        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        # Rehearsal on a one-GPU box only (never a reported number): BENCH_REHEARSAL=1 puts every rank on cuda:0 and
        # lets gloo carry the exchange, to exercise the N > 1 code path.
        self.rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.sharded = self.world > 1 or args.exchange_always
        self.n_ranks_seen = 1
        if self.sharded:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            import datetime
            to = datetime.timedelta(seconds=180)   # a rank stuck in its first collective ends the run, not the box's limit
            if self.rehearsal:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=to)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, timeout=to,
                                        device_id=torch.device("cuda", self.local_rank))
            self.n_ranks_seen = dist.get_world_size()
        self.capi = importlib.import_module("computer-graphics-engine_amd.capi")
        self.scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
        self.sharding = importlib.import_module("computer-graphics-engine_amd.sharding")
        self.rt = self.capi.Library()
        self.rt.set_device(self.local_rank)
        self.ndev, self.devname = self.rt.device_info()
        self.settings = Settings(self.rt, args, self.world)
        self.stream = torch.cuda.current_stream().cuda_stream

    # ------------------------------------------------------------------------------------------------------
    def run(self, workload: str, steps: int, warmup: int, detail: bool) -> dict:
        """The protocol for one config: untimed counting pass (one pose per frame), steady state, `warmup` untimed
        and `steps` timed frames (barrier + synchronize on both sides, MAX over ranks). detail: also the
        one-launch-in-flight pass behind the roofline, and the per-launch records."""
        torch, dist, rt, args, world = self.torch, self.dist, self.rt, self.args, self.world
        scenes, sharding = self.scenes, self.sharding
        cfg = scenes.CONFIGS[workload]
        W, H = cfg.width, cfg.height
        f, p = cfg.make_features(), cfg.make_params()
        data = scenes.get_scene(cfg.scene)
        t0 = time.time()
        scene = rt.scene(data)
        bvh = rt.bvh(scene, f)
        build_s = time.time() - t0
        cam = rt.camera(*scenes.default_camera_args(W, H))
        stream = self.stream

        fpl_arg = max(1, args.frames_per_launch)
        mode = "throughput" if fpl_arg > 1 else ("flight" if args.frames_in_flight > 1 else "latency")
        setting = self.settings.apply(mode, cfg.features)
        tiles = list(self.settings.tiles)
        renderer = sharding.StripRenderer(rt, bvh, cam, f, p, W, H, self.rank, world, strip_rows=STRIP_ROWS,
                                          frames_in_flight=args.frames_in_flight, exchange_always=args.exchange_always,
                                          exchange_batch=args.exchange_batch, exchange=args.exchange,
                                          window=not args.no_window, loopback=args.exchange_always)

        # camera poses: every frame its own — the default trackball orbits the scene once over the timed steps
        # (rotation.y += 360 deg * k / P, P = min(steps, 256) distinct poses)
        fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
        n_poses = max(1, min(steps, 256))
        cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2.0 * math.pi * k / n_poses, rot0[2]))
                for k in range(n_poses)]

        # untimed counting pass: rays and tests of THIS rank's strips, per pose
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
        reps = np.bincount(np.arange(steps) % n_poses, minlength=n_poses).astype(np.int64)  # timed uses of each pose
        total = dict(zip(keys, [int(v) for v in (per_pose_all * reps[:, None]).sum(0)]))   # over the K timed frames
        mine = dict(zip(keys, [int(v) for v in (per_pose * reps[:, None]).sum(0)]))

        # steady state before anything is timed: every buffer, every stream, every launch shape
        fpl = fpl_arg
        if renderer.sharded and renderer.batch % fpl:
            fpl = 1

        def launch_sizes(n):
            """n frames as launches of fpl frames; when fpl does not divide n, as equal launches of at most fpl
            frames whose count is a multiple of the launches in flight (20 frames, 8 per launch, 3 in flight:
            7 + 7 + 6 — never single frames, and every stream ends at about the same time)"""
            if n <= 0:
                return []
            if n % fpl == 0:
                return [fpl] * (n // fpl)
            m = -(-n // fpl)
            m = -(-m // renderer.flight) * renderer.flight
            m = min(m, n)
            return [n // m + (1 if i < n % m else 0) for i in range(m)]

        sizes = launch_sizes(steps)
        main_size = max(set(sizes), key=sizes.count)
        renderer.prepare(stream, sorted(set(sizes + launch_sizes(warmup))))

        def run_frames(first, n):
            """frames first .. first + n - 1 of the orbit"""
            k = first
            for m in launch_sizes(n):
                renderer.step(stream, frames=m, cams=[cams[q % n_poses] for q in range(k, k + m)])
                renderer.flush()   # several ranks: a launch's frames go to the root while the next launch renders
                k += m

        run_frames(-warmup, warmup)
        renderer.finish()     # nothing of the warmup (a partial exchange batch) is left for the timed region

        # ---- timed steps ----
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
        run_frames(0, steps)
        renderer.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        exch_bytes = (renderer.exchanges["bytes"] - exch0) if hasattr(renderer, "exchanges") else 0
        launch_ms = []   # (frames of the launch, trace-kernel milliseconds, kernels per launch), every timed launch
        if use_events:
            for j_, s_ in enumerate(launch_streams):
                ms_list = rt.profile_read_launches(bvh, s_, reset=True)
                # with rt_set_option("fused", 0) a frame is depth + 1 launches: sum them per step() call
                per = len(ms_list) // max(1, len(renderer.log[j_]))
                for q, fr in enumerate(renderer.log[j_]):
                    launch_ms.append((fr, sum(ms_list[q * per:(q + 1) * per]), per))
        tvec = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tvec, op=dist.ReduceOp.MAX)
        elapsed = float(tvec.item())

        # ---- ONE launch in flight, the main shape: a launch's duration is its own (what the PMC passes measure) ----
        alone_ms = []
        if detail and use_events:
            renderer.log = None
            # one discarded launch, then launches spread evenly over the orbit (a launch's duration depends on its
            # poses: the mean over the orbit is what repeats from run to run)
            n_alone = min(8, -(-n_poses // main_size))
            for j in [-1] + list(range(n_alone)):
                k = max(j, 0) * n_poses // n_alone
                renderer.step(stream, frames=main_size, cams=[cams[q % n_poses] for q in range(k, k + main_size)])
                renderer.finish()
                if j < 0:   # the discarded one
                    for s_ in launch_streams:
                        rt.profile_read_launches(bvh, s_, reset=True)
            for s_ in launch_streams:
                ms_list = rt.profile_read_launches(bvh, s_, reset=True)
                per = max(1, launch_ms[0][2]) if launch_ms else 1
                alone_ms += [sum(ms_list[q:q + per]) for q in range(0, len(ms_list) - per + 1, per)]
        rt.profile_enable(bvh, False)

        rays_bvh = total["rays"] - total.get("rays_culled", 0)
        res = {"workload": workload, "cfg": cfg, "data": data, "features": f, "W": W, "H": H, "steps": steps,
               "elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3, "value": total["rays"] / elapsed / 1e6,
               "mrays_bvh": rays_bvh / elapsed / 1e6, "total": total, "mine": mine, "rays_bvh": rays_bvh,
               "launch_ms": launch_ms, "alone_ms": alone_ms, "sizes": sizes, "main_size": main_size, "fpl": fpl,
               "tiles": tiles, "setting": setting, "n_poses": n_poses, "build_s": build_s, "exch_bytes": exch_bytes,
               "flight": renderer.flight, "batch": renderer.batch, "renderer_sharded": renderer.sharded,
               "window": bool(getattr(renderer, "window", False)), "fallback": renderer.fallback,
               "has_exchanges": hasattr(renderer, "exchanges"), "serial": None, "host_call": None}

        # ---- one frame at a time (N = 1): the library's latency setting, no launches in flight ----
        if world == 1 and not args.exchange_always and not args.no_serial:
            lat = self.settings.apply("latency")
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
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lone.step(stream, cams=[cam])
                torch.cuda.synchronize()
                td.append(time.perf_counter() - t0)
            res["serial"] = {"mean": round(float(np.mean(ts)) * 1e3, 4), "min": round(min(ts) * 1e3, 4),
                             "max": round(max(ts) * 1e3, 4), "frames": n_serial,
                             "default_pose": round(float(np.mean(td[1:])) * 1e3, 4),
                             "setting": f"one frame per launch, one launch at a time, host-synchronised; {lat}"}
            # the drop-in call: rt_render into the caller's host buffer (renderRayTracing -> Screen), same pose
            # every call like the GUI's re-render (src/main.cpp:465-470); PCIe-inclusive, never `value`
            host = np.zeros((H, W, 3), np.float32)
            th = []
            import ctypes as C
            for _ in range(6):
                t0 = time.perf_counter()
                rt._check(rt.c.rt_render(bvh.handle, C.byref(cam), C.byref(f), C.byref(p), W, H, 0, H,
                                         host.ctypes.data, None))
                th.append(time.perf_counter() - t0)
            res["host_call"] = {"mean": round(float(np.mean(th[2:])) * 1e3, 4), "min": round(min(th[2:]) * 1e3, 4),
                                "first_call": round(th[0] * 1e3, 4), "calls": len(th) - 2,
                                "setting": "rt_render(bvh, camera, ..., float* host_rgb): render + copy into the caller's "
                                           f"pageable buffer ({W * H * 12 / 1e6:.1f} MB), default pose; {lat}"}
            del lone
            self.settings.apply(mode, cfg.features)
        del renderer
        bvh.close()
        scene.close()
        torch.cuda.empty_cache()
        return res

    # ------------------------------------------------------------------------------------------------------
    def roofline(self, res: dict):
        """The roofline object of the main workload (docstring of this file)."""
        launch_ms, main_size, steps, elapsed = res["launch_ms"], res["main_size"], res["steps"], res["elapsed"]
        full = [x for x in launch_ms if x[0] == main_size]
        if not full:
            return None
        ab_rank = algorithmic_bytes(res["mine"], res["features"], 0)["trace_kernel"]
        per_launch_bytes = ab_rank / steps * main_size
        avg_timed_s = sum(x[1] for x in full) / len(full) / 1e3
        alone = res["alone_ms"]
        # the launches of the alone pass differ only by their poses (spread over the orbit): their mean
        alone_s = (sum(alone) / len(alone) / 1e3) if alone else None
        overlap = sum(x[1] for x in launch_ms) / 1e3 / elapsed
        dur = alone_s or avg_timed_s
        out = {"bound": "hbm", "kernel": "k_trace_level", "achieved": None, "peak": None, "unit": None, "frac": None,
               "traffic": None,
               "alone": {"launch_us": round(dur * 1e6, 2), "launches": len(alone),
                         "frames_per_launch": main_size, "each_us": [round(x * 1e3, 1) for x in res["alone_ms"]],
                         "note": "ONE launch of the main shape in flight, host-synchronised between launches: the "
                                 "duration a launch has to itself (mean over launches spread over the orbit, after one discarded launch)"
                         if alone_s else "no alone pass: the timed launches' own average (they overlap)"},
               "algorithmic": {"bytes_per_launch": int(per_launch_bytes),
                               "rate_gbs": round(per_launch_bytes / dur / 1e9, 2),
                               "vs_hbm_peak": round(per_launch_bytes / dur / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "SURVEY.md 8(d): 32 B per box test + 36 B per triangle test + per-hit attributes, "
                                       "whatever cache serves them — above 1 means the bytes never came from HBM; "
                                       "a yardstick for the traversal's work, not a fraction of a ceiling"},
               "timed": {"avg_launch_us": round(avg_timed_s * 1e6, 2), "launches": len(full),
                         "frames_per_launch": main_size, "kernel_launches_per_frame_launch": full[0][2],
                         "other_launches": [[x[0], round(x[1] * 1e3, 2)] for x in launch_ms if x[0] != main_size],
                         "launches_in_flight": res["flight"], "launch_overlap": round(overlap, 3),
                         "note": "launches of the timed region share the chip: their durations overlap"}}
        pj = ROOT / "profiles" / f"pmc_{res['workload']}.json"
        pm = json.load(open(pj)) if pj.exists() else None
        if not (pm and self.world == 1 and pm.get("tile_log") == res["tiles"] and pm.get("frames_per_launch")):
            out["note"] = "no PMC passes of this launch shape under profiles/: only the algorithmic figure is available"
            out["achieved"], out["peak"], out["unit"] = out["algorithmic"]["rate_gbs"], HBM_PEAK_GBS, "GB/s"
            out["frac"] = None
            return out
        # PMC passes of the bench's launch shape (same tiles, ONE launch in flight); launches of another frame count
        # are priced per frame
        k = main_size / pm["frames_per_launch"]
        traffic = pm["hbm_bytes_per_launch"] * k
        l2_bytes = pm.get("l2_requests_per_launch", 0) * 128.0 * k
        valu = pm["valu_insts_per_launch"] * k
        insts = pm.get("all_insts_per_launch", 0) * k
        ceil = {"hbm": {"achieved": round(traffic / dur / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(traffic / dur / 1e9 / HBM_PEAK_GBS, 4),
                        "what": "PMC traffic (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction) "
                                "per launch / the launch's duration"},
                "l2": {"achieved": round(l2_bytes / dur / 1e9, 2), "peak": L2_PEAK_GBS, "unit": "GB/s",
                       "frac": round(l2_bytes / dur / 1e9 / L2_PEAK_GBS, 4),
                       "what": "(TCC_HIT + TCC_MISS) x 128 B per launch / duration", "hit_rate": pm.get("l2_hit_rate")},
                "valu_issue": {"achieved": round(valu / dur / 1e9, 2), "peak": VALU_PEAK_PER_S / 1e9,
                               "unit": "G wave-instructions/s", "frac": round(valu / dur / VALU_PEAK_PER_S, 4),
                               "what": "SQ_INSTS_VALU per launch / duration against 1024 SIMDs x 2.4 GHz / 2 cycles per "
                                       "wave64 VALU instruction", "active_lane_frac": pm.get("valu_active_lane_frac")},
                "issue_all": {"achieved": round(insts / dur / 1e9, 2), "peak": ISSUE_PEAK_PER_S / 1e9,
                              "unit": "G wave-instructions/s", "frac": round(insts / dur / ISSUE_PEAK_PER_S, 4),
                              "what": "VALU + SALU + LDS + VMEM + SMEM wave-instructions per launch / duration against "
                                      "one instruction per SIMD and cycle (four resident waves x one per 4 cycles)"}}
        out["ceilings"] = ceil
        bound = max(ceil, key=lambda c: ceil[c]["frac"])
        out["bound"] = bound
        for f_ in ("achieved", "peak", "unit", "frac"):
            out[f_] = ceil[bound][f_]
        out["traffic"] = int(traffic)
        out["wave_time"] = pm.get("wave_time")   # where a resident wave's cycles go (issuing / waiting / stalled)
        out["insts_per_launch"] = {"valu": int(valu), "salu": int(pm.get("salu_insts_per_launch", 0) * k),
                                   "all": int(insts)}
        out["source"] = (f"profiles/{pj.name}: rocprofv3 --pmc of {pm.get('command', '')}; "
                         f"{pm['frames_per_launch']}-frame launches, priced per frame")
        out["profile_alone_launch_us"] = pm.get("alone_launch_us")   # the same figure when the profile was taken
        # the whole timed region against the same measured ceilings (frames / wall time)
        per_frame = 1.0 / pm["frames_per_launch"] * steps / elapsed
        out["timed"]["chip_wide"] = {
            "hbm_frac": round(pm["hbm_bytes_per_launch"] * per_frame / 1e9 / HBM_PEAK_GBS, 4),
            "l2_frac": round(pm.get("l2_requests_per_launch", 0) * 128.0 * per_frame / 1e9 / L2_PEAK_GBS, 4),
            "valu_issue_frac": round(pm["valu_insts_per_launch"] * per_frame / VALU_PEAK_PER_S, 4),
            "issue_all_frac": round(pm.get("all_insts_per_launch", 0) * per_frame / ISSUE_PEAK_PER_S, 4),
            "note": "the PMC passes' per-frame counters x the timed frames / the wall time of the timed region"}
        out["note"] = ("no bandwidth ceiling is near: the kernel's bytes come from L1 / L2 / Infinity Cache; what binds is "
                       "instruction issue per wave (one wave issues at most one instruction per ~4 cycles, four waves "
                       "per SIMD at 128 VGPRs) together with the latency of dependent node fetches (wave_time) — "
                       "DESIGN.md 3")
        return out

    # ------------------------------------------------------------------------------------------------------
    def config_of(self, res: dict) -> dict:
        total, steps, world, args = res["total"], res["steps"], self.world, self.args
        per_frame = lambda v: int(round(v / steps))  # noqa: E731
        ab = algorithmic_bytes(total, res["features"], res["W"] * res["H"] * steps)
        sizes, cfg = res["sizes"], res["cfg"]
        return {"workload": res["workload"], "scene": res["data"].name, "triangles": res["data"].n_triangles,
                "resolution": [res["W"], res["H"]], "depth": cfg.depth, "features": cfg.features,
                "camera": f"default trackball orbiting the scene: {res['n_poses']} distinct poses over the timed steps, "
                          "one per frame (counters below: mean per frame over the timed poses)",
                "rays_per_frame": per_frame(total["rays"]),
                "rays_bvh_traversed": per_frame(res["rays_bvh"]),
                "rays_culled_before_traversal": per_frame(total.get("rays_culled", 0)),
                "rays_exact_engine": round(total.get("rays_exact", 0) / steps, 2),
                "rays_shadow_skipped": per_frame(total.get("rays_shadow_skipped", 0)),
                "rays_reference_equivalent": per_frame(total["rays_ref_equiv"]),
                "rays_primary": per_frame(total["rays_primary"]), "rays_shadow": per_frame(total["rays_shadow"]),
                "rays_reflect": per_frame(total["rays_reflect"]), "box_tests": per_frame(total["box_tests"]),
                "tri_tests": per_frame(total["tri_tests"]), "group_tests": per_frame(total.get("group_tests", 0)),
                "node_fetches": per_frame(total["node_fetches"]),
                "tri_fetches": per_frame(total["tri_fetches"]), "algorithmic_bytes_per_frame": per_frame(ab["frame"]),
                "parallelism": f"strips{STRIP_ROWS}x{world}"
                               + ((f"+{'allgather' if res['fallback'] else args.exchange}(overlapped"
                                   + (f",{res['batch']}-frames-each)" if res["batch"] > 1 else ")"))
                                  if res["renderer_sharded"] else "")
                               + (f"+{res['flight']}-launches-in-flight" if res["flight"] > 1 else "")
                               + (f"+{res['fpl']}-frames-per-launch" if res["fpl"] > 1 else "")
                               + ("+REHEARSAL-gloo-one-device" if self.rehearsal else ""),
                "exchange": ({"form": "allgather" if res["fallback"] else args.exchange, "window": res["window"],
                              "fallback": res["fallback"],
                              "root_receives_bytes_per_frame": int(res["exch_bytes"] / max(steps, 1)),
                              "whole_strips_would_be": int(res["W"] * res["H"] * 12 * (world - 1) / world)}
                             if res["renderer_sharded"] and res["has_exchanges"] else None),
                "launch_frames": sizes if len(sizes) <= 8 else {str(z): sizes.count(z) for z in sorted(set(sizes))},
                "settings": res["setting"], "device": self.devname, "bvh_build_s": round(res["build_s"], 3)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    b = Bench(args)
    main_res = b.run(args.workload, args.steps, args.warmup, detail=True)
    others = {}
    if args.workload_steps > 0 and args.workload == "C3":
        for name in (("C2", "C4", "C5") if b.world == 1 and not args.exchange_always else ("C4",)):
            r = b.run(name, args.workload_steps, min(args.warmup, 8), detail=False)
            c = b.config_of(r)
            others[name] = {"ms_per_step": round(r["ms_per_step"], 4), "value": round(r["value"], 2),
                            "mrays_bvh_traversed": round(r["mrays_bvh"], 2), "steps": r["steps"],
                            "frame_ms_serial": r["serial"], "frame_ms_host_call": r["host_call"],
                            "resolution": c["resolution"], "rays_per_frame": c["rays_per_frame"],
                            "rays_bvh_traversed": c["rays_bvh_traversed"],
                            "rays_shadow_skipped": c["rays_shadow_skipped"], "box_tests": c["box_tests"],
                            "tri_tests": c["tri_tests"], "launch_frames": c["launch_frames"],
                            "parallelism": c["parallelism"], "exchange": c["exchange"]}
    if b.rank == 0:
        cpu = None
        if not args.no_cpu_baseline and b.world == 1:
            cpu = cpu_baseline(main_res["cfg"], main_res["data"], b.scenes)
        out = {
            "metric": "Mrays/s (primary+shadow+reflect) at 1920x1080",
            "value": round(main_res["value"], 2), "unit": "Mrays/s", "n_gpus": b.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(main_res["ms_per_step"], 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "n_ranks_seen": b.n_ranks_seen,
            "frame_ms_serial": main_res["serial"], "frame_ms_host_call": main_res["host_call"],
            "mrays_bvh_traversed": round(main_res["mrays_bvh"], 2),
            "config": b.config_of(main_res),
            "roofline": b.roofline(main_res) if not args.no_profile_events else None,
            "workloads": others or None,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if b.sharded:
        b.dist.barrier()
        b.dist.destroy_process_group()


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
