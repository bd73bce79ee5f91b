#!/usr/bin/env python3
"""One GPU plays every rank of an N-rank job in turn: the time of each rank's share of the bench frames (render only:
8 frames per launch, three launches in flight, 8x8 tiles, orbiting camera) for several strip heights — how much a
thicker strip buys in locality and what it costs in balance (the job runs at the pace of its slowest rank).
  WORLD=8 STRIPS=8,16,32,64 python tools/share_partition.py"""
import importlib
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np  # noqa: E402
import torch  # noqa: E402

capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
sharding = importlib.import_module("computer-graphics-engine_amd.sharding")

WORLD = int(os.environ.get("WORLD", "8"))
STRIPS = [int(x) for x in os.environ.get("STRIPS", "8,16,32,64").split(",")]
FPL, FLIGHT, POSES = int(os.environ.get("FPL", "8")), int(os.environ.get("FLIGHT", "3")), 96
rt = capi.Library()
cfg = scenes.CONFIGS[os.environ.get("TUNE_WORKLOAD", "C3")]
f, p = cfg.make_features(), cfg.make_params()
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
W, H = cfg.width, cfg.height
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / POSES, rot0[2])) for k in range(POSES)]
rt.set_option("tile_wlog", 3), rt.set_option("tile_hlog", 3), rt.set_option("leaf_flush", 32)
for kv in sys.argv[1:]:
    rt.set_option(kv.split("=")[0], int(kv.split("=")[1]))
streams = sharding.new_streams(FLIGHT)
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
for sr in STRIPS:
    times = []
    for rank in range(WORLD):
        SKEW = int(os.environ.get("SKEW", "0"))   # strip s of round k = s // WORLD goes to rank (s + SKEW * k) % WORLD
        n_strips = -(-H // sr)
        rows = np.asarray([y for s_ in range(n_strips) if (s_ + SKEW * (s_ // WORLD)) % WORLD == rank
                           for y in range(s_ * sr, min((s_ + 1) * sr, H))], np.int32)
        pad = (-len(rows)) % 8
        rows_l = np.concatenate([rows, np.full(pad, rows[-1] if len(rows) else 0, np.int32)]) if pad else rows
        bufs = [torch.zeros((FPL, len(rows_l), W, 3), dtype=torch.float32, device="cuda") for _ in range(FLIGHT)]

        def run(reps):
            n = 0
            for rep in range(reps):
                for k in range(0, POSES, FPL):
                    j = (k // FPL) % FLIGHT
                    cs = cams[k:k + FPL]
                    rt.render_device_frames(bvh, cs, f, p, W, H, bufs[j].data_ptr(), rows=rows_l, packed=True,
                                            frame_stride_floats=bufs[j][0].numel(), stream=streams[j].cuda_stream)
                    n += len(cs)
            return n
        run(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run(int(os.environ.get("REPS", "10")))
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / n * 1e3)
    print(f"skew {os.environ.get('SKEW', '0')} {sys.argv[1:]} fpl {FPL} flight {FLIGHT} world {WORLD} strips of {sr:3d} rows: share ms/frame per rank " + " ".join(f"{t:.4f}" for t in times)
          + f" | mean {np.mean(times):.4f} max {max(times):.4f} (x{WORLD}: {max(times) * WORLD:.3f})", flush=True)
