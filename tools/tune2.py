#!/usr/bin/env python3
"""Sweep runtime tuning options of the library on the bench workload with the bench's moving camera (one process,
interleaved rounds). Each argument is one setting: "a=1,b=2" (rt_set_option pairs; "base" = nothing).
Reports, per setting: the serial frame (one frame per launch, host-synchronised, 4x4 tiles) and the throughput
setting (8 frames per launch, two launches in flight, 8x8 tiles), ms per frame.
  TUNE_WORKLOAD=C3 TUNE_ROUNDS=3 TUNE_POSES=32 python tools/tune2.py base xcd_order=1 prefetch=1"""
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

settings = sys.argv[1:] or ["base"]
workload = os.environ.get("TUNE_WORKLOAD", "C3")
ROUNDS = int(os.environ.get("TUNE_ROUNDS", "3"))
POSES = int(os.environ.get("TUNE_POSES", "32"))
rt = capi.Library()
cfg = scenes.CONFIGS[workload]
f, p = cfg.make_features(), cfg.make_params()
data = scenes.get_scene(cfg.scene)
bvh = rt.bvh(rt.scene(data), f)
W, H = cfg.width, cfg.height
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / POSES, rot0[2])) for k in range(POSES)]
stream = torch.cuda.current_stream().cuda_stream
lone = sharding.StripRenderer(rt, bvh, cams[0], f, p, W, H, frames_in_flight=1)
thru = sharding.StripRenderer(rt, bvh, cams[0], f, p, W, H, frames_in_flight=2)
defaults = {}


def apply(s, tiles):
    for k, v in defaults.items():
        rt.set_option(k, v)
    rt.set_option("tile_wlog", tiles[0])
    rt.set_option("tile_hlog", tiles[1])
    rt.set_option("leaf_flush", 16 if tiles == (2, 2) else 32)
    if s != "base":
        for kv in s.split(","):
            k, x = kv.split("=")
            if k not in ("tile_wlog", "tile_hlog", "leaf_flush"):  # those are set per mode above
                defaults.setdefault(k, {"xcd_order": 1, "xcd_block": 4, "wide": 1, "fused": 1, "merge_next": 1, "skip_unlit": 1, "frame_major": 1, "tile_cull": 1, "launch_rect": 1, "tile_order": 1, "order_one_launch": 1}.get(k, int(os.environ.get("TUNE_DEFAULT_" + k.upper(), "0"))))
            rt.set_option(k, int(x))


res = {s: {"serial": [], "thru": []} for s in settings}
ref_frame = None
for rnd in range(ROUNDS):
    for s in settings:
        apply(s, (2, 2))
        lone.prepare(stream, (1,))
        ts = []
        for k in range(0, POSES, max(1, POSES // 16)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lone.step(stream, cams=[cams[k]])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[s]["serial"].append(float(np.mean(ts)) * 1e3)
        if rnd == 0:  # results must not depend on the setting
            lone.step(stream, cams=[cams[3]])
            torch.cuda.synchronize()
            fr = lone.frame.cpu().numpy().view(np.uint32)
            if ref_frame is None:
                ref_frame = fr
            res[s]["same"] = bool(np.array_equal(fr, ref_frame))
        apply(s, (3, 3))
        thru.prepare(stream, (8,))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for rep in range(3):
            for k in range(0, POSES - 7, 8):
                thru.step(stream, frames=8, cams=cams[k:k + 8])
                n += 8
        thru.finish()
        res[s]["thru"].append((time.perf_counter() - t0) / n * 1e3)
for s in settings:
    r = res[s]
    print(f"{s:40s} serial ms {np.median(r['serial']):.4f} (min {min(r['serial']):.4f})   throughput ms/frame {np.median(r['thru']):.4f} "
          f"(min {min(r['thru']):.4f})   frame bits == first: {r.get('same')}")
