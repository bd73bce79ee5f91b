#!/usr/bin/env python3
"""The driver's short timed region (20 frames, synchronised before and after) as one rank of an N-rank job sees it:
rank RANK's share of 20 orbit frames issued as launches of the given sizes on three streams, render only.
  WORLD=8 RANK=5 python tools/share_region.py 7,7,6 20 10,10 4,4,4,4,4"""
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

WORLD, RANK, FLIGHT = int(os.environ.get("WORLD", "8")), int(os.environ.get("RANK", "5")), int(os.environ.get("FLIGHT", "3"))
plans = [[int(x) for x in a.split(",")] for a in sys.argv[1:]] or [[7, 7, 6]]
rt = capi.Library()
cfg = scenes.CONFIGS[os.environ.get("WL", "C3")]
f, p = cfg.make_features(), cfg.make_params()
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
W, H = cfg.width, cfg.height
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
N = 20
cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / N, rot0[2])) for k in range(N)]
TW, TH, FLUSH = int(os.environ.get("TILE_W", "3")), int(os.environ.get("TILE_H", "3")), int(os.environ.get("FLUSH", "32"))
rt.set_option("tile_hlog", 0), rt.set_option("tile_wlog", TW), rt.set_option("tile_hlog", TH), rt.set_option("leaf_flush", FLUSH)
rt.set_option("order_one_launch", int(os.environ.get("ONE", "2")))
streams = sharding.new_streams(FLIGHT)
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
rows = sharding.strip_rows_of_rank(H, 8, RANK, WORLD)
bufs = [torch.zeros((32, len(rows), W, 3), dtype=torch.float32, device="cuda") for _ in range(FLIGHT)]
for plan in plans:
    assert sum(plan) == N

    def region():
        k = 0
        for j, m in enumerate(plan):
            b = bufs[j % FLIGHT]
            rt.render_device_frames(bvh, cams[k:k + m], f, p, W, H, b.data_ptr(), rows=rows, packed=True,
                                    frame_stride_floats=b[0].numel(), stream=streams[j % FLIGHT].cuda_stream)
            k += m
    for _ in range(3):
        region()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        region()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"tile 2^{TW} x 2^{TH} flush {FLUSH} world {WORLD} rank {RANK} order_one_launch {os.environ.get('ONE', '2')} plan {plan}: region {np.median(ts) * 1e3:.3f} ms "
          f"= {np.median(ts) * 1e3 / N:.4f} ms per frame (min {min(ts) * 1e3 / N:.4f})", flush=True)
