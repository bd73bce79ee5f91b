#!/usr/bin/env python3
"""The bench workload (C3: same shape, camera orbit, features) at different tessellations of the knot: how much of a
frame's time is the scene's memory footprint (and tree depth) rather than the ray count. Throughput setting."""
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

rt = capi.Library()
cfg = scenes.CONFIGS["C3"]
f, p = cfg.make_features(), cfg.make_params()
W, H = cfg.width, cfg.height
POSES = 32
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / POSES, rot0[2])) for k in range(POSES)]
stream = torch.cuda.current_stream().cuda_stream
rt.set_option("tile_wlog", 3), rt.set_option("tile_hlog", 3), rt.set_option("leaf_flush", 32)
for n_u, n_v in ((100, 4), (200, 8), (400, 16), (800, 32), (1600, 64), (3200, 128), (6400, 256)):
    bvh = rt.bvh(rt.scene(scenes.get_scene("knot", n_u, n_v)), f)
    R = sharding.StripRenderer(rt, bvh, cams[0], f, p, W, H, frames_in_flight=3)
    c = R.count_rays(stream, cam=cams[0])
    R.prepare(stream, (8,))
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for k in range(0, POSES, 8):
            R.step(stream, frames=8, cams=cams[k:k + 8])
            n += 8
        R.finish()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    bvh_rays = c["rays"] - c["rays_culled"]
    print(f"knot {n_u}x{n_v}: {2 * n_u * n_v:8d} triangles  {best:.4f} ms/frame  rays {c['rays']} (bvh {bvh_rays})  "
          f"{bvh_rays / best / 1e3:.0f} M bvh rays/s  box tests per bvh ray {c['box_tests'] / bvh_rays:.1f}  tri {c['tri_tests'] / bvh_rays:.1f}", flush=True)
    del R, bvh
