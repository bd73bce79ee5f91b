#!/usr/bin/env python3
"""Lone frame (one launch, host-synchronised, 16 poses of the orbit) by primary tile shape and leaf-queue threshold."""
import importlib, math, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
sharding = importlib.import_module("computer-graphics-engine_amd.sharding")
rt = capi.Library()
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = scenes.CONFIGS[wl]
f, p = cfg.make_features(), cfg.make_params()
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
W, H = cfg.width, cfg.height
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / 16, rot0[2])) for k in range(16)]
st = torch.cuda.current_stream().cuda_stream
R = sharding.StripRenderer(rt, bvh, cams[0], f, p, W, H)
res = {}
per_pose = {}
settings = [(2, 2, 16), (3, 1, 16), (4, 0, 16), (4, 1, 24), (5, 0, 24), (6, 0, 32), (3, 2, 24), (3, 3, 32)]  # (tile_wlog, tile_hlog, leaf_flush[, xcd_block])
for rnd in range(3):
    for s in settings:
        rt.set_option("tile_hlog", 0); rt.set_option("tile_wlog", s[0]); rt.set_option("tile_hlog", s[1]); rt.set_option("leaf_flush", s[2]); rt.set_option("xcd_block", s[3] if len(s) > 3 else 4)
        R.step(st, cams=[cams[0]]); torch.cuda.synchronize()
        ts = []
        for c in cams:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            R.step(st, cams=[c]); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res.setdefault(s, []).append(np.mean(ts) * 1e3)
        per_pose.setdefault(s, []).append(ts)
for s in settings:
    pp = np.median(np.array(per_pose[s]), axis=0) * 1e3
    print(f"{wl} tile 2^{s[0]} x 2^{s[1]} leaf_flush {s[2]:2d} xcd_block {s[3] if len(s) > 3 else 4}: lone frame mean over the orbit {np.median(res[s]):.4f} ms (min {min(res[s]):.4f})  by pose: " + " ".join(f"{v:.2f}" for v in pp))
