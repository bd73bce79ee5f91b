#!/usr/bin/env python3
"""A/B several builds of the library in ONE process, interleaved rounds (guide §5.4 rule 24).
Usage: tools/ab.py [--workload C3] libA.so libB.so ...   (paths relative to csrc/)"""
import importlib, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
sharding = importlib.import_module("computer-graphics-engine_amd.sharding")
args = sys.argv[1:]
THRU = os.environ.get("AB_THRU", "0") == "1"
workload = "C3"
if args and args[0] == "--workload":
    workload = args[1]; args = args[2:]
cfg = scenes.CONFIGS[workload]
f, p = cfg.make_features(), cfg.make_params()
data = scenes.get_scene(cfg.scene)
stream = torch.cuda.current_stream().cuda_stream
runs = []
for lib in args:
    os.environ["RT_WHITTED_LIB"] = str(ROOT / "computer-graphics-engine_amd/csrc" / lib)
    rt = capi.Library(build=False)
    bvh = rt.bvh(rt.scene(data), f)
    cam = rt.camera(*scenes.default_camera_args(cfg.width, cfg.height))
    if THRU:
        rt.set_option("tile_wlog", 3), rt.set_option("tile_hlog", 3), rt.set_option("leaf_flush", 32)
    R_ = sharding.StripRenderer(rt, bvh, cam, f, p, cfg.width, cfg.height, frames_in_flight=3 if THRU else 1)
    if THRU:
        import math
        fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(cfg.width, cfg.height)
        R_.orbit = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / 32, rot0[2])) for k in range(32)]
        R_.prepare(stream, (8,))
    runs.append((lib, rt, R_))
res = {lib: [] for lib, _, _ in runs}
frames = {}
for rnd in range(5):
    for lib, rt, R in runs:
        if THRU:  # AB_THRU=1: the bench's throughput setting (8 frames per launch, three launches in flight, orbit)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for rep in range(2):
                for k in range(0, 32, 8):
                    R.step(stream, frames=8, cams=R.orbit[k:k + 8])
            R.finish()
            res[lib].append((time.perf_counter() - t0) / 64 * 1e3)
            frames[lib] = R.frame.cpu().numpy().view(np.uint32)
            continue
        for _ in range(3):
            R.step(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            R.step(stream)
        torch.cuda.synchronize()
        res[lib].append((time.perf_counter() - t0) / 20 * 1e3)
        frames[lib] = R.frame.cpu().numpy().view(np.uint32)
base = frames[runs[0][0]]
for lib, _, _ in runs:
    same = np.array_equal(frames[lib], base)
    print(f"{lib:32s} ms/frame median {np.median(res[lib]):.3f} min {min(res[lib]):.3f}  frame bits == first: {same}")
