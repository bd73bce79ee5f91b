#!/usr/bin/env python3
"""The reference's CLI mode from the boundary's point of view: N host threads, one camera each, every thread calling
rt_render (≙ renderRayTracing -> Screen) on ONE BVH handle. ms per frame over all threads."""
import ctypes as C, importlib, math, sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
rt = capi.Library()
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = scenes.CONFIGS[wl]
f, p = cfg.make_features(), cfg.make_params()
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
W, H = cfg.width, cfg.height
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
K = 12
for n in (1, 2, 3, 4, 6, 8):
    cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * i / n, rot0[2])) for i in range(n)]
    bufs = [np.zeros((H, W, 3), np.float32) for _ in range(n)]

    def work(i):
        for _ in range(K):
            rt._check(rt.c.rt_render(bvh.handle, C.byref(cams[i]), C.byref(f), C.byref(p), W, H, 0, H, bufs[i].ctypes.data, None))

    for i in range(n):
        work_i = threading.Thread(target=lambda i=i: rt._check(rt.c.rt_render(bvh.handle, C.byref(cams[i]), C.byref(f), C.byref(p), W, H, 0, H, bufs[i].ctypes.data, None)))
        work_i.start(); work_i.join()   # warm: buffers of every slot
    ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    print(f"{wl}: {n} host threads x {K} frames each: {dt / (n * K) * 1e3:.3f} ms per frame ({n * K / dt:.0f} frames/s)")
