#!/usr/bin/env python3
"""The drop-in call's cost: rt_render into a host buffer (renderRayTracing -> Screen), by number of row bands."""
import ctypes as C, importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
rt = capi.Library()
for wl in (sys.argv[1:] or ["C3"]):
    cfg = scenes.CONFIGS[wl]
    f, p = cfg.make_features(), cfg.make_params()
    bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
    W, H = cfg.width, cfg.height
    cam = rt.camera(*scenes.default_camera_args(W, H))
    host = np.zeros((H, W, 3), np.float32)
    ref = None
    for bands in (1, 2, 3, 4, 6, 8):
        rt.set_option("host_bands", bands)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            rt._check(rt.c.rt_render(bvh.handle, C.byref(cam), C.byref(f), C.byref(p), W, H, 0, H, host.ctypes.data, None))
            ts.append(time.perf_counter() - t0)
        same = True if ref is None else bool(np.array_equal(host.view(np.uint32), ref))
        ref = host.view(np.uint32).copy() if ref is None else ref
        print(f"{wl} host_bands {bands}: mean {np.mean(ts[2:]) * 1e3:.3f} ms  min {min(ts[2:]) * 1e3:.3f}  first {ts[0] * 1e3:.2f}  bits == one launch: {same}")
