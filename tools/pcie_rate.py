#!/usr/bin/env python3
"""The boundary's host-pointer call: rt_render (frame copied back to the caller's buffer) on the bench
workload — the PCIe-inclusive rate DESIGN.md quotes next to the HBM-resident bench value."""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
f, p = cfg.make_features(), cfg.make_params()
rt = capi.Library(build=False)
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
cam = rt.camera(*scenes.default_camera_args(cfg.width, cfg.height))
W, H = cfg.width, cfg.height
cnt = None
for _ in range(3):
    rgb, cnt = rt.render(bvh, cam, f, p, W, H, count=True)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); rt.render(bvh, cam, f, p, W, H); ts.append(time.perf_counter() - t0)
ms = 1e3 * float(np.median(ts))
print(f"{cfg.name if hasattr(cfg,'name') else sys.argv[1:]}: rt_render to a pageable host buffer: {ms:.3f} ms per frame "
      f"({cnt.rays / ms / 1e3:.0f} Mrays/s, {W*H*12/ms/1e6:.1f} GB/s of frame bytes)")
cams = [cam] * 8
buf = np.zeros((8, H, W, 3), np.float32)   # the caller's buffer, reused (a fresh 200 MB array per call is page faults)
for _ in range(2):
    rt.render_frames(bvh, cams, f, p, W, H, out=buf)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); rt.render_frames(bvh, cams, f, p, W, H, out=buf); ts.append(time.perf_counter() - t0)
ms8 = 1e3 * float(np.median(ts)) / 8
print(f"rt_render_frames, 8 cameras per call: {ms8:.3f} ms per frame ({cnt.rays / ms8 / 1e3:.0f} Mrays/s, {W*H*12/ms8/1e6:.1f} GB/s of frame bytes)")
