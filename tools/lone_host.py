#!/usr/bin/env python3
"""Where a lone, host-synchronised frame spends its time: host enqueue (python + C ABI + HIP launches), the trace
kernel (HIP events), the rest (other kernels, launch latency, synchronisation). C3, orbiting camera, 4x4 tiles."""
import importlib
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")

rt = capi.Library()
cfg = scenes.CONFIGS[os.environ.get("TUNE_WORKLOAD", "C3")]
f, p = cfg.make_features(), cfg.make_params()
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
W, H = cfg.width, cfg.height
fovy, aspect, look_at, dist_, rot0 = scenes.default_camera_args(W, H)
POSES = 16
cams = [rt.camera(fovy, aspect, look_at, dist_, (rot0[0], rot0[1] + 2 * math.pi * k / POSES, rot0[2])) for k in range(POSES)]
stream = torch.cuda.current_stream().cuda_stream
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for s in sys.argv[1:]:
    k, v = s.split("=")
    rt.set_option(k, int(v))
rt.profile_enable(bvh, True)
for rep in range(3):
    enq, tot, trace = [], [], []
    for c in cams:
        torch.cuda.synchronize()
        rt.profile_read_launches(bvh, stream, reset=True)
        t0 = time.perf_counter()
        rt.render_device(bvh, c, f, p, W, H, frame.data_ptr(), stream=stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0), tot.append(t2 - t0)
        trace.append(sum(rt.profile_read_launches(bvh, stream, reset=True)))
    print(f"rep {rep}: frame {np.mean(tot) * 1e3:.4f} ms = host enqueue {np.mean(enq) * 1e3:.4f} + rest; trace kernel {np.mean(trace):.4f} ms; "
          f"outside the trace kernel {np.mean(tot) * 1e3 - np.mean(trace):.4f} ms")
