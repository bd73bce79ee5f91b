#!/usr/bin/env python3
"""Dump the bench scene + default camera and run tools/wide_sim.cpp on it (CPU statistics of the wide tree)."""
import importlib
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
capi = importlib.import_module("computer-graphics-engine_amd.capi")

W, H = 1920, 1080
stride = sys.argv[1] if len(sys.argv) > 1 else "4"
data = scenes.get_scene("knot")
m = data.meshes[0]
rt = capi.Library()
cam = rt.camera(*scenes.default_camera_args(W, H))
out = ROOT / "tools" / "bin" / "knot.bin"
out.parent.mkdir(exist_ok=True)
with open(out, "wb") as f:
    f.write(struct.pack("<II", len(m.positions), len(m.triangles)))
    f.write(struct.pack("<9f", *cam.origin, *cam.quat, cam.half_w, cam.half_h))
    f.write(struct.pack("<ii", W, H))
    f.write(np.ascontiguousarray(m.positions, np.float32).tobytes())
    f.write(np.ascontiguousarray(m.triangles, np.uint32).tobytes())
exe = ROOT / "tools" / "bin" / "wide_sim"
csrc = ROOT / "computer-graphics-engine_amd" / "csrc"
subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", str(exe),
                str(ROOT / "tools" / "wide_sim.cpp"), str(csrc / "host_scene.cpp"), str(csrc / "host_wide.cpp")], check=True)
subprocess.run([str(exe), str(out), stride], check=True)
