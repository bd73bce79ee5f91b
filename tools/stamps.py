#!/usr/bin/env python3
"""Phase shares of the traversal (diagnostic build csrc/librt_whitted_stamps.so,
hipcc ... -DRT_STAMPS). Read SHARES, not lengths (guide §7, In-kernel stamps)."""
import ctypes as C, importlib, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["RT_WHITTED_LIB"] = str(ROOT / "computer-graphics-engine_amd/csrc/librt_whitted_stamps.so")
import numpy as np, torch
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
sharding = importlib.import_module("computer-graphics-engine_amd.sharding")
rt = capi.Library(build=False)
rt.c.rt_debug_stamps.argtypes = [C.c_int, C.c_void_p]
for kv in sys.argv[2:]:
    rt.set_option(kv.split("=")[0], int(kv.split("=")[1]))
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
f, p = cfg.make_features(), cfg.make_params()
data = scenes.get_scene(cfg.scene)
bvh = rt.bvh(rt.scene(data), f)
cam = rt.camera(*scenes.default_camera_args(cfg.width, cfg.height))
R = sharding.StripRenderer(rt, bvh, cam, f, p, cfg.width, cfg.height)
st = torch.cuda.current_stream().cuda_stream
rt.c.rt_debug_stamps(1, None)
R.step(st); torch.cuda.synchronize()
rt.c.rt_debug_stamps(1, None)
R.step(st); torch.cuda.synchronize()
out = np.zeros(64, np.uint64)
rt.c.rt_debug_stamps(0, out.ctypes.data)
for name, b in (("exact engine, closest", 0), ("exact engine, shadow", 8), ("WIDE engine, closest / mixed", 32), ("WIDE engine, any hit", 40)):
    steal, node, leaf, total, n_node, n_leaf, waves, mx = [int(x) for x in out[b:b + 8]]
    if not waves: continue
    print(f"{name}: waves {waves} total cycles {total/1e6:.1f}M  avg/wave {total/waves:.0f} max wave {mx}"
          f" | steal {steal/total:.1%} node {node/total:.1%} leaf {leaf/total:.1%} other {(total-steal-node-leaf)/total:.1%}"
          f" | node iters {n_node} ({node/max(n_node,1):.0f} cyc each) leaf flushes {n_leaf} ({leaf/max(n_leaf,1):.0f} cyc each)")

pop, load, alu, push = [int(x) for x in out[16:20]]
n_node = int(out[4])
if n_node:
    print(f"closest node step split (cycles per step, lane 0 of each wave): pop+ray {pop/n_node:.0f}  node gather {load/n_node:.0f}"
          f"  slab tests {alu/n_node:.0f}  push {push/n_node:.0f}")

load, alu, push, n_w = [int(x) for x in out[48:52]]
if n_w:
    steal_w = int(out[32]) + int(out[40])
    print(f"WIDE node step split (cycles per step, lane 0 of each wave): pop + steal {steal_w/n_w:.0f}  ray slots (LDS) + node line (7 x b128) until the data is there {load/n_w:.0f}"
          f"  four conservative box tests {alu/n_w:.0f}  nearest child + pushes + leaf queue {push/n_w:.0f}")
