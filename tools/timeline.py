#!/usr/bin/env python3
"""Wave timeline of one k_trace_level launch (diagnostic build csrc/librt_timeline.so,
hipcc ... -DRT_TIMELINE): resident waves over time, per-CU spread, the tail."""
import ctypes as C, importlib, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["RT_WHITTED_LIB"] = str(ROOT / "computer-graphics-engine_amd/csrc/librt_timeline.so")
import numpy as np, torch
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
sharding = importlib.import_module("computer-graphics-engine_amd.sharding")
rt = capi.Library(build=False)
rt.c.rt_debug_stamps.argtypes = [C.c_int, C.c_void_p]
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    rt.set_option(k, int(v))
print("options:", sys.argv[2:])
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
f, p = cfg.make_features(), cfg.make_params()
data = scenes.get_scene(cfg.scene)
bvh = rt.bvh(rt.scene(data), f)
cam = rt.camera(*scenes.default_camera_args(cfg.width, cfg.height))
R = sharding.StripRenderer(rt, bvh, cam, f, p, cfg.width, cfg.height)
st = torch.cuda.current_stream().cuda_stream
rt.c.rt_debug_stamps(1, None)
for _ in range(3):
    R.step(st)
torch.cuda.synchronize()
rt.c.rt_debug_stamps(1, None)
R.step(st); torch.cuda.synchronize()
out = np.zeros(3 << 17, np.uint64)
rt.c.rt_debug_stamps(0, out.ctypes.data)
w = out.reshape(-1, 3)
w = w[w[:, 1] > 0]
t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
base = t0.min()
t0 -= base; t1 -= base
span = t1.max()
us = lambda x: x / 100.0  # 100 MHz
dur = t1 - t0
hw = w[:, 2]
hw = hw.astype(np.uint64)
cuid = (hw & np.uint64(0x3fff)).astype(np.int64)          # CU/SH/SE bits of HW_ID + XCC id
node_it = ((hw >> np.uint64(14)) & np.uint64(0xfffff)).astype(np.int64)
leaf_st = ((hw >> np.uint64(34)) & np.uint64(0xfffff)).astype(np.int64)
trav = ((hw >> np.uint64(54)) & np.uint64(0x3ff)).astype(np.int64)
print(f"waves {len(w)}  kernel span {us(span):.1f} us  sum of wave durations {us(dur.sum())/1e3:.2f} ms "
      f"=> mean resident waves {dur.sum()/span:.0f} of 4096 slots ({dur.sum()/span/4096:.1%})")
print(f"wave duration us: median {us(np.median(dur)):.1f}  p90 {us(np.percentile(dur,90)):.1f}  p99 {us(np.percentile(dur,99)):.1f}  max {us(dur.max()):.1f}")
# resident waves over time in 20 bins
bins = 20
edges = np.linspace(0, span, bins + 1)
occ = []
for i in range(bins):
    a, b = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
    occ.append(ov)
print("resident waves per 5% time bin:", " ".join(f"{o:.0f}" for o in occ))
heavy = dur > np.percentile(dur, 90)
print(f"heavy waves (top 10%): start-time median {us(np.median(t0[heavy])):.0f} us, p90 {us(np.percentile(t0[heavy],90)):.0f} us; end p99 {us(np.percentile(t1[heavy],99)):.0f}")
ncu = len(np.unique(cuid))
busy = np.bincount(np.unique(cuid, return_inverse=True)[1], weights=dur)
print(f"distinct CU ids {ncu}; per-CU summed wave time us: min {us(busy.min()):.0f} median {us(np.median(busy)):.0f} max {us(busy.max()):.0f}")
last = np.bincount(np.unique(cuid, return_inverse=True)[1], weights=None, minlength=ncu)
cu_end = np.array([t1[cuid == c].max() for c in np.unique(cuid)])
print(f"per-CU last wave end us: min {us(cu_end.min()):.0f} median {us(np.median(cu_end)):.0f} max {us(cu_end.max()):.0f}")
top = np.argsort(-dur)[:12]
print("longest waves: dur_us start_us traversals node_iters leaf_steps us/step")
for i in top:
    steps = node_it[i] + leaf_st[i]
    print(f"  {us(dur[i]):8.1f} {us(t0[i]):8.1f} {trav[i]:4d} {node_it[i]:7d} {leaf_st[i]:7d} {us(dur[i])/max(steps,1):6.2f}")
tot_steps = node_it + leaf_st
m = tot_steps > 0
print(f"all waves with work: mean us/step {us(dur[m]).sum()/tot_steps[m].sum():.3f}; node iters total {node_it.sum()} leaf steps total {leaf_st.sum()}")
early = m & (t1 < span * 0.4)
late = m & (t0 > span * 0.6)
for nm, sel in (("finished in first 40%", early), ("started in last 40%", late)):
    if sel.any():
        print(f"  {nm}: waves {sel.sum()} us/step {us(dur[sel]).sum()/tot_steps[sel].sum():.3f}")
np.save(ROOT / "gpurun_out" / "timeline.npy", w)
