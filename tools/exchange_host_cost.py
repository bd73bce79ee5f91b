#!/usr/bin/env python3
"""Host time to ENQUEUE one frame of the strips + exchange path (render, events, all-gather,
de-interleave) on a one-rank nccl group: the floor of ms/frame per rank whatever the GPUs do."""
import importlib, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("PG_HIGH", "1") == "1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
capi = importlib.import_module("computer-graphics-engine_amd.capi")
scenes = importlib.import_module("computer-graphics-engine_amd.scenes")
sharding = importlib.import_module("computer-graphics-engine_amd.sharding")
cfg = scenes.CONFIGS["C3"]
f, p = cfg.make_features(), cfg.make_params()
rt = capi.Library(build=False)
bvh = rt.bvh(rt.scene(scenes.get_scene(cfg.scene)), f)
cam = rt.camera(*scenes.default_camera_args(cfg.width, cfg.height))
st = torch.cuda.current_stream().cuda_stream
# EMULATE=<world>: render only the rows rank 0 of <world> would (the exchange still moves a whole frame,
# as the all-gather of <world> ranks does) -- one rank's GPU-side work in an N-rank job, minus xGMI
emulate = int(os.environ.get("EMULATE", "1"))
batch = int(os.environ.get("BATCH", "1"))
tw, th = [int(x) for x in os.environ.get("TILE", "1,3").split(",")]
rt.set_option("tile_wlog", tw); rt.set_option("tile_hlog", th)
skip = os.environ.get("SKIP", "")
if "gather" in skip:
    sharding.gather_strips = lambda g, l, root=None: None
if "copy" in skip:      # a plain device copy in place of the one-rank all-gather
    sharding.gather_strips = lambda g, l, root=None: g[0].copy_(l)
if "deint" in skip:
    rt.deinterleave_strips = lambda *a, **k: None
for flight in [int(x) for x in os.environ.get("FLIGHTS", "1,2,8,16").split(",")]:
    R = sharding.StripRenderer(rt, bvh, cam, f, p, cfg.width, cfg.height, 0, 1, frames_in_flight=flight, exchange_always=True, exchange_batch=batch)
    if emulate > 1:
        R.rows = sharding.strip_rows_of_rank(cfg.height, 8, 0, emulate)
    dup = int(os.environ.get("ROWDUP", "1"))   # the share's rows `dup` times in one launch: what a launch of dup frames would cost
    if dup > 1:
        import numpy as np
        R.rows = np.tile(R.rows, dup)
    fpl = int(os.environ.get("FPL", "1"))     # frames per launch (rt_render_device_frames)
    for _ in range(40 // fpl): R.step(st, frames=fpl)
    R.finish()
    n = 320
    t0 = time.perf_counter()
    for _ in range(n // fpl): R.step(st, frames=fpl)
    t1 = time.perf_counter(); R.finish(); t2 = time.perf_counter()
    print(f"fpl {fpl} skip [{skip}] batch {batch} emulate {emulate} tile {tw},{th} flight {flight}: host enqueue {1e6*(t1-t0)/n:.1f} us/frame, total {1e3*(t2-t0)/n:.4f} ms/launch = {1e3*(t2-t0)/n/dup:.4f} ms/frame at {dup} frames per launch", flush=True)
dist.destroy_process_group()
