#!/usr/bin/env python3
"""Summarise gpurun_out/prof_TAG/ (tools/r2_profile.sh) and copy the judged files into profiles/:
  profiles/TAG_bench_driver.json, TAG_bench_200.json        the bench lines
  profiles/TAG_rocprofv3_kernel_stats.csv                   rocprofv3 --kernel-trace --stats of the driver's command
  profiles/TAG_pmc.json                                     per-launch PMC counters of the dominant kernel
  profiles/pmc_WORKLOAD.json                                what bench.py reads: HBM traffic and VALU / SALU
                                                            instruction counts per launch of the bench's launch shape
HBM traffic as MI355X_MICROARCH.md prescribes: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KB); gfx950
reports half of the bytes of 16-B-per-lane reads: 2 x FETCH_SIZE + WRITE_SIZE (the gather pattern is uncalibrated).
Usage: tools/r2_pmc_summary.py TAG [WORKLOAD] [--print-only]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wl = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "C3"
print_only = "--print-only" in sys.argv
R = Path(__file__).resolve().parent.parent
src = R / "gpurun_out" / f"prof_{tag}"
dst = R / "profiles"


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def trace_kernel(name):
    return "k_trace_level<false" in name or "k_trace_general<false" in name


pm = {}
waves_per_launch = None
for pd in sorted((src / "pmc").iterdir()):
    try:
        f = newest(str(pd / "*" / "*_counter_collection.csv"))
    except ValueError:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    grid = {}
    for r in csv.DictReader(open(f)):
        if trace_kernel(r["Kernel_Name"]):
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            grid[int(r["Dispatch_Id"])] = int(r.get("Grid_Size", 0) or 0)
    for c, d in per.items():
        ids = sorted(d)
        # launches of the bench's main shape = the largest grid; the timed ones are the last of them
        big = max(grid[i] for i in ids)
        vals = [d[i] for i in ids if grid[i] == big]
        pm[c] = {"per_launch": sum(vals[-2:]) / len(vals[-2:]), "launches_seen": len(vals), "grid_threads": big}

out = {"workload": wl, "counters": pm}
bench = None
for name in ("bench_200.json", "bench_driver.json"):
    try:
        bench = json.loads(open(src / name).read().strip().splitlines()[-1])
        break
    except Exception:
        pass
if bench:
    r = bench["roofline"]
    out["frames_per_launch"] = r["frames_per_launch"]
    out["tile_log"] = [3, 3] if r["frames_per_launch"] > 1 else [1, 3]
g = lambda k: pm.get(k, {}).get("per_launch")  # noqa: E731
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    out["hbm_bytes_per_launch"] = int((2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0)
    out["hbm_correction"] = "2 x FETCH_SIZE (gfx950: half of 16-B-per-lane reads is reported; gather pattern uncalibrated) + WRITE_SIZE, KB -> bytes"
if g("SQ_INSTS_VALU"):
    out["valu_insts_per_launch"] = int(g("SQ_INSTS_VALU"))
    out["salu_insts_per_launch"] = int(g("SQ_INSTS_SALU") or 0)
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    # thread-cycles / (64 x instruction-cycles): the share of lanes active per VALU instruction
    out["valu_active_lane_frac"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
out["command"] = f"bench.py --steps 16 --warmup 8 --workload {wl} --frames-in-flight 1 (one launch in flight: a launch's counters are its own)"
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
for k, v in sorted(pm.items()):
    print(f"{k:32s} {v['per_launch']:16.0f}  ({v['launches_seen']} launches, grid {v['grid_threads']})")
try:
    ks = newest(str(src / "stats" / "*" / "*_kernel_stats.csv"))
    print(open(ks).read()[:1500])
except ValueError:
    ks = None
if not print_only:
    dst.mkdir(exist_ok=True)
    for name in ("bench_driver.json", "bench_200.json"):
        if (src / name).exists():
            shutil.copy(src / name, dst / f"{tag}_{name}")
    if ks:
        shutil.copy(ks, dst / f"{tag}_rocprofv3_kernel_stats.csv")
    json.dump(out, open(dst / f"{tag}_pmc.json", "w"), indent=1)
    json.dump({k: v for k, v in out.items() if k != "counters"} | {"source": f"profiles/{tag}_pmc.json"},
              open(dst / f"pmc_{wl}.json", "w"), indent=1)
