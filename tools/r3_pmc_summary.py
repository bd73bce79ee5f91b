#!/usr/bin/env python3
"""Summarise gpurun_out/prof_TAG/ (tools/r3_profile.sh) and copy the judged files into profiles/:
  profiles/TAG_bench_driver.json, TAG_bench_200.json        the bench lines
  profiles/TAG_rocprofv3_kernel_stats.csv                   rocprofv3 --kernel-trace --stats of the driver's command
  profiles/TAG_kernel_trace_by_shape.csv                    the same run's trace kernel, launch by launch: grid, start, duration
  profiles/TAG_pmc.json                                     per-launch PMC counters of the dominant kernel
  profiles/pmc_WORKLOAD.json                                what bench.py reads: HBM traffic, L2 requests and instruction
                                                            counts per launch of the bench's launch shape (ONE in flight)
HBM traffic as MI355X_MICROARCH.md prescribes: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KB); gfx950
reports half of the bytes of 16-B-per-lane reads: 2 x FETCH_SIZE + WRITE_SIZE (the gather pattern is uncalibrated).
Usage: tools/r3_pmc_summary.py TAG [WORKLOAD] [--print-only]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
wl = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "C3"
print_only = "--print-only" in sys.argv
R = Path(__file__).resolve().parent.parent
src = R / "gpurun_out" / f"prof_{tag}"
dst = R / "profiles"


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def trace_kernel(name):
    return "k_trace_level<false" in name or "k_trace_general<false" in name or "k_trace_level<0" in name \
        or "k_trace_general<0" in name


pm = {}
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
for name in ("bench_driver.json", "bench_200.json"):
    try:
        bench = json.loads(open(src / name).read().strip().splitlines()[-1])
        break
    except Exception:
        pass
out["frames_per_launch"] = 8
out["tile_log"] = [3, 3]
g = lambda k: pm.get(k, {}).get("per_launch")  # noqa: E731
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    out["hbm_bytes_per_launch"] = int((2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0)
    out["hbm_correction"] = "2 x FETCH_SIZE (gfx950: half of 16-B-per-lane reads is reported; gather pattern uncalibrated) + WRITE_SIZE, KB -> bytes"
if g("TCC_HIT_sum") is not None:
    out["l2_requests_per_launch"] = int(g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    out["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
if g("SQ_INSTS_VALU"):
    out["valu_insts_per_launch"] = int(g("SQ_INSTS_VALU"))
    out["salu_insts_per_launch"] = int(g("SQ_INSTS_SALU") or 0)
    out["all_insts_per_launch"] = int(sum((g(k) or 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                                                 "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")))
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    # thread-cycles / (64 x instruction-cycles): the share of lanes active per VALU instruction
    out["valu_active_lane_frac"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") and g("SQ_ACTIVE_INST_ANY"):
    wc = g("SQ_WAVE_CYCLES")
    out["wave_time"] = {"waiting_s_waitcnt": round(g("SQ_WAIT_ANY") / wc, 4), "issuing": round(g("SQ_ACTIVE_INST_ANY") / wc, 4),
                        "stalled_at_issue": round((g("SQ_WAIT_INST_ANY") or 0) / wc, 4),
                        "what": "SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES: where a resident wave's cycles go"}
if g("SQC_ICACHE_REQ"):
    out["icache"] = {"requests_per_launch": int(g("SQC_ICACHE_REQ")), "misses_per_launch": int(g("SQC_ICACHE_MISSES") or 0),
                     "miss_rate": round((g("SQC_ICACHE_MISSES") or 0) / g("SQC_ICACHE_REQ"), 5),
                     "duplicate_misses_per_launch": int(g("SQC_ICACHE_MISSES_DUPLICATE") or 0)}
if bench and bench.get("roofline") and bench["roofline"].get("alone"):
    out["alone_launch_us"] = bench["roofline"]["alone"]["launch_us"] * 8.0 / bench["roofline"]["alone"]["frames_per_launch"]
out["command"] = f"bench.py --steps 16 --warmup 8 --workload {wl} --frames-in-flight 1 (one launch in flight: a launch's counters are its own)"
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
for k, v in sorted(pm.items()):
    print(f"{k:32s} {v['per_launch']:16.0f}  ({v['launches_seen']} launches, grid {v['grid_threads']})")
ks = kt = None
try:
    ks = newest(str(src / "stats" / "*" / "*_kernel_stats.csv"))
    kt = newest(str(src / "stats" / "*" / "*_kernel_trace.csv"))
    print(open(ks).read()[:1500])
except ValueError:
    pass
shape_rows = []
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if trace_kernel(r["Kernel_Name"])]
    t0 = min(int(r["Start_Timestamp"]) for r in rows) if rows else 0
    for r in rows:
        gsz = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        shape_rows.append({"kernel": r["Kernel_Name"][:60], "grid_threads": gsz,
                           "frames": None, "stream_or_queue": r.get("Queue_Id", ""),
                           "start_us": round((int(r["Start_Timestamp"]) - t0) / 1e3, 1),
                           "duration_us": round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)})
    by = collections.defaultdict(list)
    for r in shape_rows:
        by[r["grid_threads"]].append(r["duration_us"])
    print("trace kernel by launch shape (grid threads: launches, mean us, min, max):")
    for gsz, v in sorted(by.items()):
        print(f"  {gsz:10d}: {len(v):3d}  {sum(v) / len(v):9.1f} {min(v):9.1f} {max(v):9.1f}")
if not print_only:
    dst.mkdir(exist_ok=True)
    for name in ("bench_driver.json", "bench_200.json"):
        if (src / name).exists():
            shutil.copy(src / name, dst / f"{tag}_{name}")
    if ks:
        shutil.copy(ks, dst / f"{tag}_rocprofv3_kernel_stats.csv")
    if shape_rows:
        with open(dst / f"{tag}_kernel_trace_by_shape.csv", "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(shape_rows[0].keys()))
            w.writeheader()
            w.writerows(shape_rows)
    json.dump(out, open(dst / f"{tag}_pmc.json", "w"), indent=1)
    json.dump({k: v for k, v in out.items() if k != "counters"} | {"source": f"profiles/{tag}_pmc.json"},
              open(dst / f"pmc_{wl}.json", "w"), indent=1)
