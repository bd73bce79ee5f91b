#!/usr/bin/env python3
"""Copy the judged summaries of gpurun_out/round_profile/ into profiles/ (tracked):
rocprofv3 kernel stats, per-level launch durations, PMC table, and hbm_traffic.json
(bench.py reads it for roofline.traffic). Usage: tools/collect_profiles.py r01"""
import csv, glob, json, os, shutil, sys, collections

def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the latest of each kind"""
    return max(glob.glob(pattern), key=os.path.getmtime)

from pathlib import Path
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = Path(__file__).resolve().parent.parent
src = R / "gpurun_out" / "round_profile"
dst = R / "profiles"
dst.mkdir(exist_ok=True)
shutil.copy(src / "bench_default.json", dst / f"{tag}_bench_default.json")
ks = newest(str(src / "stats" / "*" / "*_kernel_stats.csv"))
shutil.copy(ks, dst / f"{tag}_rocprofv3_kernel_stats.csv")
# per-level launch durations from the kernel trace (last 20 frames)
rows = list(csv.DictReader(open(newest(str(src / "stats" / "*" / "*_kernel_trace.csv")))))
tr = [r for r in rows if "k_trace_level<false" in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
bench = json.loads(open(src / "bench_default.json").read().strip().splitlines()[-1])
LPF = max(1, int(round(bench["roofline"]["launches_per_frame"])))  # k_trace_level launches per frame (1 = fused)
levels = [dur[i::LPF][-20:] for i in range(LPF)] if len(dur) % LPF == 0 else []
pm = {}
for f in sorted(newest(str(pd / "*" / "*_counter_collection.csv")) for pd in (src / "pmc").iterdir() if pd.is_dir()):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_trace_level<false" in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, d in per.items():
        ids = sorted(d)
        pm[c] = [d[i] for i in ids][-LPF:]        # the last frame's level launches
summary = {"bench": {k: bench[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline")},
           "trace_level_launch_us_by_level": [round(sum(l) / len(l), 1) for l in levels] if levels else None,
           "pmc_last_frame_by_level": pm}
json.dump(summary, open(dst / f"{tag}_profile_summary.json", "w"), indent=1)
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports 1/2 of
    # the bytes of 16 B/lane reads → doubled; WRITE_SIZE exact. Per launch = frame total / launches per frame.
    fetch, write = sum(pm["FETCH_SIZE"]), sum(pm["WRITE_SIZE"])
    per_launch = (2.0 * fetch + write) * 1024.0 / LPF
    json.dump({"workload": "C3", "hbm_bytes_per_launch": int(per_launch),
               "fetch_size_kb_frame": fetch, "write_size_kb_frame": write,
               "correction": "2 x FETCH_SIZE (gfx950, 16 B/lane loads; gather pattern uncalibrated) + WRITE_SIZE, x1024, / launches per frame",
               "source": f"profiles/{tag}_profile_summary.json"}, open(dst / "hbm_traffic.json", "w"), indent=1)
print(json.dumps(summary["bench"]["roofline"]), summary["trace_level_launch_us_by_level"])
print(open(dst / f"{tag}_rocprofv3_kernel_stats.csv").read()[:900])
