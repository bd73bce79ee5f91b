#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_passes.sh) per kernel launch of a frame."""
import csv, glob, sys, collections, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
LPF = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # k_trace_level launches per frame (1 = fused recursion)
out = {}
for f in sorted(glob.glob(f"{root}/*/*/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        if "k_trace_level<false>" not in k:
            continue
        per[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    for cname, d in per.items():
        ids = sorted(d)
        vals = [sum(d[i]) for i in ids]
        out[cname] = vals[-LPF:]  # the last frame's launches
for k, v in out.items():
    print(f"{k:28s}", " ".join(f"{x:14.0f}" for x in v), " | frame sum", f"{sum(v):.4g}")
json.dump(out, open(f"{root}/summary.json", "w"), indent=1)
