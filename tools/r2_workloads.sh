# bench lines of the other BASELINE configs (C2, C4, C5), 60 steps each
mkdir -p gpurun_out/r2
tag=${1:-w}
for wl in C2 C4 C5; do
  timeout -k 10 400 python bench.py --workload $wl --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r2/b_${tag}_$wl.json 2> gpurun_out/r2/b_${tag}_$wl.err
  python - $wl $tag <<'PY'
import json, sys
wl, tag = sys.argv[1:3]
try:
    d = json.load(open(f"gpurun_out/r2/b_{tag}_{wl}.json"))
    r = d["roofline"]; c = d["config"]
    print(wl, "Mrays/s", d["value"], "ms", d["ms_per_step"], "serial", d["frame_ms_serial"]["mean"], "rays/frame", c["rays_per_frame"], "exact-engine rays", c["rays_exact_engine"],
          "| per bvh ray: box", round(c["box_tests"]/max(1,c["rays_bvh_traversed"]),1), "tri", round(c["tri_tests"]/max(1,c["rays_bvh_traversed"]),1))
except Exception as e:
    print(wl, "FAILED", e); print(open(f"gpurun_out/r2/b_{tag}_{wl}.err").read()[-1200:])
PY
done
