# quick numbers: driver arguments + 200 steps, no cpu baseline; optional extra args
mkdir -p gpurun_out/r2
tag=${1:-q}; shift
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r2/b_${tag}_20.json 2> gpurun_out/r2/b_${tag}_20.err
python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/r2/b_${tag}_200.json 2> gpurun_out/r2/b_${tag}_200.err
for f in b_${tag}_20 b_${tag}_200; do python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r2/{f}.json"))
    r = d["roofline"]; c = d["config"]
    print(f, "Mrays/s", d["value"], "ms", d["ms_per_step"], "launch_us", r["avg_launch_us"], "frac", r["frac"], "serial", d["frame_ms_serial"],
          "| per bvh ray: box", round(c["box_tests"]/c["rays_bvh_traversed"],1), "grp", round(c["group_tests"]/c["rays_bvh_traversed"],1), "tri", round(c["tri_tests"]/c["rays_bvh_traversed"],1))
except Exception as e:
    print(f, "FAILED", e); print(open(f"gpurun_out/r2/{f}.err").read()[-1500:])
PY
done
