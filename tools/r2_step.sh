#!/bin/bash
# one development step on the GPU box: the GPU test suite, an A/B of option settings, the two bench commands
# usage: tools/r2_step.sh TAG "setting1 setting2 ..."
TAG=${1:-step}
SETTINGS=${2:-base}
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/${TAG}_tests.txt 2>&1 || { tail -n 30 gpurun_out/r2/${TAG}_tests.txt; exit 1; }
tail -n 3 gpurun_out/r2/${TAG}_tests.txt
TUNE_ROUNDS=3 timeout -k 10 600 python tools/tune2.py $SETTINGS > gpurun_out/r2/${TAG}_tune.txt 2>&1 && tail -n 8 gpurun_out/r2/${TAG}_tune.txt &&
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/${TAG}_b20.json 2> gpurun_out/r2/${TAG}_b20.err &&
timeout -k 10 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r2/${TAG}_b200.json 2> gpurun_out/r2/${TAG}_b200.err &&
python - <<PY
import json
for n in ("b20", "b200"):
    d = json.loads(open("gpurun_out/r2/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], "serial", d["config"].get("frame_ms_serial"), "launch_us", d["roofline"].get("avg_launch_us"))
PY
