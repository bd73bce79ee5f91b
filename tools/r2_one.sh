#!/bin/bash
mkdir -p gpurun_out/r2
for rep in 1 2 3; do for one in 1 2; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --set order_one_launch=$one > gpurun_out/r2/one.json 2> gpurun_out/r2/one.err || { tail -n 5 gpurun_out/r2/one.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r2/one.json').read().strip().splitlines()[-1])
print('one_launch $one steps 20:', d['value'], d['ms_per_step'], 'launch_us', d['roofline'].get('avg_launch_us'), d['roofline']['chip_wide'])"
done; done
WORLD=1 RANK=0 ONE=1 timeout -k 10 200 python tools/share_region.py 7,7,6 2>&1 | grep world
WORLD=1 RANK=0 ONE=2 timeout -k 10 200 python tools/share_region.py 7,7,6 2>&1 | grep world
