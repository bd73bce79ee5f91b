#!/bin/bash
# The judged bench lines again (after profiles/pmc_*.json changed): driver's command and 200 steps, every workload.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for WL in C3 C2 C4 C5; do
  TAG=r03; [ $WL != C3 ] && TAG=r03_$WL
  O=$R/gpurun_out/prof_$TAG; mkdir -p $O
  timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --workload $WL > $O/bench_driver.json 2> $O/bench_driver.err || echo "driver bench $WL failed"
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --workload $WL --no-cpu-baseline --workload-steps 0 > $O/bench_200.json 2> $O/bench_200.err || echo "200-step bench $WL failed"
  echo "$WL done"
done
