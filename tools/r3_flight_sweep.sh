#!/bin/bash
# frames per launch x launches in flight, per workload (bench.py, 40 timed frames, no serial / cpu parts)
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for WL in "$@"; do
  for FPL in 4 8 16; do for FL in 2 3 4; do
    python3 bench.py --steps 48 --warmup 8 --workload $WL --no-cpu-baseline --no-serial --workload-steps 0 --no-profile-events --frames-per-launch $FPL --frames-in-flight $FL 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL fpl $FPL flight $FL:', d['ms_per_step'], 'ms/frame', d['value'])"
  done; done
done
