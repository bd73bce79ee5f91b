#!/bin/bash
# One rank's share of the driver's 20-frame region at N ranks, by primary tile shape (tools/share_region.py)
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for W in "$@"; do
  for S in "3 3 32" "4 2 32" "5 1 32" "6 0 32" "3 2 24" "4 1 24" "5 0 24" "2 3 24"; do
    set -- $S
    WORLD=$W RANK=$((W / 2)) TILE_W=$1 TILE_H=$2 FLUSH=$3 python3 tools/share_region.py 7,7,6 2>&1 | grep "^tile"
  done
done
