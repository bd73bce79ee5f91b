#!/bin/bash
# lone-frame sweep: tile shapes and level splitting (serial column of tools/tune2.py)
mkdir -p gpurun_out/r2
TUNE_ROUNDS=2 TUNE_POSES=32 timeout -k 10 900 python tools/tune2.py base \
  tile_wlog=3,tile_hlog=2 tile_wlog=2,tile_hlog=3 tile_wlog=3,tile_hlog=3 tile_wlog=2,tile_hlog=1 tile_wlog=1,tile_hlog=1 \
  fused=0 fused_from=1 fused_from=2 \
  tile_wlog=3,tile_hlog=3,fused=0 tile_wlog=3,tile_hlog=3,fused_from=1 tile_wlog=3,tile_hlog=3,fused_from=2 \
  tile_wlog=3,tile_hlog=2,fused_from=1 leaf_flush=8 leaf_flush=24 \
  > gpurun_out/r2/lone_sweep.txt 2>&1
tail -n 20 gpurun_out/r2/lone_sweep.txt
