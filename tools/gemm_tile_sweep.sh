#!/bin/bash
# Tile-configuration sweep of the step's GEMM shapes at a given row count (B*T): default heuristic vs each forced tile
# (NEKO_GEMM_TILE: 0 = 128x128 3-stage, 2 = 256x128, 3 = 256x256).  Usage: tools/gemm_tile_sweep.sh "1920 7680 15808"
cd $GRAFT_REPO_ROOT
for rows in ${1:-7680}; do
  for t in -1 0 2 3; do
    echo "== rows $rows tile $t"
    if [ $t -lt 0 ]; then python3 tools/gemm_bench.py --rows $rows --iters 30
    else NEKO_GEMM_TILE=$t python3 tools/gemm_bench.py --rows $rows --iters 30; fi
  done
done
