#!/bin/bash
# Tile-configuration sweep of the step's GEMM shapes at a given row count (B*T): default heuristic vs each forced tile
# (NEKO_GEMM_TILE: 0 = 128x128 3-stage, 2 = 256x128, 3 = 256x256, 4 = 256x256 with 4 waves).  Usage: tools/gemm_tile_sweep.sh "1920 7680 15808" [embed_dim]
cd $GRAFT_REPO_ROOT
DIM=${2:+--dim $2}
for rows in ${1:-7680}; do
  for t in ${TILES:--1 0 2 3 4}; do
    echo "== rows $rows tile $t"
    if [ $t -lt 0 ]; then python3 tools/gemm_bench.py --rows $rows --iters 30 $DIM
    else NEKO_GEMM_TILE=$t python3 tools/gemm_bench.py --rows $rows --iters 30 $DIM; fi
  done
done
