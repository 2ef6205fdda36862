#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "sorted or patch or row_map or layernorm" 2>&1 | tail -5
ROUNDS=2 BENCH_ARGS="--steps 30" bash tools/step_ab.sh "sorted=NEKO_SORTED_SCATTER=1" "atomics=NEKO_SORTED_SCATTER=0"
