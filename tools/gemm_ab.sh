#!/bin/bash
# A/B of two library builds on the same box: alternating runs of tools/gemm_bench.py on a few shapes.
#   tools/gemm_ab.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
cd $GRAFT_REPO_ROOT
for r in $(seq $R); do
  IFS='|' read -ra SH <<< "${SHAPES:-fwd fc|dgrad pr|wgrad fc|lm logit16|lm dW|sq8k      NT}"
  for shape in "${SH[@]}"; do
    a=$(NEKO_HIP_LIB=$A python tools/gemm_bench.py --only "$shape" --iters 40 | grep TFLOP | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
    b=$(NEKO_HIP_LIB=$B python tools/gemm_bench.py --only "$shape" --iters 40 | grep TFLOP | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
    echo "round $r  $shape  A $a us  B $b us"
  done
done
