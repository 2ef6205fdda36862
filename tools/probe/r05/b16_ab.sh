#!/bin/bash
# Round 5: gemm_b16.hip (two workgroups per CU, 128 x 256 tiles) against the product's choice, 65536 rows, us per launch
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in 0 1; do
  for sh in "fwd fc" "dgrad pr" "fwd qkv" "fwd proj" "fwd prdrop" "fwd pr " "dgrad o" "dgrad fc16" "dgrad qkv16" "lm logit16"; do
    us=$(NEKO_GEMM_B16=$b timeout 300 python tools/gemm_bench.py --rows ${ROWS:-65536} --only "$sh" --iters 40 2>/dev/null | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
    echo "rep $rep b16 $b  $sh : $us us"
  done
done
done
