#!/bin/bash
# Round 5 probe: per-CU output-phase lock (NEKO_GEMM_EPILOCK=1) on the compiler-scheduled loop of gemm_glds.hip.
# tile 3 = one 8-wave 256 x 256 workgroup per CU (the lock is never contended), tile 2 = two 4-wave 256 x 128 workgroups per CU.
# 65536 rows, us per launch, NEKO_GEMM_A16=0 (every launch on gemm_glds.hip)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "3 0" "2 0" "2 1" "0 0" "0 1"; do
  set -- $cfg
  for sh in "fwd fc" "dgrad pr" "fwd qkv" "fwd proj" "fwd prdrop" "dgrad o"; do
    us=$(NEKO_GEMM_A16=0 NEKO_GEMM_TILE=$1 NEKO_GEMM_EPILOCK=$2 timeout 300 python tools/gemm_bench.py --rows 65536 --only "$sh" --iters 40 2>/dev/null | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
    echo "rep $rep tile $1 lock $2  $sh : $us us"
  done
done
done
