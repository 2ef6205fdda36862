#!/bin/bash
# Round 4 probe: do two co-resident 4-wave workgroups (256 x 128 tiles, 72 KB ring each) hide each other's output phase when the second
# one on every CU starts half a tile late?  Library built from gemm_glds.hip + tools/probe/gemm_stagger_experiment.patch (second form:
# blocks [LO, HI) wait NEKO_GEMM_STAGGER_10NS x 10 ns at entry).  65536 rows, us per launch.
cd $GRAFT_REPO_ROOT
L=neko_amd/csrc/libneko_hip_stg.so
run() { # tile stagger
  for sh in "fwd fc" "dgrad pr" "fwd qkv" "fwd proj"; do
    us=$(NEKO_GEMM_A16=0 NEKO_HIP_LIB=$L NEKO_GEMM_TILE=$1 NEKO_GEMM_STAGGER_10NS=$2 python tools/gemm_bench.py --rows 65536 --only "$sh" --iters 40 2>/dev/null | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
    echo "tile $1 stagger ${2}0 ns  $sh : $us us"
  done
}
run 3 0
for st in 0 300 500 700 900 1200; do run 2 $st; done
