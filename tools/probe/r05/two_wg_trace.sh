#!/bin/bash
# Round 5 probe: per-block phase times (tools/gemm_trace.py, -DNEKO_GEMM_DIAG=9 build) of the K = 768 shapes with ONE 8-wave 256 x 256 workgroup
# per CU (tile 3) against TWO 4-wave 256 x 128 workgroups per CU (tile 2), with and without the per-CU output-phase lock
cd $GRAFT_REPO_ROOT
export NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_trace.so NEKO_GEMM_A16=0
for cfg in "3 0" "2 0" "2 1" "0 0"; do
  set -- $cfg
  echo "=== tile $1 lock $2"
  NEKO_GEMM_TILE=$1 NEKO_GEMM_EPILOCK=$2 timeout 300 python tools/gemm_trace.py 2>&1 | tail -30
done
