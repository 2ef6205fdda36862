#!/bin/bash
# probe: the 8-wave 256 x 256 GEMM with its k-contiguous A operand in three 64-k slots of whole 128-B lines (NEKO_GEMM_KC64=1)
cd $GRAFT_REPO_ROOT
NEKO_GEMM_KC64=1 NEKO_GEMM_A16=0 NEKO_GEMM_B16=0 timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
rm -f /tmp/dig0 /tmp/dig1
for rep in 1 2; do
for v in 0 1; do
  echo "== NEKO_GEMM_KC64=$v"
  NEKO_GEMM_KC64=$v python3 tools/gemm_bench.py --rows 65536 --iters 30 --only "f" $( [ $rep = 1 ] && echo --digest /tmp/dig$v ) 2>&1 | grep -v amdgpu.ids | grep "fwd fc\|dgrad pr\|fwd qkv"
done
done
echo "== digests"; cmp /tmp/dig0 /tmp/dig1 && echo "bit-identical ($(wc -l < /tmp/dig0) outputs)"
ROUNDS=${ROUNDS:-3} BENCH_ARGS="--steps 30" bash tools/step_ab.sh "default=NEKO_NOP=1" "kc64=NEKO_GEMM_KC64=1"
