#!/bin/bash
# round 6: gemm_p16 with the early block (k-tile 0 of a workgroup's next tile requested in front of the epilogue) and the tile walk
# (NEKO_GEMM_P16_WALK=1: one resident workgroup per CU) -- parity, per-shape timing, step A/B against the library of the commit before
# (neko_amd/csrc/libneko_hip_r6base.so, NEKO_BUILD_TAG=r6base); same box, alternating
mkdir -p gpurun_out
O=gpurun_out/r06_p16_walk.txt
: > $O
BASE=$PWD/neko_amd/csrc/libneko_hip_r6base.so
for w in 0 1; do
  NEKO_GEMM_P16_WALK=$w timeout 900 python -m pytest tests/test_gemm_p16_gpu.py -x -q --timeout 300 > gpurun_out/r06_p16_walk${w}_tests.txt 2>&1
  echo "pytest (walk $w) rc $?" | tee -a $O
  tail -3 gpurun_out/r06_p16_walk${w}_tests.txt | tee -a $O
done
grep -q "rc [1-9]" $O && exit 1
for rep in 1 2; do
  for v in base walk0 walk1; do
    case $v in base) E="NEKO_HIP_LIB=$BASE";; walk0) E="NEKO_GEMM_P16_WALK=0";; walk1) E="NEKO_GEMM_P16_WALK=1";; esac
    echo "=== $v rep $rep" >> $O
    env $E timeout 300 python tools/gemm_bench.py --rows 65536 2>&1 | grep -v amdgpu.ids >> $O
    env $E timeout 300 python tools/gemm_bench.py --only "lm " 2>&1 | grep -v amdgpu.ids >> $O
  done
done
ROUNDS=3 bash tools/step_ab.sh "base=NEKO_HIP_LIB=$BASE" "walk0=NEKO_GEMM_P16_WALK=0" "walk1=NEKO_GEMM_P16_WALK=1" 2>&1 | tee -a $O
