#!/bin/bash
# round 6: epilogue inputs of the next 32-row pass requested ahead of this pass's stores (gemm_epi.h Park::PIPELINE, gemm_p16) -- parity,
# per-shape timing and step A/B against the library of the commit before (NEKO_BUILD_TAG=r6base); same box, alternating
mkdir -p gpurun_out
O=gpurun_out/r06_epi_pipeline.txt
: > $O
BASE=$PWD/neko_amd/csrc/libneko_hip_r6base.so
timeout 900 python -m pytest tests/test_gemm_p16_gpu.py tests/test_kernels_gpu.py -x -q --timeout 300 -k "gemm or p16 or epilogue" > gpurun_out/r06_epi_pipeline_tests.txt 2>&1
echo "pytest rc $?" | tee -a $O
tail -3 gpurun_out/r06_epi_pipeline_tests.txt | tee -a $O
grep -q "rc [1-9]" $O && exit 1
for rep in 1 2; do
  for v in base pipe; do
    case $v in base) E="NEKO_HIP_LIB=$BASE";; pipe) E="NEKO_X=0";; esac
    echo "=== $v rep $rep" >> $O
    env $E timeout 300 python tools/gemm_bench.py --rows 65536 2>&1 | grep -v amdgpu.ids >> $O
  done
done
ROUNDS=3 bash tools/step_ab.sh "base=NEKO_HIP_LIB=$BASE" "pipe=NEKO_X=0" 2>&1 | tee -a $O
