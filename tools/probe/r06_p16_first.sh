#!/bin/bash
# round 6: first run of gemm_p16.hip -- parity, then per-shape timing against the other loops (same box, alternating)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_p16_gpu.py -x -q --timeout 180 > gpurun_out/r06_p16_tests.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r06_p16_tests.txt
tail -5 gpurun_out/r06_p16_tests.txt
if grep -q "pytest rc 0" gpurun_out/r06_p16_tests.txt; then
  for rep in 1 2; do
    for m in 0 1; do
      echo "=== NEKO_GEMM_P16=$m rep $rep" >> gpurun_out/r06_p16_bench.txt
      NEKO_GEMM_P16=$m timeout 300 python tools/gemm_bench.py --rows 65536 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_p16_bench.txt
      NEKO_GEMM_P16=$m timeout 300 python tools/gemm_bench.py --only "lm " 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_p16_bench.txt
      NEKO_GEMM_P16=$m timeout 300 python tools/gemm_bench.py --only "sq8k" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_p16_bench.txt
    done
  done
fi
