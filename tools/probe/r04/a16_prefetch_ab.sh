#!/bin/bash
# Round 4: L2 prefetch in the hand-placed GEMM loop (gen_gemm_a16.py --pf-dist): correctness first, then the streamed-operand shapes and the
# step, library with prefetch (default) against one generated with --pf-dist 0.
cd $GRAFT_REPO_ROOT
L=neko_amd/csrc
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -2
for lib in $L/libneko_hip_nopf.so $L/libneko_hip.so; do
  echo "## $lib"
  NEKO_HIP_LIB=$lib python tools/probe/a16_stream_probe.py 2>/dev/null | grep "lda= *[1-9]"
  NEKO_HIP_LIB=$lib python tools/gemm_bench.py --iters 30 --only "sq8k" 2>/dev/null | grep TFLOP
  NEKO_HIP_LIB=$lib python tools/gemm_bench.py --iters 30 --rows 65536 --only "wgrad" 2>/dev/null | grep TFLOP
  NEKO_HIP_LIB=$lib python tools/gemm_bench.py --iters 30 --only "lm d" 2>/dev/null | grep TFLOP
done
ROUNDS=2 bash tools/step_ab.sh "nopf=NEKO_HIP_LIB=$L/libneko_hip_nopf.so" "pf=NEKO_HIP_LIB=$L/libneko_hip.so"
