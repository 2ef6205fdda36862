for rep in 1 2; do
for v in base rs56 rs24 df20 nosnake kcb32; do
  L=neko_amd/csrc/libneko_hip.so; [ $v != base ] && L=neko_amd/csrc/libneko_hip_$v.so
  NEKO_HIP_LIB=$L NEKO_GEMM_A16=1 python tools/gemm_bench.py --only sq8k --iters 30 2>&1 | grep sq8k | sed "s/^/$v /" >> gpurun_out/r04_s22.log
  NEKO_HIP_LIB=$L python tools/gemm_bench.py --rows 65536 --only "dgrad fc16" --iters 30 2>&1 | grep dgrad | sed "s/^/$v /" >> gpurun_out/r04_s22.log
  NEKO_HIP_LIB=$L python tools/gemm_bench.py --rows 65536 --only "fwd pr" --iters 30 2>&1 | grep "fwd pr" | sed "s/^/$v /" >> gpurun_out/r04_s22.log
  NEKO_HIP_LIB=$L python tools/gemm_bench.py --rows 65536 --only "wgrad fc" --iters 30 2>&1 | grep wgrad | sed "s/^/$v /" >> gpurun_out/r04_s22.log
done; done
