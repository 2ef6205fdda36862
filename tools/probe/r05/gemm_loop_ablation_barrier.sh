cd $GRAFT_REPO_ROOT
run() { for sh in "fwd qkv" "dgrad o" "dgrad fc16"; do us=$(env $2 NEKO_HIP_LIB=$1 timeout 300 python tools/gemm_bench.py --rows 65536 --only "$sh" --iters 30 2>/dev/null | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}'); echo "$3  $sh : $us us"; done; }
for rep in 1 2; do
run neko_amd/csrc/libneko_hip.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 full              "
run neko_amd/csrc/libneko_hip_abl_barrier.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 no k-tile barriers"
run neko_amd/csrc/libneko_hip.so "NEKO_GEMM_B16=1" "b16 full              "
run neko_amd/csrc/libneko_hip_abl_barrier.so "NEKO_GEMM_B16=1" "b16 no k-tile barriers"
done
