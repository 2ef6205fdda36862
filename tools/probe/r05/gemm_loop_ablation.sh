#!/bin/bash
# Round 5: what bounds the hand-placed k-loops?  Timing-only builds (WRONG results) of gemm_a16 / gemm_b16 with the MFMAs replaced by
# s_nop ("mfma": the L2 -> LDS DMA + fragment-read traffic alone), without the in-loop DMA ("dma": matrix pipe + LDS reads alone) and
# without the fragment reads ("reads") and without the per-k-tile block barriers ("barrier").  Run HERE (build) then on the GPU box (bench):
#   bash tools/probe/gemm_loop_ablation.sh build        # in the build container
#   gpurun -- bash tools/probe/gemm_loop_ablation.sh    # on the GPU
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  for abl in mfma dma reads barrier; do
    mkdir -p neko_amd/csrc/build_abl_$abl
    python tools/gen_gemm_a16.py --ablate $abl --out neko_amd/csrc/build_abl_$abl/gemm_a16_loop.inc > /dev/null
    python tools/gen_gemm_a16.py --geom b16 --ablate $abl --out neko_amd/csrc/build_abl_$abl/gemm_b16_loop.inc > /dev/null
    NEKO_BUILD_TAG=abl_$abl NEKO_EXTRA_HIPCC_FLAGS="-DNEKO_A16_LOOP_INC=\"build_abl_$abl/gemm_a16_loop.inc\" -DNEKO_B16_LOOP_INC=\"build_abl_$abl/gemm_b16_loop.inc\"" python -m neko_amd.build | tail -1
  done
  exit 0
fi
run() { # lib tag, extra env
  for sh in "fwd qkv" "dgrad o" "dgrad fc16"; do
    us=$(env $2 NEKO_HIP_LIB=$1 timeout 300 python tools/gemm_bench.py --rows 65536 --only "$sh" --iters 30 2>/dev/null | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
    echo "$3  $sh : $us us"
  done
}
for rep in 1 2; do
run neko_amd/csrc/libneko_hip.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 full          "
run neko_amd/csrc/libneko_hip_abl_mfma.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 no MFMA       "
run neko_amd/csrc/libneko_hip_abl_dma.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 no in-loop DMA"
run neko_amd/csrc/libneko_hip_abl_reads.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 no LDS reads  "
run neko_amd/csrc/libneko_hip_abl_barrier.so "NEKO_GEMM_B16=0 NEKO_GEMM_A16=1" "a16 no k-tile barriers"
run neko_amd/csrc/libneko_hip.so "NEKO_GEMM_B16=1" "b16 full          "
run neko_amd/csrc/libneko_hip_abl_mfma.so "NEKO_GEMM_B16=1" "b16 no MFMA       "
run neko_amd/csrc/libneko_hip_abl_dma.so "NEKO_GEMM_B16=1" "b16 no in-loop DMA"
run neko_amd/csrc/libneko_hip_abl_reads.so "NEKO_GEMM_B16=1" "b16 no LDS reads  "
run neko_amd/csrc/libneko_hip_abl_barrier.so "NEKO_GEMM_B16=1" "b16 no k-tile barriers"
done
