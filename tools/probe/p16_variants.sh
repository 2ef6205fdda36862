#!/bin/bash
# Round 6: schedule variants and timing-only ablations of gemm_p16's generated stream, every variant its own library (same objects, only
# gemm_p16.o differs; -DNEKO_P16_TRACE=1: s_memtime around the asm loop).
#   bash tools/probe/p16_variants.sh build                       # in the build container
#   gpurun -- bash tools/probe/p16_variants.sh run [tags...]     # on the GPU: per variant the K = 768 / long-K shapes + clocks per k-tile
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=neko_amd/csrc
declare -A V
declare -A D          # extra -D flags of a variant
D[noepi]="-DNEKO_GEMM_DIAG=4"
D[nostores]="-DNEKO_EPI_ABL=1"
D[noslab]="-DNEKO_EPI_ABL=3"
D[batchreads]="-DNEKO_EPI_BATCH_READS=1"
V[batchreads]=""
D[trace2_batchreads]="-DNEKO_EPI_BATCH_READS=1 -DNEKO_P16_TRACE=2"
V[trace2_batchreads]=""
D[stagger8]="-DNEKO_P16_STAGGER_10NS=800"
V[stagger8]=""
D[stagger12]="-DNEKO_P16_STAGGER_10NS=1200"
V[stagger12]=""
D[stagger16]="-DNEKO_P16_STAGGER_10NS=1600"
V[stagger16]=""
D[nt]="-DNEKO_EPI_STORE_POLICY=1"
V[nt]=""
D[trace2]="-DNEKO_P16_TRACE=2"
V[trace2]=""
D[trace2_nt]="-DNEKO_P16_TRACE=2 -DNEKO_EPI_STORE_POLICY=1"
V[trace2_nt]=""
D[st_nt]="-DNEKO_EPI_STORE_POLICY=1"
D[st_sc1]="-DNEKO_EPI_STORE_POLICY=2"
D[st_sc01]="-DNEKO_EPI_STORE_POLICY=3"
V[st_nt]=""
V[st_sc1]=""
V[st_sc01]=""
V[noepi]=""
V[nostores]=""
V[noslab]=""
V[default]=""
V[kcb32]="--p16-kcb 32"
V[nospread]="--p16-no-spread"
V[readsfirst]="--p16-reads-first 99"
V[rpp1]="--p16-reads-per-piece 1"
V[ns2]="--p16-nslot-a 2"
V[noprio]="--p16-no-setprio"
V[earlywait]="--p16-early-wait"
V[nob1]="--p16-no-b1"
V[nob1_noprio]="--p16-no-b1 --p16-no-setprio"
V[abl_mfma]="--ablate mfma"
V[abl_dma]="--ablate dma"
V[abl_reads]="--ablate reads"
if [ "$1" = build ]; then
  shift
  mkdir -p $C/build_p16v
  tags=${@:-${!V[@]}}
  for tag in $tags; do
    python tools/gen_gemm_a16.py --geom p16 ${V[$tag]} --out $C/build_p16v/$tag.inc > /dev/null || exit 1
    (cd $C && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-result -DNEKO_P16_TRACE=1 ${D[$tag]} \
       "-DNEKO_P16_LOOP_INC=\"build_p16v/$tag.inc\"" -c gemm_p16.hip -o build_p16v/$tag.o) || exit 1
    objs=$(ls $C/build/*.o | grep -v gemm_p16.o)
    hipcc -shared -fPIC --offload-arch=gfx950 $objs $C/build_p16v/$tag.o -o $C/libneko_hip_p16v_$tag.so || exit 1
    echo "built $tag"
  done
  exit 0
fi
shift
tags=${@:-default kcb32 nospread readsfirst rpp1 ns2 noprio abl_mfma abl_dma abl_reads}
NAMES="fwd qkv NN,fwd fc gp NN,fwd prdrop NN,dgrad pr4 NT,dgrad fc16 NT,dgrad qkv16 NT,dgrad o NT"
for rep in 1 2; do
  for tag in $tags; do
    echo "=== $tag rep $rep"
    NEKO_HIP_LIB=$C/libneko_hip_p16v_$tag.so NEKO_GEMM_P16=1 timeout 300 python tools/gemm_bench.py --rows 65536 --names "$NAMES" --p16-trace 2>&1 | grep -v amdgpu.ids
    NEKO_HIP_LIB=$C/libneko_hip_p16v_$tag.so NEKO_GEMM_P16=1 timeout 300 python tools/gemm_bench.py --names "lm logit16 NT" 2>&1 | grep TFLOP
  done
done
