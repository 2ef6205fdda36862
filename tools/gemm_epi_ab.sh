#!/bin/bash
# Ablation builds of gemm_glds.hip's epilogue (wrong results, timing only):  tools/gemm_epi_ab.sh build   (here)   /   run   (GPU box)
cd "$(dirname "$0")/.."
C=neko_amd/csrc
if [ "$1" = build ]; then
  python -m neko_amd.build > /dev/null
  for m in ${MASKS:-1 2 3}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-result -DNEKO_EPI_ABL=$m -I$C -Iinclude -c $C/gemm_glds.hip -o /tmp/gg_abl_$m.o || exit 1
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls $C/build/*.o | grep -v gemm_glds.o) /tmp/gg_abl_$m.o -o $C/libneko_hip_epiabl$m.so || exit 1
  done
  exit 0
fi
echo "full kernel:"; python tools/gemm_bench.py --rows 65536 --iters 30 2>&1 | grep -E "fwd qkv|fwd fc|dgrad pr |dgrad o"
for f in $C/libneko_hip_epiabl*.so; do echo "$(basename $f):"; NEKO_HIP_LIB=$f python tools/gemm_bench.py --rows 65536 --iters 30 2>&1 | grep -E "fwd qkv|fwd fc|dgrad pr |dgrad o"; done
