#!/bin/bash
# Ablation builds of attention_stream.hip (wrong results, timing only): one library per NEKO_AS_DIAG mask, selected with NEKO_HIP_LIB.
#   tools/attn_stream_ab.sh build      (here: cross-compiles)        tools/attn_stream_ab.sh run [bench args]   (on the GPU box)
cd "$(dirname "$0")/.."
C=neko_amd/csrc
if [ "$1" = build ]; then
  python -m neko_amd.build > /dev/null
  for m in ${MASKS:-1 2 4 8 3 6 7}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-result -DNEKO_AS_DIAG=$m -I$C -Iinclude -c $C/attention_stream.hip -o /tmp/as_diag_$m.o || exit 1
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls $C/build/*.o | grep -v attention_stream.o) /tmp/as_diag_$m.o -o $C/libneko_hip_asdiag$m.so || exit 1
  done
  exit 0
fi
shift
echo "full kernel:"; python tools/attn_bench.py "$@"
for f in $C/libneko_hip_asdiag*.so; do echo "$(basename $f):"; NEKO_HIP_LIB=$f python tools/attn_bench.py "$@"; done
