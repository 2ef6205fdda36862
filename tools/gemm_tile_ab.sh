#!/bin/bash
# Same-box comparison of tile configurations / library builds on the layer's GEMM shapes.
#   tools/gemm_tile_ab.sh <rows> "<lib>:<tile> <lib>:<tile> ..." [rounds] ["shape|shape|..."]
# lib = path of a libneko_hip*.so (NEKO_HIP_LIB), tile = NEKO_GEMM_TILE value or "d" for the library's own choice.
rows=$1; variants=$2; R=${3:-2}
cd $GRAFT_REPO_ROOT
IFS='|' read -ra SH <<< "${4:-fwd qkv|fwd proj|fwd fc|fwd pr|dgrad pr|dgrad fc16|dgrad o|dgrad qkv16|lm logit16}"
for r in $(seq $R); do
  for shape in "${SH[@]}"; do
    line="round $r  $shape :"
    for v in $variants; do
      lib=${v%%:*}; tile=${v##*:}
      extra=""; [[ "$shape" == lm* ]] || extra="--rows $rows"
      if [ "$tile" == "d" ]; then
        us=$(NEKO_HIP_LIB=$lib python tools/gemm_bench.py $extra --only "$shape" --iters 40 | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
      else
        us=$(NEKO_HIP_LIB=$lib NEKO_GEMM_TILE=$tile python tools/gemm_bench.py $extra --only "$shape" --iters 40 | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
      fi
      line="$line  [$(basename $lib .so | sed s/libneko_hip//):$tile] $us"
    done
    echo "$line"
  done
done
