#!/bin/bash
# Round 5: persistent form of the 8-wave 256 x 256 GEMM (one resident workgroup per CU walks the tiles and requests the next tile's first
# ring stages before its epilogue): outputs must be bit-identical to the one-tile-per-workgroup launch; per-shape and in-step timing.
# NEKO_GEMM_PERSIST_PRE=0: persistent, but nothing requested ahead of the epilogue (what persistence alone costs)
cd $GRAFT_REPO_ROOT
rm -f /tmp/dig0 /tmp/dig1
for rep in 1 2; do
for v in "0" "1" "1 NEKO_GEMM_PERSIST_PRE=0" "1 NEKO_GEMM_PERSIST_PRE=1"; do
  echo "== NEKO_GEMM_PERSIST=$v"
  env NEKO_GEMM_PERSIST=$v python3 tools/gemm_bench.py --rows 65536 --iters 30 --only "fc" $( [ $rep = 1 ] && echo --digest /tmp/dig${v:0:1} ) 2>&1 | grep -v amdgpu.ids
done
done
echo "== digests"; sort -u /tmp/dig0 > /tmp/d0; sort -u /tmp/dig1 > /tmp/d1; cmp /tmp/d0 /tmp/d1 && echo "bit-identical ($(wc -l < /tmp/d0) outputs)"
