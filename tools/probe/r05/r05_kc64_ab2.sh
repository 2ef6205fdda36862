#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do
  echo "== NEKO_GEMM_KC64=$v (gemm_glds only: NEKO_GEMM_A16=0 NEKO_GEMM_B16=0)"
  NEKO_GEMM_KC64=$v NEKO_GEMM_A16=0 NEKO_GEMM_B16=0 python3 tools/gemm_bench.py --rows 65536 --iters 30 2>&1 | grep -v amdgpu.ids | grep -v "wgrad\|^sum"
done
done
run() { name=$1; shift; args=$1; shift; env "$@" python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$name  %.2f ms/step' % d['ms_per_step'])"; }
for r in 1 2 3; do
run "m-mix default" "--steps 30" NEKO_NOP=1
run "m-mix kc64" "--steps 30" NEKO_GEMM_KC64=1
run "c4 default" "--workload c4 --steps 40" NEKO_NOP=1
run "c4 kc64" "--workload c4 --steps 40" NEKO_GEMM_KC64=1
run "gato-1.2b m-text b8 default" "--model gato-1.2b --workload m-text --batch 8 --steps 8 --warmup 3" NEKO_NOP=1
run "gato-1.2b m-text b8 kc64" "--model gato-1.2b --workload m-text --batch 8 --steps 8 --warmup 3" NEKO_GEMM_KC64=1
done
