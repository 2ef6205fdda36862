#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for w in c2 c3 c4; do
  for v in "default=NEKO_GEMM_B16=-1" "b16all=NEKO_GEMM_B16=1" "off=NEKO_GEMM_B16=0"; do
    name=${v%%=*}; envs=${v#*=}
    ms=$(env $envs python bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $r $w $name $ms ms/step"
  done
done
done
