#!/bin/bash
# Same-box A/B of the m-mix step under environment toggles:  tools/step_ab.sh "<name>=<env assignments>" ... (rounds via ROUNDS)
cd $GRAFT_REPO_ROOT
for r in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    name=${v%%=*}; envs=${v#*=}
    ms=$(env $envs python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $r  $name  $ms ms/step"
  done
done
