#!/bin/bash
# round 6: nt output stores / p16 tile threshold at the README-size and text workloads (same box, alternating)
cd $GRAFT_REPO_ROOT
for w in c2 c4 m-text; do
  for r in 1 2; do
    for v in "plain=NEKO_HIP_LIB=neko_amd/csrc/libneko_hip.so" "allnt=NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_allnt.so" "allnt_p192=NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_allnt.so NEKO_GEMM_P16_MIN_TILES=192" "plain_nop16=NEKO_HIP_LIB=neko_amd/csrc/libneko_hip.so NEKO_GEMM_P16_RULE=0"; do
      name=${v%%=*}; envs=${v#*=}
      ms=$(env $envs python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])")
      echo "$w round $r  $name  $ms ms/step"
    done
  done
done
