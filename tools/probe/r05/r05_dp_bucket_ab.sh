#!/bin/bash
# data-parallel reducer in a world of one (every collective through RCCL): round 4 (64 MB message slices, the weight-gradient side stream
# joined into the compute stream at every reduce point) against one message per run, and against the collective ordered behind the side stream
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
for w in c3 m-mix; do
  for v in "r04=NEKO_DP_BUCKET_MB=64 NEKO_DP_JOIN_MAIN=1" "whole-runs-join=NEKO_DP_JOIN_MAIN=1" "default=NEKO_NOP=1" "no-reducer=NOFORCE=1"; do
    name=${v%%=*}; envs=${v#*=}
    fl="--force-dp"; [ "$name" = "no-reducer" ] && fl=""
    env $envs python3 bench.py --workload $w --no-cpu-baseline --steps 40 $fl 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('round $r  $w  $name  %.2f ms/step  exposed %.3f' % (d['ms_per_step'], d.get('exposed_comm_ms_per_step') or 0))"
  done
done
done
python3 tools/probe/r05_dp_host_timers.py --workload c3 --no-cpu-baseline --steps 40 --force-dp 2>&1 | grep "calls/step"
