#!/bin/bash
# c3 with the reducer in a world of one: which part of the reducer costs the 0.9 ms?
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --workload c3 --no-cpu-baseline --steps 40 $FL 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$name  %.2f ms/step  exposed %.3f' % (d['ms_per_step'], d.get('exposed_comm_ms_per_step') or 0))"; }
for r in 1 2; do
FL="" run "no-reducer" NEKO_NOP=1
FL="--force-dp" run "reducer" NEKO_NOP=1
FL="--force-dp" run "reducer, collectives skipped" NEKO_DP_DRY=1
FL="--force-dp" run "reducer, 8 hardware queues" GPU_MAX_HW_QUEUES=8
FL="--force-dp" run "reducer, one stream" NEKO_WGRAD_STREAM=0
FL="" run "no-reducer, one stream" NEKO_WGRAD_STREAM=0
done
