#!/bin/bash
# the reducer in a world of one after round 5's three changes (one message per run, collectives ordered behind the side stream instead of
# joining it into the compute stream, 8 hardware queues) against round 4's behaviour, and the step without the reducer
cd $GRAFT_REPO_ROOT
run() { name=$1; w=$2; shift 2; env "$@" python3 bench.py --workload $w --no-cpu-baseline --steps 40 $FL 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w  $name  %.2f ms/step  exposed %.3f' % (d['ms_per_step'], d.get('exposed_comm_ms_per_step') or 0))"; }
for r in 1 2 3; do
for w in c3 m-mix; do
FL="--force-dp" run "reducer r04 (64 MB slices, join, 4 queues)" $w NEKO_DP_BUCKET_MB=64 NEKO_DP_JOIN_MAIN=1 GPU_MAX_HW_QUEUES=4
FL="--force-dp" run "reducer r05" $w NEKO_NOP=1
FL="--force-dp" run "reducer r05 bf16 payload" $w NEKO_DP_PAYLOAD=bf16
FL="" run "no reducer" $w NEKO_NOP=1
done
done
