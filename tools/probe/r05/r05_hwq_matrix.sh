#!/bin/bash
# HIP streams share GPU_MAX_HW_QUEUES (default 4) hardware queues; the c3 step uses the compute stream, the weight-gradient side stream
# and (with the reducer) c10d's communication stream.  Step time against the number of hardware queues, alternating.
cd $GRAFT_REPO_ROOT
run() { name=$1; w=$2; shift 2; env "$@" python3 bench.py --workload $w --no-cpu-baseline --steps 40 $FL 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w  $name  %.2f ms/step' % d['ms_per_step'])"; }
for r in 1 2 3; do
for q in 4 8 16; do
FL="" run "no-reducer q=$q" c3 GPU_MAX_HW_QUEUES=$q
FL="--force-dp" run "reducer q=$q" c3 GPU_MAX_HW_QUEUES=$q
done
done
for r in 1 2; do
for q in 4 8; do
FL="" run "no-reducer q=$q" m-mix GPU_MAX_HW_QUEUES=$q
FL="" run "no-reducer q=$q" c4 GPU_MAX_HW_QUEUES=$q
done
done
