#!/bin/bash
# structural memo of GatoPolicy._prepare (descriptor table, sorted tail, loss rows of a batch structure): README-size steps with / without
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "layout_memo or g3_pack or g6" 2>&1 | tail -3
run() { name=$1; w=$2; shift 2; env "$@" python3 bench.py --workload $w --no-cpu-baseline --steps 60 $FL 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w  $name  %.2f ms/step' % d['ms_per_step'])"; }
for r in 1 2 3; do
for w in c2 c3 c4 m-mix; do
FL="" run "memo off" $w NEKO_LAYOUT_CACHE=0
FL="" run "memo on" $w NEKO_NOP=1
done
FL="--force-dp" run "memo off, reducer" c3 NEKO_LAYOUT_CACHE=0
FL="--force-dp" run "memo on, reducer" c3 NEKO_NOP=1
done
