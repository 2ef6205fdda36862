#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 64 96 128 192; do
  python bench.py --no-cpu-baseline --batch $b --steps 15 --warmup 4 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); print('batch $b: %.2f ms/step  %.0f tokens/s  step_mfma_frac %.4f' % (j['ms_per_step'], j['value'], j['step_mfma_frac']))"
done
