#!/bin/bash
# where the data-parallel reducer's fixed cost at README batch sizes goes: c3 with / without --force-dp, host enqueue time per step
# (NEKO_BENCH_STEP_TIMES=2) and a cProfile of the forced-DP run
cd $GRAFT_REPO_ROOT
for v in "" "--force-dp"; do
  echo "== c3 $v"
  NEKO_BENCH_STEP_TIMES=2 python3 bench.py --workload c3 --no-cpu-baseline --steps 40 $v 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('ms_per_step', round(d['ms_per_step'], 3), 'exposed', d.get('exposed_comm_ms_per_step'), {k: v for k, v in d.items() if 'host' in k or 'enqueue' in k})
    elif ln.startswith('step wall times'):
        v = [float(x) for x in ln.split(':')[1].split()][10:]
        print('host enqueue ms per step (timed steps): mean %.2f  median %.2f' % (sum(v) / len(v), sorted(v)[len(v) // 2]))
"
done
python3 -c "
import cProfile, pstats, sys, runpy
sys.argv = ['bench.py', '--workload', 'c3', '--no-cpu-baseline', '--steps', '40', '--force-dp']
cProfile.run('runpy.run_path(\"bench.py\", run_name=\"__main__\")', '/tmp/prof.out')
" > /tmp/bench_out.txt 2>/tmp/bench_err.txt
python3 - <<'PY'
import pstats
p = pstats.Stats('/tmp/prof.out')
p.sort_stats('cumulative').print_stats(45)
PY
