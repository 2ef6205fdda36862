#!/bin/bash
# Per-kernel hardware counters of whole training steps (m-mix and m-text), one counter group per pass as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE | WRITE_SIZE | SQ busy/MFMA group.  3 warm-up + 2 counted steps per pass.
#   tools/pmc_step.sh <tag>      then   python tools/pmc_step_summarise.py <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in m-mix m-text; do
  s=${w#m-}
  for pass in F W Q; do
    case $pass in
      F) C="FETCH_SIZE";;
      W) C="WRITE_SIZE";;
      Q) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE";;
    esac
    d=gpurun_out/pmc_${tag}_${s}_$pass
    rm -rf $d; mkdir -p $d
    rocprofv3 --kernel-trace --pmc $C -d $d -o p --output-format csv -- python3 bench.py --workload $w --steps 2 --warmup 3 --no-cpu-baseline > $d/log.txt 2>&1
    # keep the merge small: the counter csv is all the summariser needs
    find $d -name "*kernel_trace.csv" -delete; find $d -name "*agent_info.csv" -delete
    python3 - $d <<'PY'
import csv, collections, glob, json, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'][:160], r['Grid_Size'], r['Workgroup_Size'])
    a = agg['|'.join(k)][r['Counter_Name']]
    a[0] += 1; a[1] += float(r['Counter_Value'])
json.dump({k: {c: v for c, v in cs.items()} for k, cs in agg.items()}, open(d + '/summary.json', 'w'))
PY
    find $d -name "*counter_collection.csv" -delete
  done
done
echo done
