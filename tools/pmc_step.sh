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
    find $d -name "*agent_info.csv" -delete
    python3 - $d <<'PY'
import csv, collections, glob, json, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
# wall time of every dispatch of THIS pass (kernel-trace csv): with GRBM_GUI_ACTIVE it gives the clock each kernel ran at
dur = {}
for kt in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(kt)):
        dur[r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'][:160], r['Grid_Size'], r['Workgroup_Size'])
    a = agg['|'.join(k)][r['Counter_Name']]
    a[0] += 1; a[1] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r.get('Dispatch_Id') in dur:
        b = agg['|'.join(k)]['DURATION_NS']
        b[0] += 1; b[1] += dur[r['Dispatch_Id']]
json.dump({k: {c: v for c, v in cs.items()} for k, cs in agg.items()}, open(d + '/summary.json', 'w'))
PY
    # keep the merge small: the summary is all the summariser needs
    find $d -name "*kernel_trace.csv" -delete
    find $d -name "*counter_collection.csv" -delete
  done
done
echo done
