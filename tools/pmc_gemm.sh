#!/bin/bash
# SQ counters of the GEMM main loop on two shapes (8192^3 NT and the LM-head logits); separate --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in "sq8k      NT" "lm logit16"; do
  tag=$(echo $shape | tr -d ' ')
  for pass in A B; do
    if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM";
    else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; fi
    d=gpurun_out/pmcG_${tag}_$pass
    rm -rf $d; mkdir -p $d
    rocprofv3 --kernel-trace --pmc $C -d $d -o g --output-format csv -- python3 tools/gemm_bench.py --only "$shape" --iters 2 > $d/log.txt 2>&1
  done
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmcG_*/**/g_counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'gemm_glds' not in r['Kernel_Name']: continue
        agg['gemm'][r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(f)
    for c, x in agg['gemm'].items(): print(f'   {c:32s} {x / n[c]:16.0f}  per dispatch ({n[c]} dispatches)')
PY
