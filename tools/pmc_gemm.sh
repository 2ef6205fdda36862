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
# one summary line per shape (the format bench.py greps: "<tag>: kernel cycles N; MFMA pipe busy X %; ...") + the raw counters
# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
import csv, collections, glob, os, re
per_tag = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmcG_*/**/g_counter_collection.csv', recursive=True)):
    tag = re.search(r'pmcG_(.+?)_[AB]/', f).group(1)
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if not any(k in r['Kernel_Name'] for k in ('gemm_glds', 'gemm_a16', 'gemm_p16', 'gemm_b16')): continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for c, x in agg.items(): per_tag[tag][c] = x / n[c]
print('# tools/pmc_gemm.sh: SQ counters of the GEMM kernel of each shape (whichever main loop served it: gemm_p16 / gemm_a16 / gemm_glds), per dispatch (two --pmc passes per shape)')
print('# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs')
print('# wave-cycle split = SQ_WAIT_ANY (parked) | SQ_WAIT_INST_ANY (issue-stalled) | SQ_ACTIVE_INST_ANY, of SQ_WAVE_CYCLES')
for tag, c in per_tag.items():
    try:
        kc = c['GRBM_GUI_ACTIVE'] / 8
        wc = c['SQ_WAVE_CYCLES']
        print(f"{tag}: kernel cycles {kc:.0f}; MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * kc):.1f} %; "
              f"parked {100 * c['SQ_WAIT_ANY'] / wc:.1f} % | issue-stalled {100 * c['SQ_WAIT_INST_ANY'] / wc:.1f} % | "
              f"issuing {100 * c['SQ_ACTIVE_INST_ANY'] / wc:.1f} %; LDS bank-conflict cycles "
              f"{100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.1f} % of LDS-active; LDS-issue stall "
              f"{100 * c['SQ_WAIT_INST_LDS'] / wc:.1f} %")
    except KeyError as e:
        print(f"{tag}: incomplete counter set ({e})")
    for k in sorted(c): print(f'   {k:32s} {c[k]:16.0f}')
PY
