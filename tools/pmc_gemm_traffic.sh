#!/bin/bash
# Fabric-side bytes per launch (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and L2 hit rate of one GEMM shape of tools/gemm_bench.py:
#   tools/pmc_gemm_traffic.sh "<shape substring>" [rows]      (environment toggles such as NEKO_GEMM_PERS pass through)
# FETCH_SIZE is in KiB units of 64-B requests; the gfx950 wide-read correction (x2, MI355X_MICROARCH.md) is NOT applied here: compare variants.
shape=$1; rows=${2:-65536}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmcT; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --pmc $c -d $d -o t --output-format csv -- python3 tools/gemm_bench.py --rows $rows --only "$shape" --iters 3 > $d/log.txt 2>&1
  python3 - "$c" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('gpurun_out/pmcT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm' not in r['Kernel_Name']: continue
        a = agg[(r['Kernel_Name'][:60], r['Counter_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in agg.items(): print(f"   {c:14s} {v / n:14.0f} per launch ({n} launches)  {k}")
PY
done
rm -rf gpurun_out/pmcT
