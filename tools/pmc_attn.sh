#!/bin/bash
# SQ counters of the attention kernels at the metric shape (B=32, T=1024, H=24, hd=32), dropout as given:
#   tools/pmc_attn.sh <tag> [drop] [more attn_bench.py arguments, e.g. --B 8 --H 16 --hd 128]   -> gpurun_out/<tag>_attn_counters.txt
# Two passes of 8 SQ counters (MI355X_MICROARCH.md: 8 SQ slots per pass); no tracing besides --kernel-trace.
tag=${1:-rXX}; drop=${2:-0.1}; shift; shift; extra="$*"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pass in A B; do
  case $pass in
    A) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS";;
    B) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM GRBM_GUI_ACTIVE";;
  esac
  d=gpurun_out/pmcattn_$pass; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --pmc $C -d $d -o p --output-format csv -- python3 tools/attn_bench.py --iters 2 --drop $drop $extra > $d/log.txt 2>&1
done
python3 - $tag $drop "$extra" <<'PY' > gpurun_out/${tag}_attn_counters.txt
import csv, glob, collections, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("gpurun_out/pmcattn_A", "gpurun_out/pmcattn_B"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(attn_\w+?_kernel(<\w+>)?)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1)
            a = agg[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
print(f"# attention kernels, tools/attn_bench.py {sys.argv[3] or '(B=32 T=1024 H=24 hd=32)'}, dropout {sys.argv[2]}; per-launch averages (rocprofv3 --pmc, 2 passes)")
print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3: kernel cycles = GRBM_GUI_ACTIVE / 8 (MI355X_MICROARCH.md, counters section)")
for k, cs in agg.items():
    if "attn" not in k: continue
    g = lambda n: (cs[n][1] / cs[n][0]) if n in cs and cs[n][0] else float("nan")
    wc = g("SQ_WAVE_CYCLES")
    print(f"{k}")
    print(f"   wave cycles (quad) {wc:.3e}; issuing {100*g('SQ_ACTIVE_INST_ANY')/wc:.1f} %  issue-stalled {100*g('SQ_WAIT_INST_ANY')/wc:.1f} % (of it LDS-issue {100*g('SQ_WAIT_INST_LDS')/wc:.1f} %)  parked {100*g('SQ_WAIT_ANY')/wc:.1f} %")
    print(f"   VALU-active {100*g('SQ_ACTIVE_INST_VALU')/wc:.1f} % of wave cycles; LDS-active {100*g('SQ_ACTIVE_INST_LDS')/wc:.1f} %")
    print(f"   per launch: VALU insts {g('SQ_INSTS_VALU'):.3e}  LDS insts {g('SQ_INSTS_LDS'):.3e}  SALU {g('SQ_INSTS_SALU'):.3e}  SMEM {g('SQ_INSTS_SMEM'):.3e}")
    cyc = g('GRBM_GUI_ACTIVE') / 8
    print(f"   MFMA busy cycles {g('SQ_VALU_MFMA_BUSY_CYCLES'):.3e} over kernel cycles {cyc:.3e} x 1024 SIMDs -> pipe busy {100*g('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*1024):.1f} %")
    print(f"   LDS bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT'):.3e} of LDS-index-active {g('SQ_LDS_IDX_ACTIVE'):.3e} = {100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):.1f} %")
PY
cat gpurun_out/${tag}_attn_counters.txt
rm -rf gpurun_out/pmcattn_A gpurun_out/pmcattn_B
