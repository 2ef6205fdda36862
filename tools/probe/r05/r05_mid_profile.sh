#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/patch_bench.py 20 546 > gpurun_out/r05_patch_bench.txt 2>&1
python tools/patch_bench.py 20 546 >> gpurun_out/r05_patch_bench.txt 2>&1
rm -rf gpurun_out/prof_r05mid
NEKO_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d gpurun_out/prof_r05mid -o mix -- python3 bench.py --workload m-mix --steps 3 --warmup 10 --no-cpu-baseline > gpurun_out/prof_r05mid.log 2>&1
db=$(find gpurun_out/prof_r05mid -name "*.db" | head -1)
python3 tools/rocpd_stats.py $db 60 > gpurun_out/r05mid_mmix_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_r05mid
for i in 1 2; do python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; print('%.2f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])"; NEKO_PATCH_STATS=0 NEKO_GEMM_B16=0 NEKO_LNF_ROWS=0 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; print('r04-equivalent toggles %.2f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])"; done
