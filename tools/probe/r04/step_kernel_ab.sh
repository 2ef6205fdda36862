#!/bin/bash
# per-kernel table of the m-mix step under two environment settings:  tools/probe/step_kernel_ab.sh "<envA>" "<envB>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for envs in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/prof_ab$i
  env $envs NEKO_WGRAD_STREAM=0 NEKO_AB_TAG=$i python3 -c "
import os, subprocess, sys
sys.exit(subprocess.call(['rocprofv3', '--kernel-trace', '-d', 'gpurun_out/prof_ab$i', '-o', 'ab', '--', 'python3', 'bench.py', '--steps', '3', '--warmup', '10', '--no-cpu-baseline']))
" > gpurun_out/prof_ab$i.log 2>&1
  db=$(find gpurun_out/prof_ab$i -name "*.db" | head -1)
  { echo "# $envs"; python3 tools/rocpd_stats.py $db 40; } > gpurun_out/ab_kernels_$i.txt 2>&1
  rm -rf gpurun_out/prof_ab$i
done
