#!/bin/bash
# kernel table of the c3 step with the reducer attached in a world of one: what does the reducer add on the GPU?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in nodp dp; do
  fl=""; [ $v = dp ] && fl="--force-dp"
  rm -rf gpurun_out/prof_c3_$v
  rocprofv3 --kernel-trace -d gpurun_out/prof_c3_$v -o c3 -- python3 bench.py --workload c3 --steps 10 --warmup 5 --no-cpu-baseline $fl > gpurun_out/prof_c3_$v.log 2>&1
  echo "== c3 $v"; python3 tools/rocpd_stats.py $(find gpurun_out/prof_c3_$v -name "*.db" | head -1) 28 | cut -c1-200
  rm -rf gpurun_out/prof_c3_$v
done
