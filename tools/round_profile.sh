#!/bin/bash
# Round profile on the GPU box: default bench line, m-text bench line, rocprofv3 kernel-trace of both workloads
# (10 warm-up + 3 traced steps), summaries into gpurun_out/.  Usage: tools/round_profile.sh <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python3 bench.py --workload m-text --no-cpu-baseline > gpurun_out/bench_text.json 2>> gpurun_out/bench_default.err
for w in m-mix m-text; do
  s=${w#m-}
  rm -rf gpurun_out/prof_${tag}_$s
  # single-stream run for the per-kernel table: with the weight gradients on their side stream kernels overlap and
  # their individual durations (not the step time) inflate
  NEKO_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d gpurun_out/prof_${tag}_$s -o $s -- python3 bench.py --workload $w --steps 3 --warmup 10 --no-cpu-baseline > gpurun_out/prof_${tag}_$s.log 2>&1
  db=$(find gpurun_out/prof_${tag}_$s -name "*.db" | head -1)
  { echo "# NEKO_WGRAD_STREAM=0 (one stream: kernels do not overlap; the bench lines of this round use the default two streams)"; python3 tools/rocpd_stats.py $db 45; } > gpurun_out/${tag}_m${s}_kernel_stats.txt 2>&1
done
echo done
