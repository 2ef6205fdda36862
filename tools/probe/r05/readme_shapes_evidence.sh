tag=r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c2 c3 c4; do python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_bench.json 2>> gpurun_out/${tag}_bench.err; done
rm -rf gpurun_out/prof_${tag}_c2
rocprofv3 --kernel-trace -d gpurun_out/prof_${tag}_c2 -o c2 -- python3 bench.py --workload c2 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${tag}_c2.log 2>&1
{ echo "# c2 (README halfcheetah shape: 32 x 240 tokens per step); weight gradients on the side stream at this size (kernels overlap: the sum exceeds the step time)"; python3 tools/rocpd_stats.py $(find gpurun_out/prof_${tag}_c2 -name "*.db" | head -1) 45; } > gpurun_out/${tag}_c2_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_${tag}_c2
python3 tools/capture_probe.py c2 > gpurun_out/${tag}_capture_probe.txt 2>&1
