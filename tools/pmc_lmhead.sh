cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcL1 gpurun_out/pmcL2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcL1 -o f --output-format csv -- python3 tools/lmhead_probe.py 5 ${1:-22784} > gpurun_out/pmcL1/log.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcL2 -o w --output-format csv -- python3 tools/lmhead_probe.py 5 ${1:-22784} > gpurun_out/pmcL2/log.txt 2>&1
