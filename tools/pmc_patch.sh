cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/patch_bench.py 2>&1 | tail -2
mkdir -p gpurun_out/pmcP1 gpurun_out/pmcP2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/pmcP1 -o a --output-format csv -- python3 tools/patch_bench.py 2 > gpurun_out/pmcP1/log.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d gpurun_out/pmcP2 -o b --output-format csv -- python3 tools/patch_bench.py 2 > gpurun_out/pmcP2/log.txt 2>&1
