cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/attn_bench.py 2>&1 | tail -2
mkdir -p gpurun_out/pmcA
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/pmcA -o a --output-format csv -- python3 tools/attn_bench.py --iters 2 > gpurun_out/pmcA/log.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmcB -o b --output-format csv -- python3 tools/attn_bench.py --iters 2 > gpurun_out/pmcB/log.txt 2>&1
ls gpurun_out/pmcA gpurun_out/pmcB | head
