#!/bin/bash
# Everything the round's DESIGN / profiles cite, from ONE box: bench lines (m-mix default, m-mix B=32, m-text), rocprofv3
# kernel-trace tables of both workloads (single stream), per-kernel PMC passes of the step (FETCH / WRITE / SQ), the LM-head
# traffic calibration, GEMM and attention SQ counters.   tools/round_evidence.sh <tag>   -> gpurun_out/<tag>_*
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${tag}_mmix_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --batch 32 --no-cpu-baseline > gpurun_out/${tag}_mmix_b32_bench.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --workload m-text --no-cpu-baseline > gpurun_out/${tag}_mtext_bench.json 2>> gpurun_out/${tag}_bench.err
for w in c2 c3 c4; do python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_bench.json 2>> gpurun_out/${tag}_bench.err; done
python3 bench.py --model gato-1.2b --workload m-text --batch 8 --steps 10 --warmup 3 > gpurun_out/${tag}_gato1p2b_mtext_b8_bench.json 2>> gpurun_out/${tag}_bench.err
# configs[4] as BASELINE.json states it: Gato-1.2B on the full mix (padded m-mix, and c5-mix in 4 length groups = one varlen attention launch, hd = 128)
python3 bench.py --model gato-1.2b --workload m-mix --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_gato1p2b_mmix_b32_bench.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --model gato-1.2b --workload c5-mix --batch 32 --ragged-groups 4 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_gato1p2b_c5mix_rag4_b32_bench.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --model gato-1.2b --workload m-mix --batch 64 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_gato1p2b_mmix_b64_bench.json 2>> gpurun_out/${tag}_bench.err
# data-parallel path in a world of one (RCCL executes every collective): regression anchor for exposed_comm_ms_per_step, both payloads
python3 bench.py --force-dp --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_mmix_forcedp_fp32_bench.json 2>> gpurun_out/${tag}_bench.err
NEKO_DP_PAYLOAD=bf16 python3 bench.py --force-dp --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_mmix_forcedp_bf16_bench.json 2>> gpurun_out/${tag}_bench.err
# ... and on configs[2] (c3: the configuration BASELINE names for data-parallel training; control-only, so the 50257 text rows of embed_token
# are declared unused and leave the reduction), both payloads
python3 bench.py --workload c3 --force-dp --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/${tag}_c3_forcedp_fp32_bench.json 2>> gpurun_out/${tag}_bench.err
NEKO_DP_PAYLOAD=bf16 python3 bench.py --workload c3 --force-dp --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/${tag}_c3_forcedp_bf16_bench.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --workload c5-mix --batch 32 --no-cpu-baseline > gpurun_out/${tag}_c5mix_pad_bench.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --workload c5-mix --batch 32 --ragged-groups 4 --no-cpu-baseline > gpurun_out/${tag}_c5mix_rag4_bench.json 2>> gpurun_out/${tag}_bench.err
NEKO_ATTN_VARLEN=0 python3 bench.py --workload c5-mix --batch 32 --ragged-groups 4 --no-cpu-baseline > gpurun_out/${tag}_c5mix_rag4_buckets_bench.json 2>> gpurun_out/${tag}_bench.err
for w in m-mix m-text; do
  s=${w#m-}
  rm -rf gpurun_out/prof_${tag}_$s
  NEKO_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d gpurun_out/prof_${tag}_$s -o $s -- python3 bench.py --workload $w --steps 3 --warmup 10 --no-cpu-baseline > gpurun_out/prof_${tag}_$s.log 2>&1
  db=$(find gpurun_out/prof_${tag}_$s -name "*.db" | head -1)
  { echo "# NEKO_WGRAD_STREAM=0 (one stream: kernels do not overlap; the default (auto) is one stream too at these 65536 rows per step); B = 64 x T = 1024 per step"; python3 tools/rocpd_stats.py $db 45; } > gpurun_out/${tag}_m${s}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_${tag}_$s
done
rm -rf gpurun_out/prof_${tag}_c2
rocprofv3 --kernel-trace -d gpurun_out/prof_${tag}_c2 -o c2 -- python3 bench.py --workload c2 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${tag}_c2.log 2>&1
{ echo "# c2 (README halfcheetah shape: 32 x 240 tokens per step); weight gradients on the side stream at this size (kernels overlap: the sum exceeds the step time)"; python3 tools/rocpd_stats.py $(find gpurun_out/prof_${tag}_c2 -name "*.db" | head -1) 45; } > gpurun_out/${tag}_c2_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_${tag}_c2
bash tools/pmc_step.sh $tag > gpurun_out/${tag}_pmc_step.log 2>&1
bash tools/pmc_lmhead.sh > gpurun_out/${tag}_pmc_lmhead.log 2>&1
python3 tools/pmc_lmhead_summarise.py $tag > /dev/null 2>> gpurun_out/${tag}_pmc_lmhead.log
rm -rf gpurun_out/pmcL1 gpurun_out/pmcL2
bash tools/pmc_gemm.sh > gpurun_out/${tag}_gemm_counters.txt 2>&1
rm -rf gpurun_out/pmcG_*
bash tools/pmc_attn.sh $tag 0.1 --B 64 > /dev/null 2>&1
bash tools/pmc_attn.sh ${tag}_hd128 0.1 --B 8 --H 16 --hd 128 > /dev/null 2>&1
python3 tools/gemm_bench.py --iters 30 > gpurun_out/${tag}_gemm_bench.txt 2>&1
{ echo "# B = 64, T = 1024, 24 heads of 32; 5 input sets cycled (HBM-cold operands); left padding 0 / 16 / 36 by thirds as in the m-mix batch, dO = 0 on padded rows"
  echo "# default path (two-kernel backward above 512 positions), dropout 0.1 / 0"; python3 tools/attn_bench.py --B 64 --drop 0.1 --iters 30 --rotate 5 --mix-pad --zero-pad-grad; python3 tools/attn_bench.py --B 64 --iters 30 --rotate 5 --mix-pad --zero-pad-grad
  echo "# one-pass backward (--path 3), dropout 0.1 / 0"; python3 tools/attn_bench.py --B 64 --drop 0.1 --iters 30 --rotate 5 --mix-pad --zero-pad-grad --path 3 --only bwd; python3 tools/attn_bench.py --B 64 --iters 30 --rotate 5 --mix-pad --zero-pad-grad --path 3 --only bwd; } > gpurun_out/${tag}_attn_bench.txt 2>&1
{ echo "# hd = 128 (configs[4]: 2048d x 16 heads), B = 8, T = 1024: DMA-ring kernels (attention_stream.hip), then the register-staged kernels they replace (--path 1)"
  python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30; python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30 --drop 0.1
  python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30 --path 1; python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30 --drop 0.1 --path 1
  echo "# hd = 64, B = 8, H = 32"; python3 tools/attn_bench.py --B 8 --H 32 --hd 64 --iters 30; python3 tools/attn_bench.py --B 8 --H 32 --hd 64 --iters 30 --path 1; } > gpurun_out/${tag}_attn_hd128_bench.txt 2>&1
{ tools/power_probe.sh step python3 bench.py --no-cpu-baseline --steps 150; tools/power_probe.sh sq8k python3 tools/gemm_bench.py --only sq8k --iters 3000; tools/power_probe.sh attn_hd32 python3 tools/attn_bench.py --B 64 --iters 2000 --drop 0.1; } > gpurun_out/${tag}_power_clocks.txt 2>&1
python3 tools/decode_bench.py > gpurun_out/${tag}_decode_bench.txt 2>&1
python3 tools/capture_probe.py c2 > gpurun_out/${tag}_capture_probe.txt 2>&1
echo done
