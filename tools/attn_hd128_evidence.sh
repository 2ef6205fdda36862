#!/bin/bash
# hd = 128 / 64 attention evidence of a round:  tools/attn_hd128_evidence.sh <tag>  -> gpurun_out/<tag>_attn_hd128_bench.txt, <tag>_hd128_attn_counters.txt,
# <tag>_gato1p2b_mtext_b8_bench.json
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ echo "# hd = 128 (configs[4]: 2048d x 16 heads), B = 8, T = 1024: DMA-ring kernels (attention_stream.hip), then the register-staged kernels they replace (--path 1)"
  python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30; python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30 --drop 0.1
  python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30 --path 1; python3 tools/attn_bench.py --B 8 --H 16 --hd 128 --iters 30 --drop 0.1 --path 1
  echo "# hd = 64, B = 8, H = 32: DMA ring, dropout 0 / 0.1, then register-staged"
  python3 tools/attn_bench.py --B 8 --H 32 --hd 64 --iters 30; python3 tools/attn_bench.py --B 8 --H 32 --hd 64 --iters 30 --drop 0.1; python3 tools/attn_bench.py --B 8 --H 32 --hd 64 --iters 30 --path 1; } 2>&1 | grep -v amdgpu > gpurun_out/${tag}_attn_hd128_bench.txt
python3 bench.py --model gato-1.2b --workload m-text --batch 8 --steps 10 --warmup 3 > gpurun_out/${tag}_gato1p2b_mtext_b8_bench.json 2>/dev/null
bash tools/pmc_attn.sh ${tag}_hd128 0.1 --B 8 --H 16 --hd 128 > /dev/null 2>&1
cat gpurun_out/${tag}_attn_hd128_bench.txt; grep -o '"ms_per_step[^,]*' gpurun_out/${tag}_gato1p2b_mtext_b8_bench.json; grep -o '"step_mfma_frac[^,]*' gpurun_out/${tag}_gato1p2b_mtext_b8_bench.json
grep -A6 "stream_kernel" gpurun_out/${tag}_hd128_attn_counters.txt | grep -E "kernel$|pipe busy|parked"
