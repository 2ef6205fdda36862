#!/bin/bash
# Round 5 probe: 32-k ring stages of gemm_a16 requested FOUR k-tiles ahead instead of three (tools/gen_gemm_a16.py --ring-lead 4), us per launch
cd $GRAFT_REPO_ROOT
run() { for sh in "wgrad pr" "wgrad fc" "wgrad qkv" "wgrad o" "dgrad fc16" "dgrad qkv16" "dgrad o" "fwd qkv" "fwd pr " "lm dH" "lm dW" "sq8k      NT" "sq8k      TN"; do us=$(NEKO_GEMM_B16=0 NEKO_HIP_LIB=$1 timeout 300 python tools/gemm_bench.py --rows 65536 --only "$sh" --iters 30 2>/dev/null | grep TFLOP | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}'); echo "$2  $sh : $us us"; done; }
for rep in 1 2; do
run neko_amd/csrc/libneko_hip.so "lead 3 (product)"
run neko_amd/csrc/libneko_hip_lead4.so "lead 4          "
done
for r in 1 2; do for v in "onepass=NEKO_ATTN_PATH=0" "twokernel=NEKO_ATTN_PATH=2"; do name=${v%%=*}; envs=${v#*=}
  ms=$(env $envs python bench.py --workload c4 --no-cpu-baseline --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])"); echo "c4 round $r $name $ms ms/step"; done; done
