#!/bin/bash
# Round 4: rasterisation group (row panels per group) per shape at 65536 rows; libraries built with -DNEKO_GEMM_GROUP_M=1 / 2 vs the default 8
cd $GRAFT_REPO_ROOT
L=neko_amd/csrc
bash tools/gemm_tile_ab.sh 65536 "$L/libneko_hip.so:d $L/libneko_hip_gm1.so:d $L/libneko_hip_gm2.so:d" 2 "fwd proj|fwd pr |fwd prdrop|dgrad fc16|dgrad o|dgrad qkv16|fwd qkv|fwd fc|dgrad pr|lm logit16" > gpurun_out/r04_group_m_ab.txt 2>&1
python3 bench.py --workload c5-mix --batch 32 --ragged-groups 4 --no-cpu-baseline > gpurun_out/r04_c5mix_rag4_bench.json 2>> gpurun_out/r04_bench2.err
python3 bench.py --model gato-1.2b --workload c5-mix --batch 32 --ragged-groups 4 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04_gato1p2b_c5mix_rag4_bench.json 2>> gpurun_out/r04_bench2.err
ROUNDS=2 bash tools/step_ab.sh "gm8=NEKO_HIP_LIB=$L/libneko_hip.so" "gm1=NEKO_HIP_LIB=$L/libneko_hip_gm1.so" "gm2=NEKO_HIP_LIB=$L/libneko_hip_gm2.so" > gpurun_out/r04_group_m_step_ab.txt 2>&1
