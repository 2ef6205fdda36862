#!/bin/bash
# forward c_fc (+ GELU + gelu') on the hand-placed 4-wave loop (gemm_a16) instead of the 8-wave compiler-scheduled one, inside the step
cd $GRAFT_REPO_ROOT
ROUNDS=${ROUNDS:-3} BENCH_ARGS="--steps 30" bash tools/step_ab.sh "default=NEKO_NOP=1" "a16-gelu-fwd=NEKO_GEMM_A16_RULE=1"
