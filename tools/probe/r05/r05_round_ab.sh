#!/bin/bash
# Round 5 against round 4 on ONE box, alternating: every switch of the round off (= round 4's step: compiler-scheduled residual GEMMs,
# atomics for the position / separator / patch-position gradients, ln_f backward over a zero-filled gradient, patch backward recomputing
# its statistics, one-pass attention backward at 1024 positions), then each switch alone off, then the default.
cd $GRAFT_REPO_ROOT
R04="NEKO_GEMM_B16=0 NEKO_SORTED_SCATTER=0 NEKO_LNF_ROWS=0 NEKO_PATCH_STATS=0 NEKO_ATTN_PATH=3"
ROUNDS=${ROUNDS:-3} BENCH_ARGS="--steps 30" bash tools/step_ab.sh "r04-equivalent=$R04" "r05-default=NEKO_NOP=1" \
  "no-gemm_b16=NEKO_GEMM_B16=0" "atomic-scatter=NEKO_SORTED_SCATTER=0" "lnf-dense=NEKO_LNF_ROWS=0" "onepass-attn-bwd=NEKO_ATTN_PATH=3"
