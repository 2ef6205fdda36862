#!/bin/bash
# Round 5: which launch classes gemm_b16.hip should take inside the m-mix step (NEKO_GEMM_B16_RULE bit mask), alternating on one box
cd $GRAFT_REPO_ROOT
ROUNDS=${ROUNDS:-2} BENCH_ARGS="--steps 30" bash tools/step_ab.sh "off=NEKO_GEMM_B16=0" "resid=NEKO_GEMM_B16_RULE=1" "resid+lm=NEKO_GEMM_B16_RULE=17" "resid+gelu=NEKO_GEMM_B16_RULE=3" "resid+dgelu=NEKO_GEMM_B16_RULE=5" "resid+plain=NEKO_GEMM_B16_RULE=9" "all=NEKO_GEMM_B16=1" "lnfrows0=NEKO_LNF_ROWS=0 NEKO_GEMM_B16=0" "gmhuge8=NEKO_GEMM_GM_HUGE=8 NEKO_GEMM_B16=0"
