#!/bin/bash
# one-pass (0) against two-kernel (2) hd = 32 attention backward inside the m-mix step, alternating on one box
cd $GRAFT_REPO_ROOT
ROUNDS=3 BENCH_ARGS="--steps 30" bash tools/step_ab.sh "onepass=NEKO_ATTN_PATH=0" "twokernel=NEKO_ATTN_PATH=2"
