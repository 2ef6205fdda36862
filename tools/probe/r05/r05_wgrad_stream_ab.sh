#!/bin/bash
# weight gradients on the side stream at 65536 rows (off by default since round 3: one stream measured 0.4 ms faster) again, now that the
# streams have 8 hardware queues
cd $GRAFT_REPO_ROOT
ROUNDS=${ROUNDS:-3} BENCH_ARGS="--steps 30" bash tools/step_ab.sh "one-stream(default)=NEKO_NOP=1" "side-stream-q8=NEKO_WGRAD_STREAM=1" "side-stream-q4=NEKO_WGRAD_STREAM=1 GPU_MAX_HW_QUEUES=4" "one-stream-q4=GPU_MAX_HW_QUEUES=4"
