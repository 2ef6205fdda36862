#!/bin/bash
# what the stored bf16 gelu' factor (NEKO_GELU_FACTOR, default on) and the bf16 LayerNorm dy (NEKO_LN_DY_BF16, default on)
# cost against the CPU oracle at 6 and at 24 layers: the parity tests' own report lines under each setting.
cd $GRAFT_REPO_ROOT
for v in "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  echo "## NEKO_GELU_FACTOR=$1 NEKO_LN_DY_BF16=$2"
  NEKO_GELU_FACTOR=$1 NEKO_LN_DY_BF16=$2 python -m pytest tests/test_metric_parity_gpu.py -q -s -k "24_layers or metric_shape_768d" 2>&1 | grep -o "\[parity.*\|[0-9]* passed.*\|[0-9]* failed.*"
done
