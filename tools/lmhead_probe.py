#!/usr/bin/env python
"""The dominant kernel of bench.py's `roofline` object, alone: the LM-head logits GEMM (ROWS x 768 @ 768 x 52305; ROWS = argv[2],
default 22784 = the loss rows of the m-mix bench batch padded to whole 256-row tiles, one launch per step since round 3),
bf16 out into a Vpad-strided buffer, the call engine.lm_head_loss makes), plus a calibration kernel with a known
byte count (neko_cast_f32_bf16 over 256 Mi elements: 1 GiB read, 0.5 GiB written).  Run under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, tools/pmc_lmhead.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops  # noqa: E402

D, V, VP = 768, 52305, 52480
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 22784
dev = "cuda"
a = torch.randn(ROWS, D, device=dev).to(torch.bfloat16)
w = (torch.randn(VP, D, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(ROWS, VP, dtype=torch.bfloat16, device=dev)
x = torch.randn(256 * 1024 * 1024, device=dev)
y = torch.empty_like(x, dtype=torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.gemm(a, w, ROWS, VP, D, ldb=D, out_bf16=out, ldcb=VP)
    ops.cast_f32_bf16(x, y)
torch.cuda.synchronize()
