#!/usr/bin/env python
"""Timing of the in-place bf16 cross-entropy kernel on one LM-head chunk (4096 rows x 52352 columns)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops  # noqa: E402

R, V, VP = 4096, 52305, 52352
z0 = (torch.randn(R, VP, device="cuda") * 3).to(torch.bfloat16)
tgt = torch.randint(0, V, (R,), device="cuda")
w = torch.full((R,), 1.0 / R, device="cuda")
loss = torch.empty(R, device="cuda")
z = z0.clone()
for _ in range(3):
    z.copy_(z0); ops.ce_bf16_inplace(z, V, VP, tgt, w, loss_row=loss)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(20):
    z.copy_(z0)
    e0.record(); ops.ce_bf16_inplace(z, V, VP, tgt, w, loss_row=loss); e1.record()
    torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
us = tot / 20 * 1e3
print(f"ce_bf16_inplace {R} x {VP}: {us:.1f} us, {2 * R * VP * 2 / us / 1e6:.2f} TB/s (read + write)")
