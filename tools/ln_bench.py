#!/usr/bin/env python
"""LayerNorm forward / backward at the metric shape (32768 rows x 768, or argv[1] rows): time and effective HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops
M, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 768
dev = "cuda"
x = torch.randn(M, d, device=dev); w = torch.randn(d, device=dev); b = torch.randn(d, device=dev)
dy = torch.randn(M, d, device=dev); gin = torch.randn(M, d, device=dev)
y16 = torch.empty(M, d, dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
dx = torch.empty(M, d, device=dev); dx16 = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
def t(fn, name, nbytes):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s")
t(lambda: ops.layernorm_fwd(x, w, b, y16=y16, mean=mean, rstd=rstd), "ln fwd", M * d * 6)
t(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dg, db, g_in=gin, dx=dx, dx16=dx16), "ln bwd", M * d * 18)
dy16 = dy.to(torch.bfloat16)
t(lambda: ops.layernorm_bwd(dy16, x, w, mean, rstd, dg, db, g_in=gin, dx=dx, dx16=dx16), "ln bwd (bf16 dy)", M * d * 16)
cs = torch.zeros(d, device=dev)
t(lambda: ops.layernorm_bwd(dy16, x, w, mean, rstd, dg, db, g_in=gin, dx=dx, dx16=dx16, colsum16=cs), "ln bwd (bf16 dy, + column sums)", M * d * 16)
