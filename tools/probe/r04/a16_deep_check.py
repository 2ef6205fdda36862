#!/usr/bin/env python
"""Deep-ring variant of the hand-placed GEMM loop (NEKO_GEMM_A16_DEEP=1) against the standard one: same bits expected (the k order of the
products is the same), run in two processes and compared through a file."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
out_path = sys.argv[1]
res = {}
g = torch.Generator(device=dev).manual_seed(1)
for (m, n, k, bks) in [(1024, 768, 768, False), (1024, 768, 3072, True), (512, 2304, 768, True), (2048, 768, 2304, False), (768, 1024, 384, False)]:
    A = torch.randn(m, k, device=dev, generator=g).to(BF)
    Bm = (torch.randn((k, n) if bks else (n, k), device=dev, generator=g) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    prev = ops.gemm_set_mainloop(1)
    ops.gemm(A, Bm, m, n, k, b_kstrided=bks, out_bf16=out)
    ops.gemm_set_mainloop(prev)
    ref = (A.float() @ (Bm.float() if bks else Bm.float().t()))
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    res[(m, n, k, bks)] = out.cpu()
    print(m, n, k, bks, "rel err vs fp32", f"{err:.2e}")
if os.path.exists(out_path):
    other = torch.load(out_path)
    for key, v in res.items():
        print(key, "bit-identical to the other run:", bool(torch.equal(v, other[key])))
else:
    torch.save(res, out_path)
