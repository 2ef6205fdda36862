#!/usr/bin/env python
"""Round 4 probe: does the row stride of the streamed operand matter (L2 channel mapping)?  dgrad fc16 / dgrad qkv16 / dgrad o shapes with the
activation operand allocated with padded rows (lda = K + pad elements)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
M = 65536
shapes = [(768, 3072, False), (768, 3072, True), (768, 2304, False), (768, 768, False), (2304, 768, True)]
pads = [(0, 0), (32, 0), (64, 0), (128, 0), (192, 0), (256, 0), (320, 0)]
if len(sys.argv) > 1:          # second experiment: the k-contiguous B operand (the weight) padded as well
    shapes = [(768, 3072, False), (768, 2304, False), (768, 768, False), (3072, 768, False), (52480, 768, False)]
    pads = [(0, 0), (64, 0), (0, 64), (64, 64), (64, 128), (128, 64)]
for (n, k, bks) in shapes:
    rows = 22784 if n > 4096 else M
    for pad, padb in pads:
        Afull = torch.randn(rows, k + pad, device=dev).to(BF)
        A = Afull[:, :k]
        Bfull = (torch.randn((k, n) if bks else (n, k + padb), device=dev) * 0.05).to(BF)
        Bm = Bfull if bks else Bfull[:, :k]
        out = torch.empty(rows, n, dtype=BF, device=dev)
        kw = dict(b_kstrided=bks, lda=k + pad, ldb=(n if bks else k + padb), out_bf16=out)
        for _ in range(3):
            ops.gemm(A, Bm, rows, n, k, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.gemm(A, Bm, rows, n, k, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"{'NN' if bks else 'NT'} {rows} x {n} x {k}  lda = K + {pad:3d}  ldb = {'N' if bks else 'K + %3d' % padb}   {us:8.1f} us  {2.0 * rows * n * k / us / 1e6:7.1f} TFLOP/s")
        del Afull, A, Bfull, Bm
