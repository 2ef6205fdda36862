#!/usr/bin/env python
"""Round 4 probe, weight-gradient (TN) shapes: the 3072-wide activation operand ([65536 rows = k][3072]) with padded rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
K = 65536
for (name, m, n) in [("wgrad pr  dW[3072,768] = h^T dY", 3072, 768), ("wgrad fc  dW[768,3072] = a2^T dH", 768, 3072),
                     ("wgrad qkv dW[768,2304]", 768, 2304), ("wgrad o   dW[768,768]", 768, 768)]:
    for pad in (0, 64, 128):
        pa = pad if m == 3072 or (m == 768 and n != 3072) else 0
        pb = pad if n >= 2304 else 0
        A = torch.randn(K, m + pa, device=dev).to(BF)
        Bm = torch.randn(K, n + pb, device=dev).to(BF)
        out = torch.zeros(m, n, device=dev)
        sk, kps = ops.pick_splitk(m, n, K)
        kw = dict(a_kstrided=True, b_kstrided=True, lda=m + pa, ldb=n + pb, out_f32=out, ldcf=n, accumulate=(sk == 1), splitk=sk, k_per_split=kps)
        for _ in range(3):
            ops.gemm(A, Bm, m, n, K, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.gemm(A, Bm, m, n, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"{name:34s} lda = M + {pa:3d}  ldb = N + {pb:3d}  splitk {sk:2d}  {us:8.1f} us  {2.0 * m * n * K / us / 1e6:7.1f} TFLOP/s")
        del A, Bm
