#!/usr/bin/env python
"""Round 4 probe: is the hand-placed GEMM loop waiting for its STREAMED operand?  The N = 768 GEMMs of the step read their 65536-row
activation operand once from HBM (3 column tiles share a row panel); 8192^3 re-reads both operands from L2.  Same launch twice: the real
operand (lda = K) and lda = 0 (every row of the operand is row 0: the whole A stream hits in L2; results are garbage, timing only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
M = 65536


def run(name, m, n, k, bks, lda, iters=30):
    A = torch.randn(m, k, device=dev).to(BF)
    Bm = (torch.randn((k, n) if bks else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    kw = dict(b_kstrided=bks, lda=lda, out_bf16=out)
    for _ in range(3):
        ops.gemm(A, Bm, m, n, k, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, Bm, m, n, k, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:28s} lda={lda:5d}  {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")


if len(sys.argv) > 1:          # power / clock sampling: one shape for a long time:  a16_stream_probe.py <lda0: 0|1> <iters>
    z = int(sys.argv[1])
    run("dgrad fc16 NT 65536x768x3072", M, 768, 3072, False, 0 if z else 3072, iters=int(sys.argv[2]))
    sys.exit(0)
for lda_of in (lambda k: k, lambda k: 0):
    run("dgrad fc16 NT 65536x768x3072", M, 768, 3072, False, lda_of(3072))
    run("fwd pr-like NN 65536x768x3072", M, 768, 3072, True, lda_of(3072))
    run("dgrad qkv16 NT 65536x768x2304", M, 768, 2304, False, lda_of(2304))
    run("dgrad o NT 65536x768x768", M, 768, 768, False, lda_of(768))
    run("fwd qkv-like NN 65536x2304x768", M, 2304, 768, True, lda_of(768))
    run("lm logits NT 22784x52480x768", 22784, 52480, 768, False, lda_of(768))
