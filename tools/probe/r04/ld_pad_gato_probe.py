#!/usr/bin/env python
"""Round 4 probe, 2048d geometry (Gato-1.2B, 32768 rows): activation operands with power-of-two row strides (4096 / 16384 B) against the same
operands with padded rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
M, D = 32768, 2048
# name, m, n, k, a_ks, b_ks: which operand is the activation (padded): A for NN / NT, both for TN
SH = [("fwd qkv NN", M, 3 * D, D, False, True), ("fwd fc NN", M, 4 * D, D, False, True), ("fwd pr NN", M, D, 4 * D, False, True),
      ("dgrad pr NT", M, 4 * D, D, False, False), ("dgrad fc NT", M, D, 4 * D, False, False), ("dgrad qkv NT", M, D, 3 * D, False, False),
      ("wgrad pr TN", 4 * D, D, M, True, True), ("wgrad fc TN", D, 4 * D, M, True, True), ("wgrad qkv TN", D, 3 * D, M, True, True)]
for (name, m, n, k, aks, bks) in SH:
    for pad in (0, 64, 128):
        if aks:   # TN: both operands are activations [k rows][m or n]
            A = torch.randn(k, m + pad, device=dev).to(BF)
            Bm = torch.randn(k, n + pad, device=dev).to(BF)
            lda, ldb = m + pad, n + pad
            sk, kps = ops.pick_splitk(m, n, k)
            out = torch.zeros(m, n, device=dev)
            kw = dict(a_kstrided=True, b_kstrided=True, lda=lda, ldb=ldb, out_f32=out, ldcf=n, accumulate=(sk == 1), splitk=sk, k_per_split=kps)
        else:
            A = torch.randn(m, k + pad, device=dev).to(BF)
            Bm = (torch.randn((k, n) if bks else (n, k), device=dev) * 0.05).to(BF)
            out = torch.empty(m, n, dtype=BF, device=dev)
            kw = dict(b_kstrided=bks, lda=k + pad, out_bf16=out)
        for _ in range(3):
            ops.gemm(A, Bm, m, n, k, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, Bm, m, n, k, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:14s} {m:6d} x {n:5d} x {k:6d}  activation rows + {pad:3d}   {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
        del A, Bm, out
