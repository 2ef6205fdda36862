#!/usr/bin/env python
"""Reference point, NOT product code: the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16 in, bf16 out, no epilogue)
on the GEMM shapes of the 768d step, next to neko_gemm_bf16 with its fused epilogues (tools/gemm_bench.py).  Tells how much of
the gap to the MFMA peak is this kernel and how much is what bf16 GEMMs of these shapes reach on the chip at all."""
import sys, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
D, V = 768, 52352
shapes = [("fwd qkv   NN", M, 3 * D, D, False, True), ("fwd fc    NN", M, 4 * D, D, False, True), ("fwd pr    NN", M, D, 4 * D, False, True),
          ("dgrad pr  NT", M, 4 * D, D, False, False), ("dgrad fc  NT", M, D, 4 * D, False, False), ("dgrad o   NT", M, D, D, False, False),
          ("wgrad fc  TN", D, 4 * D, M, True, True), ("wgrad qkv TN", D, 3 * D, M, True, True),
          ("lm logits NT", 4096, V, D, False, False), ("lm dH     NN", M, D, V, False, True), ("sq8k      NT", 8192, 8192, 8192, False, False)]
for name, m, n, k, at, bks in shapes:
    a = torch.randn((k, m) if at else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if bks else (n, k), device="cuda").to(torch.bfloat16)
    A = a.t() if at else a
    B = b if bks else b.t()
    for _ in range(3): c = A @ B
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = A @ B
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:14s} M={m:6d} N={n:6d} K={k:6d}  vendor library {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
