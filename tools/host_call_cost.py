#!/usr/bin/env python
"""Host cost of one ABI call: the ops.gemm wrapper vs the raw ctypes call with prebuilt arguments vs torch.empty,
on shapes small enough that the device never limits (queue drained every 200 calls)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import _lib, ops  # noqa: E402

dev = "cuda"
M, N, K = 64, 128, 64
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = torch.randn(K, N, device=dev).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
x = torch.randn(M, 768, device=dev)
g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
y = torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)


def bench(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % 200 == 199:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


lib = _lib.load()
s = ops._stream()
raw_args = (A.data_ptr(), K, 0, B.data_ptr(), N, 1, M, N, K, 1.0, None, bias.data_ptr(), None, 0, 0, None, 0, None, 0,
            None, 0, 0, out.data_ptr(), N, 1, 0, None, 0, 0, 1.0, 0, s)
print(f"ops.gemm wrapper           {bench(lambda: ops.gemm(A, B, M, N, K, b_kstrided=True, bias=bias, out_bf16=out)):6.2f} us")
print(f"raw lib.neko_gemm_bf16     {bench(lambda: lib.neko_gemm_bf16(*raw_args)):6.2f} us")
print(f"ops.layernorm_fwd wrapper  {bench(lambda: ops.layernorm_fwd(x, g, b, y16=y, mean=mean, rstd=rstd)):6.2f} us")
print(f"ops._stream()              {bench(lambda: ops._stream()):6.2f} us")
print(f"torch.empty(M, N)          {bench(lambda: torch.empty(M, N, dtype=torch.bfloat16, device=dev)):6.2f} us")
print(f"tensor slice view          {bench(lambda: out[8:16]):6.2f} us")
print(f"data_ptr()                 {bench(lambda: out.data_ptr()):6.2f} us")
