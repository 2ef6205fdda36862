#!/usr/bin/env python
"""Phase timing of the GEMM kernel inside a launch (diagnostic build with -DNEKO_GEMM_DIAG=9):
    NEKO_BUILD_TAG=trace NEKO_EXTRA_HIPCC_FLAGS=-DNEKO_GEMM_DIAG=9 python -m neko_amd.build
    NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_trace.so python tools/gemm_trace.py
Per block: prologue (launch -> first k-tile landed), k-loop, epilogue (incl. waiting for its stores), in us; plus the
spread of block start times (the rounds)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops, _lib  # noqa: E402

BF = torch.bfloat16
SHAPES = [("lm logit16 NT", 4096, 52352, 768, False, False, "bf16"),
          ("fwd fc NN", 32768, 3072, 768, False, True, "bias,gelu,bf16"),
          ("fwd qkv NN", 32768, 2304, 768, False, True, "bias,bf16"),
          ("dgrad pr NT", 32768, 3072, 768, False, False, "gelubwd,bf16"),
          ("dgrad fc NT", 32768, 768, 3072, False, False, "f32"),
          ("sq8k NT", 8192, 8192, 8192, False, False, "bf16")]
if os.environ.get("TRACE_EPILOGUES"):      # what the forward-fc epilogue's 10 us are made of
    SHAPES = [("fc plain", 32768, 3072, 768, False, True, "bf16"),
              ("fc bias", 32768, 3072, 768, False, True, "bias,bf16"),
              ("fc gelu nopre", 32768, 3072, 768, False, True, "bias,gelunopre,bf16"),
              ("fc gelu+pre", 32768, 3072, 768, False, True, "bias,gelu,bf16"),
              ("fc f32 out", 32768, 3072, 768, False, True, "f32"),
              ("pr gelubwd", 32768, 3072, 768, False, False, "gelubwd,bf16"),
              ("pr plain", 32768, 3072, 768, False, False, "bf16")]


def main():
    lib = _lib.load()
    lib.neko_gemm_diag_trace.argtypes = [C.c_void_p]
    dev = "cuda"
    for name, m, n, k, aks, bks, ex in SHAPES:
        A = torch.randn((k, m) if aks else (m, k), device=dev).to(BF)
        Bm = (torch.randn((k, n) if bks else (n, k), device=dev) * 0.05).to(BF)
        kw = dict(a_kstrided=aks, b_kstrided=bks)
        if "bias" in ex: kw["bias"] = torch.randn(n, device=dev)
        if "gelu," in ex: kw["act"] = 1; kw["pre_out"] = torch.empty(m, n, dtype=BF, device=dev)
        if "gelunopre" in ex: kw["act"] = 1
        if "gelubwd" in ex: kw["act"] = 2; kw["act_in"] = torch.randn(m, n, device=dev).to(BF)
        if "bf16" in ex: kw["out_bf16"] = torch.empty(m, n, dtype=BF, device=dev)
        else: kw["out_f32"] = torch.zeros(m, n, device=dev)
        for _ in range(3):
            ops.gemm(A, Bm, m, n, k, **kw)
        trace = torch.zeros(65536 * 4, dtype=torch.int64, device=dev)
        lib.neko_gemm_diag_trace(trace.data_ptr())
        torch.cuda.synchronize()
        ops.gemm(A, Bm, m, n, k, **kw)
        torch.cuda.synchronize()
        lib.neko_gemm_diag_trace(None)
        raw = trace.cpu().numpy()
        t = raw[:131072].reshape(-1, 4)
        c = raw[131072:].reshape(-1, 4)
        live = t[:, 0] != 0
        c = c[live[:len(c)]].astype(np.float64)
        t = t[live].astype(np.float64) / 100.0          # us
        if len(c) and c[:, 0].min() > 0:
            nkt = k // 32
            cyc = c[:, 2] - c[:, 1]
            us = (t[:len(c), 2] - t[:len(c), 1])
            print(f"               s_memtime: k-loop {cyc.mean():9.0f} ticks = {cyc.mean() / nkt:7.1f} per k-tile of 32 "
                  f"(split-K launches: per slice); ticks per us {np.median(cyc / np.maximum(us, 1e-3)):7.1f}")
        t0 = t[:, 0].min()
        pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        print(f"{name:14s} blocks {len(t):5d}  wall {t[:, 3].max() - t0:8.1f} us | prologue {pro.mean():6.2f} (p90 {np.percentile(pro, 90):6.2f})"
              f"  k-loop {loop.mean():6.2f} (p90 {np.percentile(loop, 90):6.2f})  epilogue {epi.mean():6.2f} (p90 {np.percentile(epi, 90):6.2f}) us")
        # how synchronised are the rounds: histogram of epilogue start times modulo the mean tile time
        per = (t[:, 3] - t[:, 0]).mean()
        ph = ((t[:, 2] - t0) % per) / per
        hist, _ = np.histogram(ph, bins=10, range=(0, 1))
        print("               epilogue-start phase histogram (10 bins of the mean tile time):", hist.tolist())


if __name__ == "__main__":
    main()
