#!/usr/bin/env python
"""Round 6: where a 256 x 256 tile of gemm_p16 spends its time.  Needs a -DNEKO_P16_TRACE=2 build (tools/probe/p16_variants.sh build trace2):
    NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_p16v_trace2.so NEKO_GEMM_P16=1 python tools/probe/p16_phase_trace.py
Per block (wave 0): entry -> loop start (argument loads, address arithmetic) -> loop end (prologue requests + k-loop) -> stores issued
(epilogue) -> stores drained; per CU (HW_ID, XCC_ID): the gap between one workgroup's last stamp and the next workgroup's entry."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops, _lib  # noqa: E402

BF = torch.bfloat16
M = int(os.environ.get("ROWS", "65536"))
SHAPES = [("fwd qkv NN", M, 2304, 768, False, True, "bias,bf16"),
          ("fwd fc gp NN", M, 3072, 768, False, True, "bias,gelugp,bf16"),
          ("fwd proj NN", M, 768, 768, False, True, "bias,resid,f32"),
          ("fwd pr NN", M, 768, 3072, False, True, "bias,resid,f32"),
          ("dgrad pr4 NT", M, 3072, 768, False, False, "mulact,bf16"),
          ("dgrad o NT", M, 768, 768, False, False, "bf16"),
          ("dgrad fc16 NT", M, 768, 3072, False, False, "bf16")]


def main():
    lib = _lib.load()
    lib.neko_gemm_p16_trace.argtypes = [C.c_void_p]
    dev = "cuda"
    for name, m, n, k, aks, bks, ex in SHAPES:
        A = torch.randn((k, m) if aks else (m, k), device=dev).to(BF)
        Bm = (torch.randn((k, n) if bks else (n, k), device=dev) * 0.05).to(BF)
        kw = dict(a_kstrided=aks, b_kstrided=bks)
        if "bias" in ex: kw["bias"] = torch.randn(n, device=dev)
        if "resid" in ex: kw["resid"] = torch.randn(m, n, device=dev)
        if "gelugp" in ex: kw["act"] = 3; kw["pre_out"] = torch.empty(m, n, dtype=BF, device=dev)
        if "mulact" in ex: kw["act"] = 4; kw["act_in"] = torch.randn(m, n, device=dev).to(BF)
        if "bf16" in ex: kw["out_bf16"] = torch.empty(m, n, dtype=BF, device=dev)
        else: kw["out_f32"] = torch.zeros(m, n, device=dev)
        for _ in range(3):
            ops.gemm(A, Bm, m, n, k, **kw)
        assert ops.gemm_last_mainloop() == 5, ops.MAINLOOP_NAMES[ops.gemm_last_mainloop()]
        nblk = (m // 256) * (n // 256)
        trace = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        lib.neko_gemm_p16_trace(trace.data_ptr())
        torch.cuda.synchronize()
        ops.gemm(A, Bm, m, n, k, **kw)
        torch.cuda.synchronize()
        lib.neko_gemm_p16_trace(None)
        r = trace.cpu().numpy().reshape(-1, 8)
        t = r[:, :5].astype(np.float64) / 100.0            # us (100 MHz)
        t0 = t[:, 0].min()
        setup, loop, epi, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
        clk = (r[:, 7] - r[:, 6]).astype(np.float64)
        ghz = np.median(clk / np.maximum(loop, 1e-3)) / 1e3
        # gaps per CU
        gaps = []
        # HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID in the high word
        cukey = ((r[:, 5] >> 8) & 0xff) | ((r[:, 5] >> 32) << 8)
        for cu in np.unique(cukey):
            rows = np.flatnonzero(cukey == cu)
            rows = rows[np.argsort(t[rows, 0])]
            for a, b in zip(rows[:-1], rows[1:]):
                gaps.append(t[b, 0] - t[a, 4])
        gaps = np.array(gaps) if gaps else np.zeros(1)
        ncu = len(np.unique(cukey))
        wall = t[:, 4].max() - t0
        print(f"{name:14s} blocks {nblk:5d} on {ncu} CUs  wall {wall:7.1f} us = {wall / np.ceil(nblk / ncu):5.2f} per round | entry->loop {setup.mean():5.2f}  "
              f"loop {loop.mean():6.2f} ({clk.mean() / (k // 32):6.1f} clk per k-tile at {ghz:4.2f} GHz)  epilogue {epi.mean():5.2f}  store drain {drain.mean():5.2f}  "
              f"gap to the next workgroup {np.median(gaps):5.2f} (mean {gaps.mean():5.2f}) us")


if __name__ == "__main__":
    main()
