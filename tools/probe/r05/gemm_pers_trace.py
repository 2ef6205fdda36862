#!/usr/bin/env python
"""Where a tile's time goes inside the persistent GEMM (gemm_pers.hip), from a diagnostic build:
    cd neko_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DNEKO_PERS_TRACE -c gemm_pers.hip -o build/gemm_pers_trace.o
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v gemm_pers) build/gemm_pers_trace.o -o libneko_hip_ptrace.so
    NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_ptrace.so python tools/gemm_pers_trace.py
Per block and role (wave 0: DMA wave, wave 4: store wave) and tile: s_memrealtime at 8 points; prints mean microseconds per phase
over the tiles 1.. of every block (tile 0 has no parked predecessor)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops, _lib  # noqa: E402

BF = torch.bfloat16
SHAPES = [("fwd qkv NN", 65536, 2304, 768, True, "bias"), ("fwd fc NN", 65536, 3072, 768, True, "bias,gelu"),
          ("dgrad o NT", 65536, 768, 768, False, ""), ("plain NT", 65536, 2304, 768, False, "")]


def main():
    lib = _lib.load()
    lib.neko_gemm_pers_trace.argtypes = [C.c_void_p]
    dev = "cuda"
    for name, m, n, k, bks, ex in SHAPES:
        A = torch.randn(m, k, device=dev).to(BF)
        Bm = (torch.randn((k, n) if bks else (n, k), device=dev) * 0.05).to(BF)
        kw = dict(b_kstrided=bks, out_bf16=torch.empty(m, n, dtype=BF, device=dev))
        if "bias" in ex: kw["bias"] = torch.randn(n, device=dev)
        if "gelu" in ex: kw["act"] = 3; kw["pre_out"] = torch.empty(m, n, dtype=BF, device=dev)
        for _ in range(3):
            ops.gemm(A, Bm, m, n, k, **kw)
        trace = torch.zeros(256 * 2 * 16 * 8, dtype=torch.int64, device=dev)
        lib.neko_gemm_pers_trace(trace.data_ptr())
        torch.cuda.synchronize()
        ops.gemm(A, Bm, m, n, k, **kw)
        torch.cuda.synchronize()
        lib.neko_gemm_pers_trace(None)
        t = trace.cpu().numpy().reshape(256, 2, 16, 8).astype(np.float64) / 100.0     # us
        ntile = int((t[0, 0, :, 0] != 0).sum())
        print(f"{name}: {ntile} tiles per block; wall {t[:, :, :ntile, 7].max() - t[:, :, 0, 0].min():.1f} us")
        for role, rn in ((0, "DMA wave  "), (1, "store wave")):
            x = t[:, role, 1:ntile, :]                       # tiles 1..
            d = np.diff(x, axis=-1)                          # 7 phases
            names = ["k-tile 0", "park 0", "8 units", "park 1", "8 units", f"{k // 32 - 19} plain k-tiles", "boundary"]
            print(f"   {rn}: " + " | ".join(f"{nm} {d[..., i].mean():5.2f}" for i, nm in enumerate(names)) +
                  f" | tile {(x[..., 7] - x[..., 0]).mean():6.2f} us")
            x0 = t[:, role, 0, :]
            print(f"   {rn} tile 0: k-loop {(x0[:, 6] - x0[:, 0]).mean():6.2f} boundary {(x0[:, 7] - x0[:, 6]).mean():5.2f}")


if __name__ == "__main__":
    main()
