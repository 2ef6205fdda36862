#!/usr/bin/env python
"""Where a dK/dV workgroup's time goes (diagnostic build -DNEKO_ATTN_TRACE of attention_res.hip, linked into a tagged
library: see the build line below):
    cd neko_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DNEKO_ATTN_TRACE -c attention_res.hip \
        -o /tmp/attn_trace.o && hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v attention_res.o) /tmp/attn_trace.o \
        -o libneko_hip_atrace.so
    NEKO_HIP_LIB=neko_amd/csrc/libneko_hip_atrace.so python tools/attn_trace.py [--drop 0.1]
Per wave: entry -> staging barrier (prologue), barrier -> last item done (work), sub-tiles processed; per workgroup: the
wait of the earliest-finishing wave for the latest (imbalance) and the gap between consecutive workgroups on a CU."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--drop", type=float, default=0.0)
    a = ap.parse_args()
    B, T, H, hd = a.B, a.T, 24, 32
    d = H * hd
    dev = "cuda"
    lib = _lib.load()
    lib.neko_attn_diag_trace.argtypes = [C.c_void_p]
    qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
    do = torch.randn(B * T, d, device=dev).to(torch.bfloat16)
    kb, ks = ops.mask_bias(torch.ones(B, T, device=dev))
    drop = ops.Drop(a.drop, 0x1234567) if a.drop > 0 else None
    out, lse, mask = ops.attn_fwd(qkv, kb, ks, B, T, H, hd, drop=drop, want_mask=True)
    for _ in range(3):
        ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop, mask=mask)
    trace = torch.zeros(4096 * 16 * 4, dtype=torch.int64, device=dev)
    lib.neko_attn_diag_trace(trace.data_ptr())
    torch.cuda.synchronize()
    ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop, mask=mask)
    torch.cuda.synchronize()
    lib.neko_attn_diag_trace(None)
    t = trace.cpu().numpy().reshape(4096, 16, 4).astype(np.float64)
    nb = min(B * H, 4096)
    t = t[:nb]
    live = t[:, :, 0] > 0                                  # waves that exist
    nw = int(live[0].sum())
    t = t[:, :nw]
    t0 = t[:, :, 0].min()
    start, bar, done, tiles = (t[:, :, 0] - t0) / 100.0, (t[:, :, 1] - t0) / 100.0, (t[:, :, 2] - t0) / 100.0, t[:, :, 3]
    pro, work = bar - start, done - bar
    wg_len = done.max(1) - start.min(1)
    idle = (done.max(1)[:, None] - done)                   # a wave's wait for the slowest wave of its workgroup
    print(f"dK/dV, B {B} T {T} drop {a.drop}: {nb} workgroups x {nw} waves; kernel span {done.max():.1f} us")
    print(f"  per workgroup: length {wg_len.mean():6.2f} us (p90 {np.percentile(wg_len, 90):6.2f}); prologue {pro.mean():5.2f} us (p90 {np.percentile(pro, 90):5.2f}); "
          f"work {work.mean():6.2f} us per wave (max wave {work.max(1).mean():6.2f}); sub-tiles per wave {tiles.mean():5.1f} (max {tiles.max(1).mean():5.1f})")
    print(f"  per sub-tile: {1e3 * (work / np.maximum(tiles, 1)).mean():6.0f} ns in the wave; wait for the slowest wave of the workgroup {idle.mean():5.2f} us = "
          f"{100 * idle.mean() / wg_len.mean():4.1f} % of the workgroup")
    rounds = nb / 256.0
    print(f"  {rounds:.1f} workgroups per CU: sum of lengths {rounds * wg_len.mean():6.1f} us of the {done.max():.1f} us span")


if __name__ == "__main__":
    main()
