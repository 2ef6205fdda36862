#!/usr/bin/env python
"""one-pass vs two-kernel hd = 32 backward on one left-padded sequence: where do dK / dV differ, and by how much?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops
DEV = "cuda"
H, hd = 3, 32
d = H * hd
for T, pad in ((512, 40), (512, 0), (384, 40), (300, 17), (200, 17), (512, 32), (512, 8)):
    g = torch.Generator(device=DEV).manual_seed(31)
    qkv = torch.randn(T, 3 * d, device=DEV, generator=g).to(torch.bfloat16)
    do = torch.randn(T, d, device=DEV, generator=g).to(torch.bfloat16)
    m = torch.ones(1, T, device=DEV); m[0, :pad] = 0
    kb, ks = ops.mask_bias(m)
    do[:pad] = 0
    res = {}
    for path in (2, 3):
        prev = ops.attn_set_path(path)
        try:
            o, l = ops.attn_fwd(qkv, kb, ks, 1, T, H, hd)
            res[path] = ops.attn_bwd(qkv, o, do, kb, ks, l, 1, T, H, hd).float()
        finally:
            ops.attn_set_path(prev)
    a, b = res[2], res[3]
    for nm, sl in (("dQ", slice(0, d)), ("dK", slice(d, 2 * d)), ("dV", slice(2 * d, 3 * d))):
        df = (a[:, sl] - b[:, sl]).abs()
        rows = torch.nonzero(df.amax(dim=1) > 0).flatten().tolist()
        print(f"T={T} pad={pad} {nm}: {int((df > 0).sum())} elements differ, max {float(df.max()):.3e} (signal {float(a[:, sl].abs().max()):.3e}), rows {rows[:8]}{'...' if len(rows) > 8 else ''} n_rows={len(rows)}")
