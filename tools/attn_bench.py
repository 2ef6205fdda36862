#!/usr/bin/env python
"""Isolated timing of the attention kernels at the metric shape (B=32, T=1024, H=24, hd=32) or --hd/--H/--T.
    python tools/attn_bench.py [--iters 20] [--pad 0]
Useful FLOPs (causal half): fwd 4*T^2/2*hd per (b,h); bwd 2.5x fwd (flash convention)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--H", type=int, default=24)
    ap.add_argument("--hd", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pad", type=int, default=0, help="left padding of every sequence")
    ap.add_argument("--only", default="")
    ap.add_argument("--drop", type=float, default=0.0, help="attention dropout probability")
    ap.add_argument("--path", type=int, default=0, help="0 auto (head-resident when applicable; backward: one pass for 256 < T <= 512), 1 streaming, 2 / 3 head-resident with the two-kernel / one-pass backward")
    ap.add_argument("--no-mask", action="store_true", help="backward re-hashes the dropout decisions instead of reusing the forward's")
    ap.add_argument("--zero-pad-grad", action="store_true", help="dO = 0 on the padded rows (training: no loss reaches a padded position)")
    ap.add_argument("--mix-pad", action="store_true", help="left padding 0 / 16 / 36 on a third of the sequences each (the m-mix batch)")
    ap.add_argument("--rotate", type=int, default=1, help="number of independent input sets cycled through (>= 4 at B = 64 defeats the 256 MB "
                    "Infinity Cache: the kernels then see HBM-cold operands as they do inside a training step)")
    a = ap.parse_args()
    ops.attn_set_path(a.path)
    B, T, H, hd = a.B, a.T, a.H, a.hd
    d = H * hd
    dev = "cuda"
    mask = torch.ones(B, T, device=dev)
    if a.pad:
        mask[:, :a.pad] = 0
    if a.mix_pad:
        for b in range(B):
            mask[b, :(0, 16, 36)[b % 3]] = 0
    kb, ks = ops.mask_bias(mask)
    drop = ops.Drop(a.drop, 0x1234567) if a.drop > 0 else None
    sets = []
    for _ in range(max(1, a.rotate)):
        qkv = (torch.randn(B * T, 3 * d, device=dev)).to(torch.bfloat16)
        do = torch.randn(B * T, d, device=dev).to(torch.bfloat16)
        if a.zero_pad_grad:
            do = (do.view(B, T, d) * mask[:, :, None].to(torch.bfloat16)).view(B * T, d).contiguous()
        out, lse, mk = ops.attn_fwd(qkv, kb, ks, B, T, H, hd, drop=drop, want_mask=True)
        sets.append((qkv, do, out, lse, None if a.no_mask else mk))
    fl = 4.0 * T * T / 2 * hd * H * B
    it = [0]

    def nxt():
        it[0] += 1
        return sets[it[0] % len(sets)]

    def timeit(fn, name, flops):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"{name:10s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s (useful, causal)")

    if "bwd" not in a.only:
        timeit(lambda: ops.attn_fwd(nxt()[0], kb, ks, B, T, H, hd, drop=drop, want_mask=not a.no_mask), "attn fwd", fl)
    if "fwd" not in a.only:
        def bwd():
            q, g, o, l, m = nxt()
            ops.attn_bwd(q, o, g, kb, ks, l, B, T, H, hd, drop=drop, mask=m)
        timeit(bwd, "attn bwd", 2.5 * fl)


if __name__ == "__main__":
    main()
