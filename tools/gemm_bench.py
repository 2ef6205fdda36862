#!/usr/bin/env python
"""Per-shape timing of neko_gemm_bf16 on the GEMM shapes of the 768d/6L step (B*T = 32768 rows).
    python tools/gemm_bench.py [--iters 20] [--only substr] [--safe N]
Prints us / TFLOP/s per shape; run under rocprofv3 --pmc for counters."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops  # noqa: E402

M, D, V, VP = 32768, 768, 52305, 52480
BF = torch.bfloat16

# name, M, N, K, a_kstrided, b_kstrided, extras
SHAPES = [
    ("fwd qkv   NN", M, 3 * D, D, False, True, "bias,bf16"),
    ("fwd proj  NN", M, D, D, False, True, "bias,resid,f32"),
    ("fwd fc    NN", M, 4 * D, D, False, True, "bias,gelu,bf16"),
    ("fwd fc gp NN", M, 4 * D, D, False, True, "bias,gelugp,bf16"),
    ("fwd pr    NN", M, D, 4 * D, False, True, "bias,resid,f32"),
    ("fwd prdrop NN", M, D, 4 * D, False, True, "bias,resid,drop,f32"),
    ("dgrad pr  NT", M, 4 * D, D, False, False, "gelubwd,bf16"),
    ("dgrad pr4 NT", M, 4 * D, D, False, False, "mulact,bf16"),
    ("dgrad fc  NT", M, D, 4 * D, False, False, "f32"),
    ("dgrad o   NT", M, D, D, False, False, "bf16"),
    ("dgrad fc16 NT", M, D, 4 * D, False, False, "bf16"),
    ("dgrad qkv16 NT", M, D, 3 * D, False, False, "bf16"),
    ("dgrad qkv NT", M, D, 3 * D, False, False, "f32"),
    ("wgrad pr  TN", 4 * D, D, M, True, True, "splitk"),
    ("wgrad fc  TN", D, 4 * D, M, True, True, "splitk"),
    ("wgrad o   TN", D, D, M, True, True, "splitk"),
    ("wgrad qkv TN", D, 3 * D, M, True, True, "splitk"),
    ("lm logits NT", 4096, V, D, False, False, "f32,ldc=%d" % VP),
    ("lm logit16 NT", 4096, VP, D, False, False, "bf16,ldc=%d" % VP),
    ("lm dH     NN", M, D, VP, False, True, "f32"),
    ("lm dW     TN", VP, D, M, True, True, "splitk"),
    ("rl k768   NN", M, 1024, 768, False, True, "bf16"),
    ("rl k768   NT", M, 1024, 768, False, False, "bf16"),
    ("rl k4096  NN", M, 1024, 4096, False, True, "bf16"),
    ("rl k4096  NT", M, 1024, 4096, False, False, "bf16"),
    ("rl k4096  TN", M, 1024, 4096, True, True, "bf16"),
    ("sq4k      NT", 4096, 4096, 4096, False, False, "bf16"),
    ("sq8k      NT", 8192, 8192, 8192, False, False, "bf16"),
    ("sq8k      NN", 8192, 8192, 8192, False, True, "bf16"),
    ("sq8k      TN", 8192, 8192, 8192, True, True, "bf16"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--names", default="", help="comma-separated exact shape names (stripped) instead of --only")
    ap.add_argument("--p16-trace", action="store_true", help="with a -DNEKO_P16_TRACE=1 build: clocks of gemm_p16's loop per k-tile (GELU' dgrad shape)")
    ap.add_argument("--safe", type=int, default=0)
    ap.add_argument("--sk", type=int, default=0, help="force this split-K factor on the split-K shapes")
    ap.add_argument("--dim", type=int, default=0, help="embed dim instead of 768 (2048 = the Gato-1.2B geometry)")
    ap.add_argument("--digest", default="", help="write sha256 of every output of every shape to this file (A/B of two builds / switches: the files must be equal)")
    ap.add_argument("--rows", type=int, default=0, help="B*T rows instead of 32768 (README batch sizes: 7680 = 32 x 240)")
    args = ap.parse_args()
    if args.dim:
        sub = lambda v: {D: args.dim, 3 * D: 3 * args.dim, 4 * D: 4 * args.dim}.get(v, v)
        SHAPES[:] = [(nm, sub(m), sub(n), sub(k), aks, bks, ex) for (nm, m, n, k, aks, bks, ex) in SHAPES
                     if not nm.startswith(("rl", "sq"))]
    if args.rows:
        shapes = []
        for (name, m, n, k, aks, bks, ex) in SHAPES:
            if name.startswith(("rl", "sq")) or (name.startswith("lm") and not args.dim):
                continue
            shapes.append((name, args.rows if m == M else m, n, args.rows if k == M else k, aks, bks, ex))
        SHAPES[:] = shapes
    dev = "cuda"
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(0)
    tot_us = 0.0
    for name, m, n, k, aks, bks, ex in SHAPES:
        if args.only and args.only not in name:
            continue
        if args.names and " ".join(name.split()) not in [" ".join(x.split()) for x in args.names.split(",")]:
            continue
        A = (torch.randn((k, m) if aks else (m, k), device=dev, generator=g)).to(BF)
        Bm = (torch.randn((k, n) if bks else (n, k), device=dev, generator=g) * 0.05).to(BF)
        kw = dict(a_kstrided=aks, b_kstrided=bks, safe_transpose=args.safe)
        ldc = VP if "ldc=" in ex else n
        if "bias" in ex:
            kw["bias"] = torch.randn(n, device=dev)
        if "resid" in ex:
            kw["resid"] = torch.randn(m, n, device=dev)
        if "drop" in ex:
            import types
            kw["drop"] = types.SimpleNamespace(thr=26, key=0x1234567, scale=256.0 / 230.0)
        if "gelu," in ex:
            kw["act"] = 1
            kw["pre_out"] = torch.empty(m, n, dtype=BF, device=dev)
        if "gelugp" in ex:
            kw["act"] = 3
            kw["pre_out"] = torch.empty(m, n, dtype=BF, device=dev)
        if "mulact" in ex:
            kw["act"] = 4
            kw["act_in"] = torch.randn(m, n, device=dev).to(BF)
        if "gelubwd" in ex:
            kw["act"] = 2
            kw["act_in"] = torch.randn(m, n, device=dev).to(BF)
        if "bf16" in ex:
            kw["out_bf16"] = torch.empty(m, ldc, dtype=BF, device=dev)
            kw["ldcb"] = ldc
        else:
            kw["out_f32"] = torch.zeros(m, ldc, device=dev)
            kw["ldcf"] = ldc
        if "splitk" in ex:
            sk, kps = ops.pick_splitk(m, n, k)
            if args.sk:
                sk = args.sk
                kps = ((k + sk - 1) // sk + 63) // 64 * 64
                sk = (k + kps - 1) // kps
            kw.update(splitk=sk, k_per_split=kps, accumulate=(sk == 1))
        if "acc" in ex:
            kw["accumulate"] = True
        for _ in range(3):
            ops.gemm(A, Bm, m, n, k, **kw)
        if args.digest:
            import hashlib
            torch.cuda.synchronize()
            with open(args.digest, "a") as fh:
                for key in ("out_bf16", "out_f32", "pre_out"):
                    if key in kw and not kw.get("accumulate"):
                        fh.write(f"{name} {key} {hashlib.sha256(kw[key].view(torch.uint8).cpu().numpy().tobytes()).hexdigest()}\n")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.gemm(A, Bm, m, n, k, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        tf = 2.0 * m * n * k / us / 1e6
        tot_us += us
        print(f"{name:14s} M={m:6d} N={n:6d} K={k:6d} {ex:22s} {us:9.1f} us  {tf:7.1f} TFLOP/s"
              + (f"  splitk={kw['splitk']}" if kw.get("splitk", 1) > 1 else ""))
        del A, Bm, kw
    print(f"sum {tot_us:.1f} us")
    if args.p16_trace:
        from neko_amd import _lib
        rows = args.rows or M
        m, n, k = rows, 4 * D, D
        dY = torch.randn(m, k, device=dev, generator=g).to(BF)
        W = (torch.randn(n, k, device=dev, generator=g) * 0.05).to(BF)
        fac = torch.rand(m, n, device=dev, generator=g).to(BF)
        out = torch.empty(m, n, dtype=BF, device=dev)
        cs = torch.zeros(n, device=dev)
        ws = torch.zeros(int(_lib.load().neko_gemm_colsum_ws_floats(m, n)), device=dev)
        for _ in range(3):
            _lib.call("neko_gemm_dgrad_gelu_colsum", dY.data_ptr(), k, W.data_ptr(), k, m, n, k, fac.data_ptr(), n, 1, out.data_ptr(), n,
                      ws.data_ptr(), cs.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ops.gemm_last_mainloop() == 5, "the trace needs gemm_p16 (NEKO_GEMM_P16=1)"
        nblk = (m // 256) * (n // 256)
        t = ws[(m // 128) * n:].view(torch.int64)[:nblk].cpu().double()
        nkt = k // 32
        print(f"p16 trace (GELU' dgrad {m} x {n} x {k}, {nblk} blocks): asm loop incl. prologue requests {float(t.mean()):.0f} clocks "
              f"(p10 {float(t.quantile(0.1)):.0f}, p90 {float(t.quantile(0.9)):.0f}) = {float(t.mean()) / nkt:.1f} per k-tile of 32")


if __name__ == "__main__":
    main()
