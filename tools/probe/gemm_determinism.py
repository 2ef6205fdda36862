#!/usr/bin/env python
"""Round 6: run-to-run bit identity and cross-loop agreement of neko_gemm_bf16 at the step's shapes (a race in a hand-placed loop shows as
rare differing elements at full size, not in the small unit tests).  NEKO_GEMM_P16 / set_mainloop select the loop."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops
BF = torch.bfloat16
dev = "cuda"
ROWS = int(os.environ.get("ROWS", "16384"))
D = 768
SH = [("fwd qkv", ROWS, 3 * D, D, False, True, "bias,bf16"), ("fwd fc gp", ROWS, 4 * D, D, False, True, "bias,gelugp,bf16"),
      ("fwd proj", ROWS, D, D, False, True, "bias,resid,f32"), ("fwd pr", ROWS, D, 4 * D, False, True, "bias,resid,f32"),
      ("dgrad pr4", ROWS, 4 * D, D, False, False, "mulact,bf16"), ("dgrad o", ROWS, D, D, False, False, "bf16"),
      ("dgrad fc16", ROWS, D, 4 * D, False, False, "bf16"), ("dgrad qkv16", ROWS, D, 3 * D, False, False, "bf16"),
      ("lm logit", 4096, 52480, D, False, False, "bf16")]
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for mode in (3, 2, 1, 0, -1):
    prev = ops.gemm_set_mainloop(mode)
    for name, m, n, k, aks, bks, ex in SH:
        A = torch.randn((k, m) if aks else (m, k), device=dev, generator=g).to(BF)
        Bm = (torch.randn((k, n) if bks else (n, k), device=dev, generator=g) * 0.05).to(BF)
        kw = dict(a_kstrided=aks, b_kstrided=bks)
        if "bias" in ex: kw["bias"] = torch.randn(n, device=dev, generator=g)
        if "resid" in ex: kw["resid"] = torch.randn(m, n, device=dev, generator=g)
        if "gelugp" in ex: kw["act"] = 3
        if "mulact" in ex: kw["act"] = 4; kw["act_in"] = torch.rand(m, n, device=dev, generator=g).to(BF)
        outs = []
        for it in range(6):
            o = torch.full((m, n), float("nan"), dtype=BF if "bf16" in ex else torch.float32, device=dev)
            kk = dict(kw)
            if "bf16" in ex: kk["out_bf16"] = o
            else: kk["out_f32"] = o
            if "gelugp" in ex: kk["pre_out"] = torch.full((m, n), float("nan"), dtype=BF, device=dev)
            ops.gemm(A, Bm, m, n, k, **kk)
            outs.append((o, kk.get("pre_out")))
        torch.cuda.synchronize()
        loop = ops.MAINLOOP_NAMES[ops.gemm_last_mainloop()]
        nd = sum(int((outs[0][0] != o).sum()) for o, _ in outs[1:])
        nan = int(torch.isnan(outs[0][0].float()).sum())
        if outs[0][1] is not None:
            nd += sum(int((outs[0][1] != p).sum()) for _, p in outs[1:])
        ref = (A.float().t() if aks else A.float()) @ (Bm.float() if bks else Bm.float().t())
        if "bias" in ex: ref = ref + kw["bias"]
        if "resid" in ex: ref = ref + kw["resid"]
        if "mulact" in ex: ref = ref * kw["act_in"].float()
        if "gelugp" in ex: ref = torch.nn.functional.gelu(ref.to(BF).float())
        err = float((outs[0][0].float() - ref).abs().max())
        tol = 2 ** -6 * float(ref.abs().max()) + 2e-4 * math.sqrt(k)
        flag = "" if (nd == 0 and nan == 0 and err < tol) else "   <<<<<< BAD"
        bad += bool(flag)
        print(f"mode {mode:2d} {name:12s} {m}x{n}x{k}  loop {loop:6s}  differing elements over 5 reruns {nd:8d}  nan {nan}  max err vs fp32 {err:.3g} (tol {tol:.3g}){flag}")
    ops.gemm_set_mainloop(prev)
print("BAD" if bad else "all identical")
