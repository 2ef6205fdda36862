#!/usr/bin/env python
"""Round 6: randomised shapes through gemm_p16 (neko_gemm_set_mainloop(3)) against the fp32 product -- row / column tile counts, every
contraction length class (whole trips, trips + 4 / + 8 k-tiles), layouts, leading dimensions, split-K with uneven last slices, outputs."""
import math, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops
BF = torch.bfloat16; dev = "cuda"
random.seed(int(os.environ.get("SEED", "1")))
g = torch.Generator(device=dev).manual_seed(7)
prev = ops.gemm_set_mainloop(3)
bad = served = 0
N_CASES = int(os.environ.get("CASES", "150"))
for case in range(N_CASES):
    layout = random.choice(["nn", "nn", "nt", "nt", "tn"])
    M, N = 256 * random.randint(1, 6), 256 * random.randint(1, 6)
    kt = random.choice([12, 16, 20, 24, 28, 36, 40, 44, 48, 64, 72, 96]) if layout != "tn" else random.choice([4, 8, 12, 20, 36, 64])
    K = 32 * kt
    splitk = 1
    if random.random() < 0.3:
        splitk = random.choice([2, 3])
        K = K * splitk - (128 * random.randint(0, 1) if kt > (16 if layout != "tn" else 8) else 0)     # uneven last slice
    kps = 32 * kt if splitk > 1 else 0
    a_ks, b_ks = layout == "tn", layout in ("nn", "tn")
    pad_a, pad_b, pad_c = 8 * random.randint(0, 4), 8 * random.randint(0, 4), 8 * random.randint(0, 4)
    A = torch.randn((K, M + pad_a) if a_ks else (M, K + pad_a), device=dev, generator=g).to(BF)
    B = (torch.randn((K, N + pad_b) if b_ks else (N, K + pad_b), device=dev, generator=g) * 0.1).to(BF)
    Av = A[:, :M] if a_ks else A[:, :K]
    Bv = B[:, :N] if b_ks else B[:, :K]
    ref = (Av.float().t() if a_ks else Av.float()) @ (Bv.float() if b_ks else Bv.float().t())
    kind = random.choice(["f32", "bf16", "bias_f32", "bias_resid"]) if splitk == 1 and not a_ks else "f32"
    kw = dict(a_kstrided=a_ks, b_kstrided=b_ks, lda=A.stride(0), ldb=B.stride(0))
    if splitk > 1: kw.update(splitk=splitk, k_per_split=kps)
    want = ref
    if "bias" in kind:
        bias = torch.randn(N, device=dev, generator=g); kw["bias"] = bias; want = want + bias
    if "resid" in kind:
        resid = torch.randn(M, N, device=dev, generator=g); kw["resid"] = resid; want = want + resid
    if kind == "bf16":
        big = torch.full((M, N + pad_c), float("nan"), dtype=BF, device=dev)
        ops.gemm(Av, Bv, M, N, K, out_bf16=big[:, :N], ldcb=N + pad_c, **kw)
        out, rtol = big[:, :N].float(), 2 ** -7
    else:
        big = torch.full((M, N + pad_c), float("nan"), device=dev)
        ops.gemm(Av, Bv, M, N, K, out_f32=big[:, :N], ldcf=N + pad_c, **kw)
        out, rtol = big[:, :N], 2e-4
    loop = ops.gemm_last_mainloop()
    served += loop == 5
    err = (out - want).abs()
    tol = 2e-4 * math.sqrt(K) + rtol * want.abs()
    nb = int((err > tol).sum()) + int(torch.isnan(out).sum())
    leak = pad_c and not bool(torch.isnan(big[:, N:].float()).all())
    if nb or leak:
        bad += 1
        print(f"BAD case {case}: {layout} {M}x{N}x{K} splitk {splitk} kps {kps} {kind} lda {A.stride(0)} ldb {B.stride(0)} loop {ops.MAINLOOP_NAMES[loop]}: {nb} bad, max err {float(err.max()):.3g}, leak {leak}")
ops.gemm_set_mainloop(prev)
print(f"{N_CASES} cases, {served} served by gemm_p16, {bad} bad")
sys.exit(1 if bad else 0)
