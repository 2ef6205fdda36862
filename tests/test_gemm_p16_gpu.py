"""gemm_p16.hip (round 6): the two-waves-per-SIMD, role-alternating main loop of neko_gemm_bf16, through the C ABI, against the fp32
product of the same bf16 operands (the Conv1D / Linear products of gato/transformers/trajectory_gpt2.py:139-141,222,253,264-278 and
gato/policy/gato_policy.py:172) -- every operand layout it serves, every compiled epilogue, split-K slices, and the dispatch itself:
`neko_gemm_last_mainloop()` must say that THIS loop served the call (a silent fall-through to another loop would return the same
numbers).  Tolerances as tests/test_kernels_gpu.py: fp32 outputs 2e-4 * sqrt(K) of the row scale, bf16 outputs one bf16 ulp on top."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
P16, A16, B16, GLDS64, GLDS = 5, 1, 2, 3, 0


def rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def bf(x):
    return x.to(torch.bfloat16).to(DEV).contiguous()


def close(a, b, rtol, atol, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float(err.max()):.4g} "
                           f"(tol at worst {float(tol[bad].min()):.4g}), ref scale {float(b.abs().max()):.4g}")


@pytest.fixture
def p16():
    """every launch the two-waves-per-SIMD main loop (gemm_p16.hip) can serve goes to it"""
    from neko_amd import ops
    prev = ops.gemm_set_mainloop(3)
    yield ops
    ops.gemm_set_mainloop(prev)


def operands(M, N, K, layout, seed=None):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K if seed is None else seed)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.25)
    # rows individually recognisable: a permutation inside a 16-row block or a swapped k-chunk changes the result
    A = rb(A * (1.0 + 0.01 * torch.arange(M).float().unsqueeze(1) % 0.37))
    a_ks, b_ks = layout == "tn", layout in ("nn", "tn")
    return A, Bm, (bf(A.t()) if a_ks else bf(A)), (bf(Bm) if b_ks else bf(Bm.t())), a_ks, b_ks, g


@pytest.mark.parametrize("layout,K", [("nn", 384), ("nn", 768), ("nn", 3072), ("nt", 384), ("nt", 768), ("nt", 2304), ("tn", 128), ("tn", 512),
                                      ("tn", 4224), ("nn", 512), ("nn", 640), ("nn", 2048), ("nt", 512), ("nt", 1024), ("nt", 8192)])
@pytest.mark.parametrize("M,N", [(256, 256), (768, 512)])
def test_gemm_p16_main_loop_layouts(p16, layout, M, N, K):
    """One loop trip (12 k-tiles with a k-contiguous A operand, 4 with both operands k-strided: the surplus requests at the end re-fetch
    the last unit), two and many trips, and contraction ranges that leave 4 or 8 k-tiles behind whole trips (512 = 12 + 4, 640 = 12 + 8,
    1024 / 2048 / 8192: the 2048d geometry of configs[4]) -- the tail runs the first groups of the same loop body; forward (A k-contiguous x weights stored (in, out)), dgrad (both k-contiguous: B in two 64-k
    slots) and weight-gradient (both k-strided) layouts; fp32 and bf16 outputs; against the fp32 product and against the default loops."""
    A, Bm, A_dev, B_dev, a_ks, b_ks, _ = operands(M, N, K, layout)
    ref = A @ Bm
    out = torch.full((M, N), float("nan"), device=DEV)
    p16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=out)
    assert p16.gemm_last_mainloop() == P16, p16.MAINLOOP_NAMES[p16.gemm_last_mainloop()]
    close(out, ref, 2e-4, 2e-4 * math.sqrt(K), f"p16 {layout} {M}x{N}x{K}")
    out16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    p16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_bf16=out16)
    assert p16.gemm_last_mainloop() == P16
    close(out16, ref, 2 ** -7, 2e-4 * math.sqrt(K), f"p16 bf16 {layout} {M}x{N}x{K}")
    prev = p16.gemm_set_mainloop(0)
    try:
        other = torch.full((M, N), float("nan"), device=DEV)
        p16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=other)
        assert p16.gemm_last_mainloop() != P16
    finally:
        p16.gemm_set_mainloop(prev)
    close(out, other, 1e-5, 4e-5 * math.sqrt(K), f"p16 vs default loop {layout}")
    # run-to-run identical
    again = torch.full((M, N), float("nan"), device=DEV)
    p16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=again)
    assert torch.equal(out, again)


def test_gemm_p16_many_tiles_and_leading_dimensions(p16):
    """More tiles than CUs (two rounds), operands that are column / row slices of wider buffers (lda, ldb, ldc larger than the extents: the
    q / k / v slices of the fused qkv buffer, the padded LM-head table)."""
    M, N, K = 4096, 2304 + 256, 768
    g = torch.Generator().manual_seed(5)
    Abig = rb(torch.randn(M, K + 384, generator=g)); Bbig = rb(torch.randn(K, N + 512, generator=g) * 0.1)
    A, Bm = Abig[:, 128:128 + K], Bbig[:, 256:256 + N]
    ref = A @ Bm
    Ad, Bd = bf(Abig), bf(Bbig)
    outbig = torch.full((M, N + 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    p16.gemm(Ad[:, 128:], Bd[:, 256:], M, N, K, b_kstrided=True, lda=K + 384, ldb=N + 512, out_bf16=outbig[:, 128:], ldcb=N + 256)
    assert p16.gemm_last_mainloop() == P16
    close(outbig[:, 128:128 + N], ref, 2 ** -7, 2e-4 * math.sqrt(K), "p16 slices")
    assert torch.isnan(outbig[:, :128].float()).all() and torch.isnan(outbig[:, 128 + N:].float()).all(), "wrote outside its columns"


@pytest.mark.parametrize("epi", ["bias_bf16", "bias_resid", "bias_resid_drop", "bias_gelu_pre", "bias_gelu_factor", "gelubwd", "gelubwd_factor_colsum",
                                 "alpha_dev", "bias_f32", "wgrad_accumulate", "wgrad_alpha_accumulate", "wgrad_splitk", "fwd_splitk", "fwd_splitk_alpha"])
def test_gemm_p16_epilogues(p16, epi):
    """Every compiled epilogue behind gemm_p16.hip (one kernel per layout and feature set; the accumulators leave the AGPRs 32 rows at a
    time): fp32 tolerance against the reference product, the dropped elements equal to the default loop's, run-to-run bit identity."""
    M, N, K = 768, 768, 768
    g = torch.Generator().manual_seed(321)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.1)
    ref = A @ Bm
    A_dev = bf(A)
    fwd = dict(b_kstrided=True)
    B_fwd, B_dg = bf(Bm), bf(Bm.t())
    bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)

    def twice(fn):
        a = fn()
        assert p16.gemm_last_mainloop() == P16, f"{epi}: served by {p16.MAINLOOP_NAMES[p16.gemm_last_mainloop()]}"
        b = fn()
        for x, y in zip(a, b):
            assert torch.equal(x, y), f"{epi}: not run-to-run identical"
        return a

    if epi == "bias_bf16":
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), out_bf16=out, **fwd)
            return (out,)
        (out,) = twice(run)
        close(out, ref + bias, 2 ** -7, 2e-4 * math.sqrt(K), epi)
    elif epi == "bias_f32":
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), out_f32=out, **fwd)
            return (out,)
        (out,) = twice(run)
        close(out, ref + bias, 2e-4, 2e-4 * math.sqrt(K), epi)
    elif epi in ("bias_resid", "bias_resid_drop"):
        from neko_amd.ops import Drop
        drop = Drop(0.1, 0x1234567) if epi.endswith("drop") else None
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=out, drop=drop, **fwd)
            return (out,)
        (out,) = twice(run)
        if drop is None:
            close(out, ref + bias + resid, 2e-4, 2e-4 * math.sqrt(K), epi)
        else:       # the same site key on the default loop drops the same elements: compare with it
            prev = p16.gemm_set_mainloop(0)
            try:
                other = torch.full((M, N), float("nan"), device=DEV)
                p16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=other, drop=drop, **fwd)
            finally:
                p16.gemm_set_mainloop(prev)
            close(out, other.cpu(), 2e-4, 4e-4 * math.sqrt(K), epi)
            kept = ((out.cpu() - resid).abs() > 0).float().mean()
            assert 0.86 < float(kept) < 0.94, float(kept)
    elif epi in ("bias_gelu_pre", "bias_gelu_factor"):
        act = 1 if epi == "bias_gelu_pre" else 3
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            second = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), act=act, pre_out=second, out_bf16=out, **fwd)
            return out, second
        out, second = twice(run)
        x = (ref + bias).to(torch.bfloat16).float()
        xd = x.double()
        err = (out.float().cpu() - torch.nn.functional.gelu(x)).abs()
        assert float((err > 2 ** -7 * x.abs() + 2e-2).float().mean()) < 1e-3, "gelu values"
        if act == 1:
            assert float((second.float().cpu() != x).float().mean()) < 2e-2, "stored pre-activation"
        else:
            gp = 0.5 * (1 + torch.erf(xd / math.sqrt(2))) + xd * torch.exp(-0.5 * xd * xd) / math.sqrt(2 * math.pi)
            errf = (second.float().cpu() - gp.float()).abs()
            assert float((errf > 2 ** -7 + 2e-2).float().mean()) < 1e-3, "gelu' factor"
    elif epi == "gelubwd":
        pre = rb(torch.randn(M, N, generator=g) * 1.5)
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            p16.gemm(A_dev, B_dg, M, N, K, act=2, act_in=bf(pre), out_bf16=out)
            return (out,)
        (out,) = twice(run)
        gprime = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
        close(out, ref * gprime, 2 ** -7, 2e-4 * math.sqrt(K), epi)
    elif epi == "gelubwd_factor_colsum":
        fac = rb(torch.rand(M, N, generator=g) * 1.2 - 0.1)
        base = torch.randn(N, generator=g)
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            cs = base.clone().to(DEV)
            p16.gemm_dgrad_gelu_colsum(A_dev, B_dg, M, N, K, bf(fac), out, cs, act_in_is_factor=True)
            return out, cs
        out, cs = twice(run)
        close(out, ref * fac, 2 ** -7, 2e-4 * math.sqrt(K), epi)
        want = base + (ref * fac).sum(0)
        assert float((cs.cpu() - want).norm() / want.norm()) < 2e-3
    elif epi == "alpha_dev":
        alpha_dev = torch.tensor([0.5], device=DEV)
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, alpha=2.0, alpha_dev=alpha_dev, out_f32=out, **fwd)
            return (out,)
        (out,) = twice(run)
        close(out, ref, 2e-4, 2e-4 * math.sqrt(K), epi)
    elif epi in ("wgrad_accumulate", "wgrad_alpha_accumulate"):
        At, Bk = bf(A.t()), bf(Bm)          # tn: both operands k-strided
        base = torch.randn(M, N, generator=g)
        alpha_dev = torch.tensor([0.25], device=DEV) if "alpha" in epi else None
        out = base.clone().to(DEV)
        p16.gemm(At, Bk, M, N, K, a_kstrided=True, b_kstrided=True, out_f32=out, accumulate=True,
                 **(dict(alpha=4.0, alpha_dev=alpha_dev) if alpha_dev is not None else {}))
        assert p16.gemm_last_mainloop() == P16
        close(out, base + ref, 2e-4, 2e-4 * math.sqrt(K), epi)
    elif epi == "wgrad_splitk":
        At, Bk = bf(A.t()), bf(Bm)
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            p16.gemm(At, Bk, M, N, K, a_kstrided=True, b_kstrided=True, out_f32=out, splitk=3, k_per_split=256)
            return (out,)
        (out,) = twice(run)
        close(out, ref, 2e-4, 2e-4 * math.sqrt(K), epi)
    elif epi == "fwd_splitk_alpha":          # the LM-head dH form: split-K slices to the workspace, each scaled by the device-side loss scale
        alpha_dev = torch.tensor([0.25], device=DEV)
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, alpha=2.0, alpha_dev=alpha_dev, out_f32=out, splitk=2, k_per_split=384, **fwd)
            return (out,)
        (out,) = twice(run)
        close(out, 0.5 * ref, 2e-4, 2e-4 * math.sqrt(K), epi)
    else:
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            p16.gemm(A_dev, B_fwd, M, N, K, out_f32=out, splitk=2, k_per_split=384, **fwd)
            return (out,)
        (out,) = twice(run)
        close(out, ref, 2e-4, 2e-4 * math.sqrt(K), epi)


def test_gemm_p16_declines_what_it_cannot_serve(p16):
    """Edge tiles, contraction ranges below one loop trip or not a multiple of 128, and the A-k-strided x B-k-contiguous layout stay with
    the other loops (same results)."""
    for (M, N, K, layout) in [(300, 256, 768, "nn"), (256, 256, 256, "nn"), (256, 256, 448, "nt"), (256, 256, 768, "tt")]:
        g = torch.Generator().manual_seed(K + M)
        A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.25)
        a_ks, b_ks = layout in ("tn", "tt"), layout in ("nn", "tn")
        Mp = (M + 7) // 8 * 8
        if a_ks and Mp != M:
            continue
        out = torch.full((M, N), float("nan"), device=DEV)
        p16.gemm(bf(A.t()) if a_ks else bf(A), bf(Bm) if b_ks else bf(Bm.t()), M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=out)
        assert p16.gemm_last_mainloop() != P16
        close(out, A @ Bm, 2e-4, 2e-4 * math.sqrt(K), f"declined {layout} {M}x{N}x{K}")


@pytest.mark.parametrize("mode", [3, 2, 1, 0])
def test_gemm_fp32_and_bf16_outputs_every_loop_at_step_size(mode):
    """Round 6 regression: the fast epilogue's fp32 outputs are 16-byte stores issued from inline asm (streaming policy); a store of more
    than 64 bits reads its data registers for a few cycles after it issued, and without the wait states behind it the NEXT step's value
    replaced the first dword -- only in launches with several row passes per wave and many tiles, which the small cases above never
    showed.  16384 x 768 outputs (192-384 tiles), every main loop, fp32 (+ residual) and bf16 outputs: against the fp32 product of the
    same operands, and bit-identical over reruns."""
    from neko_amd import ops
    M, N, K = 16384, 768, 768
    g = torch.Generator().manual_seed(17)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.05)
    bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
    ref = A @ Bm + bias
    Ad, Bd, biasd, residd = bf(A), bf(Bm), bias.to(DEV), resid.to(DEV)
    prev = ops.gemm_set_mainloop(mode)
    try:
        for what in ("f32", "f32+resid", "bf16"):
            outs = []
            for _ in range(3):
                if what == "bf16":
                    o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
                    ops.gemm(Ad, Bd, M, N, K, b_kstrided=True, bias=biasd, out_bf16=o)
                else:
                    o = torch.full((M, N), float("nan"), device=DEV)
                    ops.gemm(Ad, Bd, M, N, K, b_kstrided=True, bias=biasd, out_f32=o, resid=residd if what == "f32+resid" else None)
                outs.append(o)
            want = ref + (resid if what == "f32+resid" else 0)
            close(outs[0], want, 2 ** -7 if what == "bf16" else 2e-4, 2e-4 * math.sqrt(K), f"mode {mode} {what}")
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"mode {mode} {what}: not run-to-run identical"
    finally:
        ops.gemm_set_mainloop(prev)


def test_gemm_p16_random_shapes():
    """tools/probe/p16_fuzz.py: random tile counts, contraction lengths of every class (whole trips, + 4, + 8 k-tiles), layouts, leading
    dimensions, split-K with an uneven last slice, output kinds -- against the fp32 product, nothing written outside the output columns."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probe", "p16_fuzz.py")], env=dict(os.environ, CASES="120", SEED="11"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith("0 bad") and int(last.split("cases,")[1].split("served")[0]) >= 60, last
