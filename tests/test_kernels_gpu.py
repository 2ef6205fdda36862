"""Kernel-level parity: every HIP entry point (through the C ABI) against the CPU oracle / a plain
torch fp32 reference on the same seeded inputs.  Inputs of bf16 kernels are pre-rounded to bf16 so
the only differences are fp32 summation order and the documented bf16 output rounding.

Tolerances (stated per test): fp32 outputs of bf16-operand GEMMs rtol 2e-4 of the row scale;
bf16 outputs 2^-8 relative (one bf16 ulp) + the same; fp32 kernels (LayerNorm, CE, AdamW) 1e-5.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


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


@pytest.fixture(scope="module")
def ops():
    from neko_amd import ops as o
    return o


# ----------------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("safe", [0, 1, 2])   # 0 direct-to-LDS fast path (when K%64==0), 1 transposing stores, 2 register-staged
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 264), (77, 520, 1000), (512, 2304, 768), (333, 776, 1024),
                                   (1024, 52305, 128)])
def test_gemm_layouts(ops, layout, M, N, K, safe):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = rb(torch.randn(M, K, generator=g))          # logical [M,K]
    Bm = rb(torch.randn(K, N, generator=g) * 0.5)   # logical [K,N]  (asymmetric: transposes are visible)
    ref = A @ Bm
    if layout == "tn" and M % 8:
        M8 = (M + 7) // 8 * 8
        A = torch.cat([A, torch.zeros(M8 - M, K)]); ref = A @ Bm; M = M8
    if layout in ("nn", "tn") and N % 8:
        pytest.skip("contiguous extents must be multiples of 8")
    a_ks = layout == "tn"
    b_ks = layout in ("nn", "tn")
    A_dev = bf(A.t()) if a_ks else bf(A)
    B_dev = bf(Bm) if b_ks else bf(Bm.t())
    if (not a_ks or not b_ks) and K % 8:
        pytest.skip("K must be a multiple of 8 for k-contiguous operands")
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=out, safe_transpose=safe)
    torch.cuda.synchronize()
    close(out, ref, 2e-4, 2e-4 * math.sqrt(K), f"gemm {layout} safe={safe}")


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 448, 2112])
def test_gemm_pipelined_loop_short_and_long_k(ops, layout, K):
    """Output large enough for the 256x256 tile (4-stage ring: the software-pipelined main loop for k-contiguous operands,
    the plain loop for the doubly k-strided form) at contractions of 2, 4, 6, 8, 14 and 66 k-tiles: prologue shorter than
    the ring, main part absent / present, drain of every length."""
    g = torch.Generator().manual_seed(K)
    M, N = 2048, 2304
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.2)
    ref = A @ Bm
    a_ks, b_ks = layout in ("tn", "tt"), layout in ("nn", "tn")       # "tt": A k-strided, B k-contiguous (the fourth generated stream)
    A_dev = bf(A.t()) if a_ks else bf(A)
    B_dev = bf(Bm) if b_ks else bf(Bm.t())
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=out)
    close(out, ref, 2e-4, 2e-4 * math.sqrt(K), f"gemm 256-tile {layout} K={K}")
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_bf16=out16)
    close(out16, ref, 2 ** -7, 2e-4 * math.sqrt(K), f"gemm 256-tile bf16 {layout} K={K}")


@pytest.fixture
def a16(ops):
    """every launch the hand-placed long-contraction main loop (gemm_a16.hip) can serve goes to it"""
    prev = ops.gemm_set_mainloop(1)
    yield ops
    ops.gemm_set_mainloop(prev)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 384), (1024, 512, 3072), (256, 1024, 8192)])
def test_gemm_a16_main_loop_layouts(a16, layout, M, N, K):
    """gemm_a16.hip (4 waves x 128 x 128 on 16x16x32 MFMAs, accumulators in AGPRs, hand-placed instruction stream): one loop
    trip (4 k-tiles: the surplus DMA requests at the end re-fetch the last tile), 3, 96 and 256 trips, every operand layout,
    fp32 and bf16 outputs, against an fp32 product of the same bf16 operands.  B is asymmetric and scaled differently from A,
    so a transposed or permuted fragment shows."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.25)
    # make rows / columns individually recognisable: a permutation inside a 16-row block or a swapped k-chunk changes the result
    A = rb(A * (1.0 + 0.01 * torch.arange(M).float().unsqueeze(1) % 0.37))
    ref = A @ Bm
    a_ks, b_ks = layout == "tn", layout in ("nn", "tn")
    A_dev = bf(A.t()) if a_ks else bf(A)
    B_dev = bf(Bm) if b_ks else bf(Bm.t())
    out = torch.full((M, N), float("nan"), device=DEV)
    a16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=out)
    assert a16.gemm_last_mainloop() == 1, a16.MAINLOOP_NAMES[a16.gemm_last_mainloop()]      # served by gemm_a16, not by a fall-through
    close(out, ref, 2e-4, 2e-4 * math.sqrt(K), f"a16 {layout} {M}x{N}x{K}")
    out16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    a16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_bf16=out16)
    close(out16, ref, 2 ** -7, 2e-4 * math.sqrt(K), f"a16 bf16 {layout} {M}x{N}x{K}")
    # same launch through the default loop: both are fp32 sums of the same products
    prev = a16.gemm_set_mainloop(0)
    try:
        other = torch.full((M, N), float("nan"), device=DEV)
        a16.gemm(A_dev, B_dev, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks, out_f32=other)
        assert a16.gemm_last_mainloop() in (0, 3)
    finally:
        a16.gemm_set_mainloop(prev)
    close(out, other, 1e-5, 4e-5 * math.sqrt(K), f"a16 vs default loop {layout}")


@pytest.mark.parametrize("epi", ["bias_resid", "bias_gelu_factor", "gelubwd_factor", "alpha_accumulate", "splitk"])
def test_gemm_a16_epilogues(a16, epi):
    """The compiled epilogues behind the hand-placed loop (the accumulators leave the AGPRs 32 rows at a time): bias + residual
    (forward MLP projection), bias + GELU with the stored gelu' factor, the dgrad that multiplies by it, alpha (device scalar) +
    accumulate (LM-head dW), and split-K slices reduced in fixed order -- and a second call reproduces the first bit for bit."""
    M, N, K = 768, 512, 1536
    g = torch.Generator().manual_seed(99)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.1)
    ref = A @ Bm
    A_dev, B_dev = bf(A), bf(Bm)                     # nn
    kw = dict(b_kstrided=True)
    if epi == "bias_resid":
        bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
        outs = []
        for _ in range(2):
            out = torch.full((M, N), float("nan"), device=DEV)
            a16.gemm(A_dev, B_dev, M, N, K, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=out, **kw)
            outs.append(out)
        close(outs[0], ref + bias + resid, 2e-4, 2e-4 * math.sqrt(K), epi)
        assert torch.equal(outs[0], outs[1])
    elif epi == "bias_gelu_factor":
        bias = torch.randn(N, generator=g)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        fac = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        a16.gemm(A_dev, B_dev, M, N, K, bias=bias.to(DEV), act=3, pre_out=fac, out_bf16=out, **kw)
        x = (ref + bias).to(torch.bfloat16).float()
        xd = x.double()
        gp = 0.5 * (1 + torch.erf(xd / math.sqrt(2))) + xd * torch.exp(-0.5 * xd * xd) / math.sqrt(2 * math.pi)
        err = (out.float().cpu() - torch.nn.functional.gelu(x)).abs()
        assert float((err > 2 ** -7 * x.abs() + 2e-2).float().mean()) < 1e-3, "gelu values"       # pre-activation rounding flips
        errf = (fac.float().cpu() - gp.float()).abs()
        assert float((errf > 2 ** -7 + 2e-2).float().mean()) < 1e-3, "gelu' factor"
    elif epi == "gelubwd_factor":
        fac = rb(torch.rand(M, N, generator=g) * 1.2 - 0.1)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        a16.gemm(A_dev, B_dev, M, N, K, act=4, act_in=bf(fac), out_bf16=out, **kw)
        close(out, ref * fac, 2 ** -7, 2e-4 * math.sqrt(K), epi)
    elif epi == "alpha_accumulate":
        base = torch.randn(M, N, generator=g)
        out = base.clone().to(DEV)
        alpha_dev = torch.tensor([0.5], device=DEV)
        a16.gemm(A_dev, B_dev, M, N, K, alpha=2.0, alpha_dev=alpha_dev, out_f32=out, accumulate=True, **kw)
        close(out, base + ref, 2e-4, 2e-4 * math.sqrt(K), epi)
    else:
        outs = []
        for _ in range(2):
            out = torch.full((M, N), float("nan"), device=DEV)
            a16.gemm(A_dev, B_dev, M, N, K, out_f32=out, splitk=3, k_per_split=512, **kw)
            outs.append(out)
        close(outs[0], ref, 2e-4, 2e-4 * math.sqrt(K), epi)
        assert torch.equal(outs[0], outs[1])
    assert a16.gemm_last_mainloop() == 1, f"{epi}: served by {a16.MAINLOOP_NAMES[a16.gemm_last_mainloop()]}"


@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K,splitk", [(11000, 2304, 128, 1), (11000, 2304, 192, 1), (11012, 2176, 768, 1), (11008, 2304, 320, 1), (512, 512, 5120, 40)])
def test_gemm_glds_whole_line_slots_for_a_k_contiguous_operand(ops, layout, M, N, K, splitk):
    """gemm_glds64_kernel (round 5): the 8-wave 256 x 256 loop with its k-contiguous A operand in three 64-k slots of whole 128-B lines
    (XOR-swizzled 16-B chunks, pieces requested two slots ahead on even k-tiles).  Shapes that take it with the hand-placed loops off:
    the shortest contraction (two slots, no steady-state trip), one and several trips, edge tiles in M and N (clamped DMA rows / columns),
    B k-contiguous and k-strided, split-K slices of two slots each -- against the fp32 product of the same bf16 operands.  (Round 6: the
    row counts are 43 tiles of 256: with fewer than 384 tiles of 256 x 256 the launcher picks the 256 x 128 configuration for these wide
    outputs and the round-5 shapes never reached this kernel -- `neko_gemm_last_mainloop()` now says which loop served the call.)"""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.25)
    A = rb(A * (1.0 + 0.01 * torch.arange(M).float().unsqueeze(1) % 0.37))      # rows individually recognisable
    ref = A @ Bm
    b_ks = layout == "nn"
    A_dev = bf(A)
    B_dev = bf(Bm) if b_ks else bf(Bm.t())
    prev = ops.gemm_set_mainloop(0)
    try:
        if splitk > 1:
            outs = []
            for _ in range(2):
                out = torch.full((M, N), float("nan"), device=DEV)
                ops.gemm(A_dev, B_dev, M, N, K, b_kstrided=b_ks, out_f32=out, splitk=splitk, k_per_split=K // splitk)
                outs.append(out)
            close(outs[0], ref, 2e-4, 2e-4 * math.sqrt(K), f"kc64 split-K {layout}")
            assert torch.equal(outs[0], outs[1])
        else:
            out16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm(A_dev, B_dev, M, N, K, b_kstrided=b_ks, out_bf16=out16)
            close(out16, ref, 2 ** -7, 2e-4 * math.sqrt(K), f"kc64 bf16 {layout} {M}x{N}x{K}")
            out = torch.full((M, N), float("nan"), device=DEV)
            bias = torch.randn(N, generator=g)
            ops.gemm(A_dev, B_dev, M, N, K, b_kstrided=b_ks, bias=bias.to(DEV), out_f32=out)
            close(out, ref + bias, 2e-4, 2e-4 * math.sqrt(K), f"kc64 f32 + bias {layout} {M}x{N}x{K}")
        assert ops.gemm_last_mainloop() == 3, f"served by {ops.MAINLOOP_NAMES[ops.gemm_last_mainloop()]}, not by gemm_glds64_kernel"
    finally:
        ops.gemm_set_mainloop(prev)


@pytest.fixture
def b16(ops):
    """every launch the two-workgroups-per-CU main loop (gemm_b16.hip) can serve goes to it"""
    prev = ops.gemm_set_mainloop(2)
    yield ops
    ops.gemm_set_mainloop(prev)


@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K", [(128, 256, 384), (640, 768, 768), (256, 512, 3072), (1152, 1024, 2304)])
def test_gemm_b16_main_loop_layouts(b16, layout, M, N, K):
    """gemm_b16.hip (128 x 256 per workgroup, 4 waves x 64 x 128 on 16x16x32 MFMAs, 128 accumulators in AGPRs, two workgroups per CU,
    hand-placed 12-k-tile loop body): one loop trip, 2, 8 and 6 trips, B k-contiguous (dgrad) and k-strided (forward), bf16 and fp32
    outputs where a kernel exists, against an fp32 product of the same bf16 operands and against the default loop."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.25)
    A = rb(A * (1.0 + 0.01 * torch.arange(M).float().unsqueeze(1) % 0.37))      # rows individually recognisable
    ref = A @ Bm
    b_ks = layout == "nn"
    A_dev = bf(A)
    B_dev = bf(Bm) if b_ks else bf(Bm.t())
    out16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    b16.gemm(A_dev, B_dev, M, N, K, b_kstrided=b_ks, out_bf16=out16)
    assert b16.gemm_last_mainloop() == 2, b16.MAINLOOP_NAMES[b16.gemm_last_mainloop()]
    close(out16, ref, 2 ** -7, 2e-4 * math.sqrt(K), f"b16 bf16 {layout} {M}x{N}x{K}")
    prev = b16.gemm_set_mainloop(0)
    try:
        other = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        b16.gemm(A_dev, B_dev, M, N, K, b_kstrided=b_ks, out_bf16=other)
    finally:
        b16.gemm_set_mainloop(prev)
    # both are fp32 sums of the same products rounded once to bf16: equal up to the rare rounding flip
    assert float((out16.float() - other.float()).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    assert float((out16 != other).float().mean()) < 2e-2
    if not b_ks:                                      # the fp32-out kernel exists for the dgrad layout
        out = torch.full((M, N), float("nan"), device=DEV)
        b16.gemm(A_dev, B_dev, M, N, K, b_kstrided=False, out_f32=out)
        assert b16.gemm_last_mainloop() == 2
        close(out, ref, 2e-4, 2e-4 * math.sqrt(K), f"b16 f32 {layout} {M}x{N}x{K}")


@pytest.mark.parametrize("epi", ["bias_bf16", "bias_resid", "bias_resid_drop", "bias_gelu_pre", "bias_gelu_factor", "gelubwd", "gelubwd_factor_colsum"])
def test_gemm_b16_epilogues(b16, epi):
    """Every compiled epilogue behind gemm_b16.hip (one kernel per feature set; the accumulators leave the AGPRs 32 rows at a time with
    at most 8 of a pass's 16 steps of epilogue inputs in flight) against the same call on the default loops -- bit-identical where the
    epilogue arithmetic is elementwise on identical fp32 sums is not guaranteed (different summation order), so: fp32 tolerance against
    the reference product, and run-to-run bit identity."""
    M, N, K = 640, 768, 768
    g = torch.Generator().manual_seed(123)
    A = rb(torch.randn(M, K, generator=g)); Bm = rb(torch.randn(K, N, generator=g) * 0.1)
    ref = A @ Bm
    A_dev = bf(A)
    fwd = dict(b_kstrided=True)
    B_fwd, B_dg = bf(Bm), bf(Bm.t())
    bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)

    def twice(fn):
        a = fn()
        assert b16.gemm_last_mainloop() == 2, f"{epi}: served by {b16.MAINLOOP_NAMES[b16.gemm_last_mainloop()]}"
        b = fn()
        for x, y in zip(a, b):
            assert torch.equal(x, y), f"{epi}: not run-to-run identical"
        return a

    if epi == "bias_bf16":
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            b16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), out_bf16=out, **fwd)
            return (out,)
        (out,) = twice(run)
        close(out, ref + bias, 2 ** -7, 2e-4 * math.sqrt(K), epi)
    elif epi in ("bias_resid", "bias_resid_drop"):
        from neko_amd.ops import Drop
        drop = Drop(0.1, 0x1234567) if epi.endswith("drop") else None
        def run():
            out = torch.full((M, N), float("nan"), device=DEV)
            b16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=out, drop=drop, **fwd)
            return (out,)
        (out,) = twice(run)
        if drop is None:
            close(out, ref + bias + resid, 2e-4, 2e-4 * math.sqrt(K), epi)
        else:       # the same site key on the default loop drops the same elements: compare with it
            prev = b16.gemm_set_mainloop(0)
            try:
                other = torch.full((M, N), float("nan"), device=DEV)
                b16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=other, drop=drop, **fwd)
            finally:
                b16.gemm_set_mainloop(prev)
            close(out, other.cpu(), 2e-4, 4e-4 * math.sqrt(K), epi)
            kept = ((out.cpu() - resid).abs() > 0).float().mean()
            assert 0.86 < float(kept) < 0.94, float(kept)
    elif epi in ("bias_gelu_pre", "bias_gelu_factor"):
        act = 1 if epi == "bias_gelu_pre" else 3
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            second = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            b16.gemm(A_dev, B_fwd, M, N, K, bias=bias.to(DEV), act=act, pre_out=second, out_bf16=out, **fwd)
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
            b16.gemm(A_dev, B_dg, M, N, K, act=2, act_in=bf(pre), out_bf16=out)
            return (out,)
        (out,) = twice(run)
        gprime = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
        close(out, ref * gprime, 2 ** -7, 2e-4 * math.sqrt(K), epi)
    else:
        fac = rb(torch.rand(M, N, generator=g) * 1.2 - 0.1)
        base = torch.randn(N, generator=g)
        def run():
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            cs = base.clone().to(DEV)
            b16.gemm_dgrad_gelu_colsum(A_dev, B_dg, M, N, K, bf(fac), out, cs, act_in_is_factor=True)
            return out, cs
        out, cs = twice(run)
        close(out, ref * fac, 2 ** -7, 2e-4 * math.sqrt(K), epi)
        want = base + (ref * fac).sum(0)
        assert float((cs.cpu() - want).norm() / want.norm()) < 2e-3


def test_gemm_many_tiles_fast_epilogues(ops):
    """A shape with several hundred 256x256 / 128x128 tiles (more than one round of the chip) through the specialised
    epilogues: forward-like (bias + bf16 / GELU + pre-activation / bias + residual f32) and dgrad-like (both operands
    k-contiguous) launches against an fp32 matmul of the same bf16 inputs."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 8192 + 256, 2304, 192                  # 33 x 9 = 297 tiles, 6 k-tiles each
    A = rb(torch.randn(M, K, generator=g)); W = rb(torch.randn(K, N, generator=g) * 0.1)
    bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
    Ad, Wd = bf(A), bf(W)
    ref = (Ad.float() @ Wd.float()).cpu() + bias
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(Ad, Wd, M, N, K, b_kstrided=True, bias=bias.to(DEV), out_bf16=out16)
    close(out16, ref, 2 ** -7, 2e-3, "many-tiles bias+bf16")
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(Ad, Wd, M, N, K, b_kstrided=True, bias=bias.to(DEV), act=1, pre_out=pre, out_bf16=out16)
    close(pre, ref, 2 ** -7, 2e-3, "many-tiles pre")
    close(out16, torch.nn.functional.gelu(pre.float().cpu()), 2 ** -7, 2e-3, "many-tiles gelu")
    out32 = torch.empty(M, N, device=DEV)
    ops.gemm(Ad, Wd, M, N, K, b_kstrided=True, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=out32)
    close(out32, ref + resid, 1e-4, 2e-3, "many-tiles bias+resid")
    Wt = bf(W.t().contiguous())                      # [N, K]: both operands k-contiguous (dgrad form)
    ops.gemm(Ad, Wt, M, N, K, out_bf16=out16)
    close(out16, ref - bias, 2 ** -7, 2e-3, "many-tiles NT")
    # run-to-run identical
    o2 = torch.empty_like(out16)
    ops.gemm(Ad, Wt, M, N, K, out_bf16=o2)
    assert torch.equal(o2, out16)


def test_gemm_epilogues(ops):
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 256, 192
    A, W = rb(torch.randn(M, K, generator=g)), rb(torch.randn(K, N, generator=g) * 0.1)
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Wd = bf(A), bf(W)
    # bias + GELU (+ pre-activation store), bf16 out
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    h = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(Ad, Wd, M, N, K, b_kstrided=True, bias=bias.to(DEV), act=1, pre_out=pre, out_bf16=h)
    pre_ref = rb(A @ W + bias)
    close(pre, pre_ref, 2 ** -8, 1e-3, "pre")
    close(h, torch.nn.functional.gelu(pre.float().cpu()), 2 ** -8, 1e-3, "gelu")
    # bias + residual, f32 out
    out = torch.empty(M, N, device=DEV)
    ops.gemm(Ad, Wd, M, N, K, b_kstrided=True, bias=bias.to(DEV), resid=resid.to(DEV), out_f32=out)
    close(out, A @ W + bias + resid, 1e-4, 1e-3, "bias+resid")
    # accumulate + alpha + device alpha
    acc0 = torch.randn(M, N, generator=g)
    out = acc0.clone().to(DEV)
    ad = torch.tensor([0.25], device=DEV)
    ops.gemm(Ad, Wd, M, N, K, b_kstrided=True, alpha=2.0, alpha_dev=ad, out_f32=out, accumulate=True)
    close(out, acc0 + 0.5 * (A @ W), 1e-4, 1e-3, "accumulate/alpha")
    # gelu backward epilogue: dY @ W2^T * gelu'(pre)
    dY = rb(torch.randn(M, K, generator=g))
    W2 = rb(torch.randn(N, K, generator=g) * 0.1)         # (in=N, out=K) Conv1D layout -> dgrad uses it k-contiguous
    dpre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(bf(dY), bf(W2), M, N, K, act=2, act_in=pre, out_bf16=dpre)
    x = pre.float().cpu()
    gprime = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    close(dpre, (dY @ W2.t()) * gprime, 2 ** -8, 2e-3, "gelu bwd")
    # split-K atomics (wgrad form)
    Mk = 4096
    X, dYk = rb(torch.randn(Mk, 64, generator=g)), rb(torch.randn(Mk, 136, generator=g))
    dW = torch.zeros(64, 136, device=DEV)
    sk, kps = ops.pick_splitk(64, 136, Mk)
    assert sk > 1
    ops.gemm(bf(X), bf(dYk), 64, 136, Mk, a_kstrided=True, b_kstrided=True, out_f32=dW, splitk=sk, k_per_split=kps)
    close(dW, X.t() @ dYk, 1e-4, 2e-2, "split-K wgrad")


@pytest.mark.parametrize("M,N,K", [(1024, 3072, 768), (8192, 768, 192), (300, 256, 192), (512, 3072 + 8, 64)])
def test_gemm_dgrad_gelu_colsum(ops, M, N, K):
    """d_pre = (dY @ W^T) * gelu'(pre) with the bias gradient sum_rows d_pre from the same launch
    (neko_gemm_dgrad_gelu_colsum; trajectory_gpt2.py:266,274 backward): full tiles fold the column sums into the epilogue
    (first two shapes), ragged M / N take the stand-alone pass over the stored result (last two); `+=` semantics and
    run-to-run bit-identity either way."""
    g = torch.Generator().manual_seed(M + N)
    dY, W = rb(torch.randn(M, K, generator=g)), rb(torch.randn(N, K, generator=g) * 0.1)
    pre = rb(torch.randn(M, N, generator=g) * 1.5)
    x = pre
    gprime = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    ref = (dY @ W.t()) * gprime
    base = torch.randn(N, generator=g)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    cs = base.clone().to(DEV)
    ops.gemm_dgrad_gelu_colsum(bf(dY), bf(W), M, N, K, bf(pre), out, cs)
    close(out, ref, 2 ** -8, 2e-3, "d_pre")
    want = base + ref.sum(0)
    err = float((cs.cpu() - want).norm() / want.norm())
    assert err < 2e-3, err
    worst = float((cs.cpu() - want).abs().max() / ref.abs().sum(0).max())
    assert worst < 2e-3, worst
    out2 = torch.empty_like(out)
    cs2 = base.clone().to(DEV)
    ops.gemm_dgrad_gelu_colsum(bf(dY), bf(W), M, N, K, bf(pre), out2, cs2)
    assert torch.equal(out, out2)
    if M % 256 == 0 and N % 256 == 0:        # folded: band partials added in a fixed order
        assert torch.equal(cs, cs2)
    else:                                    # stand-alone column-sum pass: fp32 atomics across row blocks
        assert float((cs - cs2).abs().max()) <= 1e-4 * float(cs.abs().max())
    # the plain launch stores the same d_pre
    out3 = torch.empty_like(out)
    ops.gemm(bf(dY), bf(W), M, N, K, act=2, act_in=bf(pre), out_bf16=out3)
    assert torch.equal(out, out3)


@pytest.mark.parametrize("M,N,K", [(1024, 3072, 768), (300, 256, 192)])
def test_gemm_gelu_factor_epilogues(ops, M, N, K):
    """ABI v14: the forward epilogue act = 3 leaves gelu'(pre) (bf16) in its second output, the backward epilogue act = 4
    multiplies by that stored factor (trajectory_gpt2.py:266,274 forward / backward of h = gelu(c_fc x)): interior
    (compiled fast epilogue) and ragged (generic epilogue) tiles, both against torch's erf GELU."""
    g = torch.Generator().manual_seed(M + 7 * N)
    A, W = rb(torch.randn(M, K, generator=g)), rb(torch.randn(K, N, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    fac = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    h = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(bf(A), bf(W), M, N, K, b_kstrided=True, bias=bias.to(DEV), act=3, pre_out=fac, out_bf16=h)
    x = rb(A @ W + bias)                      # the pre-activation is rounded to bf16 first (as with act = 1)
    gprime = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    # the bf16 rounding of the pre-activation may differ by one ulp from the host's: compare through act = 1's stored value
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    h1 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(bf(A), bf(W), M, N, K, b_kstrided=True, bias=bias.to(DEV), act=1, pre_out=pre, out_bf16=h1)
    assert torch.equal(h, h1), "act = 3 must produce the same activation as act = 1"
    xd = pre.float().cpu()
    gprime = 0.5 * (1 + torch.erf(xd / math.sqrt(2))) + xd * torch.exp(-0.5 * xd * xd) / math.sqrt(2 * math.pi)
    close(fac, gprime, 2 ** -8, 1e-3, "stored gelu'")
    dY = rb(torch.randn(M, K, generator=g))
    W2 = rb(torch.randn(N, K, generator=g) * 0.1)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    base = torch.randn(N, generator=g)
    cs = base.clone().to(DEV)
    ops.gemm_dgrad_gelu_colsum(bf(dY), bf(W2), M, N, K, fac, out, cs, act_in_is_factor=True)
    ref = (dY @ W2.t()) * fac.float().cpu()
    close(out, ref, 2 ** -8, 2e-3, "d_pre from the stored factor")
    close(out, (dY @ W2.t()) * gprime, 2 ** -7, 4e-3, "d_pre vs exact gelu'")
    want = base + ref.sum(0)
    assert float((cs.cpu() - want).norm() / want.norm()) < 2e-3
    out2 = torch.empty_like(out)
    ops.gemm(bf(dY), bf(W2), M, N, K, act=4, act_in=fac, out_bf16=out2)
    assert torch.equal(out, out2)
    # act = 4 (stored bf16 factor) against act = 2 (gelu' evaluated in fp32 from the bf16 pre-activation) on the SAME inputs:
    # the stored factor costs exactly one more bf16 rounding per element (2^-9 relative on top of the output's own rounding)
    out_a2 = torch.empty_like(out)
    ops.gemm(bf(dY), bf(W2), M, N, K, act=2, act_in=pre, out_bf16=out_a2)
    a4, a2 = out.float().cpu(), out_a2.float().cpu()
    close(out, a2, 2 ** -7, 1e-3, "act = 4 vs act = 2")
    rel = float((a4 - a2).norm() / a2.norm())
    assert rel < 2 ** -8, rel          # one more bf16 rounding per element (2^-9 relative, unbiased); profiles/r04_gelu_factor_bound.txt


# ----------------------------------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dy_bf16", [False, True])
@pytest.mark.parametrize("M,d", [(5, 64), (333, 128), (1000, 768), (64, 2048)])
def test_layernorm_fwd_bwd(ops, M, d, dy_bf16):
    g = torch.Generator().manual_seed(d + M)
    x = torch.randn(M, d, generator=g) * 2 + 0.3
    w, b = torch.randn(d, generator=g), torch.randn(d, generator=g)
    dy, gin = torch.randn(M, d, generator=g), torch.randn(M, d, generator=g)
    if dy_bf16:
        dy = rb(dy)                                     # the kernel reads the same bf16 values the reference math gets
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5)
    y.backward(dy)
    xd = x.to(DEV)
    y16 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV); y32 = torch.empty(M, d, device=DEV)
    mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    ops.layernorm_fwd(xd, w.to(DEV), b.to(DEV), y16=y16, y32=y32, mean=mean, rstd=rstd)
    close(y32, y, 1e-5, 1e-5, "ln fwd f32")
    close(y16, y, 2 ** -8, 1e-5, "ln fwd bf16")
    dg = torch.ones(d, device=DEV); db = torch.ones(d, device=DEV)       # accumulate onto ones
    dx = torch.empty(M, d, device=DEV); dx16 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    cs = torch.full((d,), 2.0, device=DEV)                               # the folded column sums accumulate too
    ops.layernorm_bwd(bf(dy) if dy_bf16 else dy.to(DEV), xd, w.to(DEV), mean, rstd, dg, db, g_in=gin.to(DEV), dx=dx,
                      dx16=dx16, colsum16=cs)
    # column sums of the bf16 output as stored (the bias gradient of the Linear behind it): fp32 summation noise only
    close(cs, dx16.float().sum(dim=0).cpu() + 2, 1e-5, 1e-4 * M ** 0.5, "ln colsum16")
    close(dx, xr.grad + gin, 1e-4, 1e-4, "ln dx")
    close(dx16, xr.grad + gin, 2 ** -8, 1e-4, "ln dx16")
    close(dg, wr.grad + 1, 1e-4, 1e-3, "ln dgamma")
    close(db, br.grad + 1, 1e-4, 1e-3, "ln dbeta")


@pytest.mark.parametrize("M,d,frac", [(37, 64, 0.4), (1000, 768, 0.35), (4099, 768, 0.0), (515, 2048, 1.0)])
def test_layernorm_bwd_row_map_equals_the_expanded_gradient(ops, M, d, frac):
    """neko_layernorm_bwd_rows (ABI v17): dy given for the mapped rows only (the LM head's loss positions) == the same call on the
    zero-filled [M, d] expansion, bit for bit (same kernel arithmetic, zeros read from registers instead of memory); includes no
    mapped row at all and every row mapped, in a permuted order."""
    g = torch.Generator().manual_seed(M + d)
    x = (torch.randn(M, d, generator=g) * 2 + 0.3).to(DEV)
    w, b = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    gin = torch.randn(M, d, generator=g).to(DEV)
    sel = torch.nonzero(torch.rand(M, generator=g) < frac).flatten() if 0.0 < frac < 1.0 else (torch.arange(M) if frac >= 1.0 else torch.zeros(0, dtype=torch.long))
    n = int(sel.numel())
    perm = torch.randperm(n, generator=g)
    rows = torch.randn(max(n, 1) + 3, d, generator=g)                 # compact rows (a few unused ones behind them)
    row_map = torch.full((M,), -1, dtype=torch.int32)
    row_map[sel] = perm.to(torch.int32)
    dense = torch.zeros(M, d)
    if n:
        dense[sel] = rows[perm]
    mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, w, b, y32=torch.empty(M, d, device=DEV), mean=mean, rstd=rstd)
    outs = []
    for kw in (dict(dy=dense.to(DEV)), dict(dy=rows.to(DEV), row_map=row_map.to(DEV))):
        dg = torch.ones(d, device=DEV); db = torch.ones(d, device=DEV); cs = torch.zeros(d, device=DEV)
        dx = torch.empty(M, d, device=DEV); dx16 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
        ops.layernorm_bwd(kw["dy"], x, w, mean, rstd, dg, db, g_in=gin, dx=dx, dx16=dx16, colsum16=cs, row_map=kw.get("row_map"))
        outs.append((dx, dx16, dg, db, cs))
    for a, c, what in zip(outs[0], outs[1], ("dx", "dx16", "dgamma", "dbeta", "colsum16")):
        assert torch.equal(a, c), f"row-map LayerNorm backward differs from the expanded call in {what}"


def test_wgrad_splitk_accumulates(ops):
    """engine._wgrad at a contraction long enough for split-K (workspace slices + fixed-order reduce): the product is ADDED
    to the gradient buffer, twice in a row (gradient accumulation; two image-shape groups in one backward)."""
    from neko_amd import engine
    g = torch.Generator().manual_seed(3)
    K, Mo, N = 8192, 256, 512
    A = rb(torch.randn(K, Mo, generator=g) * 0.1)
    Bm = rb(torch.randn(K, N, generator=g) * 0.1)
    assert ops.pick_splitk(Mo, N, K)[0] > 1
    out = torch.ones(Mo, N, device=DEV)
    engine._wgrad(bf(A), bf(Bm), Mo, N, K, out)
    engine._wgrad(bf(A), bf(Bm), Mo, N, K, out)
    close(out, 1.0 + 2.0 * (A.t() @ Bm), 1e-4, 2e-3, "split-K wgrad accumulate")


# ----------------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------------
def _masks(B, T, kind):
    m = torch.ones(B, T)
    if kind == "left":
        for b in range(B):
            m[b, : (7 + 37 * b) % max(1, T - 1)] = 0
    elif kind == "right":                      # pad_seq=True style right padding (gato_policy.py:423-431)
        m[0, T - 5:] = 0
        if B > 1:
            m[1, :3] = 0; m[1, T - 9:] = 0
    elif kind == "holes":
        m[0, 3] = 0; m[0, T // 2] = 0
    return m


@pytest.fixture(params=["auto", "onepass", "split", "streaming"])
def attn_path(request, ops):
    """Every attention schedule against the oracle: head-resident for hd = 32 / T <= 1024 with the library's own choice of backward
    (auto: one pass above 256 positions), with the backward in ONE pass at every length (onepass: neko_attn_set_path(3)), with the
    two-kernel backward (split: neko_attn_set_path(2), the bit-reproducible form), streaming for everything."""
    prev = ops.attn_set_path({"auto": 0, "onepass": 3, "split": 2, "streaming": 1}[request.param])
    yield request.param
    ops.attn_set_path(prev)


@pytest.mark.parametrize("mask_kind", ["none", "left", "right", "holes"])
@pytest.mark.parametrize("B,T,H,hd", [(2, 40, 2, 32), (3, 200, 4, 32), (2, 333, 2, 64), (1, 130, 2, 128),
                                      (2, 256, 3, 32), (1, 1, 2, 32), (2, 31, 1, 32), (1, 1024, 2, 32),
                                      (2, 1000, 3, 32), (17, 97, 3, 32), (2, 1024, 2, 128), (1, 600, 3, 64), (3, 257, 1, 128)])
def test_attention_fwd_bwd(ops, attn_path, B, T, H, hd, mask_kind):
    if attn_path == "streaming" and hd == 32 and T >= 1000 and mask_kind in ("right", "holes"):
        pytest.skip("large streaming cases are covered by the 'none' and 'left' masks")
    if attn_path in ("split", "onepass") and hd != 32:
        pytest.skip("the split / one-pass choice only exists for the head-resident kernels (hd = 32)")
    if T < 16 and mask_kind != "none":
        pytest.skip("mask pattern needs T >= 16")
    g = torch.Generator().manual_seed(B * 1000 + T + hd)
    d = H * hd
    qkv = rb(torch.randn(B, T, 3 * d, generator=g))
    mask = _masks(B, T, mask_kind)
    do = rb(torch.randn(B, T, d, generator=g))
    q, k, v = qkv.clone().requires_grad_(True).split(d, dim=2)
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.split(d, dim=2)
    sh = lambda t: t.view(B, T, H, hd).permute(0, 2, 1, 3)
    o_ref = O.attention_core(sh(q), sh(k), sh(v), mask).permute(0, 2, 1, 3).reshape(B, T, d)
    o_ref.backward(do)
    kb, ks = ops.mask_bias(mask.to(DEV))
    close(kb, (1 - mask) * -10000.0, 0, 0, "kbias")
    first = torch.tensor([int((mask[b] != 0).nonzero()[0]) for b in range(B)])
    assert torch.equal(ks.cpu().long(), first)
    qkv_d = bf(qkv.view(B * T, 3 * d))
    out, lse = ops.attn_fwd(qkv_d, kb, ks, B, T, H, hd)
    scale = float(o_ref.abs().max())
    # every row, including padded query rows (the reference's finite -1e4 semantics)
    close(out.view(B, T, d), o_ref, 2 ** -7, 4e-3 * scale, "attn out")
    dqkv = ops.attn_bwd(qkv_d, out, bf(do.view(B * T, d)), kb, ks, lse, B, T, H, hd)
    gs = float(leaf.grad.abs().max())
    close(dqkv.view(B, T, 3 * d), leaf.grad, 2 ** -6, 1e-2 * gs, "attn dqkv")


@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_attention_fwd_lazy_reference_survives_a_score_spike(ops, drop_p):
    """The head-resident forward takes its exponentials against a LAZILY moved reference (no row maximum per sub-tile): a
    row whose later scores climb far above everything seen before must trigger the rescale, not overflow.  One late key is
    made huge against the late queries: raw q.k = 512 -> 130 in log2 units above the early scores (2^130 is inf in fp32).
    Out, lse and the backward (which consumes lse) against the oracle (cdna_hip_programming.md rule 26: a rare
    data-dependent branch needs an input that forces it)."""
    B, T, H, hd = 2, 160, 2, 32
    d = H * hd
    g = torch.Generator().manual_seed(77)
    qkv = rb(torch.randn(B, T, 3 * d, generator=g) * 0.5)
    qkv[:, 100:, 0:hd] = 4.0                 # queries 100.. of head 0
    qkv[:, 100, d:d + hd] = 4.0              # key 100 of head 0
    mask = torch.ones(B, T)
    do = rb(torch.randn(B, T, d, generator=g))
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.split(d, dim=2)
    sh = lambda t: t.view(B, T, H, hd).permute(0, 2, 1, 3)
    o_ref = O.attention_core(sh(q), sh(k), sh(v), mask).permute(0, 2, 1, 3).reshape(B, T, d)
    kb, ks = ops.mask_bias(mask.to(DEV))
    qkv_d = bf(qkv.view(B * T, 3 * d))
    if drop_p == 0.0:
        o_ref.backward(do)
        out, lse = ops.attn_fwd(qkv_d, kb, ks, B, T, H, hd)
        assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(lse).all())
        close(out.view(B, T, d), o_ref, 2 ** -7, 4e-3 * float(o_ref.abs().max()), "attn out (spike)")
        dqkv = ops.attn_bwd(qkv_d, out, bf(do.view(B * T, d)), kb, ks, lse, B, T, H, hd)
        close(dqkv.view(B, T, 3 * d), leaf.grad, 2 ** -6, 1e-2 * float(leaf.grad.abs().max()), "attn dqkv (spike)")
    else:      # with dropout: finite, and identical to the streaming schedule's result up to bf16 rounding
        drop = ops.Drop(drop_p, 0x13579B)
        out, lse = ops.attn_fwd(qkv_d, kb, ks, B, T, H, hd, drop=drop)
        prev = ops.attn_set_path(1)
        try:
            out_s, lse_s = ops.attn_fwd(qkv_d, kb, ks, B, T, H, hd, drop=drop)
        finally:
            ops.attn_set_path(prev)
        assert bool(torch.isfinite(out.float()).all())
        assert float((out.float() - out_s.float()).abs().max()) < 2 ** -6 * float(out_s.float().abs().max())
        assert float((lse - lse_s).abs().max()) < 1e-3 * float(lse_s.abs().max())


@pytest.mark.parametrize("B,T,H", [(3, 200, 2), (2, 1024, 2), (5, 97, 3)])
def test_attention_bwd_zero_grad_on_masked_rows(ops, B, T, H):
    """Training case: dO is exactly zero on padded query rows.  The head-resident backward then skips the keys beyond the
    diagonal for those rows (their dS vanishes); the result must still match the oracle on every row."""
    hd, d = 32, H * 32
    g = torch.Generator().manual_seed(T + H)
    qkv = rb(torch.randn(B, T, 3 * d, generator=g))
    mask = _masks(B, T, "left")
    mask[B - 1, T // 2] = 0                                    # one hole too
    do = rb(torch.randn(B, T, d, generator=g)) * mask[:, :, None]
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.split(d, dim=2)
    sh = lambda t: t.view(B, T, H, hd).permute(0, 2, 1, 3)
    o_ref = O.attention_core(sh(q), sh(k), sh(v), mask).permute(0, 2, 1, 3).reshape(B, T, d)
    o_ref.backward(do)
    kb, ks = ops.mask_bias(mask.to(DEV))
    qkv_d = bf(qkv.view(B * T, 3 * d))
    out, lse = ops.attn_fwd(qkv_d, kb, ks, B, T, H, hd)
    dqkv = ops.attn_bwd(qkv_d, out, bf(do.view(B * T, d)), kb, ks, lse, B, T, H, hd)
    gs = float(leaf.grad.abs().max())
    close(dqkv.view(B, T, 3 * d), leaf.grad, 2 ** -6, 1e-2 * gs, "attn dqkv (zero dO on masked rows)")


# ----------------------------------------------------------------------------------------------------
# cross entropy
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,V", [(17, 1000), (64, 2176), (9, 52305)])
def test_cross_entropy(ops, R, V):
    g = torch.Generator().manual_seed(V)
    Vpad = (V + 127) // 128 * 128
    logits = torch.randn(R, V, generator=g) * 3
    target = torch.randint(0, V, (R,), generator=g)
    sel = (torch.rand(R, generator=g) > 0.3).float()
    sel[0] = 1.0
    weight = sel / sel.sum()
    lr = logits.clone().requires_grad_(True)
    loss_ref = (torch.nn.functional.cross_entropy(lr, target, reduction="none") * weight).sum()
    loss_ref.backward()
    buf = torch.zeros(R, Vpad, device=DEV); buf[:, :V] = logits.to(DEV)
    loss_row = torch.empty(R, device=DEV)
    dl = torch.full((R, Vpad), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.ce_fwd_bwd(buf, V, Vpad, target.to(DEV), weight.to(DEV), loss_row=loss_row, dlogits=dl)
    row_ref = torch.nn.functional.cross_entropy(logits, target, reduction="none") * sel
    close(loss_row, row_ref, 1e-5, 1e-5, "ce rows")
    close(dl[:, :V], lr.grad, 2 ** -8, 1e-7, "dlogits")
    assert float(dl[:, V:].float().abs().max()) == 0.0 if Vpad > V else True


# ----------------------------------------------------------------------------------------------------
# elementwise / optimiser
# ----------------------------------------------------------------------------------------------------
def test_cast_colsum_sqnorm(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(100003, generator=g)
    y = torch.empty(100003, dtype=torch.bfloat16, device=DEV)
    ops.cast_f32_bf16(x.to(DEV), y)
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))
    m = rb(torch.randn(1000, 776, generator=g))
    out = torch.ones(776, device=DEV)
    ops.colsum_bf16(bf(m), 1000, 776, out, accumulate=True)
    close(out, m.sum(0) + 1, 1e-4, 1e-3, "colsum")
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    ops.sqnorm_f32(x.to(DEV), acc)
    assert abs(float(acc) - float((x.double() ** 2).sum())) < 1e-6 * float((x.double() ** 2).sum())


def test_adamw_matches_torch(ops):
    g = torch.Generator().manual_seed(9)
    n = 70001
    p0 = torch.randn(n, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for it in range(5):
        grad = torch.randn(n, generator=g) * (3.0 if it % 2 == 0 else 0.01)
        p_ref.grad = grad.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gsq = torch.zeros(1, dtype=torch.float64, device=DEV)
        gd = grad.to(DEV)
        ops.sqnorm_f32(gd, gsq)
        ops.adamw_step(p, gd, m, v, p16, 3e-3, 0.9, 0.95, 1e-8, 0.1, gsq, 1.0, None, step, None)
        assert abs(math.sqrt(float(gsq)) - float(norm_ref)) < 1e-5 * float(norm_ref)
        close(p, p_ref.data, 1e-5, 1e-6, f"adamw step {it}")
    assert int(step) == 5
    assert torch.equal(p16.cpu(), p.cpu().to(torch.bfloat16))
    # inactive range is skipped entirely
    before = p.clone()
    active = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.adamw_step(p, gd, m, v, p16, 3e-3, 0.9, 0.95, 1e-8, 0.1, None, 1.0, None, step, active)
    assert torch.equal(p, before) and int(step) == 5


# ----------------------------------------------------------------------------------------------------
# continuous tokenizer (bit-exact) -- golden vectors of the reference + a wide random sweep vs the oracle
# ----------------------------------------------------------------------------------------------------
def test_tokenizer_golden_and_sweep(ops, golden):
    f = golden("g1_tokenizer")
    off = f["offset"]
    for name in ("edge", "rnd"):
        x = f[name].to(DEV)
        assert torch.equal(ops.tokenize_continuous(x, False, 100, 256, 1024, off).cpu(), f[name + "_act"])
        assert torch.equal(ops.tokenize_continuous(x, True, 100, 256, 1024, off).cpu(), f[name + "_obs"])
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(200000, generator=g) * 5, torch.rand(200000, generator=g) * 2 - 1,
                   torch.randn(100000, generator=g) * 300])
    for mu_law in (False, True):
        ref = O.tokenize_continuous(x, mu_law, 100, 256, 1024, 50257)
        got = ops.tokenize_continuous(x.to(DEV), mu_law, 100, 256, 1024, 50257).cpu()
        assert torch.equal(got, ref), f"mu_law={mu_law}: {int((got != ref).sum())} of {x.numel()} bins differ"


@pytest.mark.parametrize("R,V", [(5, 1000), (3, 52305), (4, 77)])
def test_ce_bf16_inplace(ops, R, V):
    """Training-path CE: bf16 logits in, gradient out in the same buffer; reference = fp32 math on the SAME bf16 logits."""
    g = torch.Generator().manual_seed(V)
    Vpad = (V + 63) // 64 * 64
    z16 = (torch.randn(R, V, generator=g) * 3).to(torch.bfloat16)
    buf = torch.full((R, Vpad), float("nan"), dtype=torch.bfloat16)        # pad columns arrive uninitialised
    buf[:, :V] = z16
    tgt = torch.randint(0, V, (R,), generator=g)
    w = torch.rand(R, generator=g)
    w[1] = 0.0
    z = z16.float()
    lse = torch.logsumexp(z, dim=1)
    loss_ref = (lse - z.gather(1, tgt[:, None])[:, 0]) * (w != 0)
    d_ref = (torch.softmax(z, dim=1) - torch.nn.functional.one_hot(tgt, V)) * w[:, None]
    zb = buf.to(DEV)
    loss = torch.empty(R, device=DEV)
    ops.ce_bf16_inplace(zb, V, Vpad, tgt.to(DEV), w.to(DEV), loss_row=loss, want_grad=False)
    close(loss, loss_ref, 1e-5, 1e-5, "ce loss (no grad)")
    assert torch.equal(zb.cpu()[:, :V], z16)                               # untouched without a gradient
    ops.ce_bf16_inplace(zb, V, Vpad, tgt.to(DEV), w.to(DEV), loss_row=loss, want_grad=True)
    close(loss, loss_ref, 1e-5, 1e-5, "ce loss")
    out = zb.cpu().float()
    assert float(out[:, V:].abs().max()) == 0.0 if Vpad > V else True
    close(out[:, :V], d_ref, 2 ** -7, 1e-6, "ce dlogits")


# ----------------------------------------------------------------------------------------------------
# patch embedding residual block
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("u8", [False, True])
def test_patch_resblock_fwd_bwd(ops, u8):
    cfg = O.OracleConfig(embed_dim=64, layers=1, heads=2, text_tokens=128, context_len=64)
    sd = O.init_state_dict(cfg, 21)
    g = torch.Generator().manual_seed(8)
    imgs = torch.floor(torch.rand(3, 3, 32, 48, generator=g) * 256)
    pe = "image_embedding.patch_embedding."
    names = ["conv1.weight", "conv1.bias", "gn2.weight", "gn2.bias", "conv2.weight", "conv2.bias"]
    leaf = {n: sd[pe + n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd); sd2.update({pe + n: leaf[n] for n in names})
    x = (imgs / 255.0 * 2 - 1) / 4.0
    n, c, H, W = x.shape
    xp = x.reshape(n, c, H // 16, 16, W // 16, 16).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3, 16, 16)
    y_ref = O.residual_block_v2(sd2, xp, 32).reshape(-1, 768)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    dev = {n: sd[pe + n].to(DEV).contiguous() for n in names}
    img_in = imgs.to(torch.uint8).to(DEV) if u8 else imgs.to(DEV)
    y16, xpd = ops.patch_resblock_fwd(img_in, *[dev[n] for n in names], 128, 32)
    close(xpd, xp.reshape(-1, 768), 1e-6, 1e-6, "normalised patches")
    # convolutions run on bf16 MFMA (fp32 accumulate), the block output is bf16: bf16-level tolerance
    close(y16, y_ref.detach(), 2 ** -7, 3e-3 * float(y_ref.detach().abs().max()), "resblock fwd")
    grads = {n: torch.zeros_like(dev[n]) for n in names}
    ops.patch_resblock_bwd(xpd, dy.to(DEV), *[dev[n] for n in names], 128, 32, *[grads[n] for n in names])
    for nme in names:
        ref = leaf[nme].grad
        # bf16 MFMA operands (h2, d_h1, dy, GELU(x) rounded to bf16), fp32 accumulation over up to 18 x 256 pixels
        close(grads[nme], ref, 1e-2, 1e-2 * float(ref.abs().max()) + 1e-5, f"resblock d{nme}")
    # ABI v17: the forward hands its GroupNorm statistics over and the backward takes them instead of recomputing them -- the same
    # numbers (the recomputation repeats the forward's arithmetic), so every gradient is bit-identical to the recomputing kernel's
    y16s, xps, stats = ops.patch_resblock_fwd(img_in, *[dev[n] for n in names], 128, 32, want_stats=True)
    assert torch.equal(y16s, y16) and torch.equal(xps, xpd) and stats is not None and stats.shape == (xp.shape[0], 64)
    h1 = torch.nn.functional.conv2d(torch.nn.functional.gelu(xp), sd[pe + "conv1.weight"], sd[pe + "conv1.bias"], padding=1)
    grp = h1.reshape(h1.shape[0], 32, -1)
    close(stats[:, :32], grp.mean(-1), 2e-2, 2e-2, "group means")         # bf16 convolution operands on the device side
    close(stats[:, 32:], (grp.var(-1, unbiased=False) + 1e-5).rsqrt(), 2e-2, 2e-2, "group rstd")
    grads_s = {n: torch.zeros_like(dev[n]) for n in names}
    ops.patch_resblock_bwd(xpd, dy.to(DEV), *[dev[n] for n in names], 128, 32, *[grads_s[n] for n in names], stats=stats)
    for nme in names:
        assert torch.equal(grads_s[nme], grads[nme]), f"stats-fed backward differs from the recomputing one in d{nme}"


def test_patch_pos_add(ops):
    g = torch.Generator().manual_seed(2)
    P, d = 37, 64
    out = torch.randn(P, d, generator=g); row = torch.randn(128, d, generator=g); col = torch.randn(128, d, generator=g)
    hp = torch.randint(0, 128, (P,), generator=g, dtype=torch.int32); wp = torch.randint(0, 128, (P,), generator=g, dtype=torch.int32)
    o = out.clone().to(DEV)
    ops.patch_pos_add(o, hp.to(DEV), wp.to(DEV), row.to(DEV), col.to(DEV))
    close(o, out + (row[hp.long()] + col[wp.long()]), 1e-6, 1e-6, "pos add")
    dr = torch.zeros(128, d, device=DEV); dc = torch.zeros(128, d, device=DEV)
    ops.patch_pos_add_bwd(out.to(DEV), hp.to(DEV), wp.to(DEV), dr, dc)
    ref_r = torch.zeros(128, d).index_add_(0, hp.long(), out); ref_c = torch.zeros(128, d).index_add_(0, wp.long(), out)
    close(dr, ref_r, 1e-5, 1e-5, "pos bwd row"); close(dc, ref_c, 1e-5, 1e-5, "pos bwd col")


@pytest.mark.parametrize("P,d,nrows", [(12289, 768, 128), (500, 72, 37), (1, 768, 128), (97, 64, 3), (4099, 2048, 128)])
def test_patch_pos_add_bwd_sorted_sums_are_exact_in_order_and_reproducible(ops, P, d, nrows):
    """The atomic-free table gradients (ABI v15, segsum.hip: stable sort by destination row, runs summed in patch order, chunk
    partials in chunk order): against an fp64 reference, accumulating onto an existing gradient, bit-identical from run to run, and
    agreeing with the atomic kernel to fp32 summation noise."""
    g = torch.Generator().manual_seed(P + d)
    out = torch.randn(P, d, generator=g)
    hp = torch.randint(0, nrows, (P,), generator=g, dtype=torch.int32)
    wp = torch.randint(0, nrows, (P,), generator=g, dtype=torch.int32)
    if P > 1000:
        hp[: P // 2] = 1                       # one hot row: a run that spans hundreds of chunks
    base_r, base_c = torch.randn(nrows, d, generator=g), torch.randn(nrows, d, generator=g)
    prev, ops.SCATTER_DET = ops.SCATTER_DET, True
    runs = []
    try:
        for _ in range(3):
            dr, dc = base_r.clone().to(DEV), base_c.clone().to(DEV)
            ops.patch_pos_add_bwd(out.to(DEV), hp.to(DEV), wp.to(DEV), dr, dc)
            runs.append((dr.cpu(), dc.cpu()))
    finally:
        ops.SCATTER_DET = prev
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    ref_r = base_r.double().index_add_(0, hp.long(), out.double())
    ref_c = base_c.double().index_add_(0, wp.long(), out.double())
    tol = 4e-6 * (P / 2 + 8) ** 0.5
    close(runs[0][0], ref_r.float(), 1e-5, tol, "pos bwd row"); close(runs[0][1], ref_c.float(), 1e-5, tol, "pos bwd col")
    prev, ops.SCATTER_DET = ops.SCATTER_DET, False
    try:
        dr, dc = base_r.clone().to(DEV), base_c.clone().to(DEV)
        ops.patch_pos_add_bwd(out.to(DEV), hp.to(DEV), wp.to(DEV), dr, dc)
    finally:
        ops.SCATTER_DET = prev
    # the atomic form adds in arrival order: its error on the hot row (P / 2 addends) is a random walk that differs from run to run
    # (seen once outside 1 x tol in ~6 runs of this file), hence the wider allowance for it
    close(dr, ref_r.float(), 1e-5, 3 * tol, "atomic row"); close(dc, ref_c.float(), 1e-5, 3 * tol, "atomic col")


@pytest.mark.parametrize("P,d,nrows", [(12289, 768, 128), (500, 72, 37), (1, 768, 128), (4099, 2048, 128)])
def test_patch_pos_add_bwd_host_sorted_equals_device_sorted(ops, P, d, nrows):
    """ABI v17: the patch-position table gradients from HOST-sorted (position, patch) pairs (ops.sorted_pairs: numpy stable sort) are the
    same fixed-order sums as the device-sorted ones of ABI v15 -- bit for bit, onto an existing gradient -- and match an fp64 reference."""
    g = torch.Generator().manual_seed(P + 3 * d)
    out = torch.randn(P, d, generator=g)
    hp = torch.randint(0, nrows, (P,), generator=g, dtype=torch.int32)
    wp = torch.randint(0, nrows, (P,), generator=g, dtype=torch.int32)
    if P > 1000:
        hp[: P // 2] = 1
    base_r, base_c = torch.randn(nrows, d, generator=g), torch.randn(nrows, d, generator=g)
    import numpy as np
    rows = torch.from_numpy(np.stack([*ops.sorted_pairs(hp.numpy()), *ops.sorted_pairs(wp.numpy())]).astype(np.int32)).to(DEV)
    prev_s, prev_d = ops.SORTED_SCATTER, ops.SCATTER_DET
    try:
        ops.SORTED_SCATTER, ops.SCATTER_DET = True, False
        dr, dc = base_r.clone().to(DEV), base_c.clone().to(DEV)
        ops.patch_pos_add_bwd(out.to(DEV), hp.to(DEV), wp.to(DEV), dr, dc, sorted_rows=rows)
        ops.SCATTER_DET = True
        er, ec = base_r.clone().to(DEV), base_c.clone().to(DEV)
        ops.patch_pos_add_bwd(out.to(DEV), hp.to(DEV), wp.to(DEV), er, ec)
    finally:
        ops.SORTED_SCATTER, ops.SCATTER_DET = prev_s, prev_d
    assert torch.equal(dr, er) and torch.equal(dc, ec)
    ref_r = base_r.double().index_add_(0, hp.long(), out.double())
    tol = 4e-6 * (P / 2 + 8) ** 0.5
    close(dr, ref_r.float(), 1e-5, tol, "host-sorted pos bwd row")


@pytest.mark.parametrize("ntok,d", [(4096, 768), (777, 64)])
def test_pack_embed_bwd_host_sorted_position_and_separator_sums(ops, ntok, d):
    """ABI v17 (neko_pack_embed_bwd_sorted): local-position and separator gradients of the packing backward as fixed-order segment sums
    over host-sorted (key, token) pairs behind the descriptors; the embedding rows keep their atomics, image rows are copied.  Against
    the deterministic ABI v15 path: d_pos / d_img bit-identical, d_sep and d_embed to summation-order noise; twice for bit reproducibility."""
    import numpy as np
    g = torch.Generator().manual_seed(ntok + d)
    V, pos_rows, n_img = 3000, 64, ntok // 4
    kind = torch.randint(0, 7, (ntok,), generator=g)                 # K_PAD .. K_IMAGE
    kind[kind == 4] = 2
    desc = torch.zeros(ntok, 4, dtype=torch.int32)
    desc[:, 0] = kind.to(torch.int32)
    desc[:, 2] = torch.randint(-1, pos_rows, (ntok,), generator=g).to(torch.int32)
    desc[kind == 5, 2] = -1                                          # separators carry no local position (build_layout)
    desc[kind == 0, 2] = -1
    img = (kind == 6).nonzero().flatten()
    desc[img, 1] = torch.arange(img.numel(), dtype=torch.int32) % max(n_img, 1)
    img_rows = int(min(img.numel(), n_img))
    if img.numel() > n_img:                                          # every image row is written once: keep the sources unique
        desc[img[n_img:], 0] = 0
        desc[img[n_img:], 2] = -1
    tokens = torch.randint(0, V, (ntok,), generator=g)
    tokens[: ntok // 3] = torch.randint(0, 8, (ntok // 3,), generator=g)      # contended embedding rows
    dx = torch.randn(ntok, d, generator=g)
    dn = desc.numpy()
    key = np.where(dn[:, 0] == 5, pos_rows, np.where((dn[:, 2] >= 0) & (dn[:, 0] != 0), dn[:, 2], -1))
    ks, ix = ops.sorted_pairs(key)
    ext = torch.from_numpy(np.concatenate([dn.reshape(-1), ks, ix]).astype(np.int32)).to(DEV)

    def run(sorted_tail, det):
        prev_s, prev_d = ops.SORTED_SCATTER, ops.SCATTER_DET
        try:
            ops.SORTED_SCATTER, ops.SCATTER_DET = True, det
            de = torch.ones(V, d, device=DEV); dp = torch.ones(pos_rows, d, device=DEV); ds = torch.ones(d, device=DEV)
            di = torch.zeros(max(img_rows, 1), d, device=DEV)
            ops.pack_embed_bwd(ext if sorted_tail else desc.to(DEV), tokens.to(DEV), dx.to(DEV), de, dp, ds, di, ntok, d, sorted_tail=sorted_tail)
            return de, dp, ds, di
        finally:
            ops.SORTED_SCATTER, ops.SCATTER_DET = prev_s, prev_d
    a, b, ref = run(True, False), run(True, False), run(False, True)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.equal(a[1], ref[1]), "d_pos: host-sorted sums differ from the device-sorted ones"
    # the separator run starts at another offset of the sorted array than in the ABI v15 call (there it follows the embedding rows), so
    # its 32-entry chunks group the same addends differently: fixed order and reproducible, not bit-equal to that path
    nsep = int((desc[:, 0] == 5).sum())
    close(a[2], ref[2].cpu(), 1e-5, 4e-6 * (nsep + 8) ** 0.5, "d_sep")
    close(a[2], (1.0 + dx[desc[:, 0] == 5].double().sum(0)).float(), 1e-5, 4e-6 * (nsep + 8) ** 0.5, "d_sep reference")
    assert torch.equal(a[3], ref[3]), "d_img"
    close(a[0], ref[0].cpu(), 1e-5, 4e-6 * (ntok / 3 / 8 + 8) ** 0.5, "d_embed (atomics) vs fixed-order sums")
    emb_kind = torch.isin(kind, torch.tensor([1, 2, 3]))
    want = torch.ones(V, d).double().index_add_(0, tokens[emb_kind & (desc[:, 0] != 0)], dx[emb_kind & (desc[:, 0] != 0)].double())
    close(ref[0], want.float(), 1e-5, 4e-6 * (ntok / 3 / 8 + 8) ** 0.5, "d_embed reference")


@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_attention_schedules_agree_at_metric_shape(ops, drop_p):
    """Head-resident and streaming kernels on the bench shape (32 x 1024 x 24 heads of 32, left padding on a third of the
    batch, dropout as in training): same sums in a different order -> agreement at bf16-rounding level on every row."""
    B, T, H, hd = 32, 1024, 24, 32
    d = H * hd
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(B * T, 3 * d, device=DEV, generator=g).to(torch.bfloat16)
    do = torch.randn(B * T, d, device=DEV, generator=g).to(torch.bfloat16)
    mask = torch.ones(B, T, device=DEV)
    for b in range(0, B, 3):
        mask[b, : 16 + 20 * (b % 5)] = 0
    do = (do.view(B, T, d) * mask[:, :, None].to(torch.bfloat16)).view(B * T, d).contiguous()   # training: no loss on padded rows
    kb, ks = ops.mask_bias(mask)
    drop = ops.Drop(drop_p, 0x2468ACE) if drop_p > 0 else None
    res = {}
    for path in (0, 1, 2):
        prev = ops.attn_set_path(path)
        try:
            out, lse = ops.attn_fwd(qkv, kb, ks, B, T, H, hd, drop=drop)
            dqkv = ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop)
        finally:
            ops.attn_set_path(prev)
        res[path] = (out.float(), lse, dqkv.float())
    so, sg = float(res[1][0].abs().max()), float(res[1][2].abs().max())
    assert float((res[0][0] - res[1][0]).abs().max()) < 2 ** -7 * so
    assert float((res[0][1] - res[1][1]).abs().max()) < 1e-4 * float(res[1][1].abs().max())
    assert float((res[0][2] - res[1][2]).abs().max()) < 2 ** -6 * sg
    # one-pass against two-kernel head-resident backward: dK sees D = sum dO.O added up in another order (fp32 rounding), dQ is added
    # up block by block in the order the waves arrive
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
    # (dV would be bit-equal on unpadded sequences; on the padded ones the one-pass kernel keeps padded key blocks on its fast path,
    # where the bias joins the exponent in another order)
    assert float((res[0][2] - res[2][2]).abs().max()) < 2 ** -7 * sg


def test_lm_head_chunking_is_invisible(ops):
    """engine.lm_head_loss over 4096-row chunks (the training path at the metric shape) against one chunk holding every row:
    same loss, same bf16 dlogits, and the same dH / dW from the backward GEMMs (split-K at these sizes)."""
    from neko_amd import engine
    g = torch.Generator(device=DEV).manual_seed(9)
    M, d, V = 9000, 768, 52305
    Vpad = (V + 127) // 128 * 128
    w = torch.zeros(Vpad, d, device=DEV)
    w[:V] = torch.randn(V, d, device=DEV, generator=g) * 0.02
    hp = engine.HeadParams(V=V, Vpad=Vpad, w=w.to(torch.bfloat16), g_w=torch.zeros(Vpad, d, device=DEV))
    hf = torch.randn(M, d, device=DEV, generator=g).to(torch.bfloat16)
    tgt = torch.randint(0, V, (M,), device=DEV, generator=g)
    sel = (torch.rand(M, device=DEV, generator=g) > 0.25).float()
    cnt = sel.sum()
    la, da = engine.lm_head_loss(hp, hf, tgt, sel, cnt, True, chunk_rows=4096)
    lb, db = engine.lm_head_loss(hp, hf, tgt, sel, cnt, True, chunk_rows=1 << 20)
    assert abs(float(la) - float(lb)) < 1e-6 * abs(float(lb))
    assert torch.equal(da, db)
    go = torch.ones((), device=DEV)
    dh = engine.lm_head_backward(hp, hf, da, go)
    engine.SideStream.join(DEV)
    torch.cuda.synchronize()
    dh_ref = (da.float() @ hp.w.float())
    dw_ref = da.float().t() @ hf.float()
    close(dh, dh_ref, 1e-3, 1e-3 * float(dh_ref.abs().max()), "LM-head dH")
    close(hp.g_w, dw_ref, 1e-3, 1e-3 * float(dw_ref.abs().max()), "LM-head dW")


# ----------------------------------------------------------------------------------------------------
# packed sequences of different lengths in one attention launch (neko_attn_*_varlen)
# ----------------------------------------------------------------------------------------------------
def test_attention_varlen_equals_per_sequence_calls(ops):
    """One packed launch over sequences of 1024 / 200 / 33 / 512 / 97 / 1000 positions (two of them left-padded) must give,
    for every sequence, the very bits the uniform kernels give when that sequence is run alone: out, lse, dqkv."""
    H, hd = 3, 32
    d = H * hd
    lengths = [1024, 200, 33, 512, 97, 1000]
    pads = [0, 17, 0, 40, 0, 0]
    rows = sum(lengths)
    g = torch.Generator(device=DEV).manual_seed(31)
    qkv = torch.randn(rows, 3 * d, device=DEV, generator=g).to(torch.bfloat16)
    do = torch.randn(rows, d, device=DEV, generator=g).to(torch.bfloat16)
    kbs, kss, r0 = [], [], 0
    for T, pad in zip(lengths, pads):
        m = torch.ones(1, T, device=DEV)
        m[0, :pad] = 0
        kb, ks = ops.mask_bias(m)
        kbs.append(kb.reshape(-1)); kss.append(ks.reshape(-1))
        do[r0:r0 + pad] = 0                          # training: no loss reaches a padded position
        r0 += T
    geom = ops.VarlenGeom(lengths, H, DEV)
    assert geom.rows == rows
    kb_all, ks_all = torch.cat(kbs), torch.cat(kss)
    # 2: two-kernel backward everywhere, everything bit-equal; 3: one-pass backward everywhere, dK / dV bit-equal and dQ added up in
    # arrival order; 0 (automatic, round 5): the packed launch is sized by its longest sequence (1024 positions: two kernels) while a
    # 512-position sequence alone takes the one-pass kernel -- the two forms round a partially padded key block's probabilities through
    # different instruction sequences (fma against multiply + add), so a few elements sit one bf16 ulp apart
    # (tools/probe/r05_attn_bwd_diff.py: 2 elements of one row at T = 512 with 40 padded keys)
    for path in (2, 3, 0):
        prev = ops.attn_set_path(path)
        try:
            out, lse, _ = ops.attn_fwd_varlen(qkv, kb_all, ks_all, geom, hd)
            dqkv = ops.attn_bwd_varlen(qkv, out, do, kb_all, ks_all, lse, geom, hd)
            r0 = 0
            for i, T in enumerate(lengths):
                q1 = qkv[r0:r0 + T].contiguous()
                o1, l1 = ops.attn_fwd(q1, kbs[i].view(1, T), kss[i], 1, T, H, hd)
                g1 = ops.attn_bwd(q1, o1, do[r0:r0 + T].contiguous(), kbs[i].view(1, T), kss[i], l1, 1, T, H, hd)
                assert torch.equal(out[r0:r0 + T], o1), f"out of sequence {i}"
                assert torch.equal(lse[r0 * H:(r0 + T) * H].view(H, T), l1.view(H, T)), f"lse of sequence {i}"
                if path == 2:
                    assert torch.equal(dqkv[r0:r0 + T], g1), f"dqkv of sequence {i}"
                else:
                    if path == 3:
                        assert torch.equal(dqkv[r0:r0 + T, d:], g1[:, d:]), f"dK / dV of sequence {i}"
                    for nm, sl in (("dQ", slice(0, d)), ("dK", slice(d, 2 * d)), ("dV", slice(2 * d, 3 * d))):
                        ga, gb = dqkv[r0:r0 + T, sl].float(), g1[:, sl].float()
                        assert float((ga - gb).abs().max()) <= 2 ** -7 * float(gb.abs().max()), f"{nm} of sequence {i} (path {path})"
                r0 += T
        finally:
            ops.attn_set_path(prev)


@pytest.mark.parametrize("hd,H", [(128, 2), (64, 3)])
@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_attention_varlen_wide_heads_equals_per_sequence_calls(ops, hd, H, drop_p):
    """hd = 64 / 128 (the DMA-ring kernels, ABI v16): one packed launch over sequences of 1024 / 200 / 33 / 2048 / 97 / 1000 positions
    (two left-padded; 2048 is beyond the head-resident limit) gives the very bits of running every sequence alone -- out, lse and
    dqkv without dropout; with dropout the hash walks packed row ids, so the packed result is held against the fp32 oracle with
    the host restatement of that index instead (every sequence, forward and backward)."""
    d = H * hd
    lengths = [1024, 200, 33, 2048, 97, 1000] if drop_p == 0.0 else [300, 64, 33, 520]
    pads = [0, 17, 0, 40, 0, 0][:len(lengths)]
    rows = sum(lengths)
    g = torch.Generator(device=DEV).manual_seed(41 + hd)
    qkv = (torch.randn(rows, 3 * d, device=DEV, generator=g) * 0.6).to(torch.bfloat16)
    do = torch.randn(rows, d, device=DEV, generator=g).to(torch.bfloat16)
    kbs, kss, r0 = [], [], 0
    for T, pad in zip(lengths, pads):
        m = torch.ones(1, T, device=DEV)
        m[0, :pad] = 0
        kb, ks = ops.mask_bias(m)
        kbs.append(kb.reshape(-1)); kss.append(ks.reshape(-1))
        do[r0:r0 + pad] = 0
        r0 += T
    geom = ops.VarlenGeom(lengths, H, DEV)
    kb_all, ks_all = torch.cat(kbs), torch.cat(kss)
    drop = ops.Drop(drop_p, 0xFACE) if drop_p > 0 else None
    out, lse, mk = ops.attn_fwd_varlen(qkv, kb_all, ks_all, geom, hd, drop=drop, want_mask=True)
    assert mk is None                                   # these kernels re-hash the keep decisions in the backward
    dqkv = ops.attn_bwd_varlen(qkv, out, do, kb_all, ks_all, lse, geom, hd, drop=drop)
    r0 = 0
    if drop is None:
        for i, T in enumerate(lengths):
            q1 = qkv[r0:r0 + T].contiguous()
            o1, l1 = ops.attn_fwd(q1, kbs[i].view(1, T), kss[i], 1, T, H, hd)
            g1 = ops.attn_bwd(q1, o1, do[r0:r0 + T].contiguous(), kbs[i].view(1, T), kss[i], l1, 1, T, H, hd)
            assert torch.equal(out[r0:r0 + T], o1), f"out of sequence {i}"
            assert torch.equal(lse[r0 * H:(r0 + T) * H].view(H, T), l1.view(H, T)), f"lse of sequence {i}"
            assert torch.equal(dqkv[r0:r0 + T], g1), f"dqkv of sequence {i}"
            r0 += T
        return
    from test_dropout_gpu import mask_attn_varlen          # (tests/ is on sys.path: pytest's rootdir insertion)
    dms = mask_attn_varlen(lengths, H, drop)
    for i, T in enumerate(lengths):
        x = qkv[r0:r0 + T].float().cpu().view(1, T, 3 * d).clone().requires_grad_(True)
        q, k, v = x.split(d, dim=2)
        sh = lambda t: t.view(1, T, H, hd).permute(0, 2, 1, 3)
        mask = torch.ones(1, T); mask[0, :pads[i]] = 0
        o_ref = O.attention_core(sh(q), sh(k), sh(v), mask, drop_mask=dms[i]).permute(0, 2, 1, 3).reshape(1, T, d)
        o_ref.backward(do[r0:r0 + T].float().cpu().view(1, T, d))
        sc = float(o_ref.detach().abs().max())
        assert float((out[r0:r0 + T].float().cpu() - o_ref.detach()[0]).abs().max()) < 1e-2 * sc, f"out of sequence {i}"
        gs = float(x.grad.abs().max())
        assert float((dqkv[r0:r0 + T].float().cpu() - x.grad[0]).abs().max()) < 2e-2 * gs, f"dqkv of sequence {i}"
        r0 += T


def test_attention_varlen_dropout_masks_are_consistent(ops):
    """Packed launch with attention dropout: the backward that reuses the forward's stored keep masks must produce the very bits of
    the backward that re-hashes them (same index formula on both sides: unique row id x ceil(Tmax / 4) + key / 4), the keep rate
    is the requested one, and with the masks handed to a second forward into the same buffer nothing stale survives."""
    H, hd = 2, 32
    d = H * hd
    lengths = [1024, 300, 64, 777]
    rows = sum(lengths)
    g = torch.Generator(device=DEV).manual_seed(32)
    qkv = (torch.randn(rows, 3 * d, device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    do = torch.randn(rows, d, device=DEV, generator=g).to(torch.bfloat16)
    kb = torch.zeros(rows, device=DEV)
    ks = torch.zeros(len(lengths), dtype=torch.int32, device=DEV)
    geom = ops.VarlenGeom(lengths, H, DEV)
    drop = ops.Drop(0.1, 0xC0FFEE)
    out, lse, mk = ops.attn_fwd_varlen(qkv, kb, ks, geom, hd, drop=drop, want_mask=True)
    assert mk is not None and mk.numel() == geom.mask_dwords
    prev = ops.attn_set_path(3)                                  # (the automatic choice at 1024 positions is the two-kernel form since round 5)
    try:
        g_mask = ops.attn_bwd_varlen(qkv, out, do, kb, ks, lse, geom, hd, drop=drop, mask=mk)
        g_hash = ops.attn_bwd_varlen(qkv, out, do, kb, ks, lse, geom, hd, drop=drop, mask=None)
    finally:
        ops.attn_set_path(prev)
    assert torch.equal(g_mask[:, d:], g_hash[:, d:])             # one-pass backward: dK / dV bit-equal, dQ to fp32 summation order
    assert float((g_mask[:, :d].float() - g_hash[:, :d].float()).abs().max()) <= 2 ** -7 * float(g_hash[:, :d].float().abs().max())
    prev = ops.attn_set_path(2)
    try:
        g_mask2 = ops.attn_bwd_varlen(qkv, out, do, kb, ks, lse, geom, hd, drop=drop, mask=mk)
        g_hash2 = ops.attn_bwd_varlen(qkv, out, do, kb, ks, lse, geom, hd, drop=drop, mask=None)
    finally:
        ops.attn_set_path(prev)
    assert torch.equal(g_mask2, g_hash2)
    assert torch.equal(g_mask2[:, d:], g_mask[:, d:])
    out0, _, _ = ops.attn_fwd_varlen(qkv, kb, ks, geom, hd)
    assert bool(torch.isfinite(out.float()).all())
    # dropped and undropped outputs differ, but agree in expectation: mean absolute difference well below the signal
    assert float((out.float() - out0.float()).abs().mean()) < 0.35 * float(out0.float().abs().mean())
    out2, lse2, mk2 = ops.attn_fwd_varlen(qkv, kb, ks, geom, hd, drop=drop, want_mask=True)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)


def test_attention_backward_reproducible_switch_is_thread_local_and_selects_the_two_kernel_form(ops):
    """neko_attn_bwd_reproducible (ABI v18; what NEKO_DETERMINISTIC sets around its backward calls): at a length whose automatic
    backward is the one-pass kernel (T = 384) the calling thread gets the very bits of neko_attn_set_path(2), the process-wide knob is
    left alone, and another thread still sees the automatic schedule (its switch reads 0)."""
    import threading
    from neko_amd import _lib
    lib = _lib.load()
    B, T, H, hd = 3, 384, 2, 32
    d = H * hd
    g = torch.Generator(device=DEV).manual_seed(384)
    qkv = (torch.randn(B * T, 3 * d, device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    do = torch.randn(B * T, d, device=DEV, generator=g).to(torch.bfloat16)
    kb, ks = ops.mask_bias(torch.ones(B, T, device=DEV))
    drop = ops.Drop(0.1, 0xBEEF)
    out, lse, mk = ops.attn_fwd(qkv, kb, ks, B, T, H, hd, drop=drop, want_mask=True)
    knob = ops.attn_set_path(-1)
    prev = ops.attn_set_path(2)
    try:
        want = ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop, mask=mk)
    finally:
        ops.attn_set_path(prev)
    assert lib.neko_attn_bwd_reproducible(-1) == 0
    prev_det, ops.SCATTER_DET = ops.SCATTER_DET, True
    try:
        got = [ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop, mask=mk) for _ in range(2)]
    finally:
        ops.SCATTER_DET = prev_det
    assert torch.equal(got[0], want) and torch.equal(got[1], want)
    assert lib.neko_attn_bwd_reproducible(-1) == 0 and ops.attn_set_path(-1) == knob        # restored / never touched
    assert lib.neko_attn_bwd_reproducible(1) == 0
    seen = []
    t = threading.Thread(target=lambda: seen.append(int(lib.neko_attn_bwd_reproducible(-1))))
    t.start(); t.join()
    assert seen == [0] and lib.neko_attn_bwd_reproducible(0) == 1


@pytest.mark.parametrize("T,hd,mask_kind", [(2048, 128, "left"), (1500, 64, "holes"), (4096, 128, "none"), (4100, 128, "left")])
def test_attention_long_sequences_wide_heads(ops, T, hd, mask_kind):
    """Sequences beyond the reference's context length on the wide-head kernels: up to T = 4096 the DMA-ring kernels keep the key
    bias / lse / D of the whole sequence in LDS (attention_stream.hip), beyond it the dispatcher falls back to the register-staged
    kernels -- both against the oracle."""
    B, H = 1, 1
    g = torch.Generator().manual_seed(T + hd)
    d = H * hd
    qkv = rb(torch.randn(B, T, 3 * d, generator=g))
    mask = _masks(B, T, mask_kind)
    do = rb(torch.randn(B, T, d, generator=g))
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.split(d, dim=2)
    sh = lambda t: t.view(B, T, H, hd).permute(0, 2, 1, 3)
    o_ref = O.attention_core(sh(q), sh(k), sh(v), mask).permute(0, 2, 1, 3).reshape(B, T, d)
    o_ref.backward(do)
    kb, ks = ops.mask_bias(mask.to(DEV))
    qkv_d = bf(qkv.view(B * T, 3 * d))
    out, lse = ops.attn_fwd(qkv_d, kb, ks, B, T, H, hd)
    close(out.view(B, T, d), o_ref, 2 ** -7, 4e-3 * float(o_ref.abs().max()), "attn out")
    dqkv = ops.attn_bwd(qkv_d, out, bf(do.view(B * T, d)), kb, ks, lse, B, T, H, hd)
    close(dqkv.view(B, T, 3 * d), leaf.grad, 2 ** -6, 1e-2 * float(leaf.grad.abs().max()), "attn dqkv")


def test_attention_backward_is_bit_reproducible_by_default_at_the_atari_length(ops):
    """configs[3] (Atari layout, T = 494): since round 6 the automatic head-resident backward is the two-kernel form at every length, so the
    default schedule -- no knob, no NEKO_DETERMINISTIC -- returns the same bits on every call and the very bits of neko_attn_set_path(2)
    (the one-pass kernel summed dQ in arrival order for 256 < T <= 512)."""
    B, T, H, hd = 4, 494, 3, 32
    d = H * hd
    g = torch.Generator(device=DEV).manual_seed(494)
    qkv = (torch.randn(B * T, 3 * d, device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    do = torch.randn(B * T, d, device=DEV, generator=g).to(torch.bfloat16)
    m = torch.ones(B, T, device=DEV)
    m[1, :37] = 0                                        # a left-padded episode
    kb, ks = ops.mask_bias(m)
    drop = ops.Drop(0.1, 0xA7A21)
    assert ops.attn_set_path(-1) == 0                    # the automatic schedule
    out, lse, mk = ops.attn_fwd(qkv, kb, ks, B, T, H, hd, drop=drop, want_mask=True)
    got = [ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop, mask=mk) for _ in range(4)]
    assert all(torch.equal(got[0], x) for x in got[1:])
    prev = ops.attn_set_path(2)
    try:
        want = ops.attn_bwd(qkv, out, do, kb, ks, lse, B, T, H, hd, drop=drop, mask=mk)
    finally:
        ops.attn_set_path(prev)
    assert torch.equal(got[0], want)
