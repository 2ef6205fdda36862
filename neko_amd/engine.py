"""Kernel orchestration of the hot path: transformer stack forward/backward and LM head + masked CE.

This is the host side of the HIP path: it only allocates device buffers (torch caching allocator)
and enqueues libneko_hip.so kernels on the current stream.  Follows the op sequence of SURVEY.md 2.2:
Block.forward (gato/transformers/trajectory_gpt2.py:311-359), GPT2Model.forward (:611-795), LM head and
loss (gato/policy/gato_policy.py:172-186).

Numerics ("bf16 autocast-equivalent, fp32 accumulate everywhere"): the residual stream, LayerNorm
statistics, softmax, logits, loss and all gradients of parameters are fp32; MFMA operands
(LayerNorm outputs, qkv, attention output, MLP hidden, weights, upstream gradients) are bf16.
"""
from __future__ import annotations

import contextlib
import os
import threading
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch

from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


@dataclass
class LayerParams:
    """Views into the flat buffers for one Block (fp32 params, bf16 weight shadows, fp32 grads)."""
    ln1_w: torch.Tensor; ln1_b: torch.Tensor; ln2_w: torch.Tensor; ln2_b: torch.Tensor
    b_qkv: torch.Tensor; b_o: torch.Tensor; b_fc: torch.Tensor; b_pr: torch.Tensor
    w_qkv: torch.Tensor; w_o: torch.Tensor; w_fc: torch.Tensor; w_pr: torch.Tensor   # bf16 shadows, (in,out)
    g_ln1_w: torch.Tensor; g_ln1_b: torch.Tensor; g_ln2_w: torch.Tensor; g_ln2_b: torch.Tensor
    g_b_qkv: torch.Tensor; g_b_o: torch.Tensor; g_b_fc: torch.Tensor; g_b_pr: torch.Tensor
    g_w_qkv: torch.Tensor; g_w_o: torch.Tensor; g_w_fc: torch.Tensor; g_w_pr: torch.Tensor
    # GEGLU gate (activation_fn='geglu', trajectory_gpt2.py:267-276): nn.Linear (4d, d) weight + bias, or None
    w_gate: Optional[torch.Tensor] = None; b_gate: Optional[torch.Tensor] = None
    g_w_gate: Optional[torch.Tensor] = None; g_b_gate: Optional[torch.Tensor] = None


@dataclass
class StackParams:
    d: int
    heads: int
    eps: float
    layers: List[LayerParams]
    lnf_w: torch.Tensor; lnf_b: torch.Tensor; g_lnf_w: torch.Tensor; g_lnf_b: torch.Tensor


@dataclass
class HeadParams:
    V: int
    Vpad: int
    w: torch.Tensor      # bf16 shadow [Vpad, d] (rows >= V are zero)
    g_w: torch.Tensor    # fp32 grad   [Vpad, d]


@dataclass
class DropSites:
    """Dropout sites of one forward pass (None entries = off): embedding dropout, and per layer the attention-
    probability dropout and the two residual dropouts (trajectory_gpt2.py:541,707 / :142,179 / :143,254 / :271,278)."""
    embd: Optional[ops.Drop]
    attn: List[Optional[ops.Drop]]
    resid_attn: List[Optional[ops.Drop]]
    resid_mlp: List[Optional[ops.Drop]]


@dataclass
class LayerCtx:
    x: torch.Tensor = None; a1: torch.Tensor = None; mean1: torch.Tensor = None; rstd1: torch.Tensor = None
    qkv: torch.Tensor = None; o: torch.Tensor = None; lse: torch.Tensor = None
    dmask: object = None      # attention-dropout keep masks of the forward (tensor, list per segment, or None)
    x1: torch.Tensor = None; a2: torch.Tensor = None; mean2: torch.Tensor = None; rstd2: torch.Tensor = None
    pre: torch.Tensor = None; h: torch.Tensor = None; gate: torch.Tensor = None
    pre_is_factor: bool = False     # `pre` holds gelu'(pre-activation) (forward epilogue act 3) instead of the pre-activation


@dataclass
class Segment:
    """Rows [row0, row0 + B*T) of the token stream hold B sequences of T positions each.  A plain (B, T) batch is one
    segment; a length-bucketed ragged batch (GatoPolicy.ragged_groups) is several with different T: every kernel
    except attention works on the concatenated rows, attention runs once per segment."""
    row0: int
    B: int
    T: int
    kbias: torch.Tensor = None      # additive key bias (B, T) fp32
    kstart: torch.Tensor = None     # first real key per sequence (B,) int32


@dataclass
class StackCtx:
    B: int = 0
    T: int = 0
    segs: List[Segment] = field(default_factory=list)
    varlen: object = None            # Varlen: the packed single-launch view of `segs` (ragged batches on the head-resident kernels)
    layers: List[LayerCtx] = field(default_factory=list)
    drops: Optional[DropSites] = None
    xf: torch.Tensor = None          # residual stream entering ln_f
    meanf: torch.Tensor = None
    rstdf: torch.Tensor = None


#: dtype of the gradients that the dgrad GEMMs hand to the LayerNorm backward.  bf16 is what autocast leaves there in the reference
#: (the gradient of a bf16 addmm input is bf16) and 100 MB less out of each N = 768 dgrad and into each LayerNorm backward at
#: 65536 rows.  Round 1 measured it 0.2 ms per step SLOWER and kept fp32; the reason turned out to be the LayerNorm backward's
#: launch shape, not the bytes: with 8-byte loads on that operand a wave has fewer bytes in flight and the kernel wants two blocks
#: per CU (220 -> 157 us per call at 65536 rows, fp32 dy: 163; tools/ln_bench.py).  With that, bf16 is 0.18 ms per m-mix step
#: faster (39.58 -> 39.40, profiles/r03_step_ab.txt) and the default; NEKO_LN_DY_BF16=0 returns to fp32.  Both paths are covered
#: by the LayerNorm parity test.
LN_DY_DTYPE = F32 if os.environ.get("NEKO_LN_DY_BF16", "1") == "0" else BF16

#: The forward c_fc GEMM can leave gelu'(pre) (bf16) instead of the pre-activation in its second output (epilogue act 3:
#: both values come out of ONE evaluation of the erf series), so that the dgrad through the MLP projection multiplies by a
#: stored factor (act 4) instead of evaluating the series again for 65536 x 3072 elements per layer: its GELU' epilogue
#: cost 11.6 us of every 30 us tile.  The factor is rounded to bf16 (2^-9 relative, unbiased) before the
#: product is; gated MLPs (GEGLU) keep the pre-activation, their backward needs gelu(pre) as well.  NEKO_GELU_FACTOR=0: off.
#: Measured -0.35 ms per m-mix step (profiles/r03_step_ab.txt).
GELU_FACTOR = os.environ.get("NEKO_GELU_FACTOR", "1") != "0"


def _dgrad_to_ln(a, w, M, N, K, ldb):
    out = torch.empty(M, N, dtype=LN_DY_DTYPE, device=a.device)
    if LN_DY_DTYPE == BF16:
        ops.gemm(a, w, M, N, K, ldb=ldb, out_bf16=out)
    else:
        ops.gemm(a, w, M, N, K, ldb=ldb, out_f32=out)
    return out


class SideStream:
    """Weight / bias gradients on a second HIP stream.  They hang off the backward chain (nothing downstream reads them
    until the all-reduce / optimiser), while the chain itself is a sequence of launches whose last round rarely fills
    the chip: the dgrads with N = 768 are 768 tiles on 512 slots (1.5 rounds), forward-shaped qkv 4.5 rounds, the
    weight gradients themselves 252 blocks.  On their own queue the gradient kernels fill those tails.
    fork(fn, *temps): fn's launches run on the side stream after everything enqueued so far on the current stream;
    `temps` are tensors that may be freed before join() (their blocks must not be recycled under the side kernels).
    join(): the current stream waits for the side stream.  NEKO_WGRAD_STREAM=0 keeps everything on one stream."""
    enabled = os.environ.get("NEKO_WGRAD_STREAM", "auto") != "0"
    # "auto" (default): the side stream is used only for steps of fewer than ONE_STREAM_ROWS rows.  Round 3, same-box A/B: at
    # 65536 rows one stream is 0.4 ms FASTER (38.38 vs 38.78 ms; 32768 rows: 21.02 vs 21.15), at 7680 / 8192 rows the side stream
    # wins (c2 5.90 vs 6.07 ms, Gato-1.2B 78.7 vs 80.8) -- the chip runs these kernels at its power limit, so a second GEMM
    # beside a full-grid one only adds L2 contention; it pays where the chain's launches leave CUs idle (profiles/r03_step_ab.txt).
    auto = os.environ.get("NEKO_WGRAD_STREAM", "auto") == "auto"
    ONE_STREAM_ROWS = int(os.environ.get("NEKO_WGRAD_ONE_STREAM_ROWS", "20000"))
    rows_hint = 0            # rows of the step being differentiated (set by the backward entry points)
    _streams: dict = {}
    _dirty: dict = {}
    _tls = threading.local()  # .off > 0: this THREAD keeps everything on its current stream (a step being captured; ADVICE r05: the
                              # capture used to flip the class-wide `enabled`, which another thread's eager backward would have seen)

    @classmethod
    def _on(cls) -> bool:
        return cls.enabled and getattr(cls._tls, "off", 0) == 0

    @classmethod
    @contextlib.contextmanager
    def suspended(cls):
        """`with SideStream.suspended():` -- launches of the calling thread stay on its current stream (stream capture)."""
        cls._tls.off = getattr(cls._tls, "off", 0) + 1
        try:
            yield
        finally:
            cls._tls.off -= 1

    @classmethod
    def _get(cls, dev) -> "torch.cuda.Stream":
        key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=key)
        return cls._streams[key]

    @classmethod
    def fork(cls, fn: Callable[[], None], *temps: torch.Tensor) -> None:
        dev = temps[0].device if temps else torch.device("cuda", torch.cuda.current_device())
        if not cls._on() or dev.type != "cuda" or (cls.auto and cls.rows_hint >= cls.ONE_STREAM_ROWS):
            fn()
            return
        side = cls._get(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
        for t in temps:
            t.record_stream(side)
        cls._dirty[side.device.index] = True

    @classmethod
    def pending(cls, dev=None) -> Optional["torch.cuda.Stream"]:
        """The side stream if launches are outstanding on it, else None.  A consumer that runs on a stream of its own (the gradient
        all-reduce) orders itself behind it instead of making the compute stream wait for the weight gradients (join())."""
        if not cls._on() or not torch.cuda.is_available():
            return None
        key = torch.device(dev).index if (dev is not None and torch.device(dev).index is not None) else torch.cuda.current_device()
        return cls._streams[key] if cls._dirty.get(key) else None

    @classmethod
    def join(cls, dev=None) -> None:
        if not cls._on() or not torch.cuda.is_available():
            return
        key = torch.device(dev).index if (dev is not None and torch.device(dev).index is not None) else torch.cuda.current_device()
        if cls._dirty.get(key):
            torch.cuda.current_stream(key).wait_stream(cls._streams[key])
            cls._dirty[key] = False


def _wgrad(A: torch.Tensor, Bm: torch.Tensor, Mout: int, N: int, K: int, out: torch.Tensor,
           alpha_dev: Optional[torch.Tensor] = None) -> None:
    """out[Mout,N] += A^T @ B with A stored [K, Mout], B stored [K, N] (both k-strided).  Always "+=": a split-K product
    goes through the slice workspace and the fixed-order reduce adds the existing gradient (a second backward before the
    optimiser step -- gradient accumulation, trainer.py:176 -- must not overwrite the first)."""
    sk, kps = ops.pick_splitk(Mout, N, K)
    ops.gemm(A, Bm, Mout, N, K, a_kstrided=True, b_kstrided=True, lda=A.stride(0), ldb=Bm.stride(0),
             out_f32=out, ldcf=out.stride(0), accumulate=True, splitk=sk, k_per_split=kps,
             alpha_dev=alpha_dev)


def _geglu_gate(lp: LayerParams, a2: torch.Tensor, h: torch.Tensor, M: int, d: int) -> Optional[torch.Tensor]:
    """GEGLU (trajectory_gpt2.py:275-276): h (= gelu(c_fc a2), from the GEMM epilogue) *= gated_layer(a2) in place.
    Returns the gate [M, 4d] bf16 (saved for backward) or None when the MLP is not gated."""
    if lp.w_gate is None:
        return None
    gate = torch.empty(M, 4 * d, dtype=BF16, device=a2.device)
    if M <= 8 and d % 8 == 0 and d <= 3072:
        ops.gemv(a2, lp.w_gate, M, 4 * d, d, b_kstrided=False, ldw=d, bias=lp.b_gate, out_bf16=gate)
    else:
        ops.gemm(a2, lp.w_gate, M, 4 * d, d, ldb=d, bias=lp.b_gate, out_bf16=gate)
    ops.geglu_fwd(h, gate)
    return gate


def _seg_drop(drop: Optional[ops.Drop], si: int) -> Optional[ops.Drop]:
    """Attention-probability dropout indexes its mask by (sequence, head, query, key) LOCAL to a launch: every
    segment after the first gets its own key so that masks do not repeat across segments."""
    if drop is None or si == 0:
        return drop
    d2 = ops.Drop(0.0, ops.mix32(drop.key + 0x632BE5AB * si))
    d2.thr, d2.scale = drop.thr, drop.scale
    return d2


#: Length-bucketed (ragged) batches run their attention as ONE packed launch over all sequences (neko_attn_*_varlen) when the
#: head-resident kernels apply (hd = 32, every length <= 1024); NEKO_ATTN_VARLEN=0 returns to one launch per bucket.
ATTN_VARLEN = os.environ.get("NEKO_ATTN_VARLEN", "1") != "0"


@dataclass
class Varlen:
    """The packed view of a segment list: per-sequence geometry + key bias / first real key of every row / sequence."""
    geom: "ops.VarlenGeom"
    kbias: torch.Tensor
    kstart: torch.Tensor


def _varlen_of(segs: List[Segment], H: int, hd: int, device) -> Optional[Varlen]:
    if not ATTN_VARLEN or len(segs) < 2 or not ops.attn_varlen_supported(max(s.T for s in segs), hd):
        return None
    lengths = [s.T for s in segs for _ in range(s.B)]
    return Varlen(geom=ops.VarlenGeom.get(lengths, H, device), kbias=torch.cat([s.kbias.reshape(-1) for s in segs]),
                  kstart=torch.cat([s.kstart.reshape(-1) for s in segs]))


def _attn_fwd_segs(qkv, segs: List[Segment], H: int, hd: int, drop, save: bool = False, varlen: Optional[Varlen] = None):
    """-> (o, lse, keep masks).  With `save` and dropout on, the forward also hands back the keep decisions it made (one
    buffer per attention call) so that the backward does not re-hash them in both of its kernels."""
    if varlen is not None:
        o = torch.empty(qkv.shape[0], H * hd, dtype=BF16, device=qkv.device)
        if qkv.shape[0] > varlen.geom.rows:
            o[varlen.geom.rows:].zero_()           # alignment rows behind the last sequence (all padding)
        _, lse, mk = ops.attn_fwd_varlen(qkv, varlen.kbias, varlen.kstart, varlen.geom, hd, drop=drop, out=o, want_mask=save)
        return o, lse, mk
    if len(segs) == 1 and segs[0].B * segs[0].T == qkv.shape[0]:
        sg = segs[0]
        return ops.attn_fwd(qkv, sg.kbias, sg.kstart, sg.B, sg.T, H, hd, drop=drop, want_mask=True) if save else \
            ops.attn_fwd(qkv, sg.kbias, sg.kstart, sg.B, sg.T, H, hd, drop=drop) + (None,)
    o = torch.empty(qkv.shape[0], H * hd, dtype=BF16, device=qkv.device)
    o[segs[-1].row0 + segs[-1].B * segs[-1].T:].zero_()      # alignment rows behind the last segment (all padding)
    lses, masks = [], []
    for si, sg in enumerate(segs):
        r0, r1 = sg.row0, sg.row0 + sg.B * sg.T
        _, lse, mk = ops.attn_fwd(qkv[r0:r1], sg.kbias, sg.kstart, sg.B, sg.T, H, hd, drop=_seg_drop(drop, si), out=o[r0:r1],
                                  want_mask=True) if save else \
            ops.attn_fwd(qkv[r0:r1], sg.kbias, sg.kstart, sg.B, sg.T, H, hd, drop=_seg_drop(drop, si), out=o[r0:r1]) + (None,)
        lses.append(lse)
        masks.append(mk)
    return o, lses, masks


def _attn_bwd_segs(qkv, o, d_o, lse, segs: List[Segment], H: int, hd: int, drop, dmask=None, varlen: Optional[Varlen] = None):
    if varlen is not None:
        dqkv = torch.empty_like(qkv)
        if qkv.shape[0] > varlen.geom.rows:
            dqkv[varlen.geom.rows:].zero_()
        return ops.attn_bwd_varlen(qkv, o, d_o, varlen.kbias, varlen.kstart, lse, varlen.geom, hd, drop=drop, dqkv=dqkv, mask=dmask)
    if len(segs) == 1 and segs[0].B * segs[0].T == qkv.shape[0]:
        sg = segs[0]
        return ops.attn_bwd(qkv, o, d_o, sg.kbias, sg.kstart, lse, sg.B, sg.T, H, hd, drop=drop, mask=dmask)
    dqkv = torch.empty_like(qkv)
    dqkv[segs[-1].row0 + segs[-1].B * segs[-1].T:].zero_()
    for si, sg in enumerate(segs):
        r0, r1 = sg.row0, sg.row0 + sg.B * sg.T
        ops.attn_bwd(qkv[r0:r1], o[r0:r1], d_o[r0:r1], sg.kbias, sg.kstart, lse[si], sg.B, sg.T, H, hd,
                     drop=_seg_drop(drop, si), dqkv=dqkv[r0:r1], mask=dmask[si] if dmask is not None else None)
    return dqkv


def stack_forward(P: StackParams, x: torch.Tensor, mask: torch.Tensor, save: bool,
                  want_f32: bool = False, want_bf16: bool = True, drops: Optional[DropSites] = None,
                  segments: Optional[List[tuple]] = None):
    """x (B,T,d) fp32 residual stream, mask (B,T) fp32 0/1.  Returns (hf16 [M,d] bf16 | None,
    hf32 [M,d] fp32 | None, ctx | None) where hf = ln_f(h_L).
    segments: optional [(row0, B_k, T_k)] covering the rows of x in order, possibly followed by all-padding rows
    (ragged groups: x is then (1, M, d))."""
    B, T, d = x.shape
    M = B * T
    H = P.heads
    hd = d // H
    dev = x.device
    x = x.reshape(M, d).contiguous()
    mask = mask.to(F32).reshape(-1)
    segs = []
    for (r0, Bk, Tk) in (segments or [(0, B, T)]):
        kb, ks = ops.mask_bias(mask[r0:r0 + Bk * Tk].view(Bk, Tk))
        segs.append(Segment(row0=r0, B=Bk, T=Tk, kbias=kb, kstart=ks))
    # segments tile the rows in order; rows behind the last one (ragged layouts round the row count up to a multiple
    # of 64 so that the weight-gradient contractions stay on the fast GEMM path) are padding that no attention call
    # touches: their attention output / gradient rows are zero
    assert segs[0].row0 == 0 and all(a.row0 + a.B * a.T == b.row0 for a, b in zip(segs, segs[1:])) \
        and segs[-1].row0 + segs[-1].B * segs[-1].T <= M, "segments must tile the rows of x"
    varlen = _varlen_of(segs, H, hd, dev)
    ctx = StackCtx(B=B, T=T, segs=segs, drops=drops, varlen=varlen) if save else None
    dr = drops
    if dr is not None and dr.embd is not None:
        x = ops.dropout_f32(x, dr.embd)                       # embedding dropout (:707)
    for li, lp in enumerate(P.layers):
        a1 = torch.empty(M, d, dtype=BF16, device=dev)
        mean1 = torch.empty(M, dtype=F32, device=dev)
        rstd1 = torch.empty(M, dtype=F32, device=dev)
        ops.layernorm_fwd(x, lp.ln1_w, lp.ln1_b, y16=a1, mean=mean1, rstd=rstd1, eps=P.eps)
        qkv = torch.empty(M, 3 * d, dtype=BF16, device=dev)
        ops.gemm(a1, lp.w_qkv, M, 3 * d, d, b_kstrided=True, bias=lp.b_qkv, out_bf16=qkv)
        o, lse, dmask = _attn_fwd_segs(qkv, segs, H, hd, dr.attn[li] if dr else None, save=save, varlen=varlen)
        x1 = torch.empty(M, d, dtype=F32, device=dev)
        ops.gemm(o, lp.w_o, M, d, d, b_kstrided=True, bias=lp.b_o, resid=x, out_f32=x1,
                 drop=dr.resid_attn[li] if dr else None)
        a2 = torch.empty(M, d, dtype=BF16, device=dev)
        mean2 = torch.empty(M, dtype=F32, device=dev)
        rstd2 = torch.empty(M, dtype=F32, device=dev)
        ops.layernorm_fwd(x1, lp.ln2_w, lp.ln2_b, y16=a2, mean=mean2, rstd=rstd2, eps=P.eps)
        pre = torch.empty(M, 4 * d, dtype=BF16, device=dev) if save else None
        h = torch.empty(M, 4 * d, dtype=BF16, device=dev)
        factor = save and GELU_FACTOR and lp.w_gate is None       # `pre` then holds gelu'(pre), see GELU_FACTOR
        ops.gemm(a2, lp.w_fc, M, 4 * d, d, b_kstrided=True, bias=lp.b_fc, act=3 if factor else 1, pre_out=pre, out_bf16=h)
        gate = _geglu_gate(lp, a2, h, M, d)
        x2 = torch.empty(M, d, dtype=F32, device=dev)
        ops.gemm(h, lp.w_pr, M, d, 4 * d, b_kstrided=True, bias=lp.b_pr, resid=x1, out_f32=x2,
                 drop=dr.resid_mlp[li] if dr else None)
        if save:
            ctx.layers.append(LayerCtx(x=x, a1=a1, mean1=mean1, rstd1=rstd1, qkv=qkv, o=o, lse=lse, dmask=dmask, x1=x1, a2=a2,
                                       mean2=mean2, rstd2=rstd2, pre=pre, h=h, gate=gate, pre_is_factor=factor))
        x = x2
    hf16 = torch.empty(M, d, dtype=BF16, device=dev) if want_bf16 else None
    hf32 = torch.empty(M, d, dtype=F32, device=dev) if want_f32 else None
    meanf = torch.empty(M, dtype=F32, device=dev)
    rstdf = torch.empty(M, dtype=F32, device=dev)
    ops.layernorm_fwd(x, P.lnf_w, P.lnf_b, y16=hf16, y32=hf32, mean=meanf, rstd=rstdf, eps=P.eps)
    if save:
        ctx.xf, ctx.meanf, ctx.rstdf = x, meanf, rstdf
    return hf16, hf32, ctx


def stack_backward(P: StackParams, ctx: StackCtx, dhf: torch.Tensor,
                   on_layer_done: Optional[Callable[[int], None]] = None, dhf_row_map: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dhf [M,d] fp32 = gradient wrt ln_f output (with dhf_row_map int32 [M]: dhf holds the rows map[r] >= 0 only, the others are
    zero -- the LM head's selected loss rows, read in place by ln_f's backward).  Accumulates every parameter gradient of the stack
    into the flat gradient views and returns the gradient wrt the input embeddings [M,d] fp32.
    on_layer_done(i) is called after layer i's gradient kernels are enqueued (i = L for ln_f):
    the data-parallel reducer hooks its bucket launches there."""
    B, T = ctx.B, ctx.T
    M = B * T
    SideStream.rows_hint = M
    d, H = P.d, P.heads
    hd = d // H
    dev = dhf.device
    g = torch.empty(M, d, dtype=F32, device=dev)
    g16 = torch.empty(M, d, dtype=BF16, device=dev)
    dr = ctx.drops
    L = len(P.layers)
    # the bf16 gradient copy that enters layer i from above is consumed by its MLP c_proj: it carries that
    # site's residual-dropout mask; the fp32 residual-stream gradient never does
    # (every LayerNorm backward also leaves the column sums of its bf16 output = the bias gradient of the projection that
    # consumes it: b_pr of the layer above for ln_f / ln_1, b_o of the same layer for ln_2)
    ops.layernorm_bwd(dhf, ctx.xf, P.lnf_w, ctx.meanf, ctx.rstdf, P.g_lnf_w, P.g_lnf_b, g_in=None, dx=g, dx16=g16,
                      drop=dr.resid_mlp[L - 1] if dr else None, colsum16=P.layers[L - 1].g_b_pr, row_map=dhf_row_map)
    if on_layer_done:
        on_layer_done(len(P.layers))
    for i in range(len(P.layers) - 1, -1, -1):
        lp, c = P.layers[i], ctx.layers[i]
        # ---- MLP: x2 = x1 + gelu(a2 Wfc + bfc) Wpr + bpr ------------------------------------------
        d_pre = torch.empty(M, 4 * d, dtype=BF16, device=dev)
        d_gate = None
        if c.gate is None:    # dgrad * gelu', the c_fc bias gradient (column sums of d_pre) folded into its epilogue
            ops.gemm_dgrad_gelu_colsum(g16, lp.w_pr, M, 4 * d, d, c.pre, d_pre, lp.g_b_fc, ldb=d,
                                       act_in_is_factor=c.pre_is_factor)
        else:                                                                                 # GEGLU: h = gelu(pre) * gate
            ops.gemm(g16, lp.w_pr, M, 4 * d, d, ldb=d, out_bf16=d_pre)                        # d_h
            d_pre, d_gate = ops.geglu_bwd(d_pre, c.pre, c.gate)
        SideStream.fork(lambda c=c, lp=lp, g16=g16: _wgrad(c.h, g16, 4 * d, d, M, lp.g_w_pr), g16)
        # gradients wrt the LayerNorm outputs leave their dgrad GEMMs as bf16, as autocast leaves them in the reference
        # (the gradient of a bf16 addmm input is bf16): half the bytes out of the GEMM and into the LayerNorm backward
        if d_gate is None:
            d_a2 = _dgrad_to_ln(d_pre, lp.w_fc, M, d, 4 * d, 4 * d)
        else:                     # d_a2 = d_pre @ Wfc^T + d_gate @ Wgate  (fp32: the second product accumulates)
            d_a2 = torch.empty(M, d, dtype=F32, device=dev)
            ops.gemm(d_pre, lp.w_fc, M, d, 4 * d, ldb=4 * d, out_f32=d_a2)
            ops.gemm(d_gate, lp.w_gate, M, d, 4 * d, b_kstrided=True, ldb=d, out_f32=d_a2, accumulate=True)
            SideStream.fork(lambda c=c, lp=lp, d_gate=d_gate: (_wgrad(d_gate, c.a2, 4 * d, d, M, lp.g_w_gate),
                                                              ops.colsum_bf16(d_gate, M, 4 * d, lp.g_b_gate)), d_gate)
        SideStream.fork(lambda c=c, lp=lp, d_pre=d_pre, geglu=d_gate is not None: (
            _wgrad(c.a2, d_pre, d, 4 * d, M, lp.g_w_fc),
            ops.colsum_bf16(d_pre, M, 4 * d, lp.g_b_fc) if geglu else None), d_pre)
        g1 = torch.empty(M, d, dtype=F32, device=dev)
        g1_16 = torch.empty(M, d, dtype=BF16, device=dev)
        ops.layernorm_bwd(d_a2, c.x1, lp.ln2_w, c.mean2, c.rstd2, lp.g_ln2_w, lp.g_ln2_b, g_in=g, dx=g1, dx16=g1_16,
                          drop=dr.resid_attn[i] if dr else None, colsum16=lp.g_b_o)
        # ---- attention: x1 = x + attn(a1 Wqkv + bqkv) Wo + bo -----------------------------------------
        d_o = torch.empty(M, d, dtype=BF16, device=dev)
        ops.gemm(g1_16, lp.w_o, M, d, d, ldb=d, out_bf16=d_o)
        SideStream.fork(lambda c=c, lp=lp, g1_16=g1_16: _wgrad(c.o, g1_16, d, d, M, lp.g_w_o), g1_16)
        dqkv = _attn_bwd_segs(c.qkv, c.o, d_o, c.lse, ctx.segs, H, hd, dr.attn[i] if dr else None, dmask=c.dmask, varlen=ctx.varlen)
        d_a1 = _dgrad_to_ln(dqkv, lp.w_qkv, M, d, 3 * d, 3 * d)
        SideStream.fork(lambda c=c, lp=lp, dqkv=dqkv: (_wgrad(c.a1, dqkv, d, 3 * d, M, lp.g_w_qkv),
                                                      ops.colsum_bf16(dqkv, M, 3 * d, lp.g_b_qkv)), dqkv)
        g0 = torch.empty(M, d, dtype=F32, device=dev)
        g0_16 = torch.empty(M, d, dtype=BF16, device=dev) if i > 0 else None
        # layer 0 has no bf16 copy to make: its LayerNorm backward applies the EMBEDDING dropout's mask to the fp32 result instead
        # (the backward of `x = drop(x)` at the top of stack_forward), which used to be a pass of its own over [M, d]
        ops.layernorm_bwd(d_a1, c.x, lp.ln1_w, c.mean1, c.rstd1, lp.g_ln1_w, lp.g_ln1_b, g_in=g1, dx=g0, dx16=g0_16,
                          drop=(dr.resid_mlp[i - 1] if i > 0 else dr.embd) if dr else None,
                          colsum16=P.layers[i - 1].g_b_pr if i > 0 else None)
        g, g16 = g0, g0_16
        if on_layer_done:
            on_layer_done(i)
    SideStream.join(dev)          # every parameter gradient of the stack is ordered before what follows on this stream
    return g                      # (already through the embedding dropout's mask: layer 0's LayerNorm backward above)


# ------------------------------------------------------------------------------------------------------
# LM head + masked cross-entropy (gato_policy.py:172-186)
# ------------------------------------------------------------------------------------------------------
class KVDecoder:
    """Incremental inference over ONE sequence (B = 1, no padding): SURVEY.md 8(f) rank 2.  The reference's
    predict_* helpers (gato_policy.py:434-614) re-run the whole forward -- T positions through the stack and a (T, V)
    logits tensor -- for every generated token.  Here the rows already seen keep their q/k/v in a per-layer [cap, 3d]
    bf16 buffer (the layout the attention kernel reads): extend() pushes only the NEW rows through LayerNorm and the
    GEMMs, appends their q/k/v, runs the attention kernel over the buffer (B = 1: cheap) and continues with the new
    rows.  Exact as long as the window does not slide: a causal row never depends on later rows.  When the context is
    full the reference truncates on the left and recomputes every remaining row WITHOUT the dropped ones, so the
    caller resets and re-primes on the truncated window (same semantics, see GatoPolicy._decode_tokens)."""

    def __init__(self, P: StackParams, cap: int, device):
        self.P, self.cap, self.n = P, int(cap), 0
        self.qkv = [torch.zeros(self.cap, 3 * P.d, dtype=BF16, device=device) for _ in P.layers]
        self.kbias = torch.zeros(1, self.cap, dtype=F32, device=device)       # all positions real: no key bias
        self.kstart = torch.zeros(1, dtype=torch.int32, device=device)
        self.pos = torch.zeros(1, dtype=torch.int32, device=device)           # device-side index of the next row
        self.fast_row = P.d % 8 == 0 and 4 * P.d <= 3072                      # neko_gemv_bf16 limits (K <= 3072)

    def reset(self) -> None:
        self.n = 0
        self.pos.zero_()

    # ---- single-row step with a device-side position (HIP-graph capturable) -------------------------------------
    def _one_row(self, x: torch.Tensor) -> torch.Tensor:
        """x (1, d) fp32 at position *self.pos -> ln_f(hidden) (1, d) bf16.  No host-side shape depends on the position
        (the attention call reads it from device memory), so this body can be captured once; the products are weight
        streams (neko_gemv_bf16), not tiled GEMMs."""
        P = self.P
        d, H = P.d, P.heads
        hd = d // H
        dev = x.device
        for li, lp in enumerate(P.layers):
            a1 = torch.empty(1, d, dtype=BF16, device=dev)
            ops.layernorm_fwd(x, lp.ln1_w, lp.ln1_b, y16=a1, eps=P.eps)
            row = torch.empty(1, 3 * d, dtype=BF16, device=dev)
            ops.gemv(a1, lp.w_qkv, 1, 3 * d, d, b_kstrided=True, bias=lp.b_qkv, out_bf16=row)
            o = torch.empty(1, d, dtype=BF16, device=dev)
            ops.attn_decode(self.qkv[li], row, self.pos, o, H, hd)
            x1 = torch.empty(1, d, dtype=F32, device=dev)
            ops.gemv(o, lp.w_o, 1, d, d, b_kstrided=True, bias=lp.b_o, resid=x, out_f32=x1)
            a2 = torch.empty(1, d, dtype=BF16, device=dev)
            ops.layernorm_fwd(x1, lp.ln2_w, lp.ln2_b, y16=a2, eps=P.eps)
            h = torch.empty(1, 4 * d, dtype=BF16, device=dev)
            ops.gemv(a2, lp.w_fc, 1, 4 * d, d, b_kstrided=True, bias=lp.b_fc, act=1, out_bf16=h)
            _geglu_gate(lp, a2, h, 1, d)
            x2 = torch.empty(1, d, dtype=F32, device=dev)
            ops.gemv(h, lp.w_pr, 1, d, 4 * d, b_kstrided=True, bias=lp.b_pr, resid=x1, out_f32=x2)
            x = x2
        hf16 = torch.empty(1, d, dtype=BF16, device=dev)
        ops.layernorm_fwd(x, P.lnf_w, P.lnf_b, y16=hf16, eps=P.eps)
        return hf16

    def capture_greedy_step(self, Hp: "HeadParams", table: torch.Tensor, start: int, end: int) -> None:
        """Capture ONE greedy decode step as a HIP graph: embedding row g_x at position *pos -> stack -> LM head on that
        row -> argmax over [start, end] -> g_logits / g_token, then g_x <- embedding of the chosen token and pos += 1.
        At B = 1 a step is ~70 launches and launch-bound; replaying the graph costs one.  Call before priming (the
        warm-up runs write cache row 0, which the prime overwrites)."""
        d = self.P.d
        dev = self.qkv[0].device
        self.g_x = torch.zeros(1, d, dtype=F32, device=dev)
        self.g_logits = torch.zeros(end - start + 1, dtype=F32, device=dev)
        self.g_token = torch.zeros((), dtype=torch.int64, device=dev)

        def body():
            h = self._one_row(self.g_x)
            lg = lm_head_rows(Hp, h)[0, start:(end + 1)]
            tok = torch.argmax(lg, dim=-1) + start
            self.g_logits.copy_(lg)
            self.g_token.copy_(tok)
            self.g_x.copy_(torch.index_select(table, 0, tok.reshape(1)))     # no host read: capturable
            self.pos.add_(1)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up outside capture (allocator, lazy module loads)
                self.pos.zero_()
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.pos.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()
        self.pos.zero_()

    def replay_greedy_step(self):
        """One captured step; the caller has put the embedding of the previous token into g_x (the graph itself does
        that for every step after the first).  Returns clones of (logits, token) -- device tensors, no host sync."""
        if self.n >= self.cap:
            raise ValueError("KVDecoder: capacity exhausted (slide the window on the eager path)")
        self.graph.replay()
        self.n += 1
        return self.g_logits.clone(), self.g_token.clone()

    @torch.no_grad()
    def extend(self, x_new: torch.Tensor) -> torch.Tensor:
        """x_new (n, d) fp32 embeddings of the next n positions -> ln_f(hidden) of those positions, bf16 (n, d)."""
        P = self.P
        d, H = P.d, P.heads
        hd = d // H
        x = x_new.reshape(-1, d).to(F32).contiguous()
        n, dev = x.shape[0], x.device
        n0, T = self.n, self.n + n
        if T > self.cap:
            raise ValueError(f"KVDecoder: {T} positions exceed the capacity {self.cap}")
        if n == 1 and self.fast_row:                # the decode step proper: weight-streaming products, 1-query attention
            hf16 = self._one_row(x)                 # reads / appends at *pos == n0
            self.n = T
            self.pos.fill_(T)
            return hf16
        kb = self.kbias[:, :T].contiguous()
        for li, lp in enumerate(P.layers):
            a1 = torch.empty(n, d, dtype=BF16, device=dev)
            ops.layernorm_fwd(x, lp.ln1_w, lp.ln1_b, y16=a1, eps=P.eps)
            ops.gemm(a1, lp.w_qkv, n, 3 * d, d, b_kstrided=True, bias=lp.b_qkv, out_bf16=self.qkv[li][n0:T], ldcb=3 * d)
            o_all, _ = ops.attn_fwd(self.qkv[li][:T], kb, self.kstart, 1, T, H, hd)
            o = o_all[n0:T]
            x1 = torch.empty(n, d, dtype=F32, device=dev)
            ops.gemm(o, lp.w_o, n, d, d, b_kstrided=True, bias=lp.b_o, resid=x, out_f32=x1)
            a2 = torch.empty(n, d, dtype=BF16, device=dev)
            ops.layernorm_fwd(x1, lp.ln2_w, lp.ln2_b, y16=a2, eps=P.eps)
            h = torch.empty(n, 4 * d, dtype=BF16, device=dev)
            ops.gemm(a2, lp.w_fc, n, 4 * d, d, b_kstrided=True, bias=lp.b_fc, act=1, out_bf16=h)
            _geglu_gate(lp, a2, h, n, d)
            x2 = torch.empty(n, d, dtype=F32, device=dev)
            ops.gemm(h, lp.w_pr, n, d, 4 * d, b_kstrided=True, bias=lp.b_pr, resid=x1, out_f32=x2)
            x = x2
        hf16 = torch.empty(n, d, dtype=BF16, device=dev)
        ops.layernorm_fwd(x, P.lnf_w, P.lnf_b, y16=hf16, eps=P.eps)
        self.n = T
        self.pos.fill_(T)                           # device-side position of the next row
        return hf16


def lm_head_rows(Hp: HeadParams, h16: torch.Tensor) -> torch.Tensor:
    """fp32 logits (n, V) of a few rows (decode: the last position only) -- never the (T, V) tensor."""
    n, d = h16.shape
    out = torch.empty(n, Hp.Vpad, dtype=F32, device=h16.device)
    if n <= 8 and d <= 3072 and d % 8 == 0:
        ops.gemv(h16.contiguous(), Hp.w, n, Hp.Vpad, d, b_kstrided=False, ldw=d, out_f32=out)     # streams the table once
    else:
        ops.gemm(h16.contiguous(), Hp.w, n, Hp.Vpad, d, ldb=d, out_f32=out, ldcf=Hp.Vpad)
    return out[:, :Hp.V]


def shift_targets(tokens: torch.Tensor, tmask: torch.Tensor, pmask: torch.Tensor):
    """Position t predicts token t+1 (gato_policy.py:176-181): returns (target [M] int64,
    sel [M] fp32 0/1 = pad_mask[:, :-1]*target_mask[:, 1:] with the last position 0, count)."""
    B, T = tokens.shape
    target = torch.zeros_like(tokens)
    target[:, :-1] = tokens[:, 1:]
    sel = torch.zeros(B, T, dtype=F32, device=tokens.device)
    sel[:, :-1] = pmask[:, :-1] * tmask[:, 1:]
    sel = (sel > 0).to(F32)
    return target.reshape(-1).contiguous(), sel.reshape(-1).contiguous(), sel.sum()


def lm_head_logits(Hp: HeadParams, hf16: torch.Tensor) -> torch.Tensor:
    """Full logits [M,V] fp32 (predict_token, gato_policy.py:172)."""
    M, d = hf16.shape
    logits = torch.empty(M, Hp.V, dtype=F32, device=hf16.device)
    ops.gemm(hf16, Hp.w, M, Hp.V, d, ldb=d, out_f32=logits, ldcf=Hp.V)
    return logits


def lm_head_loss(Hp: HeadParams, hf16: torch.Tensor, target: torch.Tensor, sel: Optional[torch.Tensor], count: Optional[torch.Tensor],
                 want_grad: bool, chunk_rows: int = 4096, weight: Optional[torch.Tensor] = None):
    """Chunked LM head + CE.  Never materialises (B,T,V) fp32 logits nor the (N,V) gathered copy
    (gato_policy.py:184-185): per chunk of rows it runs logits GEMM -> CE (loss rows + bf16 dlogits).
    Returns (loss scalar tensor, dlogits bf16 [M,Vpad] | None)."""
    M, d = hf16.shape
    dev = hf16.device
    if weight is None:
        weight = sel / count.clamp(min=1.0)
    loss_rows = torch.empty(M, dtype=F32, device=dev)
    R = min(chunk_rows if want_grad else min(chunk_rows, 4096), M)      # without a gradient the logits live in an R-row scratch buffer
    # the GEMM writes bf16 logits straight into the gradient buffer and the CE kernel turns them into dlogits in place
    # (fp32 logits never reach HBM); without a gradient one chunk-sized scratch buffer is reused
    dlogits = torch.empty(M if want_grad else R, Hp.Vpad, dtype=BF16, device=dev)
    for r0 in range(0, M, R):
        r1 = min(M, r0 + R)
        n = r1 - r0
        z = dlogits[r0:r1] if want_grad else dlogits[:n]
        # N = Vpad, not V: the padded rows of the table are zero, the CE kernel masks columns >= V, and every tile is
        # then an interior tile with 16-B aligned rows (fast GEMM epilogue)
        ops.gemm(hf16[r0:r1], Hp.w, n, Hp.Vpad, d, ldb=d, out_bf16=z, ldcb=Hp.Vpad)
        ops.ce_bf16_inplace(z, Hp.V, Hp.Vpad, target[r0:r1], weight[r0:r1], loss_row=loss_rows[r0:r1],
                            want_grad=want_grad)
    loss = (loss_rows * weight).sum()       # (not torch.dot: a BLAS call inside a captured step)
    return loss, (dlogits if want_grad else None)


_SEL_WEIGHT: dict = {}


def _selected_weight(n: int, npad: int, dev) -> torch.Tensor:
    """fp32 [npad]: 1 / n on the n loss rows, 0 on the padding rows -- the weight of every selected row (mean over the loss positions,
    gato_policy.py:186).  A constant of (n, npad): kept per device (README-size steps are made of launches; this one replaced six)."""
    if torch.cuda.is_current_stream_capturing():        # a captured step keeps its own copy inside the graph's memory
        w = torch.zeros(npad, dtype=F32, device=dev)
        w[:n] = 1.0 / float(max(n, 1))
        return w
    key = (n, npad, str(dev))
    w = _SEL_WEIGHT.get(key)
    if w is None:
        if len(_SEL_WEIGHT) > 256:
            _SEL_WEIGHT.clear()
        w = torch.zeros(npad, dtype=F32, device=dev)
        w[:n] = 1.0 / float(max(n, 1))
        _SEL_WEIGHT[key] = w
    return w


def lm_head_loss_selected(Hp: HeadParams, hf16: torch.Tensor, target: Optional[torch.Tensor], idx: torch.Tensor, n: int,
                          want_grad: bool, chunk_rows: int = 4096, tokens_flat: Optional[torch.Tensor] = None,
                          tgt_idx: Optional[torch.Tensor] = None):
    """Same loss as lm_head_loss, but only the n selected positions (idx: int32 flat row indices, known on the
    host from the packing descriptors) go through the LM head -- the HIP counterpart of the reference's
    boolean-mask gather (gato_policy.py:183-185), done before the GEMM instead of after it.
    The targets of the selected rows come either from `target` (the shifted token tensor of shift_targets) or, when the packer handed
    over tgt_idx = idx + 1, straight from the flat token tensor (position t predicts token t + 1, gato_policy.py:176-181).
    Returns (loss, hsel bf16 [npad,d], dlogits bf16 [npad,Vpad] | None)."""
    # multiple of 64: keeps the wgrad contraction on the fast GEMM path; of 256 once the batch is large: whole 256-row tiles for
    # the dH product and a contraction of whole loop trips for dW (gemm_a16.hip); the padding rows are zero rows with weight 0
    npad = max(64, (n + 63) // 64 * 64) if n < 2048 else (n + 255) // 256 * 256
    dev = hf16.device
    hsel = ops.gather_rows_bf16(hf16, idx, n, npad)
    tsel = torch.zeros(npad, dtype=torch.int64, device=dev)
    if tgt_idx is not None:
        torch.index_select(tokens_flat, 0, tgt_idx[:n], out=tsel[:n])
    else:
        tsel[:n] = target.index_select(0, idx[:n].long())
    loss, dlogits = lm_head_loss(Hp, hsel, tsel, None, None, want_grad, chunk_rows, weight=_selected_weight(n, npad, dev))
    return loss, hsel, dlogits


#: ln_f's backward reads the LM head's gradient rows of the loss positions in place through a row map (neko_layernorm_bwd_rows)
#: instead of a zero-filled [M, d] expansion (fill + scatter + 2/3 of the rows read as zeros: ~0.12 ms per m-mix step).
#: NEKO_LNF_ROWS=0 returns to the expansion.
LNF_ROWS = os.environ.get("NEKO_LNF_ROWS", "1") != "0"


def lm_head_backward_selected(Hp: HeadParams, hsel: torch.Tensor, dlogits: torch.Tensor, grad_out: torch.Tensor,
                              idx: torch.Tensor, n: int, M: int, row_map: Optional[torch.Tensor] = None):
    """Backward of lm_head_loss_selected -> (dhf, row_map): either the gradient rows scattered back into a zero [M,d] buffer and
    None, or the compact rows [npad, d] and the int32 [M] map row -> compact row (-1: zero), see LNF_ROWS."""
    dsel = lm_head_backward(Hp, hsel, dlogits, grad_out)
    if LNF_ROWS:
        if row_map is None:          # (callers that pack on the host hand the map over with the loss rows: GatoPolicy._prepare)
            row_map = torch.full((M,), -1, dtype=torch.int32, device=hsel.device)
            row_map.index_copy_(0, idx[:n].long(), torch.arange(n, dtype=torch.int32, device=hsel.device))
        return dsel, row_map
    dhf = torch.zeros(M, hsel.shape[1], dtype=F32, device=hsel.device)
    ops.scatter_rows_f32(dsel, idx, n, dhf)
    return dhf, None


def lm_head_backward(Hp: HeadParams, hf16: torch.Tensor, dlogits: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """dH = (dlogits @ W) * grad_out  [M,d] fp32;  dW += (dlogits^T @ H) * grad_out.
    grad_out stays on the device (GEMM alpha_dev): no host sync in backward."""
    M, d = hf16.shape
    go = grad_out.reshape(1).to(F32)
    dhf = torch.empty(M, d, dtype=F32, device=hf16.device)
    sk, kps = ops.pick_splitk(M, d, Hp.Vpad)          # few output tiles, K = vocabulary: fill the CUs
    ops.gemm(dlogits, Hp.w, M, d, Hp.Vpad, b_kstrided=True, ldb=d, out_f32=dhf, alpha_dev=go, splitk=sk, k_per_split=kps)
    # dW on the side stream: it overlaps the start of the stack's backward (joined at the end of stack_backward, or by
    # the data-parallel reducer before it reads the range)
    SideStream.fork(lambda: _wgrad(dlogits, hf16, Hp.Vpad, d, M, Hp.g_w, alpha_dev=go), dlogits, hf16, go)
    return dhf
