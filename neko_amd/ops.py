"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/neko_hip.h).

PyTorch is only plumbing here: it owns device memory and the stream; every computation is a
libneko_hip.so kernel.  All functions enqueue on ``torch.cuda.current_stream()`` and return
immediately.  Shape/dtype preconditions are asserted before the call.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib

BF16 = torch.bfloat16
SAFE_TRANSPOSE = int(os.environ.get("NEKO_GEMM_SAFE_T", "0"))


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Python
    Stream object through four layers of device-index helpers (~9 us, ~400 calls per step); the raw getter is one
    C call."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, dtype, name: str):
    assert t.is_cuda, f"{name} must be a device tensor"
    assert t.dtype == dtype, f"{name} must be {dtype}, got {t.dtype}"


class Drop:
    """One dropout site: thr = round(p*256) (0 = off), key = 32-bit site key, scale = 256/(256-thr)."""
    __slots__ = ("thr", "key", "scale")

    def __init__(self, p: float, key: int):
        self.thr = max(0, min(255, int(round(p * 256))))
        self.key = key & 0xFFFFFFFF
        self.scale = 256.0 / (256 - self.thr)


def _drop(d):
    return (0, 0, 1.0) if (d is None or d.thr == 0) else (d.thr, d.key, d.scale)


def mix32(x: int) -> int:
    """Host-side 32-bit mixer (lowbias32) used to derive per-site dropout keys from (seed, step, site)."""
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_f32(x, drop, out=None):
    _chk(x, torch.float32, "x")
    x = x.contiguous()
    y = torch.empty_like(x) if out is None else out
    _lib.call("neko_dropout_f32", _p(x), _p(y), x.numel(), *_drop(drop), _stream())
    return y


def pick_splitk(M: int, N: int, K: int, target_blocks: int = 256) -> tuple:
    """Split-K factor for skinny-output / long-K GEMMs (weight gradients, LM-head dH).  Split-K launches run the
    256x256 tile, one block per CU (gemm_glds.hip: launch()), so the model is: rounds of `target_blocks` tiles,
    each k-step of 32 costing ~1.05 us per round, plus the fixed-order workspace reduce (write + read at ~4 TB/s).
    Returns (splitk, k_per_split) with k_per_split a multiple of 64 (of 128 when K is: every slice is then a whole number of
    trips of the long-contraction loop, gemm_a16.hip), or (1, 0)."""
    if K < 4096:
        return 1, 0
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles >= 4 * target_blocks:            # >= 4 rounds: the partial last round costs less than the reduce pass
        return 1, 0
    best_t, best = None, (1, 0)
    for sk in range(1, 33):
        q = 128 if K % 128 == 0 else 64
        kps = ((K + sk - 1) // sk + q - 1) // q * q
        if (K + kps - 1) // kps != sk:
            continue
        if sk > 1 and kps < 1024:
            break
        rounds = (tiles * sk + target_blocks - 1) // target_blocks
        t = rounds * kps * (1.05e-6 / 32)
        if sk > 1:
            t += 2.0 * sk * M * N * 4 / 4e12 + 4e-6
        if best_t is None or t < best_t * 0.97:          # prefer the smaller split on near-ties
            best_t, best = t, ((sk, kps) if sk > 1 else (1, 0))
    return best


def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, a_kstrided=False, b_kstrided=False,
         lda: Optional[int] = None, ldb: Optional[int] = None, alpha: float = 1.0,
         alpha_dev: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         resid: Optional[torch.Tensor] = None, act: int = 0, act_in: Optional[torch.Tensor] = None,
         pre_out: Optional[torch.Tensor] = None, out_f32: Optional[torch.Tensor] = None, ldcf: Optional[int] = None,
         accumulate: bool = False, out_bf16: Optional[torch.Tensor] = None, ldcb: Optional[int] = None,
         splitk: int = 1, k_per_split: int = 0, safe_transpose: Optional[int] = None,
         deterministic_splitk: bool = True, drop=None) -> None:
    """C[M,N] = alpha * opA(A) @ opB(B) (+bias)(act)(+resid)(+C).  See neko_gemm_bf16 in include/neko_hip.h."""
    _chk(A, BF16, "A"); _chk(B, BF16, "B")
    if lda is None:
        lda = A.stride(0) if A.dim() == 2 else (M if a_kstrided else K)
    if ldb is None:
        ldb = B.stride(0) if B.dim() == 2 else (N if b_kstrided else K)
    if bias is not None:
        _chk(bias, torch.float32, "bias"); assert bias.numel() >= N
    ldr = 0
    if resid is not None:
        _chk(resid, torch.float32, "resid"); ldr = resid.stride(0) if resid.dim() == 2 else N
    ldact = ldpre = 0
    if act_in is not None:
        _chk(act_in, BF16, "act_in"); ldact = act_in.stride(0) if act_in.dim() == 2 else N
    if pre_out is not None:
        _chk(pre_out, BF16, "pre_out"); ldpre = pre_out.stride(0) if pre_out.dim() == 2 else N
    if out_f32 is not None:
        _chk(out_f32, torch.float32, "out_f32")
        if ldcf is None:
            ldcf = out_f32.stride(0) if out_f32.dim() == 2 else N
    if out_bf16 is not None:
        _chk(out_bf16, BF16, "out_bf16")
        if ldcb is None:
            ldcb = out_bf16.stride(0) if out_bf16.dim() == 2 else N
    if alpha_dev is not None:
        _chk(alpha_dev, torch.float32, "alpha_dev")
    st = SAFE_TRANSPOSE if safe_transpose is None else safe_transpose
    ws = None
    if splitk > 1 and deterministic_splitk and N % 4 == 0:
        ws = torch.empty(splitk * M * N, dtype=torch.float32, device=A.device)
    _lib.call("neko_gemm_bf16", _p(A), lda, int(a_kstrided), _p(B), ldb, int(b_kstrided), M, N, K, float(alpha),
              _p(alpha_dev), _p(bias), _p(resid), ldr, act, _p(act_in), ldact, _p(pre_out), ldpre, _p(out_f32),
              ldcf or 0, int(accumulate), _p(out_bf16), ldcb or 0, splitk, k_per_split, _p(ws), *_drop(drop), st, _stream())


def gemm_dgrad_gelu_colsum(dY: torch.Tensor, W: torch.Tensor, M: int, N: int, K: int, act_in: torch.Tensor,
                           out_bf16: torch.Tensor, colsum_out: torch.Tensor, ldb: Optional[int] = None,
                           act_in_is_factor: bool = False) -> None:
    """out_bf16[M,N] = (dY[M,K] @ W[N,K]^T) * gelu'(act_in); colsum_out[N] += its column sums (the c_fc bias gradient),
    one launch (+ a 10 us band reduction).  act_in_is_factor: act_in already holds gelu'(pre) (forward ran with act=3).
    See neko_gemm_dgrad_gelu_colsum in include/neko_hip.h."""
    _chk(dY, BF16, "dY"); _chk(W, BF16, "W"); _chk(act_in, BF16, "act_in"); _chk(out_bf16, BF16, "out_bf16")
    _chk(colsum_out, torch.float32, "colsum_out"); assert colsum_out.numel() >= N
    ws = torch.empty(int(_lib.load().neko_gemm_colsum_ws_floats(M, N)), dtype=torch.float32, device=dY.device)
    _lib.call("neko_gemm_dgrad_gelu_colsum", _p(dY), dY.stride(0) if dY.dim() == 2 else K, _p(W),
              ldb if ldb is not None else (W.stride(0) if W.dim() == 2 else K), M, N, K, _p(act_in),
              act_in.stride(0) if act_in.dim() == 2 else N, int(bool(act_in_is_factor)), _p(out_bf16),
              out_bf16.stride(0) if out_bf16.dim() == 2 else N, _p(ws), _p(colsum_out), _stream())


def layernorm_fwd(x, gamma, beta, y16=None, y32=None, mean=None, rstd=None, eps: float = 1e-5):
    _chk(x, torch.float32, "x")
    M, d = x.shape[0], x.shape[1]
    assert x.is_contiguous()
    _lib.call("neko_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y16), _p(y32), _p(mean), _p(rstd), M, d,
              float(eps), _stream())


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, g_in=None, dx=None, dx16=None, accumulate=True, drop=None,
                  colsum16=None, row_map=None):
    """dy fp32 or bf16 [M,d] (neko_layernorm_bwd / neko_layernorm_bwd_bf16dy).  colsum16: fp32 [d] that receives (+=) the
    column sums of dx16 -- the bias gradient of the Linear that consumes dx16.  row_map (int32 [M]): dy is fp32 [n, d] and holds the
    gradient of row r at dy[row_map[r]], zero where row_map[r] < 0 (neko_layernorm_bwd_rows)."""
    assert dy.is_cuda and dy.dtype in (torch.float32, BF16) and dy.is_contiguous(), "dy must be a contiguous f32 / bf16 device tensor"
    M, d = x.shape[0], x.shape[1]
    nblk = _lib.load().neko_layernorm_bwd_blocks(M)
    ws = torch.empty(nblk * 3 * d, dtype=torch.float32, device=x.device)
    if colsum16 is not None:
        _chk(colsum16, torch.float32, "colsum16"); assert dx16 is not None and colsum16.numel() >= d
    if row_map is not None:
        _chk(dy, torch.float32, "dy"); _chk(row_map, torch.int32, "row_map")
        assert row_map.numel() == M and row_map.is_contiguous() and dy.shape[1] == d
        _lib.call("neko_layernorm_bwd_rows", _p(dy), _p(row_map), _p(x), _p(gamma), _p(mean), _p(rstd), _p(g_in), _p(dx), _p(dx16),
                  _p(dgamma), _p(dbeta), int(accumulate), _p(ws), M, d, *_drop(drop), _p(colsum16), _stream())
        return
    _lib.call("neko_layernorm_bwd_bf16dy" if dy.dtype == BF16 else "neko_layernorm_bwd", _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(g_in), _p(dx), _p(dx16),
              _p(dgamma), _p(dbeta), int(accumulate), _p(ws), M, d, *_drop(drop), _p(colsum16), _stream())


def mask_bias(mask: torch.Tensor):
    """(B,T) 0/1 float mask -> additive key bias (B,T) f32 and first-real-key index (B,) int32."""
    _chk(mask, torch.float32, "mask")
    B, T = mask.shape
    mask = mask.contiguous()
    kb = torch.empty(B, T, dtype=torch.float32, device=mask.device)
    ks = torch.empty(B, dtype=torch.int32, device=mask.device)
    _lib.call("neko_mask_bias", _p(mask), _p(kb), _p(ks), B, T, _stream())
    return kb, ks


def gemm_set_mainloop(mode: int) -> int:
    """1 / 0: send every launch the hand-placed long-contraction main loop (gemm_a16.hip) can serve to it / none, -1: the
    built-in per-shape choice; returns the previous mode (neko_gemm_set_mainloop)."""
    return int(_lib.load().neko_gemm_set_mainloop(int(mode)))


MAINLOOP_NAMES = {0: "glds", 1: "a16", 2: "b16", 3: "glds64", 4: "bf16", 5: "p16", -1: "none"}


def gemm_last_mainloop() -> int:
    """Which main loop served this thread's last `gemm` launch (neko_gemm_last_mainloop; names: MAINLOOP_NAMES)."""
    return int(_lib.load().neko_gemm_last_mainloop())


def attn_set_path(mode: int) -> int:
    """0 = automatic (head-resident kernels when hd = 32 and T <= 1024; their backward in one pass for 256 < T <= 512, as two kernels
    otherwise), 1 = always the streaming kernels, 2 / 3 = head-resident with the two-kernel (bit-reproducible) / one-pass backward at
    every length; returns the previous mode (neko_attn_set_path)."""
    return int(_lib.load().neko_attn_set_path(int(mode)))


def attn_fwd(qkv, kbias, kstart, B, T, H, hd, drop=None, out=None, want_mask=False, mask_buf=None):
    """out (optional): a contiguous [B*T, H*hd] bf16 row range to write into (ragged groups share one buffer).
    want_mask: with dropout on, also return the keep-mask buffer the backward of this call can reuse (int32 tensor, or
    None when the schedule in use does not exchange masks): (out, lse, mask) instead of (out, lse).
    mask_buf (optional, tests): a caller-provided int32 buffer of neko_attn_mask_dwords elements to use as that mask buffer."""
    _chk(qkv, BF16, "qkv")
    assert qkv.is_contiguous() and qkv.shape[0] == B * T
    if out is None:
        out = torch.empty(B * T, H * hd, dtype=BF16, device=qkv.device)
    else:
        _chk(out, BF16, "out"); assert out.is_contiguous() and out.shape == (B * T, H * hd)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    mask = None
    if want_mask and drop is not None and drop.thr > 0:
        n = int(_lib.load().neko_attn_mask_dwords(B, T, H, hd))
        if n > 0:
            if mask_buf is not None:
                _chk(mask_buf, torch.int32, "mask_buf"); assert mask_buf.numel() == n
            mask = mask_buf if mask_buf is not None else torch.empty(n, dtype=torch.int32, device=qkv.device)
    _lib.call("neko_attn_fwd", _p(qkv), _p(kbias), _p(kstart), _p(out), _p(lse), B, T, H, hd, *_drop(drop), _p(mask), _stream())
    return (out, lse, mask) if want_mask else (out, lse)


class VarlenGeom:
    """Packed sequences for the single-launch attention (neko_attn_*_varlen): lengths (host ints, in row order) -> row offsets,
    keep-mask offsets and sizes; the two offset arrays live on the device.

    Geometries are built once per (lengths, H, device) and kept (`VarlenGeom.get`): the two uploads go from PINNED host buffers
    that the object owns, so a training step that is being captured into a HIP graph either finds the device arrays ready or
    records copies whose source stays alive and constant for every replay -- never a copy from a temporary pageable tensor
    (that form stalls the host in eager mode and leaves a dangling host pointer in a captured graph)."""

    _cache: "dict" = {}
    _CACHE_MAX = 256          # unpinned entries
    _PINNED_WARN = 64         # warn once when captured steps hold this many
    _warned = False

    def __init__(self, lengths, H: int, device):
        self.lengths = [int(t) for t in lengths]
        self.nseq, self.Tmax, self.H = len(self.lengths), max(self.lengths), int(H)
        off, moff = [0], [0]
        for t in self.lengths:
            off.append(off[-1] + t)
            nb = (t + 31) // 32
            moff.append(moff[-1] + self.H * nb * nb * 32)
        self.rows, self.mask_dwords = off[-1], moff[-1]
        self.pinned_by_capture = False          # set on first use under stream capture; never cleared (the graph may be replayed any time)
        pin = torch.device(device).type == "cuda"
        self._host_off = torch.tensor(off, dtype=torch.int32)
        self._host_moff = torch.tensor(moff[:-1], dtype=torch.int64)
        if pin:
            self._host_off, self._host_moff = self._host_off.pin_memory(), self._host_moff.pin_memory()
        self.seq_off = self._host_off.to(device, non_blocking=True)
        self.mask_off = self._host_moff.to(device, non_blocking=True)

    @classmethod
    def get(cls, lengths, H: int, device) -> "VarlenGeom":
        key = (tuple(int(t) for t in lengths), int(H), str(device))
        g = cls._cache.get(key)
        if g is None:
            # only geometries that no captured graph points at count against the limit and are evicted (a HIP graph bakes
            # seq_off / mask_off and, when the geometry was built inside the capture, the pinned sources of their uploads into its nodes;
            # with every entry pinned nothing was evicted and the cache grew without a bound that anybody saw)
            free = [k for k, old in cls._cache.items() if not old.pinned_by_capture]
            if len(free) >= cls._CACHE_MAX:
                del cls._cache[free[0]]
            npinned = len(cls._cache) - len(free)
            if npinned >= cls._PINNED_WARN and not cls._warned:
                cls._warned = True
                import warnings
                warnings.warn(f"VarlenGeom: {npinned} packed-attention geometries are held by captured steps (each keeps pinned host "
                              "buffers and device tensors for the life of the process)")
            g = cls._cache[key] = cls(lengths, H, device)
        if not g.pinned_by_capture and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            g.pinned_by_capture = True           # used (or built) while a step is being captured: lives as long as the process
        return g


def attn_varlen_supported(Tmax: int, hd: int) -> bool:
    return bool(_lib.load().neko_attn_varlen_supported(int(Tmax), int(hd)))


def attn_fwd_varlen(qkv, kbias, kstart, geom: VarlenGeom, hd, drop=None, out=None, want_mask=False):
    """Packed counterpart of attn_fwd: qkv [rows, 3 H hd], kbias [rows], kstart [nseq] -> (out [rows, H hd], lse [rows * H]
    laid out [sequence][head][position], keep masks or None)."""
    _chk(qkv, BF16, "qkv")
    H = geom.H
    assert qkv.is_contiguous() and qkv.shape[0] >= geom.rows and kbias.numel() >= geom.rows
    if out is None:
        out = torch.empty(qkv.shape[0], H * hd, dtype=BF16, device=qkv.device)
    lse = torch.empty(geom.rows * H, dtype=torch.float32, device=qkv.device)
    mask = None
    if want_mask and drop is not None and drop.thr > 0 and hd == 32:      # only the head-resident kernels hand keep masks on
        mask = torch.empty(geom.mask_dwords, dtype=torch.int32, device=qkv.device)
    _lib.call("neko_attn_fwd_varlen", _p(qkv), _p(kbias), _p(kstart), _p(geom.seq_off), _p(geom.mask_off), _p(out), _p(lse),
              geom.nseq, geom.Tmax, H, hd, *_drop(drop), _p(mask), _stream())
    return out, lse, mask


class _DetAttnPath:
    """Under NEKO_DETERMINISTIC (SCATTER_DET) the head-resident attention backward of THIS thread runs as its two kernels at every length
    (neko_attn_bwd_reproducible, thread-local in the library: the process-wide schedule knob is not touched): the one-pass form adds dQ
    up block by block in arrival order, the two-kernel form is bit-reproducible."""

    def __enter__(self):
        self.prev = int(_lib.load().neko_attn_bwd_reproducible(1)) if SCATTER_DET else None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            _lib.load().neko_attn_bwd_reproducible(self.prev)
        return False


def attn_bwd_varlen(qkv, out, dout, kbias, kstart, lse, geom: VarlenGeom, hd, drop=None, dqkv=None, mask=None):
    _chk(dout, BF16, "dout")
    H = geom.H
    D = torch.empty(geom.rows * H, dtype=torch.float32, device=qkv.device)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if mask is not None:
        _chk(mask, torch.int32, "mask"); assert mask.numel() == geom.mask_dwords
    with _DetAttnPath():
        _lib.call("neko_attn_bwd_varlen", _p(qkv), _p(out), _p(dout), _p(kbias), _p(kstart), _p(geom.seq_off), _p(geom.mask_off),
                  _p(lse), _p(D), _p(dqkv), geom.nseq, geom.rows, geom.Tmax, H, hd, *_drop(drop), _p(mask), _stream())
    return dqkv


def attn_bwd(qkv, out, dout, kbias, kstart, lse, B, T, H, hd, drop=None, dqkv=None, mask=None):
    """dqkv (optional): a contiguous [B*T, 3*H*hd] bf16 row range to write into.
    mask: the keep-mask buffer attn_fwd(..., want_mask=True) returned for the SAME call (or None: decisions are re-hashed)."""
    _chk(dout, BF16, "dout")
    dev = qkv.device
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and qkv.shape[0] == B * T
    D = torch.empty(B * H * T, dtype=torch.float32, device=dev)
    qflags = torch.empty(B * ((T + 63) // 64), dtype=torch.int32, device=dev)
    if dqkv is None:
        dqkv = torch.empty(B * T, 3 * H * hd, dtype=BF16, device=dev)
    else:
        _chk(dqkv, BF16, "dqkv"); assert dqkv.is_contiguous() and dqkv.shape == (B * T, 3 * H * hd)
    if mask is not None:
        _chk(mask, torch.int32, "mask")
        assert mask.numel() == int(_lib.load().neko_attn_mask_dwords(B, T, H, hd)), "mask buffer of another call / schedule"
    with _DetAttnPath():
        _lib.call("neko_attn_bwd", _p(qkv), _p(out), _p(dout), _p(kbias), _p(kstart), _p(lse), _p(D), _p(qflags),
                  _p(dqkv), B, T, H, hd, *_drop(drop), _p(mask), _stream())
    return dqkv


def gemv(x, W, M, N, K, *, b_kstrided, ldw=None, bias=None, resid=None, act=0, out_f32=None, out_bf16=None):
    """y[M<=8, N] = x[M,K] . W (+bias)(GELU)(+resid): weight-streaming product for the decode step (neko_gemv_bf16)."""
    _chk(x, BF16, "x"); _chk(W, BF16, "W")
    assert x.stride(-1) == 1 and (out_f32 is not None or out_bf16 is not None)
    _lib.call("neko_gemv_bf16", _p(x), x.stride(0) if x.dim() == 2 else K, _p(W), ldw if ldw is not None else W.stride(0),
              int(b_kstrided), M, N, K, _p(bias), _p(resid), resid.stride(0) if resid is not None else 0, int(act),
              _p(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
              _p(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0, _stream())


def attn_decode(cache, row, pos, out, H, hd):
    """Newest row (index *pos, device int32) of one sequence against its [cap, 3d] q|k|v cache; see neko_attn_decode."""
    _chk(cache, BF16, "cache"); _chk(row, BF16, "row"); _chk(out, BF16, "out"); _chk(pos, torch.int32, "pos")
    assert cache.is_contiguous() and cache.shape[1] == 3 * H * hd and row.numel() == 3 * H * hd and out.numel() == H * hd
    _lib.call("neko_attn_decode", _p(cache), _p(row), _p(pos), _p(out), H, hd, cache.shape[0], _stream())


def ce_bf16_inplace(z, V, Vpad, target, weight, loss_row=None, want_grad=True):
    """z bf16 [R, >=Vpad]: logits in, weight * (softmax - onehot) out (in place); see neko_ce_bf16_inplace."""
    _chk(z, BF16, "z"); _chk(target, torch.int64, "target"); _chk(weight, torch.float32, "weight")
    assert z.stride(1) == 1
    _lib.call("neko_ce_bf16_inplace", _p(z), z.stride(0), V, Vpad, _p(target), _p(weight), _p(loss_row),
              int(bool(want_grad)), z.shape[0], _stream())


def ce_fwd_bwd(logits, V, Vpad, target, weight, loss_row=None, dlogits=None):
    _chk(logits, torch.float32, "logits"); _chk(target, torch.int64, "target"); _chk(weight, torch.float32, "weight")
    R = logits.shape[0]
    _lib.call("neko_ce_fwd_bwd", _p(logits), logits.stride(0), V, Vpad, _p(target), _p(weight), _p(loss_row),
              _p(dlogits), dlogits.stride(0) if dlogits is not None else 0, R, _stream())


def pack_embed_fwd(desc, cont_vals, disc_vals, img_emb, embed, pos_embed, sep, ntok, d, mu, M, n_bins, cont_start,
                   disc_start):
    _chk(desc, torch.int32, "desc")
    dev = embed.device
    x = torch.empty(ntok, d, dtype=torch.float32, device=dev)
    tokens = torch.empty(ntok, dtype=torch.int64, device=dev)
    tmask = torch.empty(ntok, dtype=torch.float32, device=dev)
    pmask = torch.empty(ntok, dtype=torch.float32, device=dev)
    _lib.call("neko_pack_embed_fwd", _p(desc), _p(cont_vals), _p(disc_vals), _p(img_emb), _p(embed), _p(pos_embed),
              _p(sep), _p(x), _p(tokens), _p(tmask), _p(pmask), ntok, d, float(mu), float(M), n_bins, cont_start,
              disc_start, _stream())
    return x, tokens, tmask, pmask


#: NEKO_DETERMINISTIC=1: the embedding-table gradients (token / position / separator embeddings, patch position tables) are summed in
#: sorted, fixed order (ABI v15, segsum.hip) instead of by fp32 atomics -- the only order-dependent sums of a step: with it a training run
#: is bit-reproducible (tools/trajectory_noise_probe.py: identical losses over repeated runs).  Costs 0.15 ms per m-mix step (two stable
#: sorts of 65536 keys + the ordered row sums, against 0.29 + 0.21 ms of contended atomics), hence off by default.
SCATTER_DET = os.environ.get("NEKO_DETERMINISTIC", "0") == "1"


#: The position-table / separator gradients of the packing backward and the patch-position table gradients as fixed-order segment sums
#: over HOST-sorted (key, row) pairs (neko_pack_embed_bwd_sorted / neko_patch_pos_add_bwd_sorted, ABI v17) instead of fp32 atomics on a
#: few dozen heavily contended rows; NEKO_SORTED_SCATTER=0 returns to the atomics.
SORTED_SCATTER = os.environ.get("NEKO_SORTED_SCATTER", "1") != "0"
SEGSUM_KEY_NONE = 0xFFFFF


def sorted_pairs(keys):
    """numpy int array of destination rows (negative: none) -> (keys_sorted, idx_sorted) int32, stable ascending, entries without a
    destination last with the key SEGSUM_KEY_NONE -- the input format of the *_sorted backward entry points."""
    import numpy as np
    k = np.where(keys < 0, SEGSUM_KEY_NONE, keys).astype(np.int32)
    order = np.argsort(k.astype(np.int16) if k.max(initial=0) < 32767 else k, kind="stable").astype(np.int32)
    return k[order], order


def pack_embed_bwd(desc, tokens, dx, d_embed, d_pos, d_sep, d_img, ntok, d, sorted_tail=False):
    """sorted_tail: `desc` holds, behind its ntok x 4 descriptors, ntok sorted keys and ntok row indices (sorted_pairs of the local
    position, or d_pos.shape[0] for a separator token): position / separator gradients then come from fixed-order segment sums."""
    _chk(dx, torch.float32, "dx")
    if sorted_tail and SORTED_SCATTER and not SCATTER_DET and d_pos.shape[0] < SEGSUM_KEY_NONE:
        assert desc.numel() >= 6 * ntok and desc.dtype == torch.int32 and desc.is_contiguous()
        n = int(_lib.load().neko_pack_embed_bwd_sorted_ws_bytes(ntok, d))
        ws = torch.empty(n, dtype=torch.uint8, device=dx.device)
        flat = desc.view(-1)
        _lib.call("neko_pack_embed_bwd_sorted", _p(desc), _p(tokens), _p(dx), _p(d_embed), _p(d_pos), _p(d_sep), _p(d_img), ntok, d,
                  int(d_pos.shape[0]), _p(flat[4 * ntok:]), _p(flat[5 * ntok:]), _p(ws), n, _stream())
        return
    if SCATTER_DET and d_embed.shape[0] < 0xFFFFF and d_pos.shape[0] < 0xFFFFF:
        n = int(_lib.load().neko_pack_embed_bwd_det_ws_bytes(ntok, d))
        ws = torch.empty(n, dtype=torch.uint8, device=dx.device)
        _lib.call("neko_pack_embed_bwd_det", _p(desc), _p(tokens), _p(dx), _p(d_embed), _p(d_pos), _p(d_sep), _p(d_img),
                  ntok, d, int(d_embed.shape[0]), int(d_pos.shape[0]), _p(ws), n, _stream())
        return
    _lib.call("neko_pack_embed_bwd", _p(desc), _p(tokens), _p(dx), _p(d_embed), _p(d_pos), _p(d_sep), _p(d_img),
              ntok, d, _stream())


def tokenize_continuous(x, use_mu_law, mu, M, n_bins, offset):
    _chk(x, torch.float32, "x")
    x = x.contiguous()
    ids = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    _lib.call("neko_tokenize_continuous", _p(x), _p(ids), x.numel(), int(use_mu_law), float(mu), float(M), n_bins,
              offset if offset is not None else 0, _stream())
    return ids


def gather_rows_bf16(src, idx, n, npad):
    """dst[r] = src[idx[r]] for r < n, zero rows up to npad."""
    _chk(src, BF16, "src"); _chk(idx, torch.int32, "idx")
    d = src.shape[1]
    dst = torch.empty(npad, d, dtype=BF16, device=src.device)
    _lib.call("neko_gather_rows_bf16", _p(src), _p(idx), _p(dst), n, npad, d, _stream())
    return dst


def scatter_rows_f32(src, idx, n, dst):
    _chk(src, torch.float32, "src"); _chk(dst, torch.float32, "dst")
    _lib.call("neko_scatter_rows_f32", _p(src), _p(idx), _p(dst), n, src.shape[1], _stream())


def cast_f32_bf16(x, y):
    _chk(x, torch.float32, "x"); _chk(y, BF16, "y")
    _lib.call("neko_cast_f32_bf16", _p(x), _p(y), x.numel(), _stream())


def colsum_bf16(x, M, N, out, accumulate=True, ld=None):
    _chk(x, BF16, "x"); _chk(out, torch.float32, "out")
    _lib.call("neko_colsum_bf16", _p(x), ld if ld is not None else x.stride(0), M, N, _p(out), int(accumulate),
              _stream())


def geglu_fwd(h, gate):
    """h (= gelu(pre), bf16, contiguous) *= gate in place (neko_geglu_fwd; MLP.forward trajectory_gpt2.py:275-276)."""
    _chk(h, BF16, "h"); _chk(gate, BF16, "gate")
    assert h.is_contiguous() and gate.is_contiguous() and h.numel() == gate.numel()
    _lib.call("neko_geglu_fwd", _p(h), _p(gate), h.numel(), _stream())


def geglu_bwd(dh, pre, gate):
    """(d_pre, d_gate) = (dh * gate * gelu'(pre), dh * gelu(pre)), bf16 (neko_geglu_bwd)."""
    _chk(dh, BF16, "dh"); _chk(pre, BF16, "pre"); _chk(gate, BF16, "gate")
    assert dh.is_contiguous() and pre.is_contiguous() and gate.is_contiguous()
    d_pre, d_gate = torch.empty_like(dh), torch.empty_like(dh)
    _lib.call("neko_geglu_bwd", _p(dh), _p(pre), _p(gate), _p(d_pre), _p(d_gate), dh.numel(), _stream())
    return d_pre, d_gate


def sqnorm_f32(g, out_accum):
    _chk(g, torch.float32, "g"); _chk(out_accum, torch.float64, "out_accum")
    _lib.call("neko_sqnorm_f32", _p(g), g.numel(), _p(out_accum), _stream())


def adamw_step(p, g, m, v, p16, lr, beta1, beta2, eps, wd, gnorm_sq, max_norm, grad_scale, step, active, lr_dev=None):
    _lib.call("neko_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(p16), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), float(wd), _p(gnorm_sq), float(max_norm), _p(grad_scale), _p(step),
              _p(active), _p(lr_dev), _stream())


_salt_holder = [None]


def set_drop_salt(salt: Optional[torch.Tensor]) -> None:
    """Register (or, with None, remove) the device uint32 every dropout site adds to its key (neko_set_drop_salt): the
    per-step variation of the masks inside a captured training step.  The tensor is kept alive here."""
    if salt is not None:
        assert salt.is_cuda and salt.dtype == torch.int32 and salt.numel() == 1
    _lib.call("neko_set_drop_salt", _p(salt))
    _salt_holder[0] = salt


#: the patch forward hands its GroupNorm statistics to the backward (neko_patch_resblock_fwd_stats / _bwd_stats, ABI v17) instead of
#: letting it recompute them; NEKO_PATCH_STATS=0 returns to the recomputing backward
PATCH_STATS = os.environ.get("NEKO_PATCH_STATS", "1") != "0"


def patch_resblock_fwd(images, w1, b1, gw, gb, w2, b2, mid, groups, want_x=True, want_padded=False, want_stats=False):
    """-> (y16 [P, 768] bf16, x_patches [P, 768] fp32 | None), plus with `want_padded` the zero-padded storage y16 is the row prefix of
    ([Ppad, 768], Ppad a multiple of 128): the weight gradient of the projection contracts over ITS rows; plus with `want_stats` the
    GroupNorm statistics [P, 64] fp32 (mean | rstd per group) for patch_resblock_bwd, or None when x_patches was not asked for."""
    assert images.is_cuda and images.dim() == 4 and images.shape[1] == 3
    assert images.dtype in (torch.float32, torch.uint8)
    images = images.contiguous()
    n, _, H, W = images.shape
    if H % 16 or W % 16:
        raise AssertionError("Image dimensions must be divisible by patch size")
    P = n * (H // 16) * (W // 16)
    # the patch rows are the CONTRACTION of the projection's weight gradient: storage padded with zero rows to a multiple of 128
    # keeps that product on the LDS-DMA GEMMs (a patch count such as 12289 sent it to the register-staged fallback: 237 us)
    Ppad = (P + 127) // 128 * 128
    y16_all = torch.empty(Ppad, 768, dtype=BF16, device=images.device)
    if Ppad > P:
        y16_all[P:].zero_()
    y16 = y16_all[:P]
    xp = torch.empty(P, 768, dtype=torch.float32, device=images.device) if want_x else None
    stats = torch.empty(P, 64, dtype=torch.float32, device=images.device) if (want_stats and want_x and PATCH_STATS) else None
    if stats is not None:
        _lib.call("neko_patch_resblock_fwd_stats", _p(images), int(images.dtype == torch.uint8), n, H, W, _p(w1), _p(b1),
                  _p(gw), _p(gb), _p(w2), _p(b2), mid, groups, _p(y16), _p(xp), _p(stats), _stream())
    else:
        _lib.call("neko_patch_resblock_fwd", _p(images), int(images.dtype == torch.uint8), n, H, W, _p(w1), _p(b1),
                  _p(gw), _p(gb), _p(w2), _p(b2), mid, groups, _p(y16), _p(xp), _stream())
    out = (y16, xp)
    if want_padded:
        out += (y16_all,)
    if want_stats:
        out += (stats,)
    return out


def patch_resblock_bwd(xp, dy, w1, b1, gw, gb, w2, b2, mid, groups, dw1, db1, dgw, dgb, dw2, db2, stats=None):
    _chk(dy, torch.float32, "dy")
    ws = torch.empty(_lib.load().neko_patch_resblock_bwd_ws_floats(xp.shape[0]), dtype=torch.float32, device=xp.device)
    if stats is not None:
        _chk(stats, torch.float32, "stats"); assert stats.shape == (xp.shape[0], 64) and stats.is_contiguous()
        _lib.call("neko_patch_resblock_bwd_stats", _p(xp), _p(stats), _p(dy), xp.shape[0], _p(w1), _p(b1), _p(gw), _p(gb), _p(w2), _p(b2),
                  mid, groups, _p(dw1), _p(db1), _p(dgw), _p(dgb), _p(dw2), _p(db2), _p(ws), _stream())
        return
    _lib.call("neko_patch_resblock_bwd", _p(xp), _p(dy), xp.shape[0], _p(w1), _p(b1), _p(gw), _p(gb), _p(w2), _p(b2),
              mid, groups, _p(dw1), _p(db1), _p(dgw), _p(dgb), _p(dw2), _p(db2), _p(ws), _stream())


def patch_pos_add(out, hpos, wpos, row_emb, col_emb):
    P, d = out.shape
    _lib.call("neko_patch_pos_add", _p(out), _p(hpos), _p(wpos), _p(row_emb), _p(col_emb), P, d, _stream())


def patch_pos_add_bwd(dout, hpos, wpos, d_row, d_col, sorted_rows=None):
    """sorted_rows: int32 [4, P] = (sorted hpos, their patch indices, sorted wpos, their patch indices) from sorted_pairs, or None"""
    P, d = dout.shape
    if sorted_rows is not None and SORTED_SCATTER and not SCATTER_DET and P > 0 and d_row.shape[0] < SEGSUM_KEY_NONE:
        assert sorted_rows.dtype == torch.int32 and sorted_rows.shape == (4, P) and sorted_rows.is_contiguous()
        n = int(_lib.load().neko_patch_pos_add_bwd_sorted_ws_bytes(P, d))
        ws = torch.empty(n, dtype=torch.uint8, device=dout.device)
        _lib.call("neko_patch_pos_add_bwd_sorted", _p(dout), _p(sorted_rows[0]), _p(sorted_rows[1]), _p(sorted_rows[2]), _p(sorted_rows[3]),
                  _p(d_row), _p(d_col), P, d, int(d_row.shape[0]), _p(ws), n, _stream())
        return
    if SCATTER_DET and P > 0:
        n = int(_lib.load().neko_patch_pos_add_bwd_det_ws_bytes(P, d))
        ws = torch.empty(n, dtype=torch.uint8, device=dout.device)
        _lib.call("neko_patch_pos_add_bwd_det", _p(dout), _p(hpos), _p(wpos), _p(d_row), _p(d_col), P, d, int(d_row.shape[0]),
                  _p(ws), n, _stream())
        return
    _lib.call("neko_patch_pos_add_bwd", _p(dout), _p(hpos), _p(wpos), _p(d_row), _p(d_col), P, d, _stream())
