"""GatoPolicy on the MI355X HIP path -- drop-in for gato/policy/gato_policy.py:18-614.

Same constructor signature, attributes (``.module``, ``.device``, ``.context_len``, ``.embed_dim``,
``.transformer``, ``.text_tokenizer``, ``.token_starts/.token_ends``, ``.continuous_action_tokenizer``,
``.image_embedding``, ``.embed_token``), ``state_dict`` keys/shapes and call signatures
(``forward(inputs, compute_loss, **kwargs) -> (logits, loss)``, ``tokenize_input_dicts``, ``predict_*``).
The nn.Modules are parameter containers; all compute between the batch dicts and the loss is
hand-written HIP (neko_amd/csrc) driven by neko_amd.engine.  There is no CPU / eager fallback.

Extra (optional, keyword-only) knobs that the reference does not have:
  text_tokenizer=   an object with ``vocab_size`` (+ ``encode``/``decode``) used instead of
                    ``AutoTokenizer.from_pretrained`` (no network on the GPU box);
  forward(..., return_logits=False)  skips materialising the (B,T,V) fp32 logits that
                    ``Trainer.train_step`` discards anyway (trainer.py:178).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Optional, Sequence, Union

import os

import numpy as np
import torch
import torch.nn as nn

from .. import engine, ops
from ..flat import FlatParams
from ..transformers.trajectory_gpt2 import GPT2Config, GPT2Model
from .embeddings import ImageEmbedding
from .input_tokenizers import ContinuousTokenizer

# descriptor kinds (neko_amd/csrc/pack_embed.hip)
K_PAD, K_TOKEN, K_CONT_OBS, K_CONT_ACT, K_DISC, K_SEP, K_IMAGE, K_DEVID = 0, 1, 2, 3, 4, 5, 6, 7
VPAD_ALIGN = 256      # predict_token rows padded with zeros to whole 256-row GEMM tiles (LM-head dW takes the long-contraction loop)


class PackedBatch:
    """Host-side result of laying out a list of batch dicts (gato_policy.py:245-431)."""

    def __init__(self):
        self.B = 0
        self.T = 0
        self.desc: Optional[np.ndarray] = None      # [B*T, 4] int32
        self.cont: List[torch.Tensor] = []          # flattened continuous sources (obs and actions)
        self.disc: List[torch.Tensor] = []          # flattened discrete / device-id sources
        self.images: List[torch.Tensor] = []        # per image example: (n_ts,3,H,W)
        self.given_img_emb: List[torch.Tensor] = [] # per example with precomputed embeddings: (n, P, d)
        self.img_order: List[tuple] = []            # ('img', idx) | ('emb', idx) in example order
        self.segments: Optional[List[tuple]] = None # ragged groups: [(row0, B_k, T_k)] tiling the rows of desc
        self.order: Optional[List[int]] = None      # ragged groups: example index of every sequence, in row order


def _as_2d_ids(t):
    """gato_policy.py:264-273: list -> (1,L); 1-D tensor -> (1,L); 2-D kept."""
    if isinstance(t, list):
        a = np.asarray(t, dtype=np.int64)
        return a.reshape(1, -1) if a.ndim == 1 else a
    if t.dim() == 1:
        return t.unsqueeze(0)
    return t


#: a bucketed (ragged) layout is used only when it removes at least this fraction of the padded rows (see build_layout)
RAGGED_MIN_SAVING = float(os.environ.get("NEKO_RAGGED_MIN_SAVING", "0.10"))


def plan_ragged_groups(lengths: Sequence[int], max_groups: int) -> List[List[int]]:
    """Partition examples into at most `max_groups` length buckets minimising the padded token count
    sum_k B_k * T_k (T_k = longest member of bucket k).  Exact dynamic programme over the distinct lengths in
    descending order (an optimal bucket is a contiguous range of the sorted lengths).  Returns lists of example
    indices, longest bucket first; within a bucket the input order is kept."""
    uniq = sorted(set(int(l) for l in lengths), reverse=True)
    if len(uniq) > 64:                                   # many distinct lengths (free text): quantise to 16 first
        uniq = sorted(set((u + 15) // 16 * 16 for u in uniq), reverse=True)
        key = lambda l: (int(l) + 15) // 16 * 16
    else:
        key = int
    cnt = {u: 0 for u in uniq}
    for l in lengths:
        cnt[key(l)] += 1
    n, G = len(uniq), max(1, min(int(max_groups), len(uniq)))
    pre = [0]
    for u in uniq:
        pre.append(pre[-1] + cnt[u])
    INF = float("inf")
    dp = [[INF] * (n + 1) for _ in range(G + 1)]
    arg = [[0] * (n + 1) for _ in range(G + 1)]
    dp[0][0] = 0
    for g in range(1, G + 1):
        for j in range(1, n + 1):
            for i in range(g - 1, j):
                if dp[g - 1][i] < INF:
                    c = dp[g - 1][i] + uniq[i] * (pre[j] - pre[i])
                    if c < dp[g][j]:
                        dp[g][j], arg[g][j] = c, i
    g = min(range(1, G + 1), key=lambda k: (dp[k][n], k))
    cuts, j = [], n
    while g > 0:
        i = arg[g][j]
        cuts.append((i, j))
        j, g = i, g - 1
    groups = []
    for i, j in reversed(cuts):
        members = set(uniq[i:j])
        groups.append([e for e, l in enumerate(lengths) if key(l) in members])
    return groups


_MODALITY_KEYS = ("text", "images", "image_embeddings", "continuous_obs", "discrete_obs", "continuous_actions", "discrete_actions")


def layout_signature(inputs: Sequence[dict]):
    """What the descriptor table of a batch depends on when no token ids live on the host: which modalities every example has and
    their shapes.  None when the table is data dependent (text ids in host memory are written INTO the descriptors) or an input is of
    an unexpected type.  Fixed-shape tasks (control episodes, image observations, device-resident text) produce the same signature
    step after step, and _prepare then reuses the descriptors, their sorted tail and the loss-row indices it uploaded before:
    at README batch sizes the host enqueue IS the step (tools/probe/r05_host_profile_c3.py: 1.1 of 5.7 ms per c3 step went here)."""
    sig = []
    for ex in inputs:
        e = []
        for k in _MODALITY_KEYS:
            v = ex.get(k)
            if v is None:
                continue
            if not isinstance(v, torch.Tensor):
                return None
            if k == "text" and not v.is_cuda:
                return None
            e.append((k, tuple(v.shape)))
        sig.append(tuple(e))
    return tuple(sig)


def collect_sources(inputs: Sequence[dict]) -> PackedBatch:
    """The value-source lists of build_layout (same order, same tensors) without the descriptor table: what _prepare still needs from
    a batch whose table it has cached.  Only for batches layout_signature() accepts (device-resident text)."""
    pb = PackedBatch()
    for ex in inputs:
        if ex.get("text") is not None:
            pb.disc.append(_as_2d_ids(ex["text"]).to(torch.int32).reshape(-1))
        if ex.get("image_embeddings") is not None:
            pb.given_img_emb.append(ex["image_embeddings"])
            pb.img_order.append(("emb", len(pb.given_img_emb) - 1))
        elif ex.get("images") is not None:
            pb.images.append(ex["images"])
            pb.img_order.append(("img", len(pb.images) - 1))
        if ex.get("continuous_obs") is not None:
            pb.cont.append(ex["continuous_obs"].reshape(-1))
        if ex.get("discrete_obs") is not None:
            pb.disc.append(ex["discrete_obs"].reshape(-1))
        if ex.get("continuous_actions") is not None:
            pb.cont.append(ex["continuous_actions"].reshape(-1))
        if ex.get("discrete_actions") is not None:
            pb.disc.append(ex["discrete_actions"].reshape(-1))
    return pb


def build_layout(inputs: Sequence[dict], use_pos_encoding: bool, context_len: int, pad_seq: bool,
                 n_patches_of=None, ragged_groups: int = 0) -> PackedBatch:
    """Turn the list of example dicts into one descriptor table + source lists (host only, numpy).
    Per timestep order [image patches | text | continuous obs | discrete obs | SEP | continuous act |
    discrete act] (gato_policy.py:355); positions 0..n_obs-1 get a local position (:380-385);
    left-pad to the longest example (:408-416); optional right-pad to context_len (:423-431)."""
    pb = PackedBatch()
    per_ex = []
    cont_off = disc_off = img_off = 0
    for ex in inputs:
        n_ts = None

        def set_ts(n):
            nonlocal n_ts
            if n_ts is None:
                n_ts = n
            else:
                assert n_ts == n, "number of timesteps must be the same for all modalities"

        segs = []   # (kind, per-ts count, src_base, host_ids or None, target)
        text_seg = None
        if ex.get("text") is not None:
            ids = _as_2d_ids(ex["text"])
            if isinstance(ids, np.ndarray) or not ids.is_cuda:
                arr = ids if isinstance(ids, np.ndarray) else ids.to(torch.int64).numpy()
                text_seg = (K_TOKEN, arr.shape[1], 0, arr.astype(np.int64), 1)
            else:
                flat = ids.to(torch.int32).reshape(-1)
                text_seg = (K_DEVID, ids.shape[1], disc_off, None, 1)
                pb.disc.append(flat)
                disc_off += flat.numel()
            n_ts = ids.shape[0]
        if ex.get("images") is not None or ex.get("image_embeddings") is not None:
            if ex.get("image_embeddings") is not None:
                e = ex["image_embeddings"]
                n_img, n_patch = e.shape[0], e.shape[1]
                pb.given_img_emb.append(e)
                pb.img_order.append(("emb", len(pb.given_img_emb) - 1))
            else:
                im = ex["images"]
                n_img = im.shape[0]
                n_patch = (im.shape[2] // 16) * (im.shape[3] // 16)
                assert im.shape[2] % 16 == 0 and im.shape[3] % 16 == 0, "Image dimensions must be divisible by patch size"
                pb.images.append(im)
                pb.img_order.append(("img", len(pb.images) - 1))
            segs.append((K_IMAGE, n_patch, img_off, None, 0))
            img_off += n_img * n_patch
            set_ts(n_img)
        if text_seg is not None:
            segs.append(text_seg)
        if ex.get("continuous_obs") is not None:
            t = ex["continuous_obs"]
            segs.append((K_CONT_OBS, t.shape[1], cont_off, None, 0))
            pb.cont.append(t.reshape(-1))
            cont_off += t.numel()
            set_ts(t.shape[0])
        if ex.get("discrete_obs") is not None:
            t = ex["discrete_obs"]
            segs.append((K_DISC, t.shape[1], disc_off, None, 0))
            pb.disc.append(t.reshape(-1))
            disc_off += t.numel()
            set_ts(t.shape[0])
        n_obs = sum(s[1] for s in segs)
        segs.append((K_SEP, 1, 0, None, 0))
        if ex.get("continuous_actions") is not None:
            t = ex["continuous_actions"]
            segs.append((K_CONT_ACT, t.shape[1], cont_off, None, 1))
            pb.cont.append(t.reshape(-1))
            cont_off += t.numel()
            set_ts(t.shape[0])
        if ex.get("discrete_actions") is not None:
            t = ex["discrete_actions"]
            segs.append((K_DISC, t.shape[1], disc_off, None, 1))
            pb.disc.append(t.reshape(-1))
            disc_off += t.numel()
            set_ts(t.shape[0])
        assert n_ts is not None, "example has no modality"
        tp = sum(s[1] for s in segs)
        d = np.zeros((n_ts, tp, 4), dtype=np.int64)
        col = 0
        ts = np.arange(n_ts, dtype=np.int64)[:, None]
        for kind, cnt, base, host_ids, tgt in segs:
            j = np.arange(cnt, dtype=np.int64)[None, :]
            d[:, col:col + cnt, 0] = kind
            if kind == K_TOKEN:
                d[:, col:col + cnt, 1] = host_ids
            elif kind != K_SEP:
                d[:, col:col + cnt, 1] = base + ts * cnt + j
            d[:, col:col + cnt, 3] = tgt
            col += cnt
        d[:, :, 2] = -1
        if use_pos_encoding:
            d[:, :n_obs, 2] = np.arange(n_obs, dtype=np.int64)[None, :]
        per_ex.append(d.reshape(n_ts * tp, 4))
    if ragged_groups > 0:
        # length-bucketed layout: sequences are left-padded to the longest member of THEIR bucket,
        # buckets are concatenated along the row axis.  Descriptor rows carry absolute source offsets, so moving an
        # example's rows does not touch the value buffers.
        groups = plan_ragged_groups([e.shape[0] for e in per_ex], ragged_groups)
        lens = [e.shape[0] for e in per_ex]
        bucket_rows = sum(len(g) * max(lens[i] for i in g) for g in groups)
        # buckets pay with one attention launch each and ragged GEMM edges: with nearly full sequences (the metric's
        # m-mix: 1024 / 1008 / 988 tokens, 1.6 % padding) they measured 6 % SLOWER than the padded layout, with the
        # 1024 / 494 / 289 / 240 mix 1.54x faster -- so they are only used when they remove at least a tenth of the rows
        if bucket_rows > (1.0 - RAGGED_MIN_SAVING) * len(per_ex) * max(lens):
            groups = None
    if ragged_groups > 0 and groups is not None:
        blocks, segs, order, row0 = [], [], [], 0
        for members in groups:
            Tk = max(per_ex[i].shape[0] for i in members)
            blk = np.zeros((len(members), Tk, 4), dtype=np.int32)
            blk[:, :, 2] = -1
            for r, i in enumerate(members):
                blk[r, Tk - per_ex[i].shape[0]:] = per_ex[i]
            blocks.append(blk.reshape(-1, 4))
            segs.append((row0, len(members), Tk))
            order += members
            row0 += len(members) * Tk
        tail = (-row0) % 64          # row count up to a multiple of 64: the weight gradients contract over the rows,
        if tail:                     # and the fast GEMM path wants that contraction in whole 64-row steps
            blk = np.zeros((tail, 4), dtype=np.int32)
            blk[:, 2] = -1
            blocks.append(blk)
        pb.B, pb.T, pb.desc = 1, row0 + tail, np.concatenate(blocks, axis=0)
        pb.segments, pb.order = segs, order
        return pb
    T = max(e.shape[0] for e in per_ex)
    T_out = context_len if (pad_seq and context_len > T) else T
    B = len(per_ex)
    desc = np.zeros((B, T_out, 4), dtype=np.int32)
    desc[:, :, 2] = -1
    for i, e in enumerate(per_ex):
        desc[i, T - e.shape[0]:T] = e
    pb.B, pb.T, pb.desc = B, T_out, desc.reshape(B * T_out, 4)
    return pb


class _PolicyCoreFn(torch.autograd.Function):
    """packed embeddings -> (logits | empty, loss | empty): transformer stack + LM head + masked CE."""

    @staticmethod
    def forward(ctx, policy: "GatoPolicy", x, pmask, tokens, tmask, compute_loss: bool, return_logits: bool, pack, *params):
        """pack: the `_PackInfo` of the `_tokenize` call that produced (x, tokens, tmask, pmask) in THIS forward, or None
        when the caller handed the four tensors in (kwargs form, gato_policy.py:161-165): the loss positions are then
        derived from the masks that were passed, as the reference does (:176-183)."""
        need = bool(compute_loss) and any(ctx.needs_input_grad)   # grad mode is off inside Function.forward
        f = policy._flat
        f.ensure_shadow()
        B, T, d = x.shape
        sp = policy.transformer._stack_params()
        segments = pack.segments if pack is not None else None
        hf16, _, sctx = engine.stack_forward(sp, x.detach().to(torch.float32), pmask, save=need,
                                             drops=policy.transformer.make_drops(), segments=segments)
        hp = policy._head_params()
        logits = engine.lm_head_logits(hp, hf16).view(B, T, hp.V) if return_logits else x.new_zeros(0)
        loss = x.new_zeros(())
        dlogits = None
        ctx.sel_idx = ctx.sel_map = None
        if compute_loss:
            if pack is not None and pack.n_loss > 0 and policy.lm_head_selected_rows:
                # the loss rows and their targets' positions are known on the host (_prepare): no shifted copy of the tokens, no
                # selection mask and no count have to be formed on the device
                idx, n = pack.loss_idx, pack.n_loss
                target = None if pack.tgt_idx is not None else engine.shift_targets(tokens, tmask, pmask)[0]
                loss, hf16, dlogits = engine.lm_head_loss_selected(hp, hf16, target, idx, n, want_grad=need,
                                                                   chunk_rows=policy.lm_head_chunk_rows,
                                                                   tokens_flat=tokens.reshape(-1), tgt_idx=pack.tgt_idx)
                ctx.sel_idx, ctx.sel_n, ctx.sel_map = idx, n, pack.row_map
            else:
                target, sel, count = engine.shift_targets(tokens, tmask, pmask)
                if segments is not None:        # the shifted selection above crosses sequence boundaries in the bucketed
                    sel = torch.zeros_like(sel)  # layout; it is only reached when the batch has no loss position at all
                    count = sel.sum()
                loss, dlogits = engine.lm_head_loss(hp, hf16, target, sel, count, want_grad=need,
                                                    chunk_rows=policy.lm_head_chunk_rows)
        ctx.policy, ctx.sctx, ctx.hf16, ctx.dlogits, ctx.shape = policy, sctx, hf16 if need else None, dlogits, (B, T, d)
        ctx.mark_non_differentiable(logits)
        return logits, loss

    @staticmethod
    def backward(ctx, _g_logits, g_loss):
        policy = ctx.policy
        f = policy._flat
        B, T, d = ctx.shape
        names = policy.transformer._param_names() + ["predict_token.weight"]
        f.prepare_backward(names)
        dp = policy._dp
        engine.SideStream.rows_hint = B * T        # rows of this step: decides whether weight gradients fork onto the side stream
        row_map = None
        if ctx.sel_idx is not None:
            dhf, row_map = engine.lm_head_backward_selected(policy._head_params(), ctx.hf16, ctx.dlogits, g_loss, ctx.sel_idx,
                                                            ctx.sel_n, B * T, row_map=ctx.sel_map)
        else:
            dhf = engine.lm_head_backward(policy._head_params(), ctx.hf16, ctx.dlogits, g_loss)
        if dp is not None:
            dp.group_ready("head")

        def layer_done(i):
            if dp is not None:
                dp.group_ready("lnf" if i == len(policy.transformer.h) else f"layer{i}")

        gx = engine.stack_backward(policy.transformer._stack_params(), ctx.sctx, dhf, on_layer_done=layer_done, dhf_row_map=row_map)
        engine.SideStream.join(gx.device)       # (stack_backward joins too: explicit for the LM-head dW launched before it)
        f.attach_grads(names)
        ctx.sctx = ctx.hf16 = ctx.dlogits = None
        return (None, gx.view(B, T, d)) + (None,) * (len(ctx.needs_input_grad) - 2)


class _PackInfo:
    """What the host knows about a batch it has just packed: the loss positions (row (b, t) is selected when position t
    is real and position t+1 is a target, gato_policy.py:176-183) as a device index list, and the length buckets of the
    ragged layout.  Handed from `_tokenize` to the policy core of the SAME forward call, never cached across calls."""
    __slots__ = ("loss_idx", "n_loss", "segments", "order", "rows", "row_map", "tgt_idx")

    def __init__(self, loss_idx, n_loss, segments, order, rows, row_map=None, tgt_idx=None):
        self.loss_idx, self.n_loss, self.segments, self.order, self.rows = loss_idx, n_loss, segments, order, rows
        self.tgt_idx = tgt_idx      # int32: loss_idx + 1, the flat position of every loss row's target token
        #: int32 [rows]: index of row r among the loss rows, -1 where r carries no loss -- the inverse of loss_idx, built on the host
        #: with it (round 5: three torch launches per step made it on the device; ln_f's backward reads the loss rows' gradients through it)
        self.row_map = row_map


#: batches whose descriptor table depends on their structure only (layout_signature) keep the uploaded table, its sorted tail and the
#: loss-row indices for reuse; this many distinct structures are remembered per policy (0 = off)
LAYOUT_CACHE = int(os.environ.get("NEKO_LAYOUT_CACHE", "32"))


class _LayoutEntry:
    __slots__ = ("B", "T", "segments", "order", "has_text", "desc_dev", "idx_dev", "n_sel", "map_dev", "tgt_dev")      # (the key holds sorted_tail)


class _Prepared:
    """A batch after the host half of tokenize_input_dicts (GatoPolicy._prepare): nothing but device tensors and the
    host-known structure.  `tensors()` lists the device inputs in a fixed order (a captured step copies them into its
    static twins), `signature()` is what must be equal for two batches to share one captured graph."""

    def __init__(self):
        self.B = self.T = 0
        self.sorted_tail = False     # the tail of `desc` really holds host-sorted (key, row) pairs (not the KEY_NONE placeholder)
        self.desc = self.cont = self.disc = None
        self.img_order, self.img_ids, self.img_groups, self.given = [], [], [], []
        self.pack: Optional[_PackInfo] = None

    def tensors(self) -> List[torch.Tensor]:
        ts = [self.desc, self.pack.loss_idx] + [t for t in (self.pack.row_map, self.pack.tgt_idx) if t is not None]
        ts += [t for t in (self.cont, self.disc) if t is not None]
        for X, pos, _, _ in self.img_groups:
            ts += [X, pos]
        return ts + list(self.given)

    def signature(self):
        return (self.B, self.T, self.pack.n_loss, tuple(self.pack.segments or ()), tuple(self.img_order), tuple(self.img_ids),
                tuple((tuple(t.shape), str(t.dtype)) for t in self.tensors()),
                tuple((tuple(idxs), tuple(cnt)) for _, _, idxs, cnt in self.img_groups))


class _StubTokenizer:
    def __init__(self, vocab_size):
        self.vocab_size = vocab_size

    def encode(self, s):
        raise RuntimeError("no text tokenizer loaded (offline); pass text_tokenizer=...")

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


class GatoPolicy(nn.Module):
    def __init__(
        self,
        device: Union[torch.device, str],
        embed_dim: int,
        layers: int,
        heads: int,
        dropout: float,
        activation_fn="gelu",
        mu: int = 100,
        M: int = 256,
        patch_size: int = 16,
        resid_mid_channels: int = 132,
        num_groups: int = 32,
        position_vocab_size: int = 128,
        continuous_tokens: int = 1024,
        discrete_tokens: int = 1024,
        context_len=1024,
        use_pos_encoding: bool = True,
        use_patch_pos_encoding: bool = True,
        pretrained_lm: Optional[str] = None,
        flash: bool = False,
        tokenizer_model_name: str = "gpt2",
        pad_seq: bool = False,
        *,
        text_tokenizer=None,
    ):
        super().__init__()
        self.device = device
        self.context_len = context_len
        self.pad_seq = pad_seq
        self.mu, self.M = mu, M

        # Text tokenizer (gato_policy.py:57): only vocab_size / encode / decode are used
        if text_tokenizer is None:
            try:
                from transformers import AutoTokenizer
                text_tokenizer = AutoTokenizer.from_pretrained(tokenizer_model_name)
            except Exception as e:  # offline box
                raise RuntimeError(
                    f"could not load tokenizer '{tokenizer_model_name}' ({type(e).__name__}); pass "
                    "text_tokenizer=<object with vocab_size/encode/decode> when there is no network") from e
        elif isinstance(text_tokenizer, int):
            text_tokenizer = _StubTokenizer(text_tokenizer)
        self.text_tokenizer = text_tokenizer

        self.text_tokens = self.text_tokenizer.vocab_size                            # :60-63
        self.continuous_tokens = continuous_tokens
        self.discrete_tokens = discrete_tokens
        self.vocab_size = self.text_tokens + self.discrete_tokens + self.continuous_tokens
        self.token_starts = {"text": 0, "continuous": self.text_tokens,
                             "discrete": self.text_tokens + self.continuous_tokens}   # :66-70
        self.token_ends = {"text": self.text_tokens - 1,
                           "continuous": self.text_tokens + self.continuous_tokens - 1,
                           "discrete": self.text_tokens + self.continuous_tokens + self.discrete_tokens - 1}

        if pretrained_lm is not None:
            raise NotImplementedError("pretrained_lm / LoRA (gato_policy.py:79-95) needs downloaded weights; "
                                      "out of scope of the HIP hot path (SURVEY.md 2.1 #16)")
        gate = False
        if activation_fn == "geglu":                                        # :97-100: gated MLP around the erf GELU
            gate, activation_fn = True, "gelu"
        if activation_fn != "gelu":
            raise NotImplementedError(f"activation_fn={activation_fn!r}: the HIP MLP epilogues implement ACT2FN['gelu'] "
                                      "(erf form) and 'geglu' (arguments.py:55 default 'gelu')")
        if embed_dim % heads != 0:
            raise AssertionError("embed_dim must be divisible by heads")   # trajectory_gpt2.py:126
        if embed_dim // heads not in (32, 64, 128) or embed_dim % 8:
            raise NotImplementedError(f"head dim {embed_dim // heads} unsupported by the HIP attention kernel "
                                      "(supported: 32, 64, 128)")
        if resid_mid_channels % num_groups != 0:
            # the reference's own default (132, 32) dies inside nn.GroupNorm with this error
            raise ValueError("num_channels must be divisible by num_groups")
        config = GPT2Config(vocab_size=1, n_embd=embed_dim, n_head=heads, n_layer=layers, resid_pdrop=dropout,
                            attn_pdrop=dropout, n_positions=context_len, n_inner=embed_dim * 4,
                            activation_function=activation_fn, n_ctx=context_len, flash=flash, gate=gate)
        self.transformer = GPT2Model(config)                                         # :115
        self.embed_token = nn.Embedding(self.vocab_size, embed_dim)                  # :117
        self.embed_dim = embed_dim
        self.predict_token = nn.Linear(embed_dim, self.vocab_size, bias=False)       # :122
        self.separator_token = nn.Parameter(torch.zeros(embed_dim))                  # :124
        self.continuous_action_tokenizer = ContinuousTokenizer(
            use_mu_law=False, mu=mu, M=M, n_bins=self.continuous_tokens, offset=self.token_starts["continuous"])
        self.continuous_obs_tokenizer = ContinuousTokenizer(
            use_mu_law=True, mu=mu, M=M, n_bins=self.continuous_tokens, offset=self.token_starts["continuous"])
        self.use_patch_pos_encoding = use_patch_pos_encoding
        self.image_embedding = ImageEmbedding(embed_dim=embed_dim, patch_size=patch_size,
                                              resid_mid_channels=resid_mid_channels, num_groups=num_groups,
                                              position_vocab_size=position_vocab_size,
                                              use_pos_encoding=self.use_patch_pos_encoding)
        self.use_pos_encoding = use_pos_encoding
        self.pos_embed_observation = nn.Embedding(context_len, embed_dim)            # :149

        # rows per LM-head logits / cross-entropy launch.  Training writes the logits straight into the (already allocated)
        # dlogits buffer, so the chunk costs no memory: 32768 = one or two launches per step measured 0.2-0.3 ms ahead of 4096
        # on the m-mix step (39.28 / 39.35 -> 39.12 / 39.05 ms, profiles/r03_step_ab.txt); a forward without gradient keeps a
        # 4096-row scratch (engine.lm_head_loss)
        self.lm_head_chunk_rows = int(os.environ.get("NEKO_LM_CHUNK_ROWS", "32768"))
        self.lm_head_selected_rows = True   # LM head only at loss positions when they are known on the host
        #: > 0: training forwards (compute_loss=True, return_logits=False) pack the batch into at most this many length
        #: buckets instead of left-padding every example to the longest one (misc/todo.md:11);
        #: 0 = the reference's layout.  Same loss and gradients, fewer padded positions through the stack.
        self.ragged_groups = int(os.environ.get("NEKO_RAGGED_GROUPS", "0"))
        self.last_pack: Optional[_PackInfo] = None      # statistics of the last packed batch (bench / logging only)
        self._dp = None
        self._layout_cache: "OrderedDict" = OrderedDict()      # structural memo of _prepare (layout_signature)
        self._hp = None
        self._flat: Optional[FlatParams] = None
        self._flatten(torch.device(device))

    # ---- flat storage ----------------------------------------------------------------------------
    def _flatten(self, device: torch.device) -> None:
        named = dict(self.named_parameters())
        groups: "OrderedDict[str, list]" = OrderedDict()
        groups["frontend"] = [(n, named[n]) for n in ("embed_token.weight", "pos_embed_observation.weight",
                                                      "separator_token")]
        groups["image"] = [(n, named[n]) for n in self.image_embedding.flat_param_names("image_embedding.")]
        for g, plist in self.transformer.param_groups_for_flat("transformer.").items():
            groups[g] = plist
        groups["head"] = [("predict_token.weight", named["predict_token.weight"])]
        groups["never"] = [("transformer.wte.weight", named["transformer.wte.weight"])]
        self.Vpad = (self.vocab_size + VPAD_ALIGN - 1) // VPAD_ALIGN * VPAD_ALIGN
        self._flat = FlatParams(groups, device, padded_numel={"predict_token.weight": self.Vpad * self.embed_dim})
        self.transformer.attach_flat(self._flat, "transformer.")
        self.image_embedding.attach_flat(self._flat, "image_embedding.")
        self._hp = None

    def _apply(self, fn, *a, **k):
        """.to() / .cuda() / .float(): a call that leaves every parameter where it is (the usual `model.to(device)`
        after construction, train.py:106, accelerator.prepare) keeps the flat storage -- optimisers, gradient reducers
        and captured decode graphs hold pointers into it.  A call that really moves or converts parameters rebuilds
        the storage; objects built on the old one (NekoAdamW, GradReducer) then refuse to run."""
        super()._apply(fn, *a, **k)
        f = getattr(self, "_flat", None)
        if f is None:
            return self
        intact = all(p.device == f.device and p.dtype == torch.float32 and p.data_ptr() == f.view(n).data_ptr()
                     for n, p in f.param_of.items())
        if not intact:
            dev = next(self.parameters()).device
            self.device = dev
            self._flatten(dev)
            self._decoders = {}          # captured graphs point into the old storage
            self.last_pack = None
        return self

    def _frontend_names(self) -> List[str]:
        n = ["embed_token.weight", "separator_token"]
        if self.use_pos_encoding:
            n.append("pos_embed_observation.weight")
        return n

    def _head_params(self) -> engine.HeadParams:
        if self._hp is None:
            self._hp = engine.HeadParams(V=self.vocab_size, Vpad=self.Vpad,
                                         w=self._flat.sview("predict_token.weight", padded_rows=self.Vpad),
                                         g_w=self._flat.gview("predict_token.weight", padded_rows=self.Vpad))
        return self._hp

    @property
    def module(self):
        return self

    def _dev(self) -> torch.device:
        return self._flat.device

    def _gather_values(self, parts, dtype, dev):
        """Concatenated value buffer on the device.  Host-resident parts are concatenated on the host and go over in
        one asynchronous pinned copy (a pageable .to(device) would block the host until the stream drains)."""
        if not parts:
            return None
        if all(t.is_cuda for t in parts):
            return torch.cat([t.to(dev, dtype) for t in parts])
        if not any(t.is_cuda for t in parts):
            return self.image_embedding._upload(torch.cat([t.to(dtype).reshape(-1) for t in parts]), dev)
        return torch.cat([(t if t.is_cuda else self.image_embedding._upload(t.to(dtype).contiguous(), dev)).to(dev, dtype).reshape(-1)
                          for t in parts])

    # ---- packing (gato_policy.py:195-432) -----------------------------------------------------------
    def tokenize_input_dicts(self, inputs: list):
        """Returns (token_embeddings (B,T,d) f32, tokens (B,T) i64, token_target_masks (B,T) f32,
        token_masks (B,T) f32) like the reference; the work is one descriptor upload + HIP kernels."""
        return self._tokenize(inputs, 0)[:4]

    def _prepare(self, inputs: list, ragged_groups: int) -> "_Prepared":
        """Host half of tokenize_input_dicts: the descriptor table (numpy), the concatenated value buffers, the image
        groups with their patch positions and the loss-row indices, all on the device when this returns.  No kernel of
        the model runs here, so a captured step (training/captured.py) can replay its graph on a `_Prepared` whose
        tensors were copied into the graph's static inputs."""
        dev = self._dev()
        if dev.type != "cuda":
            raise RuntimeError("neko_amd.GatoPolicy computes on the GPU only (no CPU fallback)")
        if self.pad_seq:
            ragged_groups = 0           # pad_seq asks for context_len-wide rows (gato_policy.py:423-431)
        sorted_tail = bool(torch.is_grad_enabled() and ops.SORTED_SCATTER)
        pos_rows = self.pos_embed_observation.weight.shape[0]
        sig = layout_signature(inputs) if LAYOUT_CACHE > 0 else None
        key = None if sig is None else (sig, self.use_pos_encoding, self.context_len, self.pad_seq, ragged_groups, sorted_tail, pos_rows, str(dev))
        hit = self._layout_cache.get(key) if key is not None else None
        if hit is not None:
            self._layout_cache.move_to_end(key)
            pb = collect_sources(inputs)
            pb.B, pb.T, pb.segments, pb.order = hit.B, hit.T, hit.segments, hit.order
        else:
            pb = build_layout(inputs, self.use_pos_encoding, self.context_len, self.pad_seq, ragged_groups=ragged_groups)
        has_text = hit.has_text if hit is not None else bool(np.isin(pb.desc[:, 0], (K_TOKEN, K_DEVID)).any())
        if self._dp is not None and getattr(self._dp, "no_text_declared", False) and has_text:
            raise RuntimeError("text tokens in a batch, but the data-parallel reducer was told that the text rows of "
                               "embed_token never receive gradients (GradReducer.declare_unused_rows)")
        pr = _Prepared()
        pr.B, pr.T = pb.B, pb.T
        pr.sorted_tail = sorted_tail        # (part of the cache key: a hit was built in the same mode)
        if hit is not None:
            pr.desc, idx_dev, n_sel, map_dev, tgt_dev = hit.desc_dev, hit.idx_dev, hit.n_sel, hit.map_dev, hit.tgt_dev
        else:
            # behind the descriptors: the (local position | separator) destinations of the packing backward, sorted on the host
            # (ops.sorted_pairs; _PackEmbedV2.backward hands them to neko_pack_embed_bwd_sorted) -- one flat int32 tensor of 6 ints per row
            M = pb.desc.shape[0]
            if sorted_tail:
                kind, pos = pb.desc[:, 0], pb.desc[:, 2]
                skey = np.where(kind == K_SEP, pos_rows, np.where((pos >= 0) & (kind != K_PAD) & (pos < pos_rows), pos, -1))
                ks, ix = ops.sorted_pairs(skey)
            else:
                ks, ix = np.full(M, ops.SEGSUM_KEY_NONE, np.int32), np.zeros(M, np.int32)
            pr.desc = self.image_embedding._upload(torch.from_numpy(np.concatenate([pb.desc.reshape(-1), ks, ix])), dev)
            # loss positions are known on the host (gato_policy.py:176-183): row (b,t) is selected when position t is
            # real and position t+1 is a target.  Uploaded once; lets the LM head run on the selected rows only.
            B, T = pb.B, pb.T
            selm = np.zeros(B * T, dtype=bool)
            for (r0, Bk, Tk) in (pb.segments or [(0, B, T)]):
                dk = pb.desc[r0:r0 + Bk * Tk].reshape(Bk, Tk, 4)
                sk = np.zeros((Bk, Tk), dtype=bool)
                sk[:, :-1] = (dk[:, :-1, 0] != K_PAD) & (dk[:, 1:, 3] != 0)
                selm[r0:r0 + Bk * Tk] = sk.reshape(-1)
            sel_idx = np.flatnonzero(selm).astype(np.int32)
            n_sel = int(sel_idx.size)
            # one upload: [loss rows (at least one slot) | their target positions = row + 1 | row -> loss-row index, -1 elsewhere]
            nslot = max(n_sel, 1)
            both = np.full(2 * nslot + B * T, -1, dtype=np.int32)
            both[:2 * nslot] = 0
            both[:n_sel] = sel_idx
            both[nslot:nslot + n_sel] = sel_idx + 1
            both[2 * nslot + sel_idx] = np.arange(n_sel, dtype=np.int32)
            both_dev = self.image_embedding._upload(torch.from_numpy(both), dev)
            idx_dev, tgt_dev, map_dev = both_dev[:nslot], both_dev[nslot:2 * nslot], both_dev[2 * nslot:]
            if key is not None:
                # the cached device tensors are read-only inputs of the packing kernels and of the LM head's row gather; clones, so that
                # nothing else owns their storage
                ent = _LayoutEntry()
                ent.B, ent.T, ent.segments, ent.order, ent.has_text = pb.B, pb.T, pb.segments, pb.order, has_text
                ent.desc_dev, ent.idx_dev, ent.n_sel, ent.map_dev, ent.tgt_dev = pr.desc.clone(), idx_dev.clone(), n_sel, map_dev.clone(), tgt_dev.clone()
                self._layout_cache[key] = ent
                while len(self._layout_cache) > LAYOUT_CACHE:
                    self._layout_cache.popitem(last=False)
        pr.cont = self._gather_values(pb.cont, torch.float32, dev)
        pr.disc = self._gather_values(pb.disc, torch.int32, dev)
        pr.img_order = list(pb.img_order)
        pr.img_ids = [idx for kind, idx in pb.img_order if kind == "img"]
        # positions drawn per example (in order), kernels batched per image shape
        pr.img_groups = self.image_embedding.prepare_many([pb.images[i] for i in pr.img_ids]) if pr.img_ids else []
        pr.given = [e.to(dev, torch.float32) for e in pb.given_img_emb]
        pr.pack = _PackInfo(idx_dev, n_sel, pb.segments, pb.order, pb.B * pb.T, map_dev, tgt_dev)
        return pr

    def _embed_prepared(self, pr: "_Prepared"):
        """Device half: patch embedding of the image groups and the packing kernel.  Returns (x, tokens, target masks,
        pad masks, pack info) -- (B,T,.) tensors, or (1,M,.) tensors with pack.segments = [(row0, B_k, T_k)]."""
        img_emb = None
        if pr.img_order:
            embedded = dict(zip(pr.img_ids, self.image_embedding.embed_groups(pr.img_groups, len(pr.img_ids)))) if pr.img_ids else {}
            parts = []
            for kind, idx in pr.img_order:
                e = embedded[idx] if kind == "img" else pr.given[idx]
                parts.append(e.reshape(-1, self.embed_dim))
            img_emb = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        params = [self._flat.param_of[n] for n in self._frontend_names()]
        x, tokens, tmask, pmask = _PackEmbedV2.apply(self, pr.desc, pr.cont, pr.disc, img_emb, pr.B * pr.T, bool(pr.sorted_tail), *params)
        B, T, d = pr.B, pr.T, self.embed_dim
        self.last_pack = pr.pack
        return x.view(B, T, d), tokens.view(B, T), tmask.view(B, T), pmask.view(B, T), pr.pack

    def _tokenize(self, inputs: list, ragged_groups: int):
        """tokenize_input_dicts, optionally in the length-bucketed layout."""
        return self._embed_prepared(self._prepare(inputs, ragged_groups))

    def _loss_from_prepared(self, pr: "_Prepared"):
        """The training call `forward(inputs, compute_loss=True, return_logits=False)` on an already prepared batch."""
        x, tokens, tmask, pmask, pack = self._embed_prepared(pr)
        names = self.transformer._param_names() + ["predict_token.weight"]
        params = [self._flat.param_of[n] for n in names]
        _, loss = _PolicyCoreFn.apply(self, x, pmask.to(torch.float32), tokens, tmask, True, False, pack, *params)
        return loss

    # ---- forward (gato_policy.py:156-192) -------------------------------------------------------------
    def forward(self, inputs: Optional[list] = None, compute_loss=False, **kwargs):
        return_logits = kwargs.pop("return_logits", True)
        if inputs is not None:
            # length-bucketed layout: only when no (B, T, V) logits tensor has to be handed back (the training call,
            # trainer.py:178 discards them) -- the loss and every gradient are those of the padded layout
            ragged = self.ragged_groups if (compute_loss and not return_logits and len(inputs) > 1) else 0
            token_embeddings, tokens, token_target_masks, token_masks, pack = self._tokenize(inputs, ragged)
        else:
            pack = None
            assert ("token_embeddings" in kwargs and "tokens" in kwargs and "token_target_masks" in kwargs
                    and "token_masks" in kwargs), "if inputs is None, must provide embeddings, tokens, and masks"
            token_embeddings = kwargs["token_embeddings"]
            tokens = kwargs["tokens"]
            token_target_masks = kwargs["token_target_masks"]
            token_masks = kwargs["token_masks"]
        if not token_embeddings.is_cuda:
            raise RuntimeError("neko_amd.GatoPolicy computes on the GPU only (no CPU fallback)")
        if compute_loss:
            assert tokens is not None and token_target_masks is not None, "compute_loss needs tokens and target masks"
        names = self.transformer._param_names() + ["predict_token.weight"]
        params = [self._flat.param_of[n] for n in names]
        logits, loss = _PolicyCoreFn.apply(self, token_embeddings, token_masks.to(torch.float32), tokens,
                                           token_target_masks, bool(compute_loss), bool(return_logits), pack, *params)
        return (logits if return_logits else None), (loss if compute_loss else None)

    # ---- inference helpers (gato_policy.py:434-470, 477-544, 556-614) ----------------------------------------------
    # KV-cached decode.  The reference re-runs the whole forward for every generated token;
    # here only the new positions go through the stack (engine.KVDecoder) and the LM head runs on the last row only.
    # Semantics are the reference's, including the sliding window: once context_len positions are in flight the
    # window is truncated on the left and RE-PRIMED (every remaining row recomputed without the dropped ones, exactly
    # what a full forward over the truncated window computes).  Pinned by fixture G12 (tests/golden/
    # make_fixture_decode.py: the reference's own predict_text / predict_response / predict_control outputs).
    def _decode_tokens(self, token_embeddings, n_tokens, start_token, end_token, deterministic):
        """Autoregressive continuation of ONE unpadded sequence.  Returns (logits (n_tokens, end-start+1) fp32,
        list of 0-d token tensors in the global vocabulary)."""
        assert token_embeddings.shape[0] == 1, "decode works on a single sequence"
        emb = token_embeddings[0, -self.context_len:, :].to(torch.float32)
        hp = self._head_params()
        self._flat.ensure_shadow()
        table = self._flat.view("embed_token.weight")
        use_graph = (deterministic and n_tokens > 1 and emb.shape[0] + n_tokens - 1 <= self.context_len
                     and os.environ.get("NEKO_DECODE_GRAPH", "1") != "0")
        # decoders (cache buffers + the captured step) are kept per token range: capture costs ~100 ms, a rollout calls
        # this thousands of times.  The graph holds pointers into the flat parameter storage and its bf16 shadow, which
        # never move, so it stays valid across optimiser steps (ensure_shadow above refreshes the shadow in place).
        if not hasattr(self, "_decoders"):
            self._decoders = {}
        key = (start_token, end_token) if use_graph else None
        dec = self._decoders.get(key)
        if dec is None:
            dec = engine.KVDecoder(self.transformer._stack_params(), self.context_len, emb.device)
            if use_graph:
                dec.capture_greedy_step(hp, table, start_token, end_token)
            self._decoders[key] = dec
        dec.reset()
        if use_graph:
            # greedy continuation that never slides the window: one eager step on the primed rows, then every further
            # token is ONE replay of a captured HIP graph (stack on the new row + LM head + argmax + embedding lookup)
            h = dec.extend(emb)
            logits = engine.lm_head_rows(hp, h[-1:])[0, start_token:(end_token + 1)]
            token = torch.argmax(logits, dim=-1) + start_token
            all_logits, tokens = [logits], [token]
            dec.g_x.copy_(torch.index_select(table, 0, token.reshape(1)))
            for _ in range(n_tokens - 1):
                lg, tk = dec.replay_greedy_step()
                all_logits.append(lg)
                tokens.append(tk)
            return torch.stack(all_logits, dim=0), tokens
        h = dec.extend(emb)
        all_logits, tokens = [], []
        for _ in range(n_tokens):
            logits = engine.lm_head_rows(hp, h[-1:])[0, start_token:(end_token + 1)]
            if deterministic:
                token = torch.argmax(logits, dim=-1)
            else:
                token = torch.multinomial(torch.nn.functional.softmax(logits, dim=-1), num_samples=1)[0]
            token = token + start_token
            all_logits.append(logits)
            tokens.append(token)
            new = table[token].reshape(1, -1)
            emb = torch.cat([emb, new], dim=0)
            if emb.shape[0] > self.context_len:           # window slides: reference semantics = recompute the window
                emb = emb[-self.context_len:]
                dec.reset()
                h = dec.extend(emb)
            else:
                h = dec.extend(new)
        return torch.stack(all_logits, dim=0), tokens

    @torch.no_grad()
    def predict_text(self, batch_dict, max_length=20, deterministic=True):
        start_token, end_token = self.token_starts["text"], self.token_ends["text"]
        token_embeddings, _, _, _ = self.tokenize_input_dicts([batch_dict])
        return self._decode_tokens(token_embeddings, max_length, start_token, end_token, deterministic)

    @torch.no_grad()
    def predict_response(self, image, prompt_tokens=[], max_length=128, deterministic=True):
        start_token, end_token = self.token_starts["text"], self.token_ends["text"]
        image_embeddings = self.image_embedding(image)
        assert image_embeddings.shape[0] == 1, "number of images should always be 1 for predicting response"
        return self._predict_response_cached(image_embeddings, list(prompt_tokens), max_length, start_token, end_token,
                                             deterministic)

    def _predict_response_cached(self, image_embeddings, prompt_tokens, max_length, start_token, end_token, deterministic):
        """The packed sequence is [patches | prompt | response ... | SEP]: the logits of the last text position do not
        depend on the trailing separator (causal), so the cache holds the sequence WITHOUT it and every step appends
        the embedding row of the newly chosen token, taken from the packing kernels (token + local position embedding)
        so the rows are the very ones the full forward would see."""
        dev = self._dev()
        dec = engine.KVDecoder(self.transformer._stack_params(), self.context_len, dev)
        hp = self._head_params()
        self._flat.ensure_shadow()
        n_patches = image_embeddings.shape[1]
        pred_logits, response_tokens = [], []
        for idx in range(max_length):
            batch_dict = {"image_embeddings": image_embeddings, "text": torch.tensor(prompt_tokens + response_tokens)}
            emb, _, _, _ = self.tokenize_input_dicts([batch_dict])
            n_seq = n_patches + len(prompt_tokens) + idx          # positions before the separator
            rows = emb[0, :n_seq, :].to(torch.float32)
            if n_seq > self.context_len:
                raise ValueError("predict_response: sequence exceeds context_len")
            if n_seq == 0:
                raise ValueError("predict_response needs at least one position (image or prompt)")
            h = dec.extend(rows[dec.n:])
            nxt = engine.lm_head_rows(hp, h[-1:])[0, start_token:(end_token + 1)]
            if deterministic:
                next_token = torch.argmax(nxt).item()
            else:
                next_token = torch.multinomial(torch.nn.functional.softmax(nxt, dim=-1), num_samples=1).item()
            pred_logits.append(nxt)
            response_tokens.append(next_token)
        return torch.stack(pred_logits, dim=0), self.text_tokenizer.decode(response_tokens)

    def predict_caption(self, image, max_length=128, deterministic=True):
        return self.predict_response(image, prompt_tokens=[], max_length=max_length, deterministic=deterministic)

    def predict_answer(self, image, question, max_length=16, deterministic=True):
        return self.predict_response(image, prompt_tokens=self.text_tokenizer.encode(question), max_length=max_length,
                                     deterministic=deterministic)

    @torch.no_grad()
    def predict_control(self, input: dict, task, deterministic: bool = True):
        """gato_policy.py:556-614: the action of ONE control example whose last timestep's action slots are padding.
        ``task.action_type`` is compared by class name (gymnasium is not a dependency of the HIP path)."""
        kind = getattr(task.action_type, "__name__", str(task.action_type))
        discrete = kind in ("Discrete", "DiscreteSpace")                      # neko_amd.tasks.control_task stand-in
        n_tok = task.action_tokens
        space = "discrete" if discrete else "continuous"
        first, last = self.token_starts[space], self.token_ends[space]
        if discrete:
            assert n_tok == 1, "only support 1 discrete action token"
            assert task.env.action_space.n <= self.discrete_tokens, "discrete action space too large for model"
            last = first + task.env.action_space.n - 1
        emb = self.tokenize_input_dicts([input])[0][:, :-n_tok, :]           # history without the padded action slots
        # one priming pass over the history, then one cached step per action token
        _, chosen = self._decode_tokens(emb, n_tok, first, last, deterministic)
        if discrete:
            return chosen[0] - first
        return self.continuous_action_tokenizer.decode(torch.stack(chosen, dim=0))


class _PackEmbedV2(torch.autograd.Function):
    """Packing front-end: descriptor table -> (x, tokens, target mask, pad mask); backward scatters
    d_x into embed_token / pos_embed_observation / separator_token grads and returns d(img_emb)."""

    @staticmethod
    def forward(ctx, policy: GatoPolicy, desc, cont, disc, img_emb, ntok, sorted_tail, *params):
        f = policy._flat
        d = policy.embed_dim
        img = None if img_emb is None else img_emb.detach().contiguous()
        x, tokens, tmask, pmask = ops.pack_embed_fwd(
            desc, cont, disc, img, f.view("embed_token.weight"), f.view("pos_embed_observation.weight"),
            f.view("separator_token"), ntok, d, policy.mu, policy.M, policy.continuous_tokens,
            policy.token_starts["continuous"], policy.token_starts["discrete"])
        ctx.policy, ctx.desc, ctx.tokens, ctx.ntok = policy, desc, tokens, ntok
        # explicit, never inferred from desc.numel(): a batch prepared under no_grad carries a placeholder tail
        ctx.sorted_tail = bool(sorted_tail)
        ctx.img_rows = 0 if img_emb is None else img_emb.shape[0]
        ctx.mark_non_differentiable(tokens, tmask, pmask)
        return x, tokens, tmask, pmask

    @staticmethod
    def backward(ctx, gx, *_unused):
        policy = ctx.policy
        f, d = policy._flat, policy.embed_dim
        names = policy._frontend_names()
        f.prepare_backward(names + ([] if policy.use_pos_encoding else ["pos_embed_observation.weight"]))
        gx = gx.contiguous().to(torch.float32)
        d_img = None
        if ctx.img_rows > 0 and ctx.needs_input_grad[4]:
            d_img = torch.zeros(ctx.img_rows, d, dtype=torch.float32, device=gx.device)
        ops.pack_embed_bwd(ctx.desc, ctx.tokens, gx.view(-1, d), f.gview("embed_token.weight"),
                           f.gview("pos_embed_observation.weight"), f.gview("separator_token"), d_img, ctx.ntok, d,
                           sorted_tail=ctx.sorted_tail)
        f.attach_grads(names)
        if policy._dp is not None:
            policy._dp.group_ready("frontend")
        return (None, None, None, None, d_img, None) + (None,) * (len(ctx.needs_input_grad) - 6)
