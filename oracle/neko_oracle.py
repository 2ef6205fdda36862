"""CPU oracle for the NEKO/Gato hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32 restatement (functional, over a ``state_dict``) of the reference's
forward/backward path: continuous tokenizer, patch-position indices, image patch embedding,
``tokenize_input_dicts`` packing, the trajectory-GPT2 transformer, LM head + masked
cross-entropy, the LR schedule and one ``train_step``.  Written from SURVEY.md and the
reference's behaviour; every function cites the reference file:line it follows
(paths relative to /root/reference).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``neko_amd``) never imports it and has no CPU fallback.

PARITY PIN: the reference has no tests/golden vectors for this path,
so this oracle is pinned against fixtures produced by importing the reference itself in the
build container: ``tests/golden/make_fixtures.py`` -> ``tests/golden/*.pt``; checked by
``tests/test_oracle_golden.py`` (fixtures G1-G12; every generator script under
``tests/golden/make_fixture*.py`` re-runs the imported reference when /root/reference is present).

Floating-point gradients come from torch autograd over these differentiable functions.
``bf16=True`` arguments emulate the HIP kernels' rounding points (bf16 GEMM operands and
stored activations, fp32 accumulation/statistics) so kernel tests can use tight tolerances.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Mirror of the GatoPolicy ctor arguments that matter for the path
    (gato/policy/gato_policy.py:19-48)."""
    embed_dim: int
    layers: int
    heads: int
    text_tokens: int = 50257           # tokenizer.vocab_size (gato_policy.py:60)
    continuous_tokens: int = 1024      # :36
    discrete_tokens: int = 1024        # :37
    context_len: int = 1024            # :39
    mu: float = 100.0                  # :29
    M: float = 256.0                   # :30
    patch_size: int = 16               # :32
    num_groups: int = 32               # :34
    position_vocab_size: int = 128     # :35
    use_pos_encoding: bool = True      # :41
    use_patch_pos_encoding: bool = True  # :42
    layer_norm_eps: float = 1e-5       # HF GPT2Config default (trajectory_gpt2.py:301)
    pad_seq: bool = False              # :47
    activation_fn: str = "gelu"        # :27 ('geglu' adds the gate, trajectory_gpt2.py:267-276)

    @property
    def vocab_size(self) -> int:       # gato_policy.py:63
        return self.text_tokens + self.discrete_tokens + self.continuous_tokens

    @property
    def continuous_start(self) -> int:  # gato_policy.py:68
        return self.text_tokens

    @property
    def discrete_start(self) -> int:    # gato_policy.py:69
        return self.text_tokens + self.continuous_tokens


def _rb(x: Tensor, bf16: bool) -> Tensor:
    """Round to bf16 and back (emulates a bf16 store) when bf16 emulation is on."""
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


# --------------------------------------------------------------------------------------
# A1: continuous tokenizer   (gato/policy/input_tokenizers.py:5-30)
# --------------------------------------------------------------------------------------
def mu_law(x: Tensor, mu: float = 100.0, M: float = 256.0) -> Tensor:
    """input_tokenizers.py:5-6 -- sign(x) * log(1 + mu|x|) / log(1 + mu*M)."""
    return torch.sign(x) * torch.log(1 + mu * torch.abs(x)) / math.log(1 + mu * M)


def tokenize_continuous(x: Tensor, use_mu_law: bool, mu: float = 100.0, M: float = 256.0,
                        n_bins: int = 1024, offset: Optional[int] = None) -> Tensor:
    """input_tokenizers.py:17-30 -- optional mu-law, clamp to [-1,1], (x+1)*(n_bins/2),
    truncation toward zero to int32, + offset.  NB value +1.0 lands in bin ``n_bins``."""
    if use_mu_law:
        x = mu_law(x, mu, M)
    x = torch.clamp(x, -1, 1)
    x = (x + 1) * (n_bins / 2)
    t = x.to(torch.int32)
    if offset is not None:
        t = t + offset
    return t


def detokenize_continuous(t: Tensor, n_bins: int = 1024, offset: Optional[int] = None) -> Tensor:
    """input_tokenizers.py:32-42 (decode, actions only)."""
    if offset is not None:
        t = t - offset
    return (2 * t) / n_bins - 1


# --------------------------------------------------------------------------------------
# A4: patch position indices + image embedding   (gato/policy/embeddings.py)
# --------------------------------------------------------------------------------------
def patch_pos_intervals(n: int, vocab: int = 128) -> Tensor:
    """embeddings.py:80-89 -- int32 [n,2] (lo, hi) quantised intervals of each of the n patches."""
    lin = torch.linspace(0, 1, n + 1)
    iv = torch.stack([lin[:-1], lin[1:]]).T
    return (iv * vocab).to(torch.int32)


def patch_pos_indices_eval(n: int, vocab: int = 128) -> Tensor:
    """embeddings.py:96-100 -- eval mode: round(mean(lo, hi-1))."""
    iv = patch_pos_intervals(n, vocab).clone()
    iv[:, 1] = iv[:, 1] - 1
    return iv.mean(dim=-1, dtype=torch.float32).round().to(torch.int32)


def residual_block_v2(sd: Dict[str, Tensor], x: Tensor, num_groups: int,
                      prefix: str = "image_embedding.patch_embedding.") -> Tensor:
    """embeddings.py:111-131 -- x + conv2(GELU(GN(conv1(GELU(x))))), 3x3 convs, padding 1."""
    h = F.conv2d(F.gelu(x), sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], padding=1)
    h = F.group_norm(h, num_groups, sd[prefix + "gn2.weight"], sd[prefix + "gn2.bias"], eps=1e-5)
    h = F.conv2d(F.gelu(h), sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], padding=1)
    return x + h


def image_embedding(sd: Dict[str, Tensor], cfg: OracleConfig, images: Tensor,
                    h_pos: Optional[Tensor] = None, w_pos: Optional[Tensor] = None,
                    normalize: bool = True, bf16: bool = False) -> Tensor:
    """embeddings.py:28-61 -- (n,3,H,W) in 0..255 -> (n, H/p*W/p, d).
    h_pos/w_pos: explicit patch-position indices (train mode draws them at random,
    embeddings.py:92-94; eval mode uses patch_pos_indices_eval); None -> eval indices."""
    p = cfg.patch_size
    n, c, H, W = images.shape
    assert H % p == 0 and W % p == 0, "Image dimensions must be divisible by patch size"
    nh, nw = H // p, W // p
    x = images.to(torch.float32)
    if normalize:
        x = (x / 255.0 * 2) - 1
        x = x / math.sqrt(p)
    # 'b c (n_h p_1) (n_w p_2) -> (b n_h n_w) c p_1 p_2'
    x = x.reshape(n, c, nh, p, nw, p).permute(0, 2, 4, 1, 3, 5).reshape(n * nh * nw, c, p, p)
    x = residual_block_v2(sd, x, cfg.num_groups)
    # '(b n_h n_w) c p_1 p_2 -> b n_h n_w (c p_1 p_2)'
    x = x.reshape(n, nh, nw, c * p * p)
    W_ = sd["image_embedding.post_embedding_projection.weight"]
    x = F.linear(_rb(x, bf16), _rb(W_, bf16), sd["image_embedding.post_embedding_projection.bias"])
    if cfg.use_patch_pos_encoding:
        if h_pos is None:
            h_pos = patch_pos_indices_eval(nh, cfg.position_vocab_size)
        if w_pos is None:
            w_pos = patch_pos_indices_eval(nw, cfg.position_vocab_size)
        hp = sd["image_embedding.patch_pos_encoding.height_pos_embedding.weight"][h_pos.long()]
        wp = sd["image_embedding.patch_pos_encoding.width_pos_embedding.weight"][w_pos.long()]
        x = x + (hp[:, None, :] + wp[None, :, :])[None]
    return x.reshape(n, nh * nw, -1)


# --------------------------------------------------------------------------------------
# A2/A3: packing   (gato/policy/gato_policy.py:195-432)
# --------------------------------------------------------------------------------------
def tokenize_input_dicts(sd: Dict[str, Tensor], cfg: OracleConfig, inputs: Sequence[dict],
                         patch_positions: Optional[Sequence[Optional[Tuple[Tensor, Tensor]]]] = None,
                         bf16: bool = False):
    """gato_policy.py:195-432.  Per example: per-timestep order
    [image patches | text | continuous obs | discrete obs | SEP | continuous act | discrete act]
    (:355), local position embedding added to the observation tokens only (:380-385),
    flatten (:398-400), left-pad to the batch max (:408-416), optional right pad to context_len
    (:423-431).  Returns (embeddings (B,T,d) f32, tokens (B,T) i64, target mask (B,T) f32,
    pad mask (B,T) f32)."""
    E = sd["embed_token.weight"]
    d = cfg.embed_dim
    embs, toks, tgts = [], [], []
    for ex_i, ex in enumerate(inputs):
        tok_parts: List[Optional[Tensor]] = [None] * 7
        tgt_parts: List[Optional[Tensor]] = [None] * 7
        obs_emb: List[Tensor] = []
        act_emb: List[Tensor] = []
        n_ts = None

        def _set_ts(n):
            nonlocal n_ts
            if n_ts is None:
                n_ts = n
            else:
                assert n_ts == n, "number of timesteps must be the same for all modalities"

        text_e = img_e = None
        if ex.get("text") is not None:                                    # :264-277
            t = ex["text"]
            t = torch.tensor(t) if isinstance(t, list) else t
            t = t.long()
            if t.dim() == 1:
                t = t.unsqueeze(0)
            text_e = E[t]
            tok_parts[1], tgt_parts[1] = t, torch.ones_like(t, dtype=torch.float32)
            n_ts = t.shape[0]
        if ex.get("images") is not None or ex.get("image_embeddings") is not None:   # :282-296
            if ex.get("images") is not None:
                hp = wp = None
                if patch_positions is not None and patch_positions[ex_i] is not None:
                    hp, wp = patch_positions[ex_i]
                img_e = image_embedding(sd, cfg, ex["images"], hp, wp, bf16=bf16)
            if ex.get("image_embeddings") is not None:
                img_e = ex["image_embeddings"]
            n_img, n_patch = img_e.shape[0], img_e.shape[1]
            tok_parts[0] = torch.zeros(n_img, n_patch, dtype=torch.long)
            tgt_parts[0] = torch.zeros(n_img, n_patch)
            _set_ts(n_img)
        if img_e is not None:
            obs_emb.append(img_e)
        if text_e is not None:
            obs_emb.append(text_e)
        if ex.get("continuous_obs") is not None:                          # :298-306
            t = tokenize_continuous(ex["continuous_obs"], True, cfg.mu, cfg.M,
                                    cfg.continuous_tokens, cfg.continuous_start).long()
            tok_parts[2], tgt_parts[2] = t, torch.zeros_like(t, dtype=torch.float32)
            obs_emb.append(E[t])
            _set_ts(t.shape[0])
        if ex.get("discrete_obs") is not None:                            # :308-317
            t = ex["discrete_obs"].long() + cfg.discrete_start
            tok_parts[3], tgt_parts[3] = t, torch.zeros_like(t, dtype=torch.float32)
            obs_emb.append(E[t])
            _set_ts(t.shape[0])
        if ex.get("continuous_actions") is not None:                      # :319-327
            t = tokenize_continuous(ex["continuous_actions"], False, cfg.mu, cfg.M,
                                    cfg.continuous_tokens, cfg.continuous_start).long()
            tok_parts[5], tgt_parts[5] = t, torch.ones_like(t, dtype=torch.float32)
            act_emb.append(E[t])
            _set_ts(t.shape[0])
        if ex.get("discrete_actions") is not None:                        # :329-340
            t = ex["discrete_actions"].long() + cfg.discrete_start
            tok_parts[6], tgt_parts[6] = t, torch.ones_like(t, dtype=torch.float32)
            act_emb.append(E[t])
            _set_ts(t.shape[0])

        sep = torch.ones(n_ts, 1, d) * sd["separator_token"]              # :343-345
        tok_parts[4] = torch.zeros(n_ts, 1, dtype=torch.long)
        tgt_parts[4] = torch.zeros(n_ts, 1)

        ex_tok = torch.cat([t for t in tok_parts if t is not None], dim=1)          # :350-359
        ex_tgt = torch.cat([t.to(torch.float32) for t in tgt_parts if t is not None], dim=1)
        ex_emb = torch.cat(obs_emb, dim=1)                                          # :371-378
        n_obs = ex_emb.shape[1]
        if cfg.use_pos_encoding:                                                    # :380-385
            ex_emb = ex_emb + sd["pos_embed_observation.weight"][:n_obs].unsqueeze(0)
        a = torch.cat(act_emb, dim=1) if act_emb else torch.zeros(n_ts, 0, d)        # :387-392
        ex_emb = torch.cat([ex_emb, sep, a], dim=1)                                 # :393
        total = n_ts * ex_emb.shape[1]
        embs.append(ex_emb.reshape(1, total, d))
        toks.append(ex_tok.reshape(1, total))
        tgts.append(ex_tgt.reshape(1, total))

    T = max(e.shape[1] for e in embs)
    B = len(embs)
    out_e = torch.zeros(B, T, d)
    out_t = torch.zeros(B, T, dtype=torch.long)
    out_g = torch.zeros(B, T)
    out_m = torch.zeros(B, T)
    for i in range(B):                                                    # :408-416 left pad
        n = embs[i].shape[1]
        out_e[i, T - n:] = embs[i][0]
        out_t[i, T - n:] = toks[i][0]
        out_g[i, T - n:] = tgts[i][0]
        out_m[i, T - n:] = 1.0
    if cfg.pad_seq and cfg.context_len > T:                               # :423-431 right pad
        pad = cfg.context_len - T
        out_e = F.pad(out_e, (0, 0, 0, pad))
        out_t = F.pad(out_t, (0, pad))
        out_g = F.pad(out_g, (0, pad))
        out_m = F.pad(out_m, (0, pad))
    return out_e, out_t, out_g, out_m


# --------------------------------------------------------------------------------------
# A5-A10: transformer   (gato/transformers/trajectory_gpt2.py)
# --------------------------------------------------------------------------------------
def conv1d(x: Tensor, w: Tensor, b: Tensor, bf16: bool = False) -> Tensor:
    """HF Conv1D (trajectory_gpt2.py:139-141,264-265): addmm(b, x2d, W) with W stored (in,out)."""
    shp = x.shape[:-1] + (w.shape[1],)
    y = torch.addmm(b, _rb(x, bf16).reshape(-1, x.shape[-1]), _rb(w, bf16))
    return y.view(shp)


def attention_core(q: Tensor, k: Tensor, v: Tensor, pad_mask: Tensor, bf16: bool = False,
                   drop_mask: Optional[Tensor] = None) -> Tensor:
    """trajectory_gpt2.py:163-188 (_attn) + :663-679 (mask prep), dropout 0.
    q,k,v: (B,H,T,hd); pad_mask (B,T) 1=real 0=pad.
    w = q k^T / sqrt(hd); causal: where(tril, w, -1e4) (REPLACE); padding: w += (1-mask)*-1e4 (ADD);
    softmax; w v."""
    B, H, T, hd = q.shape
    w = torch.matmul(q, k.transpose(-1, -2)) / (float(hd) ** 0.5)
    tril = torch.tril(torch.ones(T, T, dtype=torch.bool))
    w = torch.where(tril, w, torch.tensor(-1e4, dtype=w.dtype))
    w = w + ((1.0 - pad_mask.to(w.dtype)) * -10000.0)[:, None, None, :]
    w = torch.softmax(w, dim=-1)
    if drop_mask is not None:          # attn_dropout (:179): explicit multiplicative mask (0 or 1/(1-p))
        w = w * drop_mask
    return torch.matmul(_rb(w, bf16), v)


def gelu(x: Tensor) -> Tensor:
    """ACT2FN['gelu'] == exact erf GELU (trajectory_gpt2.py:266)."""
    return F.gelu(x)


def block_forward(sd: Dict[str, Tensor], cfg: OracleConfig, i: int, x: Tensor, pad_mask: Tensor,
                  bf16: bool = False, drop_masks: Optional[dict] = None) -> Tensor:
    """trajectory_gpt2.py:311-359 (Block.forward) with Attention.forward :203-257 and
    MLP.forward :273-278, no cross-attention, no cache.  Dropout is either off or given as explicit multiplicative
    masks drop_masks[('attn', i)] (B,H,T,T), [('resid_attn', i)] / [('resid_mlp', i)] (B,T,d) -- the nn.Dropout sites
    :179, :254, :278 -- so a counter-based kernel mask can be reproduced exactly."""
    dm = drop_masks or {}
    p = f"transformer.h.{i}."
    d, H = cfg.embed_dim, cfg.heads
    B, T, _ = x.shape
    a = F.layer_norm(x, (d,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], cfg.layer_norm_eps)
    qkv = _rb(conv1d(a, sd[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"], bf16), bf16)
    q, k, v = qkv.split(d, dim=2)                                          # :222
    sh = lambda t: t.view(B, T, H, d // H).permute(0, 2, 1, 3)             # :195-201
    o = attention_core(sh(q), sh(k), sh(v), pad_mask, bf16, dm.get(("attn", i)))
    o = _rb(o.permute(0, 2, 1, 3).reshape(B, T, d), bf16)                  # :190-193
    ao = conv1d(o, sd[p + "attn.c_proj.weight"], sd[p + "attn.c_proj.bias"], bf16)       # :253
    if ("resid_attn", i) in dm:
        ao = ao * dm[("resid_attn", i)]                                                   # :254
    x = x + ao                                                                             # :333
    a2 = F.layer_norm(x, (d,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], cfg.layer_norm_eps)
    pre = _rb(conv1d(a2, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], bf16), bf16)
    h = _rb(gelu(pre), bf16)                                               # :274
    if cfg.activation_fn == "geglu":                                       # :275-276
        h = h * F.linear(a2, sd[p + "mlp.gated_layer.weight"], sd[p + "mlp.gated_layer.bias"])
    mo = conv1d(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], bf16)         # :277
    if ("resid_mlp", i) in dm:
        mo = mo * dm[("resid_mlp", i)]                                                    # :278
    x = x + mo                                                                             # :355
    return x


def transformer_forward(sd: Dict[str, Tensor], cfg: OracleConfig, x: Tensor, pad_mask: Tensor,
                        bf16: bool = False, return_all: bool = False, drop_masks: Optional[dict] = None):
    """GPT2Model.forward (trajectory_gpt2.py:611-795) for inputs_embeds + attention_mask:
    no position embedding (:700-701), embedding dropout 0, L blocks, ln_f (:779)."""
    if drop_masks and "embd" in drop_masks:            # embedding dropout (:707)
        x = x * drop_masks["embd"]
    hs = [x]
    for i in range(cfg.layers):
        x = block_forward(sd, cfg, i, x, pad_mask, bf16, drop_masks)
        hs.append(x)
    out = F.layer_norm(x, (cfg.embed_dim,), sd["transformer.ln_f.weight"],
                       sd["transformer.ln_f.bias"], cfg.layer_norm_eps)
    if return_all:
        return out, hs
    return out


# --------------------------------------------------------------------------------------
# A11: LM head + masked cross-entropy   (gato/policy/gato_policy.py:172-186)
# --------------------------------------------------------------------------------------
def lm_head(sd: Dict[str, Tensor], hidden: Tensor, bf16: bool = False) -> Tensor:
    """gato_policy.py:122,172 -- Linear(d->V, no bias), weight layout (V,d)."""
    return F.linear(_rb(hidden, bf16), _rb(sd["predict_token.weight"], bf16))


def masked_cross_entropy(logits: Tensor, tokens: Tensor, target_masks: Tensor, pad_masks: Tensor) -> Tensor:
    """gato_policy.py:174-186 -- predict token t+1 from position t; position counts when the
    source token is real (pad_mask[:, :-1]) AND the next token is a target (target_mask[:, 1:]);
    mean over all selected positions of the batch."""
    V = logits.shape[-1]
    ll = logits[:, :-1, :]
    lm = (pad_masks[:, :-1] * target_masks[:, 1:]).reshape(-1)
    tt = tokens[:, 1:].reshape(-1)
    sel = lm > 0
    return F.cross_entropy(ll.reshape(-1, V)[sel], tt[sel])


def policy_forward(sd: Dict[str, Tensor], cfg: OracleConfig, emb: Tensor, tokens: Tensor,
                   target_masks: Tensor, pad_masks: Tensor, compute_loss: bool = True,
                   bf16: bool = False, drop_masks: Optional[dict] = None):
    """GatoPolicy.forward from the packed batch on (gato_policy.py:167-192)."""
    hidden = transformer_forward(sd, cfg, emb, pad_masks, bf16, drop_masks=drop_masks)
    logits = lm_head(sd, hidden, bf16)
    loss = masked_cross_entropy(logits, tokens, target_masks, pad_masks) if compute_loss else None
    return logits, loss


def policy_forward_dicts(sd, cfg, inputs, compute_loss=True, patch_positions=None, bf16=False):
    """GatoPolicy.forward(inputs=list[dict]) (gato_policy.py:156-192)."""
    emb, tok, tgt, msk = tokenize_input_dicts(sd, cfg, inputs, patch_positions, bf16)
    return policy_forward(sd, cfg, emb, tok, tgt, msk, compute_loss, bf16)


# --------------------------------------------------------------------------------------
# A13: LR schedule   (gato/training/schedulers.py:21-32)
# --------------------------------------------------------------------------------------
def lr_ratio(step: int, warmup: int, total: int, base_lr: float, init_lr: float, min_lr: float,
             cosine_decay: bool = True) -> float:
    if step <= warmup:
        lr = init_lr + (base_lr - init_lr) * step / warmup
    elif cosine_decay:
        progress = (step - warmup) / float(max(1, total - warmup))
        lr = min_lr + 0.5 * (base_lr - min_lr) * (1 + math.cos(math.pi * progress))
    else:
        lr = base_lr
    return lr / base_lr


# --------------------------------------------------------------------------------------
# A12: one train step   (gato/training/trainer.py:176-186 + train.py:127-136)
# --------------------------------------------------------------------------------------
#: parameters (state_dict keys) that are trainable; buffers attn.bias / attn.masked_bias are not.
def trainable_keys(sd: Dict[str, Tensor]) -> List[str]:
    return [k for k in sd if not (k.endswith(".attn.bias") or k.endswith(".attn.masked_bias"))]


@dataclass
class AdamWState:
    """torch.optim.AdamW semantics (train.py:127-133): decoupled weight decay on every
    parameter, bias correction with a per-parameter step count, parameters whose grad is
    None are skipped entirely (no decay, no step increment)."""
    lr: float
    beta1: float = 0.9
    beta2: float = 0.95
    eps: float = 1e-8
    weight_decay: float = 0.1
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)
    step: Dict[str, int] = field(default_factory=dict)


def clip_grad_norm(grads: Dict[str, Optional[Tensor]], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (trainer.py:181-182): total L2 norm over params that have
    a grad; coef = max_norm/(norm+1e-6) clamped to 1; grads scaled in place. Returns pre-clip norm."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in gs)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in gs:
        g.mul_(coef)
    return float(total)


def adamw_step(sd: Dict[str, Tensor], grads: Dict[str, Optional[Tensor]], st: AdamWState, lr: float):
    for k, g in grads.items():
        if g is None:
            continue
        p = sd[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(p)
            st.v[k] = torch.zeros_like(p)
            st.step[k] = 0
        st.step[k] += 1
        t = st.step[k]
        p.mul_(1 - lr * st.weight_decay)
        st.m[k].mul_(st.beta1).add_(g, alpha=1 - st.beta1)
        st.v[k].mul_(st.beta2).addcmul_(g, g, value=1 - st.beta2)
        bc1 = 1 - st.beta1 ** t
        bc2 = 1 - st.beta2 ** t
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(st.eps)
        p.addcdiv_(st.m[k], denom, value=-lr / bc1)


def loss_and_grads(sd: Dict[str, Tensor], cfg: OracleConfig, inputs=None, packed=None,
                   patch_positions=None, bf16: bool = False, drop_masks: Optional[dict] = None):
    """loss + dict of grads (None where the parameter did not take part, e.g. transformer.wte
    and, on image-free batches, image_embedding.*) -- autograd over the oracle functions."""
    keys = trainable_keys(sd)
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    full = dict(sd)
    full.update(leaf)
    if packed is None:
        emb, tok, tgt, msk = tokenize_input_dicts(full, cfg, inputs, patch_positions, bf16)
    else:
        emb, tok, tgt, msk = packed
    logits, loss = policy_forward(full, cfg, emb, tok, tgt, msk, True, bf16, drop_masks)
    gl = torch.autograd.grad(loss, [leaf[k] for k in keys], allow_unused=True)
    return loss.detach(), logits.detach(), {k: g for k, g in zip(keys, gl)}


def train_step(sd: Dict[str, Tensor], cfg: OracleConfig, st: AdamWState, inputs, lr: float,
               grad_norm_clip: Optional[float] = 1.0, patch_positions=None):
    """trainer.py:176-186: forward, backward, clip (max 1.0), AdamW step. Mutates sd in place.
    Returns (loss, pre-clip grad norm)."""
    loss, _, grads = loss_and_grads(sd, cfg, inputs, patch_positions=patch_positions)
    gn = float("nan")
    if grad_norm_clip is not None:
        gn = clip_grad_norm(grads, grad_norm_clip)
    with torch.no_grad():
        adamw_step(sd, grads, st, lr)
    return float(loss), gn


# --------------------------------------------------------------------------------------
# deterministic weights for tests/bench (NOT the reference's RNG stream: parity tests load
# the same state_dict into both sides)
# --------------------------------------------------------------------------------------
def init_state_dict(cfg: OracleConfig, seed: int = 0, resid_mid_channels: int = 128) -> Dict[str, Tensor]:
    """Same keys/shapes as GatoPolicy.state_dict() with the reference's
    init *distributions* (trajectory_gpt2.py:375-385: N(0,0.02) inside the transformer, LN 1/0;
    torch defaults outside: Embedding N(0,1), Linear/Conv kaiming-uniform; separator zeros)."""
    g = torch.Generator().manual_seed(seed)
    d, L, V = cfg.embed_dim, cfg.layers, cfg.vocab_size
    n = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    u = lambda *s, bound=1.0: (torch.rand(*s, generator=g) * 2 - 1) * bound
    sd: Dict[str, Tensor] = {}
    sd["separator_token"] = torch.zeros(d)
    sd["transformer.wte.weight"] = n(1, d, std=0.02)
    for i in range(L):
        p = f"transformer.h.{i}."
        sd[p + "ln_1.weight"] = torch.ones(d)
        sd[p + "ln_1.bias"] = torch.zeros(d)
        sd[p + "attn.bias"] = torch.tril(torch.ones(cfg.context_len, cfg.context_len, dtype=torch.uint8)
                                         ).view(1, 1, cfg.context_len, cfg.context_len)
        sd[p + "attn.masked_bias"] = torch.tensor(-1e4)
        sd[p + "attn.c_attn.weight"] = n(d, 3 * d, std=0.02)
        sd[p + "attn.c_attn.bias"] = torch.zeros(3 * d)
        sd[p + "attn.c_proj.weight"] = n(d, d, std=0.02)
        sd[p + "attn.c_proj.bias"] = torch.zeros(d)
        sd[p + "ln_2.weight"] = torch.ones(d)
        sd[p + "ln_2.bias"] = torch.zeros(d)
        sd[p + "mlp.c_fc.weight"] = n(d, 4 * d, std=0.02)
        sd[p + "mlp.c_fc.bias"] = torch.zeros(4 * d)
        sd[p + "mlp.c_proj.weight"] = n(4 * d, d, std=0.02)
        sd[p + "mlp.c_proj.bias"] = torch.zeros(d)
    sd["transformer.ln_f.weight"] = torch.ones(d)
    sd["transformer.ln_f.bias"] = torch.zeros(d)
    sd["embed_token.weight"] = n(V, d)
    sd["predict_token.weight"] = u(V, d, bound=1 / math.sqrt(d))
    C = resid_mid_channels
    pe = "image_embedding.patch_embedding."
    sd[pe + "conv1.weight"] = u(C, 3, 3, 3, bound=1 / math.sqrt(27))
    sd[pe + "conv1.bias"] = u(C, bound=1 / math.sqrt(27))
    sd[pe + "gn2.weight"] = torch.ones(C)
    sd[pe + "gn2.bias"] = torch.zeros(C)
    sd[pe + "conv2.weight"] = u(3, C, 3, 3, bound=1 / math.sqrt(9 * C))
    sd[pe + "conv2.bias"] = u(3, bound=1 / math.sqrt(9 * C))
    pdim = cfg.patch_size * cfg.patch_size * 3
    sd["image_embedding.post_embedding_projection.weight"] = u(d, pdim, bound=1 / math.sqrt(pdim))
    sd["image_embedding.post_embedding_projection.bias"] = u(d, bound=1 / math.sqrt(pdim))
    sd["image_embedding.patch_pos_encoding.height_pos_embedding.weight"] = n(cfg.position_vocab_size, d)
    sd["image_embedding.patch_pos_encoding.width_pos_embedding.weight"] = n(cfg.position_vocab_size, d)
    sd["pos_embed_observation.weight"] = n(cfg.context_len, d)
    if cfg.activation_fn == "geglu":    # MLP.gated_layer = nn.Linear(d, 4d) inside the transformer (trajectory_gpt2.py:267-268):
        for i in range(L):              # N(0, 0.02) weight; drawn LAST so the other tensors match the ungated model's
            p = f"transformer.h.{i}."   # (a small non-zero bias so that parity checks see it; the reference zeroes it)
            sd[p + "mlp.gated_layer.weight"] = n(4 * d, d, std=0.02)
            sd[p + "mlp.gated_layer.bias"] = n(4 * d, std=0.02)
    return sd
