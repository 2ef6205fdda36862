"""Host-side mirror of gato/transformers/trajectory_gpt2.py for the hot path.

Same class names, constructor meaning, ``state_dict`` keys / shapes / layouts and call signature
(``GPT2Model(config)(inputs_embeds=(B,T,d), attention_mask=(B,T))['last_hidden_state']``,
trajectory_gpt2.py:535-795) -- but the modules are parameter containers only: the compute is the
HIP stack in neko_amd.engine (LayerNorm / bf16-MFMA GEMM / fused attention kernels).

Not mirrored (dead code in the reference for this path): TF checkpoint loader,
AdapterMLP, cross-attention, head pruning, naive model parallel, KV cache (`present`).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn

from .. import engine
from ..flat import FlatParams


@dataclass
class GPT2Config:
    """The HF GPT2Config fields the reference reads (gato_policy.py:101-114, trajectory_gpt2.py:301,541-543)."""
    vocab_size: int = 1
    n_embd: int = 768
    n_head: int = 12
    n_layer: int = 12
    n_positions: int = 1024
    n_ctx: int = 1024
    n_inner: Optional[int] = None
    activation_function: str = "gelu"
    resid_pdrop: float = 0.1
    attn_pdrop: float = 0.1
    embd_pdrop: float = 0.1           # HF default; the reference never overrides it
    layer_norm_epsilon: float = 1e-5
    initializer_range: float = 0.02
    flash: bool = False
    gate: bool = False


class Conv1D(nn.Module):
    """HF Conv1D container: weight stored (in, out), y = x @ W + b (trajectory_gpt2.py:139-141)."""

    def __init__(self, nf: int, nx: int):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        nn.init.normal_(self.weight, std=0.02)


class Attention(nn.Module):
    def __init__(self, nx: int, n_ctx: int, config: GPT2Config, scale: bool = False):
        super().__init__()
        assert nx % config.n_head == 0                                      # trajectory_gpt2.py:126
        # causal-mask buffers are kept only for state_dict compatibility (:127-130); the kernel
        # derives the causal mask from indices.
        self.register_buffer("bias", torch.tril(torch.ones((n_ctx, n_ctx), dtype=torch.uint8)).view(1, 1, n_ctx, n_ctx))
        self.register_buffer("masked_bias", torch.tensor(-1e4))
        self.n_head = config.n_head
        self.split_size = nx
        self.scale = scale
        self.c_attn = Conv1D(3 * nx, nx)
        self.c_proj = Conv1D(nx, nx)
        self.attn_dropout = nn.Dropout(config.attn_pdrop)
        self.resid_dropout = nn.Dropout(config.resid_pdrop)


class MLP(nn.Module):
    def __init__(self, n_state: int, config: GPT2Config):
        super().__init__()
        nx = config.n_embd
        self.c_fc = Conv1D(n_state, nx)
        self.c_proj = Conv1D(nx, n_state)
        # activation_fn='geglu' (gato_policy.py:97-100): h = act(c_fc x) * gated_layer(x), trajectory_gpt2.py:267-276
        self.gated_layer = nn.Linear(nx, n_state) if config.gate else None
        self.dropout = nn.Dropout(config.resid_pdrop)


class Block(nn.Module):
    def __init__(self, n_ctx: int, config: GPT2Config, scale: bool = False):
        super().__init__()
        hidden = config.n_embd
        inner = config.n_inner if config.n_inner is not None else 4 * hidden
        assert inner == 4 * hidden, "the HIP MLP path assumes n_inner = 4*n_embd (gato_policy.py:109)"
        self.ln_1 = nn.LayerNorm(hidden, eps=config.layer_norm_epsilon)
        self.attn = Attention(hidden, n_ctx, config, scale)
        self.ln_2 = nn.LayerNorm(hidden, eps=config.layer_norm_epsilon)
        self.mlp = MLP(inner, config)


class ModelOutput(dict):
    """dict with attribute access, like HF's BaseModelOutputWithPastAndCrossAttentions."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class _TransformerFn(torch.autograd.Function):
    """inputs_embeds -> ln_f(h_L) through the HIP stack; parameter gradients go straight to the flat buffer."""

    @staticmethod
    def forward(ctx, model: "GPT2Model", x: torch.Tensor, mask: torch.Tensor, *params):
        need = any(ctx.needs_input_grad)   # grad mode is off inside Function.forward
        model._flat.ensure_shadow()
        _, hf32, sctx = engine.stack_forward(model._stack_params(), x.detach().to(torch.float32), mask, save=need,
                                             want_f32=True, want_bf16=False, drops=model.make_drops())
        ctx.model, ctx.sctx = model, sctx
        return hf32.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        names = model._param_names()
        model._flat.prepare_backward(names)
        M = g.shape[0] * g.shape[1]
        gx = engine.stack_backward(model._stack_params(), ctx.sctx, g.reshape(M, -1).contiguous().to(torch.float32),
                                   on_layer_done=model._on_layer_done)
        model._flat.attach_grads(names)
        return (None, gx.view(g.shape), None) + (None,) * (len(ctx.needs_input_grad) - 3)


class GPT2Model(nn.Module):
    def __init__(self, config: GPT2Config):
        super().__init__()
        self.config = config
        self.wte = nn.Embedding(config.vocab_size, config.n_embd)          # unused, kept for state_dict parity (:538)
        self.drop = nn.Dropout(config.embd_pdrop)                             # :541
        self.h = nn.ModuleList([Block(config.n_ctx, config, scale=True) for _ in range(config.n_layer)])
        self.ln_f = nn.LayerNorm(config.n_embd, eps=config.layer_norm_epsilon)
        self.apply(self._init_weights)                                        # :545 init_weights()
        self._flat: Optional[FlatParams] = None
        self._prefix = ""
        self._sp: Optional[engine.StackParams] = None
        self._on_layer_done = None

    def _init_weights(self, module):
        """trajectory_gpt2.py:375-385."""
        if isinstance(module, (nn.Linear, nn.Embedding, Conv1D)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            if isinstance(module, (nn.Linear, Conv1D)) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    # ---- flat-parameter plumbing ---------------------------------------------------------------
    def layer_param_names(self, i: int) -> List[str]:
        p = f"h.{i}."
        gate = ("mlp.gated_layer.weight", "mlp.gated_layer.bias") if self.config.gate else ()
        return [p + s for s in ("ln_1.weight", "ln_1.bias", "attn.c_attn.weight", "attn.c_attn.bias",
                                "attn.c_proj.weight", "attn.c_proj.bias", "ln_2.weight", "ln_2.bias",
                                "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias") + gate]

    def param_groups_for_flat(self, prefix: str = "") -> "OrderedDict[str, list]":
        named = dict(self.named_parameters())
        groups: "OrderedDict[str, list]" = OrderedDict()
        for i in range(self.config.n_layer):
            groups[f"layer{i}"] = [(prefix + n, named[n]) for n in self.layer_param_names(i)]
        groups["lnf"] = [(prefix + n, named[n]) for n in ("ln_f.weight", "ln_f.bias")]
        return groups

    def attach_flat(self, flat: FlatParams, prefix: str) -> None:
        self._flat, self._prefix, self._sp = flat, prefix, None

    def _ensure_flat(self, device) -> None:
        if self._flat is None:
            groups = self.param_groups_for_flat("")
            groups["never"] = [("wte.weight", self.wte.weight)]
            self.attach_flat(FlatParams(groups, device), "")

    def _param_names(self) -> List[str]:
        names = []
        for i in range(self.config.n_layer):
            names += [self._prefix + n for n in self.layer_param_names(i)]
        return names + [self._prefix + "ln_f.weight", self._prefix + "ln_f.bias"]

    def _stack_params(self) -> engine.StackParams:
        if self._sp is not None:
            return self._sp
        f, pre = self._flat, self._prefix
        layers = []
        for i in range(self.config.n_layer):
            p = f"{pre}h.{i}."
            gate = {}
            if self.config.gate:        # nn.Linear layout (out, in) = (4d, d): a k-contiguous B operand
                gate = dict(w_gate=f.sview(p + "mlp.gated_layer.weight"), b_gate=f.view(p + "mlp.gated_layer.bias"),
                            g_w_gate=f.gview(p + "mlp.gated_layer.weight"), g_b_gate=f.gview(p + "mlp.gated_layer.bias"))
            layers.append(engine.LayerParams(
                **gate,
                ln1_w=f.view(p + "ln_1.weight"), ln1_b=f.view(p + "ln_1.bias"),
                ln2_w=f.view(p + "ln_2.weight"), ln2_b=f.view(p + "ln_2.bias"),
                b_qkv=f.view(p + "attn.c_attn.bias"), b_o=f.view(p + "attn.c_proj.bias"),
                b_fc=f.view(p + "mlp.c_fc.bias"), b_pr=f.view(p + "mlp.c_proj.bias"),
                w_qkv=f.sview(p + "attn.c_attn.weight"), w_o=f.sview(p + "attn.c_proj.weight"),
                w_fc=f.sview(p + "mlp.c_fc.weight"), w_pr=f.sview(p + "mlp.c_proj.weight"),
                g_ln1_w=f.gview(p + "ln_1.weight"), g_ln1_b=f.gview(p + "ln_1.bias"),
                g_ln2_w=f.gview(p + "ln_2.weight"), g_ln2_b=f.gview(p + "ln_2.bias"),
                g_b_qkv=f.gview(p + "attn.c_attn.bias"), g_b_o=f.gview(p + "attn.c_proj.bias"),
                g_b_fc=f.gview(p + "mlp.c_fc.bias"), g_b_pr=f.gview(p + "mlp.c_proj.bias"),
                g_w_qkv=f.gview(p + "attn.c_attn.weight"), g_w_o=f.gview(p + "attn.c_proj.weight"),
                g_w_fc=f.gview(p + "mlp.c_fc.weight"), g_w_pr=f.gview(p + "mlp.c_proj.weight")))
        self._sp = engine.StackParams(
            d=self.config.n_embd, heads=self.config.n_head, eps=self.config.layer_norm_epsilon, layers=layers,
            lnf_w=f.view(pre + "ln_f.weight"), lnf_b=f.view(pre + "ln_f.bias"),
            g_lnf_w=f.gview(pre + "ln_f.weight"), g_lnf_b=f.gview(pre + "ln_f.bias"))
        return self._sp

    def make_drops(self):
        """Dropout sites for one training-mode forward (None in eval mode or when every p is 0).  Keys are derived
        from (torch.initial_seed(), rank, a per-model forward counter, site id): masks differ per step / site / rank
        and are regenerated, not stored.  The rates are the nn.Dropout modules' p, like the reference reads them
        (embd_pdrop stays 0.1 unless transformer.drop.p is changed)."""
        if not self.training:
            return None
        ps = [self.drop.p] + [b.attn.attn_dropout.p for b in self.h] + [b.attn.resid_dropout.p for b in self.h] + \
             [b.mlp.dropout.p for b in self.h]
        if not any(p > 0 for p in ps):
            return None
        from .. import ops
        rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        self._drop_step = getattr(self, "_drop_step", 0) + 1
        base = ops.mix32((torch.initial_seed() & 0xFFFFFFFF) ^ ops.mix32(self._drop_step) ^ ops.mix32(0x51ED270B + rank))
        mk = lambda p, site: (ops.Drop(p, ops.mix32(base + site * 0x9E3779B9)) if p > 0 else None)
        return engine.DropSites(
            embd=mk(self.drop.p, 0),
            attn=[mk(b.attn.attn_dropout.p, 1 + 3 * i) for i, b in enumerate(self.h)],
            resid_attn=[mk(b.attn.resid_dropout.p, 2 + 3 * i) for i, b in enumerate(self.h)],
            resid_mlp=[mk(b.mlp.dropout.p, 3 + 3 * i) for i, b in enumerate(self.h)])

    # ---- reference call signature (trajectory_gpt2.py:611-795) ------------------------------------
    def forward(self, input_ids=None, past_key_values=None, attention_mask=None, token_type_ids=None,
                position_ids=None, head_mask=None, inputs_embeds=None, encoder_hidden_states=None,
                encoder_attention_mask=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, labels=None):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if inputs_embeds is None:
            raise ValueError("the HIP path takes inputs_embeds (the reference never passes input_ids, "
                             "gato_policy.py:169)")
        for name, v in (("past_key_values", past_key_values), ("token_type_ids", token_type_ids),
                        ("head_mask", head_mask), ("encoder_hidden_states", encoder_hidden_states)):
            if v is not None:
                raise NotImplementedError(f"{name} is not supported on the HIP path")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not materialised by the fused kernels")
        if not inputs_embeds.is_cuda:
            raise RuntimeError("neko_amd.GPT2Model runs on the GPU only (no CPU fallback)")
        B, T, _ = inputs_embeds.shape
        if attention_mask is None:
            attention_mask = torch.ones(B, T, dtype=torch.float32, device=inputs_embeds.device)
        self._ensure_flat(inputs_embeds.device)
        params = [self._flat.param_of[n] for n in self._param_names()]
        out = _TransformerFn.apply(self, inputs_embeds, attention_mask.view(B, -1), *params)
        return ModelOutput(last_hidden_state=out, past_key_values=None, hidden_states=None, attentions=None,
                           cross_attentions=None)
