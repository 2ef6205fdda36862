"""activation_fn='geglu' on the HIP path (gato_policy.py:97-100; MLP.forward trajectory_gpt2.py:273-278:
h = gelu(c_fc x) * gated_layer(x)): the two elementwise kernels against a torch fp32 reference, and the gated
policy against the G8 fixture captured from the imported reference (tests/golden/make_fixture_geglu.py).

Tolerances as in test_policy_gpu.py: hidden states / logits 3e-2 of the tensor scale, loss 2e-3 relative,
per-parameter grad norms 5e-2 relative, 30-step trace 1e-3 relative.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


def make_policy(cfg, seed, train=False):
    from neko_amd.policy.gato_policy import GatoPolicy
    m = GatoPolicy(DEV, cfg.embed_dim, cfg.layers, cfg.heads, 0.0, activation_fn=cfg.activation_fn,
                   resid_mid_channels=128, context_len=cfg.context_len, continuous_tokens=cfg.continuous_tokens,
                   discrete_tokens=cfg.discrete_tokens, text_tokenizer=cfg.text_tokens)
    m.transformer.drop.p = 0.0
    r = m.load_state_dict(O.init_state_dict(cfg, seed), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    m.train(train)
    return m


def to_dev(batch):
    return [{k: (v.to(DEV) if torch.is_tensor(v) and v.dtype != torch.uint8 else v) for k, v in ex.items()}
            for ex in batch]


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


@pytest.mark.parametrize("n", [8, 4096 + 24, 3 * 1024 * 1024 + 5])
def test_geglu_kernels_match_fp32_reference(n):
    from neko_amd import ops
    g = torch.Generator().manual_seed(n)
    pre = (torch.randn(n, generator=g) * 2).to(torch.bfloat16)
    gate = torch.randn(n, generator=g).to(torch.bfloat16)
    dh = torch.randn(n, generator=g).to(torch.bfloat16)
    pre32, gate32, dh32 = pre.float(), gate.float(), dh.float()
    act = torch.nn.functional.gelu(pre32).to(torch.bfloat16)          # what the c_fc GEMM epilogue leaves in h
    h = act.clone().to(DEV)
    ops.geglu_fwd(h, gate.to(DEV))
    ref_h = act.float() * gate32
    # one bf16 rounding of the product
    assert float((h.float().cpu() - ref_h).abs().max()) <= 2 ** -8 * float(ref_h.abs().max()) + 1e-6
    assert relerr(h, ref_h) < 2 ** -8
    d_pre, d_gate = ops.geglu_bwd(dh.to(DEV), pre.to(DEV), gate.to(DEV))
    x = pre32.clone().requires_grad_(True)
    gt = gate32.clone().requires_grad_(True)
    (torch.nn.functional.gelu(x) * gt * dh32).sum().backward()
    err_p = (d_pre.float().cpu() - x.grad).abs()
    err_g = (d_gate.float().cpu() - gt.grad).abs()
    assert bool((err_p <= 2 ** -8 * x.grad.abs() + 2e-6).all()), float(err_p.max())
    assert bool((err_g <= 2 ** -8 * gt.grad.abs() + 2e-6).all()), float(err_g.max())


def test_geglu_rejects_unaligned_or_null():
    from neko_amd import _lib, ops
    a = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.NekoHipError):
        _lib.call("neko_geglu_fwd", ops._p(a[1:]), ops._p(a), 8, ops._stream())


def test_g8_hidden_states(golden):
    f = golden("g8_geglu")
    cfg = O.OracleConfig(**f["cfg"])
    m = make_policy(cfg, f["seed"])
    assert any("mlp.gated_layer.weight" in k for k in m.state_dict())
    h = f["hidden"]
    with torch.no_grad():
        out = m.transformer(inputs_embeds=h["x"].to(DEV), attention_mask=h["mask"].to(DEV))["last_hidden_state"]
    ref = h["last_hidden_state"]
    valid = h["mask"].bool()
    assert relerr(out.cpu()[valid], ref[valid]) < 3e-2
    assert relerr(out, ref) < 5e-2
    # and the gate matters: the ungated model with the same weights is far away
    cfg0 = O.OracleConfig(**{**f["cfg"], "activation_fn": "gelu"})
    sd0 = {k: v for k, v in O.init_state_dict(cfg, f["seed"]).items() if "gated_layer" not in k}
    ref0 = O.transformer_forward(sd0, cfg0, h["x"], h["mask"])
    # (N(0, 0.02) weights: the MLP is a small part of the residual stream, so the bar is relative to the gate's effect)
    e_gate, e_gpu = relerr(ref0[valid], ref[valid]), relerr(out.cpu()[valid], ref[valid])
    assert e_gate > 2e-2 and e_gpu < 0.3 * e_gate, (e_gpu, e_gate)


def test_g8_logits_loss_grads(golden):
    f = golden("g8_geglu")
    cfg = O.OracleConfig(**f["cfg"])
    p = f["policy"]
    m = make_policy(cfg, f["seed"])
    logits, loss = m(to_dev(p["batch"]), compute_loss=True)
    assert tuple(logits.shape) == p["logits_shape"]
    assert relerr(logits[:, ::p["row_stride"], :], p["logits_rows"]) < 3e-2
    assert abs(float(loss) - p["loss"]) < 2e-3 * abs(p["loss"]), (float(loss), p["loss"])
    loss.backward()
    named = dict(m.named_parameters())
    sq = 0.0
    for k, n in p["grad_norms"].items():
        g = named[k].grad
        if n is None:
            assert g is None, k
            continue
        assert g is not None, k
        gn = float(g.float().norm())
        sq += gn * gn
        if k.endswith("c_attn.bias"):
            continue
        assert abs(gn - n) < 5e-2 * n + 1e-6, (k, gn, n)
    assert abs(math.sqrt(sq) - p["total_grad_norm"]) < 2e-2 * p["total_grad_norm"]
    for k, gref in p["small_grads"].items():
        if k.endswith("c_attn.bias"):
            continue
        assert relerr(named[k].grad, gref) < 8e-2, (k, relerr(named[k].grad, gref))


def test_g8_training_trace(golden):
    from neko_amd.training.optim import NekoAdamW
    from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler
    f = golden("g8_geglu")["train"]
    cfg = O.OracleConfig(**f["cfg"])
    m = make_policy(cfg, f["seed"], train=True)
    opt = NekoAdamW(m, lr=f["lr"], betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, f["warmup"], f["total_steps"], base_lr=f["lr"],
                                                   init_lr=f["init_lr"], min_lr=f["min_lr"])
    batches = [to_dev(b) for b in f["batches"]]
    losses, norms = [], []
    for step in range(f["total_steps"]):
        _, loss = m.forward(inputs=batches[step % len(batches)], compute_loss=True, return_logits=False)
        loss.backward()
        norms.append(opt.clip_grad_norm_(1.0))
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().tolist()
    norms = torch.stack(norms).reshape(-1).cpu().tolist()
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, f["trace"]["loss"])]
    reln = [abs(a - b) / abs(b) for a, b in zip(norms, f["trace"]["grad_norm"])]
    assert max(rel) < 1e-3, (max(rel), rel.index(max(rel)))
    assert max(reln[:5]) < 5e-3 and max(reln) < 8e-2, (reln[:5], max(reln))     # see test_g7b_training_trace_1e3


# (KV-cached decode with the gated MLP: fixture G12 case "text_geglu", tests/test_decode_gpu.py)


def test_geglu_side_stream_and_dp_ranges_cover_the_gate():
    """The gate's parameters live in their layer's flat range (what the optimiser and the gradient reducer walk)."""
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=128, context_len=64, activation_fn="geglu")
    m = make_policy(cfg, 7, train=True)
    batch = [{"text": list(range(30))}, {"continuous_obs": torch.randn(3, 4).to(DEV),
                                         "continuous_actions": (torch.rand(3, 2) * 2 - 1).to(DEV)}]
    _, loss = m(batch, compute_loss=True, return_logits=False)
    loss.backward()
    flat = m._flat
    for i in range(cfg.layers):
        a, b = flat.range_of_group(f"layer{i}")
        for nm in ("weight", "bias"):
            o, n, _ = flat.offsets[f"transformer.h.{i}.mlp.gated_layer.{nm}"]
            assert a <= o and o + n <= b
            assert float(flat.grad[o:o + n].abs().sum()) > 0
