"""Generate tests/golden/g8_geglu.pt by running the REFERENCE with activation_fn='geglu'.

Same rules as make_fixtures.py (build container only, reference imported unmodified through ref_shims.py,
weights from oracle.neko_oracle.init_state_dict(cfg, seed)); kept separate so that the G1..G7 fixtures are
not regenerated when this one is.

    python tests/golden/make_fixture_geglu.py

G8 = the gated MLP (gato_policy.py:97-100, MLP.forward trajectory_gpt2.py:273-278):
  * transformer hidden states on a left-padded batch (as G5),
  * policy logits rows + loss + per-parameter grad norms + small gradients on a mixed batch (as G6),
  * a 30-step training trace (as G7b, shorter).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
from make_fixtures import TEXT_VOCAB, mixed_batch  # noqa: E402
from oracle import neko_oracle as O  # noqa: E402


def build(GP, cfg, seed):
    m = GP("cpu", cfg.embed_dim, cfg.layers, cfg.heads, 0.0, activation_fn=cfg.activation_fn,
           resid_mid_channels=128, context_len=cfg.context_len,
           continuous_tokens=cfg.continuous_tokens, discrete_tokens=cfg.discrete_tokens)
    m.transformer.drop.p = 0.0
    sd = O.init_state_dict(cfg, seed)
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return m, sd


def main():
    torch.set_num_threads(4)
    GP = ref_shims.install(TEXT_VOCAB)
    from gato.training.schedulers import get_linear_warmup_cosine_decay_scheduler

    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=TEXT_VOCAB, context_len=64,
                         activation_fn="geglu")
    seed = 2468
    m, sd = build(GP, cfg, seed)
    assert any("gated_layer" in k for k in sd)
    m.eval()

    g = torch.Generator().manual_seed(85)
    B, T = 3, 40
    x = torch.randn(B, T, cfg.embed_dim, generator=g)
    mask = torch.ones(B, T)
    mask[1, :9] = 0
    mask[2, :31] = 0
    x = x * mask[..., None]
    with torch.no_grad():
        r = m.transformer(inputs_embeds=x, attention_mask=mask, output_hidden_states=True)
    hidden = {"x": x, "mask": mask, "hidden_states": [h.clone() for h in r["hidden_states"]],
              "last_hidden_state": r["last_hidden_state"].clone()}

    g = torch.Generator().manual_seed(86)
    batch = mixed_batch(g, cfg, with_images=True)
    m.zero_grad()
    logits, loss = m(batch, compute_loss=True)
    loss.backward()
    gn = {k: (None if p.grad is None else p.grad.norm().item()) for k, p in m.named_parameters()}
    total = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in m.parameters() if p.grad is not None)).item()
    small = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None and p.numel() <= 4096}
    policy = {"batch": batch, "logits_rows": logits[:, ::5, :].detach().clone(), "row_stride": 5,
              "logits_shape": tuple(logits.shape), "loss": loss.item(), "grad_norms": gn,
              "total_grad_norm": total, "small_grads": small}
    m.zero_grad()

    # short training trace in the reference recipe (arguments.py defaults, as G7b), text + control batches
    cfg3 = O.OracleConfig(embed_dim=128, layers=2, heads=4, text_tokens=TEXT_VOCAB, context_len=128,
                          activation_fn="geglu")
    m8, _ = build(GP, cfg3, 1357)
    m8.train()
    lr8, init8, warm8, total8 = 1e-4, 1e-7, 10, 30
    opt8 = torch.optim.AdamW(m8.parameters(), lr=lr8, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch8 = get_linear_warmup_cosine_decay_scheduler(opt8, warm8, total8, base_lr=lr8, init_lr=init8, min_lr=lr8 / 10.0)
    g = torch.Generator().manual_seed(88)
    batches8 = []
    for i in range(3):
        b = []
        for j in range(4):
            if (i + j) % 2 == 0:
                b.append({"text": torch.randint(0, TEXT_VOCAB, (90 + 5 * j,), generator=g).tolist()})
            else:
                b.append({"continuous_obs": torch.randn(5, 11, generator=g),
                          "continuous_actions": torch.rand(5, 3, generator=g) * 2 - 1})
        batches8.append(b)
    tr8 = {"loss": [], "grad_norm": [], "lr": []}
    for step in range(total8):
        tr8["lr"].append(sch8.get_last_lr()[0])
        _, loss = m8.forward(inputs=batches8[step % len(batches8)], compute_loss=True)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(m8.parameters(), 1.0)
        opt8.step(); sch8.step(); opt8.zero_grad()
        tr8["loss"].append(loss.item()); tr8["grad_norm"].append(float(gnorm))
    trace = {"cfg": cfg3.__dict__, "seed": 1357, "batches": batches8, "lr": lr8, "init_lr": init8, "warmup": warm8,
             "total_steps": total8, "min_lr": lr8 / 10.0, "trace": tr8}

    obj = {"cfg": cfg.__dict__, "seed": seed, "hidden": hidden, "policy": policy, "train": trace}
    path = os.path.join(HERE, "g8_geglu.pt")
    torch.save(obj, path)
    print(f"g8_geglu: {os.path.getsize(path) / 1024:.1f} KiB; loss {policy['loss']:.5f}, total grad norm {total:.5f}; "
          f"trace {tr8['loss'][:2]} .. {tr8['loss'][-2:]}")


if __name__ == "__main__":
    main()
