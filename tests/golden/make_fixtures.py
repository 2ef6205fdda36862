"""Generate the golden fixtures in tests/golden/*.pt by running the REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box uses the committed
fixtures.  The reference is imported unmodified through tests/golden/ref_shims.py; weights
come from oracle.neko_oracle.init_state_dict(seed) loaded with load_state_dict, so a fixture
holds only the seed, the inputs and the reference's outputs.

    python tests/golden/make_fixtures.py

Fixture list = SURVEY.md section 8(c) G1..G7.
"""
import hashlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
from oracle import neko_oracle as O  # noqa: E402

TEXT_VOCAB = 128


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def build_reference(GatoPolicy, cfg: O.OracleConfig, seed: int):
    m = GatoPolicy("cpu", cfg.embed_dim, cfg.layers, cfg.heads, 0.0,
                   resid_mid_channels=128, context_len=cfg.context_len,
                   continuous_tokens=cfg.continuous_tokens, discrete_tokens=cfg.discrete_tokens)
    m.transformer.drop.p = 0.0          # embd dropout stays 0.1 otherwise
    sd = O.init_state_dict(cfg, seed)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m, sd


def mixed_batch(g, cfg, with_images=True):
    """One batch with the four example kinds of SURVEY G3."""
    r = lambda *s: torch.rand(*s, generator=g)
    ri = lambda lo, hi, s: torch.randint(lo, hi, s, generator=g)
    batch = [
        {"continuous_obs": torch.randn(3, 5, generator=g) * 2.0,
         "continuous_actions": r(3, 2) * 2 - 1},
        {"text": ri(0, cfg.text_tokens, (9,)).tolist(), "images": None, "continuous_obs": None,
         "discrete_obs": None, "continuous_actions": None, "discrete_actions": None},
        {"discrete_obs": ri(0, 7, (2, 3)).to(torch.int32),
         "continuous_actions": r(2, 1) * 2 - 1},
    ]
    if with_images:
        batch.insert(0, {"images": torch.floor(r(2, 3, 32, 32) * 256),
                         "discrete_actions": ri(0, 4, (2, 1)).to(torch.int32)})
        batch.append({"images": torch.floor(r(1, 3, 32, 48) * 256).to(torch.uint8),
                      "text": ri(0, cfg.text_tokens, (5,)).tolist()})
    return batch


def main():
    torch.set_num_threads(4)
    GP = ref_shims.install(TEXT_VOCAB)
    import gato.policy.input_tokenizers as it
    import gato.policy.embeddings as emb
    from gato.training.schedulers import get_linear_warmup_cosine_decay_scheduler

    out = {}

    # ---- G1 continuous tokenizer -------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    edge = torch.tensor([-2, -1, -0.999, -1e-9, 0, 1e-9, 0.5, 0.998, 0.999, 1, 3], dtype=torch.float32)
    rnd = torch.cat([torch.randn(257, generator=g) * 3, torch.rand(257, generator=g) * 2 - 1])
    act_tok = it.ContinuousTokenizer(use_mu_law=False, mu=100, M=256, n_bins=1024, offset=TEXT_VOCAB)
    obs_tok = it.ContinuousTokenizer(use_mu_law=True, mu=100, M=256, n_bins=1024, offset=TEXT_VOCAB)
    out["g1_tokenizer"] = {
        "offset": TEXT_VOCAB, "edge": edge, "rnd": rnd,
        "edge_act": act_tok.encode(edge.clone()), "edge_obs": obs_tok.encode(edge.clone()),
        "rnd_act": act_tok.encode(rnd.clone()), "rnd_obs": obs_tok.encode(rnd.clone()),
        "decode_in": torch.arange(0, 1025, 37) + TEXT_VOCAB,
        "decode_out": act_tok.decode((torch.arange(0, 1025, 37) + TEXT_VOCAB).clone()),
    }

    # ---- G2 patch position indices -----------------------------------------------------
    ppe = emb.PatchPosEncoding(position_vocab_size=128, embed_dim=8).eval()
    g2 = {}
    for n in list(range(1, 17)) + [32]:
        lin = torch.linspace(0, 1, n + 1)
        iv = (torch.stack([lin[:-1], lin[1:]]).T * 128).to(torch.int32)
        # run the reference module to obtain the eval indices it actually uses: recover them by
        # matching rows of the (unique, random) embedding table
        x = torch.zeros(1, n, 1, 8)
        pe = ppe(x)  # n x 1 x 8 = h_emb + w_emb(index for n_width=1)
        w_idx0 = O.patch_pos_indices_eval(1, 128)  # single width patch
        h_rows = pe[:, 0, :] - ppe.width_pos_embedding.weight[w_idx0.long()][0]
        idx = torch.cdist(h_rows.detach(), ppe.height_pos_embedding.weight.detach()).argmin(dim=1)
        g2[n] = {"eval_idx": idx.to(torch.int32), "intervals": iv}
    out["g2_patchpos"] = g2

    # ---- tiny policy config ------------------------------------------------------------
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=TEXT_VOCAB, context_len=64)
    seed = 1234
    m, sd = build_reference(GP, cfg, seed)
    m.eval()

    # ---- G3 tokenize_input_dicts -------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    batch = mixed_batch(g, cfg, with_images=True)
    with torch.no_grad():
        e, t, tg, pm = m.tokenize_input_dicts(batch)
    out["g3_pack"] = {"cfg": cfg.__dict__, "seed": seed, "sd_digest": sd_digest(sd), "batch": batch,
                      "embeddings": e, "tokens": t, "target_masks": tg, "pad_masks": pm}

    # ---- G4 image embedding ------------------------------------------------------------
    g = torch.Generator().manual_seed(4)
    imgs = torch.floor(torch.rand(2, 3, 32, 48, generator=g) * 256)
    with torch.no_grad():
        ie = m.image_embedding(imgs)
    out["g4_image"] = {"cfg": cfg.__dict__, "seed": seed, "images": imgs, "out": ie}

    # ---- G5 transformer hidden states, left-padded -------------------------------------
    g = torch.Generator().manual_seed(5)
    B, T = 3, 40
    x = torch.randn(B, T, cfg.embed_dim, generator=g)
    mask = torch.ones(B, T)
    mask[1, :7] = 0
    mask[2, :33] = 0
    x = x * mask[..., None]
    with torch.no_grad():
        r = m.transformer(inputs_embeds=x, attention_mask=mask, output_hidden_states=True)
    out["g5_hidden"] = {"cfg": cfg.__dict__, "seed": seed, "x": x, "mask": mask,
                        "hidden_states": [h.clone() for h in r["hidden_states"]],
                        "last_hidden_state": r["last_hidden_state"].clone()}

    # second config: hd=32 with d=128, H=4, L=3 (BASELINE configs[0] geometry)
    cfg2 = O.OracleConfig(embed_dim=128, layers=3, heads=4, text_tokens=TEXT_VOCAB, context_len=96)
    m2, sd2 = build_reference(GP, cfg2, 77)
    m2.eval()
    g = torch.Generator().manual_seed(55)
    B, T = 2, 80
    x2 = torch.randn(B, T, cfg2.embed_dim, generator=g)
    mask2 = torch.ones(B, T)
    mask2[0, :19] = 0
    x2 = x2 * mask2[..., None]
    with torch.no_grad():
        r2 = m2.transformer(inputs_embeds=x2, attention_mask=mask2, output_hidden_states=True)
    out["g5b_hidden"] = {"cfg": cfg2.__dict__, "seed": 77, "x": x2, "mask": mask2,
                         "hidden_states": [h.clone() for h in r2["hidden_states"]],
                         "last_hidden_state": r2["last_hidden_state"].clone()}

    # ---- G6 policy logits + loss + grad norms ------------------------------------------
    g = torch.Generator().manual_seed(6)
    batch6 = mixed_batch(g, cfg, with_images=True)
    m.zero_grad()
    logits, loss = m(batch6, compute_loss=True)
    loss.backward()
    gn = {k: (None if p.grad is None else p.grad.norm().item()) for k, p in m.named_parameters()}
    total = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in m.parameters() if p.grad is not None)).item()
    small = {k: p.grad.clone() for k, p in m.named_parameters()
             if p.grad is not None and p.numel() <= 4096}
    out["g6_policy"] = {"cfg": cfg.__dict__, "seed": seed, "batch": batch6,
                        "logits_rows": logits[:, ::5, :].detach().clone(), "row_stride": 5,
                        "logits_shape": tuple(logits.shape), "loss": loss.item(),
                        "grad_norms": gn, "total_grad_norm": total, "small_grads": small}
    m.zero_grad()

    # ---- G7 100-step training trace ----------------------------------------------------
    # mirrors trainer.py:176-186 / train.py:127-136 (torch AdamW, the reference's LambdaLR
    # schedule, clip_grad_norm_ 1.0); dropout 0; model.train() so patch positions are drawn at
    # random -- they are replayed from the same seed and stored.
    m7, sd7 = build_reference(GP, cfg, 4321)
    m7.train()
    lr, init_lr, warm, total_steps = 3e-3, 1e-5, 10, 100
    opt = torch.optim.AdamW(m7.parameters(), lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, warm, total_steps, base_lr=lr, init_lr=init_lr,
                                                   min_lr=lr / 10.0)
    g = torch.Generator().manual_seed(7)
    batches = [mixed_batch(g, cfg, with_images=(i % 2 == 0)) for i in range(4)]
    trace = {"loss": [], "grad_norm": [], "lr": [], "patch_positions": []}
    for step in range(total_steps):
        b = batches[step % len(batches)]
        # replay the PatchPosEncoding draws (embeddings.py:92-94) for this forward
        torch.manual_seed(1000 + step)
        pos = []
        for ex in b:
            if ex.get("images") is not None:
                nh, nw = ex["images"].shape[2] // 16, ex["images"].shape[3] // 16
                hi = O.patch_pos_intervals(nh, 128)
                wi = O.patch_pos_intervals(nw, 128)
                hp = torch.tensor([torch.randint(low=int(a), high=int(bb), size=()) for a, bb in hi])
                wp = torch.tensor([torch.randint(low=int(a), high=int(bb), size=()) for a, bb in wi])
                pos.append((hp, wp))
            else:
                pos.append(None)
        torch.manual_seed(1000 + step)
        trace["lr"].append(sch.get_last_lr()[0])
        logits, loss = m7.forward(inputs=b, compute_loss=True)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(m7.parameters(), 1.0)
        opt.step()
        sch.step()
        opt.zero_grad()
        trace["loss"].append(loss.item())
        trace["grad_norm"].append(float(gnorm))
        trace["patch_positions"].append(pos)
    final_sd = {k: v.detach().clone() for k, v in m7.state_dict().items()
                if v.numel() <= 4096 and not k.endswith(".attn.bias")}
    out["g7_trace"] = {"cfg": cfg.__dict__, "seed": 4321, "batches": batches, "lr": lr, "init_lr": init_lr,
                       "warmup": warm, "total_steps": total_steps, "min_lr": lr / 10.0,
                       "trace": trace, "final_small_params": final_sd}

    # ---- G7b: 100 steps in the regime of the metric's recipe (arguments.py defaults: lr 1e-4, warm-up from
    # 1e-7, betas .9/.95, wd .1, clip 1.0) on the d=128 / 3 layers / hd=32 geometry, text + control batches.
    # This is the trace the "loss within 1e-3 rel over 100 steps" gate is held on; G7 above (lr 3e-3 on a
    # 64-d model, loss 7.8 -> 0.2 in 100 steps) is chaotic and only checked for tracking.
    cfg3 = O.OracleConfig(embed_dim=128, layers=3, heads=4, text_tokens=TEXT_VOCAB, context_len=128)
    m8, _ = build_reference(GP, cfg3, 999)
    m8.train()
    lr8, init8, warm8, total8 = 1e-4, 1e-7, 20, 100
    opt8 = torch.optim.AdamW(m8.parameters(), lr=lr8, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch8 = get_linear_warmup_cosine_decay_scheduler(opt8, warm8, total8, base_lr=lr8, init_lr=init8, min_lr=lr8 / 10.0)
    g = torch.Generator().manual_seed(8)
    batches8 = []
    for i in range(5):
        b = []
        for j in range(4):
            if (i + j) % 2 == 0:
                b.append({"text": torch.randint(0, TEXT_VOCAB, (100 + 3 * j,), generator=g).tolist()})
            else:
                b.append({"continuous_obs": torch.randn(5, 11, generator=g), "continuous_actions": torch.rand(5, 3, generator=g) * 2 - 1})
        batches8.append(b)
    tr8 = {"loss": [], "grad_norm": [], "lr": []}
    for step in range(total8):
        tr8["lr"].append(sch8.get_last_lr()[0])
        _, loss = m8.forward(inputs=batches8[step % len(batches8)], compute_loss=True)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(m8.parameters(), 1.0)
        opt8.step(); sch8.step(); opt8.zero_grad()
        tr8["loss"].append(loss.item()); tr8["grad_norm"].append(float(gnorm))
    out["g7b_trace"] = {"cfg": cfg3.__dict__, "seed": 999, "batches": batches8, "lr": lr8, "init_lr": init8,
                        "warmup": warm8, "total_steps": total8, "min_lr": lr8 / 10.0, "trace": tr8}
    print("g7b loss head:", tr8["loss"][:3], "tail:", tr8["loss"][-3:])

    for name, obj in out.items():
        path = os.path.join(HERE, name + ".pt")
        torch.save(obj, path)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")
    print("loss trace head:", trace["loss"][:5], "tail:", trace["loss"][-3:])


if __name__ == "__main__":
    main()
