"""Pin the CPU oracle (oracle/neko_oracle.py) to golden vectors produced by the imported
reference (tests/golden/make_fixtures.py).  CPU only."""
import math

import pytest
import torch

from oracle import neko_oracle as O


def _cfg(d):
    return O.OracleConfig(**d)


def test_g1_continuous_tokenizer(golden):
    f = golden("g1_tokenizer")
    off = f["offset"]
    for name in ("edge", "rnd"):
        x = f[name]
        assert torch.equal(O.tokenize_continuous(x, False, 100, 256, 1024, off), f[name + "_act"])
        assert torch.equal(O.tokenize_continuous(x, True, 100, 256, 1024, off), f[name + "_obs"])
    # SURVEY G1 known answers
    assert (f["edge_act"] - off).tolist() == [0, 0, 0, 512, 512, 512, 768, 1022, 1023, 1024, 1024]
    assert (f["edge_obs"] - off).tolist() == [244, 279, 279, 512, 512, 512, 710, 744, 744, 744, 799]
    assert torch.equal(O.detokenize_continuous(f["decode_in"].clone(), 1024, off), f["decode_out"])


def test_g2_patch_positions(golden):
    f = golden("g2_patchpos")
    for n, rec in f.items():
        assert torch.equal(O.patch_pos_indices_eval(n, 128), rec["eval_idx"]), n
        assert torch.equal(O.patch_pos_intervals(n, 128), rec["intervals"]), n
    assert O.patch_pos_indices_eval(5).tolist() == [12, 38, 63, 88, 114]
    assert O.patch_pos_indices_eval(6).tolist() == [10, 31, 52, 74, 95, 116]


def test_g3_pack(golden):
    f = golden("g3_pack")
    cfg = _cfg(f["cfg"])
    sd = O.init_state_dict(cfg, f["seed"])
    e, t, tg, pm = O.tokenize_input_dicts(sd, cfg, f["batch"])
    assert torch.equal(t, f["tokens"])
    assert torch.equal(tg, f["target_masks"])
    assert torch.equal(pm, f["pad_masks"])
    torch.testing.assert_close(e, f["embeddings"], rtol=1e-5, atol=1e-5)


def test_g4_image_embedding(golden):
    f = golden("g4_image")
    cfg = _cfg(f["cfg"])
    sd = O.init_state_dict(cfg, f["seed"])
    out = O.image_embedding(sd, cfg, f["images"])
    torch.testing.assert_close(out, f["out"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", ["g5_hidden", "g5b_hidden"])
def test_g5_transformer_hidden_states(golden, name):
    f = golden(name)
    cfg = _cfg(f["cfg"])
    sd = O.init_state_dict(cfg, f["seed"])
    out, hs = O.transformer_forward(sd, cfg, f["x"], f["mask"], return_all=True)
    # reference hidden_states = inputs of every block, then ln_f output (trajectory_gpt2.py:733,782)
    ref_hs = f["hidden_states"]
    assert len(ref_hs) == cfg.layers + 1
    for i in range(cfg.layers):
        torch.testing.assert_close(hs[i], ref_hs[i], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out, ref_hs[-1], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(out, f["last_hidden_state"], rtol=1e-5, atol=2e-5)


def test_g6_policy_logits_loss_grads(golden):
    f = golden("g6_policy")
    cfg = _cfg(f["cfg"])
    sd = O.init_state_dict(cfg, f["seed"])
    loss, logits, grads = O.loss_and_grads(sd, cfg, f["batch"])
    assert tuple(logits.shape) == f["logits_shape"]
    torch.testing.assert_close(logits[:, ::f["row_stride"], :], f["logits_rows"], rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - f["loss"]) < 1e-5 * max(1.0, abs(f["loss"]))
    for k, n in f["grad_norms"].items():
        if n is None:
            assert grads[k] is None, k
        else:
            assert grads[k] is not None, k
            assert abs(float(grads[k].norm()) - n) <= 1e-4 * max(n, 1e-6) + 1e-7, k
    for k, gref in f["small_grads"].items():
        torch.testing.assert_close(grads[k], gref, rtol=1e-4, atol=1e-5)
    tot = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values() if g is not None))
    assert abs(tot - f["total_grad_norm"]) < 1e-4 * f["total_grad_norm"]
    assert grads["transformer.wte.weight"] is None   # never receives a grad


def test_g7b_training_trace_reference_recipe(golden):
    f = golden("g7b_trace")
    cfg = _cfg(f["cfg"])
    sd = O.init_state_dict(cfg, f["seed"])
    st = O.AdamWState(lr=f["lr"])
    tr = f["trace"]
    for step in range(f["total_steps"]):
        lr = f["lr"] * O.lr_ratio(step, f["warmup"], f["total_steps"], f["lr"], f["init_lr"], f["min_lr"])
        assert abs(lr - tr["lr"][step]) < 1e-12 + 1e-9 * lr
        loss, gn = O.train_step(sd, cfg, st, f["batches"][step % len(f["batches"])], lr, 1.0)
        assert abs(loss - tr["loss"][step]) < 1e-4 * abs(tr["loss"][step]), (step, loss, tr["loss"][step])
        assert abs(gn - tr["grad_norm"][step]) < 1e-3 * tr["grad_norm"][step], (step, gn)


def test_g7_training_trace(golden):
    f = golden("g7_trace")
    cfg = _cfg(f["cfg"])
    sd = O.init_state_dict(cfg, f["seed"])
    st = O.AdamWState(lr=f["lr"])
    tr = f["trace"]
    for step in range(f["total_steps"]):
        ratio = O.lr_ratio(step, f["warmup"], f["total_steps"], f["lr"], f["init_lr"], f["min_lr"])
        lr = f["lr"] * ratio
        assert abs(lr - tr["lr"][step]) < 1e-12 + 1e-9 * lr
        b = f["batches"][step % len(f["batches"])]
        loss, gn = O.train_step(sd, cfg, st, b, lr, 1.0, patch_positions=tr["patch_positions"][step])
        assert abs(loss - tr["loss"][step]) < 1e-3 * max(1.0, abs(tr["loss"][step])), (step, loss, tr["loss"][step])
        assert abs(gn - tr["grad_norm"][step]) < 2e-3 * max(1.0, tr["grad_norm"][step]), (step, gn)
    for k, v in f["final_small_params"].items():
        if k.endswith("masked_bias"):
            continue
        if k.endswith("c_attn.bias"):
            # the K third of the qkv bias has a mathematically zero gradient (softmax is
            # invariant to a per-query shift): Adam normalises pure rounding noise there.
            d = cfg.embed_dim
            torch.testing.assert_close(sd[k][:d], v[:d], rtol=2e-3, atol=2e-5)
            torch.testing.assert_close(sd[k][2 * d:], v[2 * d:], rtol=2e-3, atol=2e-5)
            continue
        torch.testing.assert_close(sd[k], v, rtol=2e-3, atol=2e-5)


def test_g8_geglu_oracle_matches_reference(golden):
    """activation_fn='geglu' (gato_policy.py:97-100, MLP.forward trajectory_gpt2.py:273-278): hidden states, policy
    logits / loss / gradients and a 30-step training trace of the gated model, captured from the imported reference."""
    f = golden("g8_geglu")
    cfg = _cfg(f["cfg"])
    assert cfg.activation_fn == "geglu"
    sd = O.init_state_dict(cfg, f["seed"])
    h = f["hidden"]
    out, hs = O.transformer_forward(sd, cfg, h["x"], h["mask"], return_all=True)
    for i in range(cfg.layers):
        torch.testing.assert_close(hs[i], h["hidden_states"][i], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out, h["last_hidden_state"], rtol=1e-5, atol=2e-5)
    p = f["policy"]
    loss, logits, grads = O.loss_and_grads(sd, cfg, p["batch"])
    torch.testing.assert_close(logits[:, ::p["row_stride"], :], p["logits_rows"], rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - p["loss"]) < 1e-5 * max(1.0, abs(p["loss"]))
    for k, n in p["grad_norms"].items():
        if n is None:
            assert grads[k] is None, k
        else:
            assert abs(float(grads[k].norm()) - n) <= 1e-4 * max(n, 1e-6) + 1e-7, k
    assert any("gated_layer" in k for k in p["grad_norms"])
    for k, gref in p["small_grads"].items():
        torch.testing.assert_close(grads[k], gref, rtol=1e-4, atol=1e-5)
    t = f["train"]
    cfg3 = _cfg(t["cfg"])
    sd3 = O.init_state_dict(cfg3, t["seed"])
    st = O.AdamWState(lr=t["lr"])
    for step in range(t["total_steps"]):
        lr = t["lr"] * O.lr_ratio(step, t["warmup"], t["total_steps"], t["lr"], t["init_lr"], t["min_lr"])
        loss, gn = O.train_step(sd3, cfg3, st, t["batches"][step % len(t["batches"])], lr, 1.0)
        assert abs(loss - t["trace"]["loss"][step]) < 1e-4 * abs(t["trace"]["loss"][step]), (step, loss)
        assert abs(gn - t["trace"]["grad_norm"][step]) < 1e-3 * t["trace"]["grad_norm"][step], (step, gn)


@pytest.mark.parametrize("name", ["text_short", "text_sliding_window", "text_128d", "text_geglu"])
def test_g12_oracle_reproduces_reference_decode_logits(golden, name):
    """Fixture G12 holds what the reference's own predict_text returned (gato_policy.py:434-470).  The oracle has no
    decode loop; teacher-forcing the reference's tokens through its forward must give the same logits rows and
    arg-maxes: the window is [prompt ids + SEP | embed_token rows of the chosen tokens], truncated on the left to
    context_len, and the row of interest is always the last one."""
    c = next(x for x in golden("g12_decode")["cases"] if x["name"] == name)
    cfg = O.OracleConfig(**c["cfg"])
    sd = O.init_state_dict(cfg, c["weight_seed"])
    emb, _, _, mask = O.tokenize_input_dicts(sd, cfg, [{"text": list(c["prompt"])}])
    lo, hi = 0, cfg.text_tokens - 1
    for i, tok in enumerate(c["tokens"]):
        hidden = O.transformer_forward(sd, cfg, emb, torch.ones(emb.shape[:2]))
        row = O.lm_head(sd, hidden[:, -1])[0, lo:hi + 1]
        assert torch.allclose(row, c["logits"][i], rtol=1e-4, atol=2e-5), (name, i)
        assert int(torch.argmax(row)) + lo == tok
        emb = torch.cat([emb, sd["embed_token.weight"][tok].reshape(1, 1, -1)], dim=1)[:, -cfg.context_len:]
