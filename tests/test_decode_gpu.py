"""KV-cached incremental decode against the reference-style full-forward loops:
engine.KVDecoder vs stack_forward on the same rows, and predict_text / predict_control / predict_response with
kv_cache=True vs kv_cache=False (same tokens, same logits to bf16 tolerance), including the sliding window."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


def _policy(ctx=40, d=64, L=2, H=2, vocab=128, seed=5):
    from neko_amd.policy.gato_policy import GatoPolicy
    cfg = O.OracleConfig(embed_dim=d, layers=L, heads=H, text_tokens=vocab, context_len=ctx)
    m = GatoPolicy(DEV, d, L, H, 0.0, resid_mid_channels=128, context_len=ctx, text_tokenizer=vocab)
    m.transformer.drop.p = 0.0
    m.load_state_dict(O.init_state_dict(cfg, seed))
    m.eval()
    return m


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-9))


def test_decoder_matches_full_forward_rows():
    from neko_amd import engine
    m = _policy(ctx=96)
    sp = m.transformer._stack_params()
    m._flat.ensure_shadow()
    g = torch.Generator().manual_seed(0)
    T, d = 70, 64
    x = torch.randn(1, T, d, generator=g).to(DEV)
    hf16, _, _ = engine.stack_forward(sp, x, torch.ones(1, T, device=DEV), save=False)
    dec = engine.KVDecoder(sp, 96, DEV)
    h0 = dec.extend(x[0, :33])                      # prime (not a multiple of any tile)
    assert _rel(h0, hf16[:33]) < 2e-2
    outs = [dec.extend(x[0, t:t + 1]) for t in range(33, 60)]       # one position at a time
    h2 = dec.extend(x[0, 60:70])                    # a block of positions
    got = torch.cat([h0] + outs + [h2], dim=0)
    assert dec.n == T and _rel(got, hf16) < 2e-2
    with pytest.raises(ValueError):
        dec.extend(torch.zeros(40, d, device=DEV))  # beyond the capacity


@pytest.mark.parametrize("prompt_len,max_length", [(12, 10), (36, 9)])      # second case: 37 + 9 > context_len 40
def test_predict_text_cached_equals_full(prompt_len, max_length):
    m = _policy(ctx=40)
    g = torch.Generator().manual_seed(prompt_len)
    batch = {"text": torch.randint(0, 128, (prompt_len,), generator=g).tolist()}
    l_full, t_full = m.predict_text(batch, max_length=max_length, kv_cache=False)
    l_kv, t_kv = m.predict_text(batch, max_length=max_length, kv_cache=True)
    assert [int(t) for t in t_kv] == [int(t) for t in t_full]
    assert l_kv.shape == l_full.shape and _rel(l_kv, l_full) < 2e-2


@pytest.mark.parametrize("kind", ["Box", "Discrete"])
def test_predict_control_cached_equals_full(kind):
    m = _policy(ctx=48)
    g = torch.Generator().manual_seed(3)
    n_act = 3 if kind == "Box" else 1
    action_type = type(kind, (), {})
    env = types.SimpleNamespace(action_space=types.SimpleNamespace(n=5))
    task = types.SimpleNamespace(action_type=action_type, action_tokens=n_act, env=env)
    ex = {"continuous_obs": torch.randn(4, 5, generator=g).to(DEV)}
    if kind == "Box":
        ex["continuous_actions"] = (torch.rand(4, n_act, generator=g) * 2 - 1).to(DEV)
    else:
        ex["discrete_actions"] = torch.randint(0, 5, (4, 1), generator=g).to(torch.int32).to(DEV)
    a_full = m.predict_control(ex, task, kv_cache=False)
    a_kv = m.predict_control(ex, task, kv_cache=True)
    assert torch.equal(torch.as_tensor(a_kv).cpu(), torch.as_tensor(a_full).cpu())


def test_predict_response_cached_equals_full():
    m = _policy(ctx=64)
    g = torch.Generator().manual_seed(9)
    image = torch.floor(torch.rand(1, 3, 32, 48, generator=g) * 256).to(DEV)      # 6 patches
    prompt = torch.randint(0, 128, (5,), generator=g).tolist()
    l_full, s_full = m.predict_response(image, prompt_tokens=prompt, max_length=8, kv_cache=False)
    l_kv, s_kv = m.predict_response(image, prompt_tokens=prompt, max_length=8, kv_cache=True)
    assert s_kv == s_full
    assert l_kv.shape == l_full.shape and _rel(l_kv, l_full) < 2e-2
