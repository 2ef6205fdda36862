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


@pytest.mark.parametrize("H,hd,n", [(2, 32, 0), (2, 32, 37), (3, 64, 300), (2, 128, 129)])
def test_attn_decode_kernel_matches_last_row_of_full_attention(H, hd, n):
    """neko_attn_decode (position from device memory) vs neko_attn_fwd's row n on the same q|k|v rows; it must also
    append the new row's k/v to the cache."""
    from neko_amd import ops
    g = torch.Generator().manual_seed(n + hd)
    d, cap = H * hd, 320
    rows = torch.randn(n + 1, 3 * d, generator=g).to(torch.bfloat16).to(DEV)
    ref, _ = ops.attn_fwd(rows.contiguous(), *ops.mask_bias(torch.ones(1, n + 1, device=DEV)), 1, n + 1, H, hd)
    cache = torch.zeros(cap, 3 * d, dtype=torch.bfloat16, device=DEV)
    cache[:n] = rows[:n]
    pos = torch.tensor([n], dtype=torch.int32, device=DEV)
    out = torch.empty(1, d, dtype=torch.bfloat16, device=DEV)
    ops.attn_decode(cache, rows[n:n + 1].contiguous(), pos, out, H, hd)
    assert _rel(out[0], ref[n]) < 1.5e-2
    assert torch.equal(cache[n, d:], rows[n, d:])          # k and v appended
    assert float(cache[n + 1:].abs().max()) == 0.0 if n + 1 < cap else True


def test_graph_replay_decode_equals_eager_and_full(monkeypatch):
    m = _policy(ctx=64)
    g = torch.Generator().manual_seed(21)
    batch = {"text": torch.randint(0, 128, (20,), generator=g).tolist()}
    l_full, t_full = m.predict_text(batch, max_length=12, kv_cache=False)
    monkeypatch.setenv("NEKO_DECODE_GRAPH", "0")
    l_eager, t_eager = m.predict_text(batch, max_length=12, kv_cache=True)
    monkeypatch.setenv("NEKO_DECODE_GRAPH", "1")
    l_graph, t_graph = m.predict_text(batch, max_length=12, kv_cache=True)
    assert [int(t) for t in t_graph] == [int(t) for t in t_eager] == [int(t) for t in t_full]
    assert _rel(l_graph, l_full) < 2e-2 and _rel(l_graph, l_eager) < 2e-2
    # a second call re-captures on fresh buffers and gives the same answer
    l2, t2 = m.predict_text(batch, max_length=12, kv_cache=True)
    assert [int(t) for t in t2] == [int(t) for t in t_graph] and torch.equal(l2, l_graph)


@pytest.mark.parametrize("M,N,K,ks,act", [(1, 2304, 768, True, 0), (1, 768, 3072, True, 0), (3, 264, 96, True, 1),
                                          (8, 136, 64, True, 1), (1, 52352, 768, False, 0), (5, 1000, 200, False, 0)])
def test_gemv_matches_reference(M, N, K, ks, act):
    """neko_gemv_bf16 (decode products) vs fp32 math on the same bf16 operands, with bias / GELU / residual."""
    from neko_amd import ops
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = (torch.randn((K, N) if ks else (N, K), generator=g) * 0.05).to(torch.bfloat16)
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = x.float() @ (W.float() if ks else W.float().t()) + bias
    if act:
        ref = torch.nn.functional.gelu(ref.to(torch.bfloat16).float())
    ref = ref + resid
    o32 = torch.empty(M, N, device=DEV)
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemv(x.to(DEV), W.to(DEV), M, N, K, b_kstrided=ks, bias=bias.to(DEV), resid=resid.to(DEV), act=act, out_f32=o32, out_bf16=o16)
    assert float((o32.cpu() - ref).abs().max()) < 2e-3 * float(ref.abs().max()) + 2e-3
    assert _rel(o16, ref) < 1e-2
