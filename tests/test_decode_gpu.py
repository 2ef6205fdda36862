"""KV-cached incremental decode, pinned to the REFERENCE: fixture G12
(tests/golden/make_fixture_decode.py) holds what the reference's own predict_text / predict_response / predict_control
returned -- tokens, logits rows, decoded actions, one case sliding the window, one with the gated MLP -- and the
captured-graph and eager cached HIP paths must reproduce them.  Plus kernel-level checks: engine.KVDecoder vs
stack_forward on the same rows, the one-query attention kernel, the weight-streaming GEMV."""
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


def _g12(golden, name):
    for c in golden("g12_decode")["cases"]:
        if c["name"] == name:
            return c
    raise KeyError(name)


def _policy_for(case):
    from neko_amd.policy.gato_policy import GatoPolicy
    cfg = O.OracleConfig(**case["cfg"])
    m = GatoPolicy(DEV, cfg.embed_dim, cfg.layers, cfg.heads, 0.0, activation_fn=cfg.activation_fn, resid_mid_channels=128,
                   context_len=cfg.context_len, text_tokenizer=cfg.text_tokens)
    m.transformer.drop.p = 0.0
    m.load_state_dict(O.init_state_dict(cfg, case["weight_seed"]))
    m.eval()
    return m


def _row_err(got, ref):
    """max over steps of max|got - ref| / max|ref row|: the quantity fixture G12's `min_gap` is stated in."""
    got, ref = got.float().cpu(), ref.float()
    return float(((got - ref).abs().amax(dim=-1) / ref.abs().amax(dim=-1)).max())


def _gate(case):
    # 2e-2 of the row scale, and below half of the reference's own smallest top-2 gap: the arg-max of every
    # step is then forced to be the reference's
    return min(2e-2, 0.5 * case["min_gap"])


@pytest.mark.parametrize("name", ["text_short", "text_sliding_window", "text_128d", "text_geglu"])
def test_predict_text_matches_reference_fixture(golden, monkeypatch, name):
    """gato_policy.py:434-470 run by the reference itself (G12) vs the KV-cached HIP decode: captured-graph path (when the
    window does not slide) and eager cached path -- same tokens, logits inside the gate."""
    c = _g12(golden, name)
    m = _policy_for(c)
    for graph in ("1", "0"):
        monkeypatch.setenv("NEKO_DECODE_GRAPH", graph)
        logits, toks = m.predict_text({"text": list(c["prompt"])}, max_length=c["max_length"], deterministic=True)
        assert tuple(logits.shape) == tuple(c["logits"].shape)
        err = _row_err(logits, c["logits"])
        assert err < _gate(c), (name, graph, err, c["min_gap"])
        assert [int(t) for t in toks] == c["tokens"], (name, graph)


def test_predict_response_matches_reference_fixture(golden):
    """gato_policy.py:477-544 (image embeddings + prompt tokens -> greedy text)."""
    c = _g12(golden, "response_image_prompt")
    m = _policy_for(c)
    logits, text = m.predict_response(c["image"].to(DEV), prompt_tokens=list(c["prompt"]), max_length=c["max_length"])
    assert tuple(logits.shape) == tuple(c["logits"].shape)
    # the image rows come from the bf16 convolution path (embeddings.py under autocast): same gate
    assert _row_err(logits, c["logits"]) < _gate(c)
    assert text == c["text"]


@pytest.mark.parametrize("space", ["Box", "Discrete"])
def test_predict_control_matches_reference_fixture(golden, monkeypatch, space):
    """gato_policy.py:556-614 for a gymnasium Box (3 mu-law-free action tokens, decoded back to floats) and a Discrete
    action space (arg-max restricted to the env's n actions)."""
    c = _g12(golden, f"control_{space}")
    m = _policy_for(c)
    action_type = type(space, (), {})
    env = types.SimpleNamespace(action_space=types.SimpleNamespace(n=c["n_actions"]))
    task = types.SimpleNamespace(action_type=action_type, action_tokens=c["action_tokens"], env=env)
    ex = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in c["example"].items()}
    seen = []
    inner = m._decode_tokens

    def spy(*a, **k):
        r = inner(*a, **k)
        seen.append(r[0])
        return r
    m._decode_tokens = spy
    for graph in ("1", "0"):
        monkeypatch.setenv("NEKO_DECODE_GRAPH", graph)
        seen.clear()
        action = m.predict_control(ex, task, deterministic=True)
        assert _row_err(seen[0], c["logits"]) < _gate(c)
        assert torch.equal(torch.as_tensor(action).cpu().reshape(-1), torch.as_tensor(c["action"]).reshape(-1)), (space, graph)


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


def test_graph_replay_is_stable_across_calls(monkeypatch):
    m = _policy(ctx=64)
    g = torch.Generator().manual_seed(21)
    batch = {"text": torch.randint(0, 128, (20,), generator=g).tolist()}
    monkeypatch.setenv("NEKO_DECODE_GRAPH", "0")
    l_eager, t_eager = m.predict_text(batch, max_length=12)
    monkeypatch.setenv("NEKO_DECODE_GRAPH", "1")
    l_graph, t_graph = m.predict_text(batch, max_length=12)
    assert [int(t) for t in t_graph] == [int(t) for t in t_eager]
    assert _rel(l_graph, l_eager) < 2e-2
    # a second call replays the cached graph on reset buffers and gives the same answer, also after a no-op .to()
    m.to(DEV)
    l2, t2 = m.predict_text(batch, max_length=12)
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
