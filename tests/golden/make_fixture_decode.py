"""Generate tests/golden/g12_decode.pt by running the REFERENCE's own inference helpers.

Same rules as make_fixtures.py (build container only; the reference is imported unmodified through ref_shims.py;
weights from oracle.neko_oracle.init_state_dict(cfg, seed)).

    python tests/golden/make_fixture_decode.py

G12 = `GatoPolicy.predict_text` (gato/policy/gato_policy.py:434-470), `predict_response` (:477-544) and
`predict_control` (:556-614, gymnasium Box and Discrete action spaces), all with `deterministic=True`:
the chosen tokens (global vocabulary), the logits rows the reference returns, the decoded action.  One predict_text
case slides the window (prompt + generated tokens > context_len), one uses the gated MLP.

Greedy decoding compares arg-maxes, and random weights give nearly tied logits now and then, so for every case the input
seed is searched: TRIES seeds are run through the reference and the one whose SMALLEST top-2 logit gap over all steps
(relative to the row's max |logit|) is largest is recorded, together with that gap (`min_gap`, 2.5-6 %).  The GPU test
holds the HIP logits to min(2e-2, min_gap / 2) of the row scale, which forces the same arg-max at every step.
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
from oracle import neko_oracle as O  # noqa: E402

TEXT_VOCAB = 48
TRIES = 300


def build(GP, cfg, seed):
    m = GP("cpu", cfg.embed_dim, cfg.layers, cfg.heads, 0.0, activation_fn=cfg.activation_fn, resid_mid_channels=128,
           context_len=cfg.context_len, continuous_tokens=cfg.continuous_tokens, discrete_tokens=cfg.discrete_tokens)
    m.transformer.drop.p = 0.0
    sd = O.init_state_dict(cfg, seed)
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    m.eval()
    return m


def min_gap(logits: torch.Tensor) -> float:
    """logits (n, C): smallest best-vs-second gap over the rows, relative to max|row|."""
    top = torch.topk(logits, 2, dim=-1).values
    return float(((top[:, 0] - top[:, 1]) / logits.abs().amax(dim=-1)).min())


def cfg_dict(cfg):
    return dict(embed_dim=cfg.embed_dim, layers=cfg.layers, heads=cfg.heads, text_tokens=cfg.text_tokens,
                context_len=cfg.context_len, activation_fn=cfg.activation_fn)


def main():
    torch.set_num_threads(4)
    GP = ref_shims.install(TEXT_VOCAB)
    import gymnasium as gym       # the stub ref_shims installed: predict_control compares task.action_type with these classes
    out = {"cases": []}

    def search(fn, what, tries=TRIES):
        best = None
        for s in range(tries):
            rec = fn(s)
            rec["input_seed"] = s
            rec["min_gap"] = min_gap(rec["logits"])
            if best is None or rec["min_gap"] > best["min_gap"]:
                best = rec
        out["cases"].append(best)
        print(f"{what}: input seed {best['input_seed']}, smallest top-2 gap {best['min_gap']:.4f} of the row scale")

    # ---- predict_text -------------------------------------------------------------------------------------------
    def text_case(name, cfg, wseed, prompt_len, max_length):
        m = build(GP, cfg, wseed)

        def run(s):
            g = torch.Generator().manual_seed(1000 + s)
            prompt = torch.randint(0, TEXT_VOCAB, (prompt_len,), generator=g).tolist()
            with torch.no_grad():
                logits, toks = m.predict_text({"text": prompt}, max_length=max_length, deterministic=True)
            return {"kind": "text", "name": name, "cfg": cfg_dict(cfg), "weight_seed": wseed, "prompt": prompt,
                    "max_length": max_length, "logits": logits.clone(), "tokens": [int(t) for t in toks]}
        search(run, name)

    small = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=TEXT_VOCAB, context_len=40)
    text_case("text_short", small, 5, 12, 10)
    text_case("text_sliding_window", small, 5, 36, 9)            # 36 ids + SEP = 37 positions, + 9 > context_len 40
    wide = O.OracleConfig(embed_dim=128, layers=3, heads=4, text_tokens=TEXT_VOCAB, context_len=96)
    text_case("text_128d", wide, 6, 50, 16)
    geglu = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=TEXT_VOCAB, context_len=64, activation_fn="geglu")
    text_case("text_geglu", geglu, 7, 20, 10)

    # ---- predict_response (image + prompt tokens -> text) ----------------------------------------------------------
    cfg_r = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=TEXT_VOCAB, context_len=64)
    m_r = build(GP, cfg_r, 8)

    def resp(s):
        g = torch.Generator().manual_seed(2000 + s)
        image = torch.floor(torch.rand(1, 3, 32, 48, generator=g) * 256)          # 6 patches
        prompt = torch.randint(0, TEXT_VOCAB, (5,), generator=g).tolist()
        with torch.no_grad():
            logits, text = m_r.predict_response(image, prompt_tokens=list(prompt), max_length=8, deterministic=True)
        return {"kind": "response", "name": "response_image_prompt", "cfg": cfg_dict(cfg_r), "weight_seed": 8,
                "image": image, "prompt": prompt, "max_length": 8, "logits": logits.clone(), "text": text}
    search(resp, "predict_response")

    # ---- predict_control ---------------------------------------------------------------------------------------------
    cfg_c = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=TEXT_VOCAB, context_len=48)
    m_c = build(GP, cfg_c, 9)
    # the logits are not returned by predict_control: record them through a forward hook on the reference's own forward
    seen = []
    orig_forward = m_c.forward

    def spy(*a, **k):
        r = orig_forward(*a, **k)
        seen.append(r[0][0, -1].detach().clone())
        return r
    m_c.forward = spy

    def control(kind):
        def run(s):
            g = torch.Generator().manual_seed(3000 + s)
            n_ts = 4
            if kind == "Box":
                n_act = 3
                task = types.SimpleNamespace(action_type=gym.spaces.Box, action_tokens=n_act, env=None)
                ex = {"continuous_obs": torch.randn(n_ts, 5, generator=g),
                      "continuous_actions": torch.rand(n_ts, n_act, generator=g) * 2 - 1}
                lo, hi = TEXT_VOCAB, TEXT_VOCAB + 1023
            else:
                n_act, n = 1, 5
                task = types.SimpleNamespace(action_type=gym.spaces.Discrete, action_tokens=1,
                                             env=types.SimpleNamespace(action_space=types.SimpleNamespace(n=n)))
                ex = {"images": torch.floor(torch.rand(n_ts, 3, 32, 32, generator=g) * 256),
                      "discrete_actions": torch.randint(0, n, (n_ts, 1), generator=g).to(torch.int32)}
                lo, hi = TEXT_VOCAB + 1024, TEXT_VOCAB + 1024 + n - 1
            seen.clear()
            with torch.no_grad():
                action = m_c.predict_control(ex, task, deterministic=True)
            rows = torch.stack(seen)[:, lo:hi + 1]
            return {"kind": "control", "name": f"control_{kind}", "space": kind, "cfg": cfg_dict(cfg_c), "weight_seed": 9,
                    "example": ex, "action_tokens": n_act, "n_actions": 5, "logits": rows.clone(),
                    "action": action.clone() if torch.is_tensor(action) else action}
        return run
    search(control("Box"), "predict_control Box")
    search(control("Discrete"), "predict_control Discrete")

    out["tries"] = TRIES
    out["torch_version"] = torch.__version__
    path = os.path.join(HERE, "g12_decode.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
