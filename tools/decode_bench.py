#!/usr/bin/env python
"""KV-cached decode vs the reference-style full forward per token, at the metric model (768d x 6L x 24H, V = 52305).
    predict_control: 41 timesteps x 24 tokens of half-cheetah history (984 positions) + 6 action tokens
    predict_text:    900-token prompt + 64 generated tokens"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd.policy.gato_policy import GatoPolicy  # noqa: E402

torch.set_num_threads(8)
dev = "cuda"
torch.manual_seed(0)
m = GatoPolicy(dev, 768, 6, 24, 0.0, resid_mid_channels=128, context_len=1024, text_tokenizer=50257)
m.eval()
g = torch.Generator().manual_seed(1)
task = types.SimpleNamespace(action_type=type("Box", (), {}), action_tokens=6, env=None)
ex = {"continuous_obs": torch.randn(41, 17, generator=g).to(dev), "continuous_actions": (torch.rand(41, 6, generator=g) * 2 - 1).to(dev)}
prompt = {"text": torch.randint(0, 50257, (900,), generator=g).tolist()}


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def full_forward_decode(emb, n_tokens, lo, hi):
    """The cost model of the reference's loops (gato_policy.py:446-468, 585-612): one full forward over the whole window
    and a (T, V) logits tensor per generated token.  Bench-only baseline; parity of the cached path is fixture G12."""
    table = m._flat.view("embed_token.weight")
    for _ in range(n_tokens):
        mask = torch.ones(emb.shape[:2], device=emb.device)
        logits, _ = m.forward(token_embeddings=emb, token_masks=mask, token_target_masks=None, tokens=None)
        tok = torch.argmax(logits[0, -1, lo:hi + 1]) + lo
        emb = torch.cat([emb, table[tok].reshape(1, 1, -1)], dim=1)[:, -m.context_len:]


with torch.no_grad():
    emb_c = m.tokenize_input_dicts([ex])[0][:, :-6]
    emb_t = m.tokenize_input_dicts([prompt])[0]
    c0, c1 = m.token_starts["continuous"], m.token_ends["continuous"]
    tc = timeit(lambda: full_forward_decode(emb_c, 6, c0, c1), 5)
    tt = timeit(lambda: full_forward_decode(emb_t, 64, 0, m.token_ends["text"]), 2)
    print(f"full forward per token: predict_control-shaped (6 action tokens after 984 positions) {tc:8.2f} ms/env step   "
          f"predict_text-shaped (64 tokens after 900) {tt:9.1f} ms = {tt / 64:6.2f} ms/token")
    tc = timeit(lambda: m.predict_control(ex, task), 5)
    tt = timeit(lambda: m.predict_text(prompt, max_length=64), 2)
    print(f"KV-cached decode      : predict_control (6 action tokens after 984 positions) {tc:8.2f} ms/env step   "
          f"predict_text (64 tokens after 900) {tt:9.1f} ms = {tt / 64:6.2f} ms/token")
