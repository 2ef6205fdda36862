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


for kv in (False, True):
    tc = timeit(lambda: m.predict_control(ex, task, kv_cache=kv), 5)
    tt = timeit(lambda: m.predict_text(prompt, max_length=64, kv_cache=kv), 2)
    print(f"kv_cache={kv!s:5}: predict_control (6 action tokens after 984 positions) {tc:8.2f} ms/env step   "
          f"predict_text (64 tokens after 900) {tt:9.1f} ms = {tt / 64:6.2f} ms/token")
