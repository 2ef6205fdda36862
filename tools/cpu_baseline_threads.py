#!/usr/bin/env python
"""Thread sweep of bench.py's cpu_baseline leg on this box's host cores (the oracle, B = 2 x T = 1024 text, fp32,
dropout masks included): 1 warm-up + 1 timed iteration per thread count.  Justifies the thread count bench.py uses."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import neko_oracle as O  # noqa: E402

D, L, H, T, V = 768, 6, 24, 1024, 50257
cfg = O.OracleConfig(embed_dim=D, layers=L, heads=H, text_tokens=V, context_len=T)
sd = O.init_state_dict(cfg, 0)
g = torch.Generator().manual_seed(1234)
batch = [{"text": torch.randint(0, V, (T - 1,), generator=g).tolist()} for _ in range(2)]
print(f"hardware threads: {os.cpu_count()}")
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    ts = []
    for it in range(2):
        t0 = time.perf_counter()
        keep = lambda *s: (torch.rand(*s, generator=g) >= 0.1).float() / 0.9
        dm = {"embd": keep(2, T, D)}
        for i in range(L):
            dm[("attn", i)] = keep(2, H, T, T); dm[("resid_attn", i)] = keep(2, T, D); dm[("resid_mlp", i)] = keep(2, T, D)
        O.loss_and_grads(sd, cfg, batch, drop_masks=dm)
        ts.append(time.perf_counter() - t0)
    print(f"threads {n:4d}: {ts[1]:7.2f} s per fwd+bwd iteration = {2 * T / ts[1]:7.1f} tokens/s", flush=True)
