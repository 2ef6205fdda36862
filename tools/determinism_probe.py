#!/usr/bin/env python
"""Run-to-run reproducibility of one forward + backward on a fixed model and batch: padded layout vs length buckets.
Differences beyond fp32 atomic-order noise (~1e-7 relative) would point at an uninitialised read or a race."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import neko_oracle as O  # noqa: E402
import test_policy_gpu as T  # noqa: E402

f = torch.load("tests/golden/g7_trace.pt", weights_only=False)
cfg = O.OracleConfig(**f["cfg"])
for ragged in (0, 3):
    for bi, b in enumerate(f["batches"][:2]):
        m, _ = T.make_policy(cfg, f["seed"], train=False)
        m.ragged_groups = ragged
        batch = T.to_dev(b)
        runs = []
        for r in range(4):
            m.zero_grad(set_to_none=True)
            m._flat.zero_grad()
            _, loss = m(batch, compute_loss=True, return_logits=False)
            loss.backward()
            torch.cuda.synchronize()
            runs.append((float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
        worst = ("", 0.0)
        for r in runs[1:]:
            for k, g in r[1].items():
                d = float((g - runs[0][1][k]).abs().max() / runs[0][1][k].abs().max().clamp(min=1e-30))
                if d > worst[1]:
                    worst = (k, d)
        print(f"ragged={ragged} batch {bi}: losses {[x[0] for x in runs]} worst relative grad difference {worst[1]:.3e} ({worst[0]})")
