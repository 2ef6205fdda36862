#!/usr/bin/env python
"""How far do two EAGER runs of the captured-step test's recipe drift apart (same seeds, same batches)?  The only run-to-run
difference of a step is the order of the fp32 atomics in the embedding scatters (~1e-7 of a gradient entry); AdamW divides by
sqrt(v), so an entry whose gradient IS rounding noise moves by +-lr with a noise-dependent sign and the trajectories separate.
Prints the relative loss difference per step over N runs against run 0.  With NEKO_DETERMINISTIC=1 (sorted, fixed-order table
gradients, ABI v15) every difference is exactly 0."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_captured_gpu as T  # noqa: E402

runs = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    m0 = T._policy(0.0)
    m0.eval(); m0.train()
    opt0, sch0 = T._opt(m0)
    torch.manual_seed(3)
    ls = []
    for b in T._batches():
        _, loss = m0.forward(inputs=b, compute_loss=True, return_logits=False)
        loss.backward()
        opt0.clip_grad_norm_(1.0)
        opt0.step(); sch0.step(); opt0.zero_grad()
        ls.append(loss.detach())
    runs.append(torch.stack(ls).cpu())
for r in runs[1:]:
    print(" ".join(f"{float(x):.1e}" for x in ((r - runs[0]).abs() / runs[0].abs())))
