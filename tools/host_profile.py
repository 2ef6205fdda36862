#!/usr/bin/env python
"""Where the HOST time of one training step goes (the step is host-bound once the kernels are fast enough):
cProfile over a few m-mix steps, no device syncs inside the timed region.  python tools/host_profile.py [workload]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neko_amd.policy.gato_policy import GatoPolicy  # noqa: E402
from neko_amd.training.optim import NekoAdamW  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "m-mix"
dev = "cuda:0"
torch.set_num_threads(8)
torch.manual_seed(0)
model = GatoPolicy(dev, bench.D, bench.L, bench.H, 0.1, resid_mid_channels=128, context_len=bench.T, text_tokenizer=bench.V_TEXT)
model.train()
opt = NekoAdamW(model, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batches = [bench.make_batch(wl, B, 1234 + 100 * i, dev) for i in range(2)]


def step(i):
    _, loss = model.forward(inputs=batches[i % 2], compute_loss=True, return_logits=False)
    loss.backward()
    opt.clip_grad_norm_(1.0)
    opt.step()
    opt.zero_grad()


for i in range(3):
    step(i)
torch.cuda.synchronize()
N = 6
t0 = time.perf_counter()
for i in range(N):
    step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"[no profiler] B = {B}: host enqueue time {1e3 * t_host / N:.2f} ms/step; with final sync {1e3 * t_all / N:.2f} ms/step")
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue time {1e3 * t_host / N:.2f} ms/step; with final sync {1e3 * t_all / N:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
