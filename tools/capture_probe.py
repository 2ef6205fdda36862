#!/usr/bin/env python
"""Where does a captured C2 step spend its time?  (a) graph replays back to back on frozen inputs (device-side cost of the
graph), (b) the host half of a step (GatoPolicy._prepare + input copies), (c) the eager step for comparison."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd.policy.gato_policy import GatoPolicy
from neko_amd.tasks import synthetic as S
from neko_amd.training.captured import CapturedTrainStep
from neko_amd.training.optim import NekoAdamW
from neko_amd import engine

torch.set_num_threads(8)
dev = "cuda"
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
torch.manual_seed(0)
m = GatoPolicy(dev, 768, 6, 24, 0.1, resid_mid_channels=128, context_len=1024, text_tokenizer=50257)
m.train()
opt = NekoAdamW(m, lr=1e-4)
if wl == "c2":
    bs = [S.SyntheticControlTask(17, 6, 10, seed=s, device=dev).sample_batch(32) for s in (1, 2)]
else:
    bs = [S.SyntheticAtariTask(13, 96, 96, seed=s, device=dev).sample_batch(32) for s in (1, 2)]

def eager(i):
    _, loss = m.forward(inputs=bs[i % 2], compute_loss=True, return_logits=False)
    loss.backward(); opt.clip_grad_norm_(1.0); opt.step(); opt.zero_grad()

def timed(fn, n):
    for i in range(5): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print(f"[{wl}] eager step                         : {timed(eager, 50):.3f} ms")
engine.SideStream.enabled = False
print(f"[{wl}] eager step, one stream             : {timed(eager, 50):.3f} ms")
engine.SideStream.enabled = True
engine.SideStream.enabled = True
cap = CapturedTrainStep(m, opt, None)          # (captures on ONE stream whatever the eager setting: see CapturedTrainStep._capture)
for i in range(4): cap.step(bs[i % 2])
e = next(iter(cap.entries.values()))
print(f"[{wl}] captured step (prepare + copies + replay)   : {timed(lambda i: cap.step(bs[i % 2]), 50):.3f} ms")
print(f"[{wl}] graph.replay() back to back                  : {timed(lambda i: e.graph.replay(), 50):.3f} ms")
t0 = time.perf_counter()
for i in range(50): pr = m._prepare(bs[i % 2], 0)
torch.cuda.synchronize()
print(f"[{wl}] host half (_prepare) alone                   : {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
