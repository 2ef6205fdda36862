#!/usr/bin/env python
"""cProfile of the host side of c3 steps (README batch size: the host enqueue is the step).  The backward is run on the calling thread
(torch.autograd.set_multithreading_enabled(False)) so that the profiler sees it."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from neko_amd.policy.gato_policy import GatoPolicy
from neko_amd.training.optim import NekoAdamW

dev = torch.device("cuda", 0)
torch.set_num_threads(8)
torch.manual_seed(0)
model = GatoPolicy(dev, 768, 6, 24, 0.1, resid_mid_channels=128, context_len=1024, text_tokenizer=bench.V_TEXT)
model.train()
opt = NekoAdamW(model, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
batches = [bench.make_batch("c3", 32, 1234 + 100 * i, dev) for i in range(2)]


def step(i):
    _, loss = model.forward(inputs=batches[i % 2], compute_loss=True, return_logits=False)
    loss.backward()
    opt.clip_grad_norm_(1.0)
    opt.step()
    opt.zero_grad()


torch.autograd.set_multithreading_enabled(False)
for i in range(10):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
N = 40
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"# {N} steps; times below are totals over them (divide by {N})")
st.print_stats(45)
st.sort_stats("cumulative")
st.print_stats(35)
