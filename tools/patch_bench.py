#!/usr/bin/env python
"""Isolated timing of the patch-embedding kernels (936 patches = 26 Atari frames of 96x96)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neko_amd import ops
from oracle import neko_oracle as O
cfg = O.OracleConfig(embed_dim=64, layers=1, heads=2, text_tokens=128, context_len=64)
sd = O.init_state_dict(cfg, 21)
pe = "image_embedding.patch_embedding."
names = ["conv1.weight", "conv1.bias", "gn2.weight", "gn2.bias", "conv2.weight", "conv2.bias"]
dev = {n: sd[pe + n].cuda().contiguous() for n in names}
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 26      # 286 frames = 10296 patches: the Atari share of an m-mix step
imgs = torch.floor(torch.rand(frames, 3, 96, 96) * 256).cuda()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
y16, xp = ops.patch_resblock_fwd(imgs, *[dev[n] for n in names], 128, 32)
dy = torch.randn(xp.shape, device="cuda")
grads = {n: torch.zeros_like(dev[n]) for n in names}
def t(fn, name):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)*1e3/iters:.1f} us for {xp.shape[0]} patches")
t(lambda: ops.patch_resblock_fwd(imgs, *[dev[n] for n in names], 128, 32), "resblock fwd")
t(lambda: ops.patch_resblock_bwd(xp, dy, *[dev[n] for n in names], 128, 32, *[grads[n] for n in names]), "resblock bwd (recomputes the GroupNorm statistics)")
_, _, stats = ops.patch_resblock_fwd(imgs, *[dev[n] for n in names], 128, 32, want_stats=True)
if stats is not None:
    t(lambda: ops.patch_resblock_fwd(imgs, *[dev[n] for n in names], 128, 32, want_stats=True), "resblock fwd + statistics out")
    t(lambda: ops.patch_resblock_bwd(xp, dy, *[dev[n] for n in names], 128, 32, *[grads[n] for n in names], stats=stats), "resblock bwd (statistics from the forward)")
