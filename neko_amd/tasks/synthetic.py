"""Synthetic task generators that emit the reference's batch-dict format:
``ControlTask.sample_batch_configurable`` (gato/tasks/control_task.py:296-325), ``TextTask.sample_batch``
(gato/tasks/text_task.py:44-55), ``CaptionTask.sample_batch`` (gato/tasks/caption_task.py:112-118).
The reference's own tasks need Minari / gymnasium / HF datasets / network (out of scope);
these produce the same keys, dtypes and shapes from a seeded generator: fixed-shape inputs for the
benchmark and for parity tests."""
from __future__ import annotations

from typing import List, Optional

import torch


class SyntheticTask:
    name = "synthetic"
    kind = "control"     # which batch proportion the Trainer draws it for: control | text | caption | vqa

    def __init__(self, seed: int = 1234, device="cpu"):
        self.gen = torch.Generator().manual_seed(seed)
        self.device = device

    def _to(self, t):
        return t.to(self.device)


class SyntheticControlTask(SyntheticTask):
    """Continuous-observation control episodes, e.g. halfcheetah: n_obs=17, n_act=6 -> 24 tokens/timestep."""

    def __init__(self, n_obs=17, n_act=6, n_timesteps=10, name="halfcheetah-like", **kw):
        super().__init__(**kw)
        self.n_obs, self.n_act, self.n_ts, self.name = n_obs, n_act, n_timesteps, name

    def sample_batch(self, batch_size: int, *_, **__) -> List[dict]:
        out = []
        for _ in range(batch_size):
            obs = torch.randn(self.n_ts, self.n_obs, generator=self.gen)
            act = torch.rand(self.n_ts, self.n_act, generator=self.gen) * 2 - 1
            out.append({"continuous_actions": self._to(act), "continuous_obs": self._to(obs)})
        return out


class SyntheticAtariTask(SyntheticTask):
    """Image observations (n_ts,3,H,W) in 0..255 + one discrete action per timestep (Breakout: 96x96 -> 36 patches)."""

    def __init__(self, n_timesteps=13, height=96, width=96, n_actions=4, name="atari-like", **kw):
        super().__init__(**kw)
        self.n_ts, self.h, self.w, self.n_actions, self.name = n_timesteps, height, width, n_actions, name

    def sample_batch(self, batch_size: int, *_, **__) -> List[dict]:
        out = []
        for _ in range(batch_size):
            img = torch.randint(0, 256, (self.n_ts, 3, self.h, self.w), generator=self.gen).to(torch.float32)
            act = torch.randint(0, self.n_actions, (self.n_ts, 1), generator=self.gen).to(torch.int32)
            out.append({"discrete_actions": self._to(act), "images": self._to(img)})
        return out


class SyntheticTextTask(SyntheticTask):
    kind = "text"

    def __init__(self, n_tokens=1023, vocab=50257, name="text", **kw):
        super().__init__(**kw)
        self.n_tokens, self.vocab, self.name = n_tokens, vocab, name

    def sample_batch(self, batch_size: int) -> List[dict]:
        out = []
        for _ in range(batch_size):
            ids = torch.randint(0, self.vocab, (self.n_tokens,), generator=self.gen).tolist()
            out.append({"text": ids, "images": None, "continuous_obs": None, "discrete_obs": None,
                        "continuous_actions": None, "discrete_actions": None})
        return out


class SyntheticCaptionTask(SyntheticTask):
    """uint8 image (1,3,256,256) on the CPU + caption token ids (caption_task.py:112-118)."""
    kind = "caption"

    def __init__(self, n_tokens=767, vocab=50257, size=256, name="caption", **kw):
        super().__init__(**kw)
        self.n_tokens, self.vocab, self.size, self.name = n_tokens, vocab, size, name

    def sample_batch(self, batch_size: int) -> List[dict]:
        out = []
        for _ in range(batch_size):
            img = torch.randint(0, 256, (1, 3, self.size, self.size), generator=self.gen).to(torch.uint8)
            ids = torch.randint(0, self.vocab, (self.n_tokens,), generator=self.gen).tolist()
            out.append({"images": img, "text": ids})
        return out


def metric_mix_batch(batch_size: int, seed: int, device, text_vocab: int = 50257) -> List[dict]:
    """The M-mix workload of SURVEY.md 8(d): a batch cycling three example kinds that each (nearly)
    fill 1024 positions -- caption-like 256 patches + 767 ids + SEP = 1024; control-like 42x(17+1+6) = 1008
    (left-padded); Atari-like 26x(36+1+1) = 988 (left-padded)."""
    cap = SyntheticCaptionTask(seed=seed, device=device, vocab=text_vocab)
    ctl = SyntheticControlTask(17, 6, 42, seed=seed + 1, device=device)
    atr = SyntheticAtariTask(26, 96, 96, seed=seed + 2, device=device)
    out = []
    for i in range(batch_size):
        out += (cap, ctl, atr)[i % 3].sample_batch(1)
    return out


def ragged_mix_batch(batch_size: int, seed: int, device, text_vocab: int = 50257) -> List[dict]:
    """A C5-like length mix (BASELINE configs[4]: text + MuJoCo + Atari + caption in one batch): full-context text
    (1023 ids + SEP = 1024), Atari 13 x (36 + 1 + 1) = 494, caption 256 patches + 32 ids + SEP = 289, halfcheetah
    10 x (17 + 1 + 6) = 240.  The reference left-pads all of them to 1024 (gato_policy.py:408-416): half of the
    positions are padding, which is what GatoPolicy.ragged_groups removes."""
    txt = SyntheticTextTask(1023, text_vocab, seed=seed, device=device)
    atr = SyntheticAtariTask(13, 96, 96, seed=seed + 1, device=device)
    cap = SyntheticCaptionTask(32, text_vocab, seed=seed + 2, device=device)
    ctl = SyntheticControlTask(17, 6, 10, seed=seed + 3, device=device)
    out = []
    for i in range(batch_size):
        out += (txt, atr, cap, ctl)[i % 4].sample_batch(1)
    return out
