"""Fused optimiser tail of Trainer.train_step (gato/training/trainer.py:181-186, train.py:127-133):
global grad-norm -> clip -> AdamW -> bf16 weight-shadow refresh, as a few HIP launches over the flat
parameter ranges, with no host synchronisation (norm, clip coefficient, step counters and the
"range took part in this step" flags live on the device).

torch semantics kept: decoupled weight decay on every parameter (no groups), bias correction with a
per-range step count, parameters whose grad is None this step (transformer.wte always;
image_embedding.* on image-free batches; the embedding tables when the caller passed embeddings
directly) are skipped entirely -- no decay, no moment update, no step increment.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops


class NekoAdamW(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.AdamW(model.parameters(), lr, betas, eps, weight_decay)`` on a neko_amd
    GatoPolicy.  ``step()`` = clip (if ``clip_grad_norm_`` was called since the last step) + AdamW."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1):
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        f = model._flat
        self.flat = f
        dev = f.device
        self.m = torch.zeros_like(f.data)
        self.v = torch.zeros_like(f.data)
        self.groups = [g for g in f.group_ranges if g != "never"]
        self.steps: Dict[str, torch.Tensor] = {g: torch.zeros(1, dtype=torch.int32, device=dev) for g in self.groups}
        self.active = torch.zeros(len(self.groups), dtype=torch.int32, device=dev)
        self._stager = None        # pinned staging for the per-step flag upload (event-guarded reuse)
        self._last_act = None      # flags currently in `self.active`
        self.gnorm_sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._pending_clip: Optional[float] = None
        #: optional 1-element fp32 device tensor holding the learning rate: when set, the AdamW kernel reads it instead of
        #: the `lr` kernel argument (a captured step cannot change an argument between replays, training/captured.py)
        self.lr_dev: Optional[torch.Tensor] = None
        self.grad_scale: Optional[torch.Tensor] = None    # set by the DP reducer (1/world)
        self.flags_reduce = None                          # DP: callable(active tensor) -> union over ranks

    def _check_storage(self):
        if self.model._flat is not self.flat:
            raise RuntimeError("NekoAdamW: the model's flat parameter storage was rebuilt after this optimiser was created "
                               "(a .to() / .cuda() that really moved or converted parameters); build the optimiser after "
                               "the model is on its device")

    # which ranges received gradients this step (host knowledge: param.grad attached by the backward)
    def _active_groups(self):
        f = self.flat
        act = []
        for g in self.groups:
            a, b = f.group_ranges[g]
            has = any(f.param_of[n].grad is not None for n, (o, _, _) in f.offsets.items() if a <= o < b)
            act.append(1 if has else 0)
        return act

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Like ``accelerator.clip_grad_norm_(model.parameters(), max_norm)`` (trainer.py:182): computes the global
        L2 norm now (device scalar, returned without sync) and applies the clip inside the next ``step()``."""
        self._check_storage()
        f = self.flat
        act = self._active_groups()
        self.gnorm_sq.zero_()
        # adjacent ranges are one launch (the flat gradient is contiguous; alignment gaps are zero)
        runs = []
        for g, on in zip(self.groups, act):
            if on or self.flags_reduce is not None:    # DP: another rank may have used the range (zeros add 0)
                a, b = f.group_ranges[g]
                if runs and runs[-1][1] == a:
                    runs[-1][1] = b
                else:
                    runs.append([a, b])
        for a, b in runs:
            ops.sqnorm_f32(f.grad[a:b], self.gnorm_sq)
        self._pending_clip = float(max_norm)
        gs = 1.0 if self.grad_scale is None else self.grad_scale
        return self.gnorm_sq.sqrt().to(torch.float32) * gs

    @torch.no_grad()
    def step(self, closure=None):
        self._check_storage()
        f = self.flat
        grp = self.param_groups[0]
        lr, (b1, b2), eps, wd = grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"]
        act = self._active_groups()
        # the flags rarely change from step to step: upload them only when they do (with a reducer attached the device
        # copy is overwritten by the MAX over ranks every step, so it is refreshed every step)
        if act != self._last_act or self.flags_reduce is not None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("NekoAdamW.step inside a graph capture needs the 'range active' flags already on the "
                                   "device: run one eager step on a batch of the same structure first")
            if self.active.is_cuda:
                if self._stager is None:
                    from ..utils.utils import HostStager
                    self._stager = HostStager()
                self.active.copy_(self._stager.upload(torch.tensor(act, dtype=torch.int32), self.active.device))
            else:
                self.active.copy_(torch.tensor(act, dtype=torch.int32))
            self._last_act = list(act)
        if self.flags_reduce is not None:
            self.flags_reduce(self.active)
        clip = self._pending_clip
        for i, g in enumerate(self.groups):
            if not act[i] and self.flags_reduce is None:
                continue
            a, b = f.group_ranges[g]
            ops.adamw_step(f.data[a:b], f.grad[a:b], self.m[a:b], self.v[a:b], f.shadow[a:b], lr, b1, b2, eps, wd,
                           self.gnorm_sq if clip is not None else None, clip if clip is not None else 0.0,
                           self.grad_scale, self.steps[g], self.active[i:i + 1], lr_dev=self.lr_dev)
        self._pending_clip = None
        f.mark_shadow_fresh()

    # ---- checkpoint / resume ("optimizer state for true resume") -------------------------
    def state_dict(self):
        """Adam moments of the flat storage, the per-range step counters and the hyper-parameters.  The moments are
        stored per PARAMETER NAME (state_dict keys of the model), not as the flat buffer, so a checkpoint survives a
        change of the flat layout (padding, range order)."""
        f = self.flat
        m, v = {}, {}
        for name, (off, numel, shape) in f.offsets.items():
            m[name] = self.m[off:off + numel].view(shape).detach().cpu().clone()
            v[name] = self.v[off:off + numel].view(shape).detach().cpu().clone()
        return {"format": "neko_amd.NekoAdamW/1", "exp_avg": m, "exp_avg_sq": v,
                "steps": {g: int(t.item()) for g, t in self.steps.items()},
                "param_groups": [{k: val for k, val in grp.items() if k != "params"} for grp in self.param_groups]}

    def load_state_dict(self, sd):
        if sd.get("format") != "neko_amd.NekoAdamW/1":
            raise ValueError("not a NekoAdamW state_dict")
        f = self.flat
        missing = [n for n in f.offsets if n not in sd["exp_avg"]]
        if missing:
            raise KeyError(f"optimizer state lacks {missing[:3]}{'...' if len(missing) > 3 else ''}")
        with torch.no_grad():
            for name, (off, numel, shape) in f.offsets.items():
                self.m[off:off + numel].copy_(sd["exp_avg"][name].reshape(-1).to(self.m.device))
                self.v[off:off + numel].copy_(sd["exp_avg_sq"][name].reshape(-1).to(self.v.device))
            for g, t in self.steps.items():
                t.fill_(int(sd["steps"].get(g, 0)))
        for grp, saved in zip(self.param_groups, sd["param_groups"]):
            grp.update(saved)

    def zero_grad(self, set_to_none: bool = True):
        """One memset of the flat gradient; ``.grad`` of every parameter is detached again (set to None) so the
        next backward knows which parameters took part (torch's set_to_none semantics)."""
        self.flat.zero_grad()
        for p in self.flat.param_of.values():
            p.grad = None
