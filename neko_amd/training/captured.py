"""A whole training step as ONE HIP graph per batch structure (the brief's "capture launch-bound inner
loops in hipGraphs").

Why: at the README batch sizes (configs[1]: 32 x 240 halfcheetah tokens per step) a step is ~400 kernel launches of a few
microseconds each and the host needs as long to enqueue them through ctypes as the device needs to run them
(profiles/r01_step19_c2_host_profile.txt: 5.9 ms of enqueue per 6.3 ms step).  Replaying a captured step costs one launch.

What has to hold for a step to be replayable:
  * every kernel ARGUMENT is frozen at capture time.  The three per-step host scalars therefore live in device memory:
    the learning rate (NekoAdamW.lr_dev, read by the AdamW kernel), the dropout variation (ops.set_drop_salt: one uint32
    every dropout site adds to its key, advanced by the graph itself at the end of each replay) and the optimiser's step
    counters / clip norm (already device-side).
  * every device INPUT lives at a fixed address: GatoPolicy._prepare() does the host half of tokenize_input_dicts
    (descriptor table, value buffers, image groups, patch positions, loss rows -- numpy and a handful of uploads)
    OUTSIDE the graph; its tensors are copied into the graph's static twins before each replay.
  * the batch STRUCTURE (shapes, number of loss rows, image groups, length buckets) is the graph's key: a new structure
    runs one eager step (which also warms the allocator) and is captured at its second occurrence.

Semantics are those of Trainer.train_step (trainer.py:176-186): forward, backward, clip, AdamW, scheduler step,
zero_grad.  With dropout off the losses are those of the eager path up to the order of the fp32 atomics in the
embedding scatter (tests/test_captured_gpu.py)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops


class _Entry:
    __slots__ = ("static", "graph", "loss", "gnorm", "seen", "act", "active")

    def __init__(self):
        self.static = None
        self.graph = None
        self.loss = None
        self.gnorm = None
        self.seen = 0
        self.act = None         # which parameter ranges take part in a step of this structure (host list) ...
        self.active = None      # ... and the device copy the graph's AdamW launches point at


class CapturedTrainStep:
    def __init__(self, model, optimizer, scheduler=None, grad_norm_clip: Optional[float] = 1.0, max_graphs: int = 8):
        if getattr(model, "_dp", None) is not None:
            raise NotImplementedError("CapturedTrainStep: the data-parallel reducer's collectives are not captured; "
                                      "run multi-rank jobs on the eager path")
        dev = model._dev()
        self.model, self.opt, self.sch, self.clip = model, optimizer, scheduler, grad_norm_clip
        self.entries: Dict[tuple, _Entry] = {}
        self.max_graphs = max_graphs
        self.salt = torch.zeros(1, dtype=torch.int32, device=dev)
        optimizer.lr_dev = torch.full((1,), float(optimizer.param_groups[0]["lr"]), dtype=torch.float32, device=dev)
        ops.set_drop_salt(self.salt)                 # eager steps of this model read the same salt: one code path
        self._eager_active = optimizer.active        # the optimiser's own flag tensor (eager steps rewrite it)
        self._flat = model._flat                     # the graphs hold pointers into this storage
        self.replays = self.eager_steps = 0

    def close(self) -> None:
        """Back to plain eager training: the salt is unregistered, the optimiser takes `lr` from its param group again."""
        torch.cuda.synchronize()
        ops.set_drop_salt(None)
        self.opt.lr_dev = None
        self.opt.active, self.opt._last_act = self._eager_active, None
        self.entries.clear()

    # ---- one step, eager: exactly what the graph holds ---------------------------------------------------------------
    def _body(self, pr):
        loss = self.model._loss_from_prepared(pr)
        loss.backward()
        gnorm = self.opt.clip_grad_norm_(self.clip) if self.clip is not None else None
        self.opt.step()
        self.opt.zero_grad()
        self.salt.add_(0x61C88647)                   # int32 wrap-around: next step, other masks at every site
        return loss.detach(), gnorm

    def step(self, batch, ragged_groups: Optional[int] = None):
        """Returns (loss, pre-clip gradient norm or None) as device tensors (no host sync)."""
        m = self.model
        assert m.training, "CapturedTrainStep.step is the training step (model.train())"
        if m._flat is not self._flat:
            raise RuntimeError("CapturedTrainStep: the model's flat parameter storage was rebuilt (a .to() / .cuda() that really "
                               "moved parameters) -- the captured graphs point into the old one; build a new CapturedTrainStep")
        rg = m.ragged_groups if ragged_groups is None else ragged_groups
        pr = m._prepare(batch, rg if len(batch) > 1 else 0)
        lr = float(self.sch.get_last_lr()[0]) if self.sch is not None else float(self.opt.param_groups[0]["lr"])
        self.opt.param_groups[0]["lr"] = lr
        # the value travels as the ARGUMENT of a fill launch outside the graph: a reusable pinned host word could be overwritten
        # by a later step() before an earlier asynchronous upload has read it (the host runs ahead of the device in replay mode)
        self.opt.lr_dev.fill_(lr)
        key = pr.signature()
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= self.max_graphs:            # bounded: a stream of ever-new structures stays eager
                self.opt.active, self.opt._last_act = self._eager_active, None
                out = self._body(pr)
                self.eager_steps += 1
                self._after()
                return out
            e = self.entries[key] = _Entry()
        e.seen += 1
        if e.seen == 1:
            # first batch of this structure: a plain eager step (allocator warm-up, lazy module loads) that also tells which
            # parameter ranges a step of this structure touches: the graph gets its own device copy of those flags
            self.opt.active, self.opt._last_act = self._eager_active, None
            out = self._body(pr)
            self.eager_steps += 1
            e.act = list(self.opt._last_act)
            e.active = self._eager_active.clone()
        else:
            self.opt.active, self.opt._last_act = e.active, list(e.act)
            if e.graph is None:
                self._capture(e, pr)
            m._flat.ensure_shadow()          # weights edited from Python since the last step (load_state_dict): re-cast eagerly
            for dst, src in zip(e.static.tensors(), pr.tensors()):
                dst.copy_(src, non_blocking=True)
            e.graph.replay()
            self.replays += 1
            out = (e.loss.clone(), e.gnorm.clone() if e.gnorm is not None else None)
        self._after()
        return out

    def _after(self):
        if self.sch is not None:
            self.sch.step()

    def _capture(self, e: _Entry, pr) -> None:
        import copy
        st = copy.copy(pr)
        # static twins of every device input (same shapes / dtypes, own storage)
        st.desc = pr.desc.clone()
        st.cont = pr.cont.clone() if pr.cont is not None else None
        st.disc = pr.disc.clone() if pr.disc is not None else None
        st.img_groups = [(X.clone(), pos.clone(), idxs, cnt) for X, pos, idxs, cnt in pr.img_groups]
        st.given = [g.clone() for g in pr.given]
        st.pack = copy.copy(pr.pack)
        st.pack.loss_idx = pr.pack.loss_idx.clone()
        st.pack.row_map = pr.pack.row_map.clone() if pr.pack.row_map is not None else None
        st.pack.tgt_idx = pr.pack.tgt_idx.clone() if pr.pack.tgt_idx is not None else None
        e.static = st
        torch.cuda.synchronize()
        e.graph = torch.cuda.CUDAGraph()
        # One stream inside the graph: a captured fork onto the weight-gradient side stream becomes graph branches that the runtime
        # replays on queues of its own -- level with one stream at best (c2: 6.2 vs 6.0 ms with 4 hardware queues) and 1.5x SLOWER
        # with the 8 queues the package asks for (9.3 ms; profiles/r05_capture_hwq.txt)
        from .. import engine
        with engine.SideStream.suspended():          # thread-local, not the class-wide switch
            with torch.cuda.graph(e.graph):
                e.loss, e.gnorm = self._body(st)
