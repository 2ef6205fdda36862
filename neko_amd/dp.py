"""Data-parallel gradient reduction over RCCL/xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

Replaces what Accelerate/DDP does implicitly for the reference: one all-reduce of the
gradients per step, averaged with equal weight per rank.  MI355X-first differences:
  * gradients already live in ONE flat fp32 buffer laid out in reverse order of completion, so a
    bucket is a contiguous slice -- no bucket packing copies, no autograd hooks;
  * a bucket's all-reduce is enqueued on torch.distributed's communication stream the moment the
    backward of its layers has been *enqueued* (stream-ordered by an event, never a host wait), so
    the reduction of layer i overlaps the backward kernels of layers < i;
  * the per-forward buffer broadcast (6 MB of causal masks) and the per-step "unused parameter"
    bitmap all-reduce of DDP are gone: masks are computed from indices, and the set of ranges that
    took part is a handful of ints reduced (MAX) asynchronously and read on the device by the optimiser;
  * averaging is folded into the optimiser's gradient scale (1/world) instead of a divide pass.
Works with backend "gloo" on CPU tensors too (used by the world_size-2 CPU tests).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class _Done:
    """a finished piece of work (marks where a staged payload is widened back once the handles in front of it have been waited for)"""
    def wait(self):
        return True


class GradReducer:
    def __init__(self, flat, bucket_bytes: Optional[int] = None, group=None, payload: str = "fp32", force_collectives: bool = False,
                 collective: Optional[str] = None):
        """payload: "fp32" (default: what DDP sends for the reference's fp32 master gradients) or "bf16" -- every bucket is
        rounded to bf16 into a staging buffer, summed over the ranks in bf16 and widened back into the fp32 gradient:
        half the bytes on each xGMI link (249 instead of 498 MB per step at 768d) for one extra rounding of each rank's
        gradient (relative 2^-9; the sum itself is still exact to bf16 per hop).  NEKO_DP_PAYLOAD selects it in bench.py /
        train.py."""
        assert payload in ("fp32", "bf16"), payload
        self.payload = payload
        #: "allreduce" (default) or "rs_ag" (NEKO_DP_COLLECTIVE): every message as reduce-scatter + all-gather.  The sums are
        #: the same; what differs is what RCCL may do with them on the 7-link xGMI mesh: an all-reduce that RCCL runs as a ring carries the
        #: whole payload over ONE link per direction, while reduce-scatter / all-gather of 1/N shards can use every direct link at once.
        #: Unmeasured here (the pool hands out one GPU per box): opt-in, and tools/scale_check.sh records which one a real node prefers.
        #: A backend without reduce_scatter_tensor falls back to all-reduce, with one warning.
        collective = collective or os.environ.get("NEKO_DP_COLLECTIVE", "allreduce")
        assert collective in ("allreduce", "rs_ag"), collective
        self.collective = collective
        self._rs_ag_ok: Optional[bool] = None
        self._keep: List = []                # shards of in-flight reduce-scatter / all-gather pairs
        #: issue every collective even in a world of ONE rank (a sum over one rank is the identity): the only way to run
        #: the reducer's RCCL calls, streams and events on a one-GPU box (tests/test_dp_gpu.py, bench.py --force-dp)
        self.force = bool(force_collectives)
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # One collective per contiguous live slice of a run: RCCL pipelines a large message itself, and every extra call is ~60 us of host
        # time (c10d bookkeeping + an event pair + the wait) that README-size steps, whose host enqueue IS the step, pay in full --
        # tools/probe/r05_dp_host_timers.py: 13.8 all-reduce calls per c3 step with 64 MB slices, 0.65 ms of host time.  NEKO_DP_BUCKET_MB
        # (or bucket_bytes) restores an upper bound on the message size.
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("NEKO_DP_BUCKET_MB", "4096")) * (1 << 20))
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._dirty_names: Dict[tuple, List[str]] = {}
        self.handles: List = []
        self.pending: Dict[str, bool] = {}
        self.grad_scale = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=flat.grad.device)
        #: flat-gradient slices known to stay zero on EVERY rank for the whole run (see declare_unused_rows): not reduced
        self.zero_slices: List[tuple] = []
        self.no_text_declared = False       # set with declare_unused_rows("embed_token.weight", 0, text_tokens)
        #: False on the accumulating micro-steps of gradient accumulation: nothing is reduced (the flat gradient keeps
        #: adding up locally) until the last micro-step, whose backward reduces every range once
        self.sync = True
        self.model = None
        #: finished ranges wait until a contiguous run of at least this many bytes has built up (or backward ends) before their
        #: all-reduce is issued: a transformer layer of the 768d model is 28 MB, and at README batch sizes (c3: 5.8 ms per step) every
        #: collective's fixed cost -- an RCCL launch, an event pair and a join of the weight-gradient side stream -- showed in the
        #: one-GPU anchor (6.89 ms with the reducer against 5.77 without it, profiles/r05_c3_forcedp_*).  NEKO_DP_MIN_RUN_MB=0: one per range.
        self.min_run_elems = int(float(os.environ.get("NEKO_DP_MIN_RUN_MB", "64")) * (1 << 20)) // 4
        self._run: Optional[List[int]] = None          # [a, b) of the finished, not yet reduced contiguous ranges

    def broadcast_parameters(self, src: int = 0) -> None:
        """One-time parameter broadcast rank0 -> all (DDP constructor semantics)."""
        if self.world > 1 or self.force:
            dist.broadcast(self.flat.data, src=src, group=self.group)

    #: ranges that do not take part in every step on every rank (no image in the batch, embeddings
    #: passed in directly).  Collectives must be issued in the same order on all ranks, so these are
    #: never reduced from inside backward: `flush()` reduces them unconditionally at the end.
    DEFERRED = ("frontend", "image")

    def group_ready(self, gname: str) -> None:
        """All gradient kernels of flat group `gname` are enqueued: launch its all-reduce(s)."""
        if (self.world == 1 and not self.force) or not self.sync or gname not in self.flat.group_ranges or gname in self.DEFERRED:
            return
        a, b = self.flat.group_ranges[gname]
        if gname == "head":
            # first range of a backward: a run left behind by a backward that raised (or by a caller that skipped flush()) must not be
            # issued now, on this rank only -- the collective sequences of the ranks would diverge
            self._run = None
        # ranges finish from the end of the flat buffer towards its start (head, ln_f, layer L-1 .. 0): extend the waiting run downwards,
        # or start a new one behind a gap.  The decision depends on the range order and sizes only, so every rank issues the same collectives.
        if self._run is not None and self._run[0] == b:
            self._run[0] = a
        elif self._run is not None and self._run[1] == a:
            self._run[1] = b
        else:
            self._issue_run()
            self._run = [a, b]
        if self._run[1] - self._run[0] >= self.min_run_elems:
            self._issue_run()

    def _issue_run(self) -> None:
        if self._run is not None:
            a, b = self._run
            self._run = None
            self._reduce_range(a, b)

    def flush(self) -> None:
        """After backward: reduce what is still waiting, then the ranges that are not guaranteed to be touched on every rank.
        MANDATORY after every synchronised backward: a trailing run below `min_run_elems` and the DEFERRED ranges are only reduced here
        (`finish()` refuses to go on when a run is still waiting)."""
        if (self.world == 1 and not self.force) or not self.sync:
            return
        self._issue_run()
        for g in self.DEFERRED:
            if g in self.flat.group_ranges:
                self._reduce(g)

    def declare_unused_rows(self, name: str, row0: int, row1: int) -> None:
        """Rows [row0, row1) of parameter `name` never receive a gradient on ANY rank in this run -- a property of the
        run's configuration that every rank must declare identically (the collective layout depends on it).  The use:
        control-only training (BASELINE configs[1..3], text_prop = caption_prop = vqa_prop = 0) never touches the
        50257 text rows of `embed_token` (154 MB of the 160 MB `frontend` range at 768d), and that range is the one
        reduction that cannot overlap with backward (its gradient is produced last).  Summing zeros is the identity,
        so skipping the slice changes nothing.  GatoPolicy refuses text input while the declaration is active."""
        off, numel, shape = self.flat.offsets[name]
        cols = numel // shape[0]
        assert 0 <= row0 <= row1 <= shape[0]
        if row1 > row0:
            self.zero_slices.append((off + row0 * cols, off + row1 * cols))
            self.zero_slices.sort()

    def _live_ranges(self, a: int, b: int) -> List[tuple]:
        """[a, b) minus the declared zero slices."""
        out, cur = [], a
        for (z0, z1) in self.zero_slices:
            if z1 <= cur or z0 >= b:
                continue
            if z0 > cur:
                out.append((cur, z0))
            cur = max(cur, z1)
        if cur < b:
            out.append((cur, b))
        return out

    def _reduce(self, gname: str) -> None:
        a, b = self.flat.group_ranges[gname]
        self._reduce_range(a, b)

    def _reduce_range(self, a: int, b: int) -> None:
        if self.model is not None and self.model._flat is not self.flat:
            raise RuntimeError("GradReducer: the model's flat parameter storage was rebuilt after attach() "
                               "(a .to() / .cuda() that really moved parameters); attach after the model is on its device")
        # Weight gradients may still be running on the side stream (README-size steps).  The collective has to wait for them, the backward
        # chain on the compute stream does not: the collective is issued with the side stream current (c10d orders its communication
        # stream behind the current stream), after the side stream has been ordered behind everything enqueued on the compute stream so far
        # (bias / LayerNorm gradients of the same range).  Round 4 joined the side stream into the compute stream here instead, which
        # serialised the weight gradients with the chain at every reduce point.
        import contextlib
        issue_on = contextlib.nullcontext()
        if self.flat.grad.is_cuda:
            from .engine import SideStream
            side = SideStream.pending(self.flat.grad.device)
            if side is not None and os.environ.get("NEKO_DP_JOIN_MAIN") == "1":      # round-4 behaviour, for A/B runs
                SideStream.join(self.flat.grad.device)
                side = None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(self.flat.grad.device))
                issue_on = torch.cuda.stream(side)
        dirty = getattr(self.flat, "_dirty", None)
        if dirty is not None:       # a reduced slice may hold other ranks' gradients even if this rank never wrote it
            names = self._dirty_names.get((a, b))
            if names is None:
                names = self._dirty_names[(a, b)] = [n for n, (o, _, _) in self.flat.offsets.items() if a <= o < b]
            dirty.update(names)
        if os.environ.get("NEKO_DP_DRY") == "1":      # probe: hooks, stream ordering and bookkeeping without the collectives themselves
            return
        with issue_on:
            for (la, lb) in self._live_ranges(a, b):
                for s in range(la, lb, self.bucket_elems):
                    e = min(lb, s + self.bucket_elems)
                    g = self.flat.grad[s:e]
                    if self.payload == "bf16":
                        st = torch.empty(e - s, dtype=torch.bfloat16, device=g.device)
                        if g.is_cuda:
                            from . import ops
                            ops.cast_f32_bf16(g, st)
                        else:
                            st.copy_(g)
                        for h in self._sum_over_ranks(st):
                            self.handles.append((h, None, g))
                        self.handles.append((_Done(), st, g))          # widened back into g by finish(), behind the handles above
                    else:
                        for h in self._sum_over_ranks(g):
                            self.handles.append((h, None, g))

    def _sum_over_ranks(self, t: torch.Tensor) -> List:
        """SUM of `t` (1-D, contiguous) over the ranks, in place; returns the async work handles in issue order."""
        n, w = t.numel(), max(self.world, 1)
        if self.collective == "rs_ag" and n >= 1024 * w and self._rs_ag_ok is not False:
            n0 = n // w * w                                   # the part that divides into equal shards; < w elements are left over
            shard = torch.empty(n0 // w, dtype=t.dtype, device=t.device)
            try:
                hs = [dist.reduce_scatter_tensor(shard, t[:n0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
                self._rs_ag_ok = True
            except (RuntimeError, NotImplementedError) as e:   # raised at issue time, on every rank alike (same backend): nothing was sent
                self._rs_ag_ok = False
                import warnings
                warnings.warn(f"GradReducer(collective='rs_ag'): the backend has no reduce_scatter_tensor for these tensors ({e}); using all-reduce")
                return [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
            self._keep.append(shard)
            # NCCL / RCCL: the collectives of one process group run on one communication stream, so the all-gather is ordered behind the
            # reduce-scatter that fills `shard` without any wait here (a wait would order the COMPUTE stream behind it).  Other backends
            # (gloo: a pool of worker threads) give no such order: wait for the first before issuing the second.
            if not (dist.get_backend(self.group) == "nccl" and t.is_cuda):
                hs[0].wait()
            hs.append(dist.all_gather_into_tensor(t[:n0], shard, group=self.group, async_op=True))
            if n0 < n:
                hs.append(dist.all_reduce(t[n0:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return hs
        return [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]

    def reduce_flags(self, flags: torch.Tensor) -> None:
        if self.world > 1 or self.force:   # stream-ordered (NCCL: the current stream waits on the comm stream, the host does not)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)

    def finish(self) -> None:
        """Make the current stream wait for every outstanding reduction (no host block on CUDA)."""
        if self._run is not None:
            raise RuntimeError("GradReducer.finish(): finished gradient ranges are still waiting for their all-reduce -- flush() has to be "
                               "called after backward and before finish()")
        for h, staged, g in self.handles:
            h.wait()
            if staged is not None:
                g.copy_(staged)          # widen the summed bf16 payload back into the fp32 gradient (stream-ordered)
                if staged.is_cuda:       # (allocated while the side stream was current, read here on the compute stream)
                    staged.record_stream(torch.cuda.current_stream(staged.device))
        self.handles.clear()
        self._keep.clear()
        if self.flat.grad.is_cuda:
            from .engine import SideStream
            SideStream.join(self.flat.grad.device)      # nothing of the backward is left on the side stream when the optimiser starts

    def attach(self, model, optimizer) -> None:
        model._dp = self
        self.model = model
        model.image_embedding._on_grads_ready = lambda: self.group_ready("image")
        optimizer.grad_scale = self.grad_scale
        optimizer.flags_reduce = self.reduce_flags
