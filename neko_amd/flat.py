"""Flat parameter / gradient / bf16-shadow storage.

MI355X-first layout: every trainable tensor of GatoPolicy lives in ONE fp32 buffer (``data``),
its gradient in a parallel buffer (``grad``) and a bf16 mirror for the MFMA kernels (``shadow``).
``nn.Parameter``s are views, so ``state_dict()`` keeps the reference's keys and shapes
(gato/utils/utils.py:31-32).  Consequences:
  * the optimiser tail (norm, clip, AdamW, shadow refresh) is a handful of launches over ranges,
  * the data-parallel gradient reduction works on contiguous slices with no packing copies,
  * HIP kernels accumulate straight into ``grad`` (autograd never materialises per-parameter grads).
Ranges ("groups") are laid out in reverse order of gradient completion so buckets become ready
back-to-front during backward.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

ALIGN = 64  # elements: every tensor starts on a 256-B (fp32) / 128-B (bf16) boundary


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class FlatParams:
    def __init__(self, groups: "OrderedDict[str, List[Tuple[str, nn.Parameter]]]", device,
                 padded_numel: Optional[Dict[str, int]] = None):
        padded_numel = padded_numel or {}
        self.device = torch.device(device)
        self.offsets: Dict[str, Tuple[int, int, Tuple[int, ...]]] = {}
        self.group_ranges: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        self.param_of: Dict[str, nn.Parameter] = {}
        off = 0
        for gname, plist in groups.items():
            start = off
            for name, p in plist:
                n = p.numel()
                self.offsets[name] = (off, n, tuple(p.shape))
                self.param_of[name] = p
                off += _round_up(max(n, padded_numel.get(name, n)), ALIGN)
            self.group_ranges[gname] = (start, off)
        self.total = off
        self.data = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.shadow = (torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
                       if self.device.type == "cuda" else None)
        with torch.no_grad():
            for name, (o, n, shape) in self.offsets.items():
                p = self.param_of[name]
                v = self.data[o:o + n].view(shape)
                v.copy_(p.data)
                p.data = v
        self._shadow_version = None
        #: names whose gradient slice may hold data since the last whole-buffer memset (zero_grad): only those need a
        #: memset when a backward finds their ``.grad`` detached -- the step used to zero the 498 MB buffer twice
        self._dirty: set = set(self.offsets)

    # ---- views ---------------------------------------------------------------------------------
    def view(self, name: str, padded_rows: Optional[int] = None) -> torch.Tensor:
        return self._v(self.data, name, padded_rows)

    def gview(self, name: str, padded_rows: Optional[int] = None) -> torch.Tensor:
        return self._v(self.grad, name, padded_rows)

    def sview(self, name: str, padded_rows: Optional[int] = None) -> torch.Tensor:
        return self._v(self.shadow, name, padded_rows)

    def _v(self, buf, name, padded_rows):
        o, n, shape = self.offsets[name]
        if padded_rows is not None:
            cols = shape[-1]
            return buf[o:o + padded_rows * cols].view(padded_rows, cols)
        return buf[o:o + n].view(shape)

    def range_of_group(self, g: str) -> Tuple[int, int]:
        return self.group_ranges[g]

    # ---- bf16 shadow ---------------------------------------------------------------------------
    def _version(self) -> int:
        return sum(p._version for p in self.param_of.values())

    def ensure_shadow(self) -> None:
        """Re-cast the bf16 mirror when any parameter was modified in place from Python
        (load_state_dict, a torch optimiser, manual edits).  The fused optimiser refreshes it itself."""
        v = self._version()
        if v != self._shadow_version:
            from . import ops
            ops.cast_f32_bf16(self.data, self.shadow)
            self._shadow_version = v

    def mark_shadow_fresh(self) -> None:
        self._shadow_version = self._version()

    # ---- gradients -------------------------------------------------------------------------------
    def zero_grad(self) -> None:
        self.grad.zero_()
        self._dirty.clear()

    def attach_grads(self, names: Sequence[str]) -> None:
        """Point ``param.grad`` of the named parameters at their slice of the flat gradient."""
        for name in names:
            p = self.param_of[name]
            g = self.gview(name)
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)

    def prepare_backward(self, names: Sequence[str]) -> None:
        """Kernels accumulate (+=) into the flat gradient.  Parameters whose ``.grad`` is None
        (fresh step, or zero_grad(set_to_none=True) from a torch optimiser) must start from zero."""
        fresh = sorted(self.offsets[n][0] for n in names if self.param_of[n].grad is None and n in self._dirty)
        self._dirty.update(names)
        if not fresh:
            return
        ends = {o: o + _round_up(n, ALIGN) for (o, n, _) in self.offsets.values()}
        # merge adjacent slices so a whole layer (or the whole model) is one memset
        runs: List[List[int]] = []
        for o in fresh:
            if runs and runs[-1][1] >= o:
                runs[-1][1] = max(runs[-1][1], ends[o])
            else:
                runs.append([o, ends[o]])
        for a, b in runs:
            self.grad[a:b].zero_()
