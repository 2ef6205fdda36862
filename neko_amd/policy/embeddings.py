"""Host-side mirror of gato/policy/embeddings.py (ImageEmbedding, PatchPosEncoding, ResidualBlock_V2).

Same class names / constructor arguments / state_dict keys; the compute is HIP:
  patchify + normalise + residual conv block  -> neko_patch_resblock_fwd/bwd  (patch_embed.hip)
  Linear(768 -> d)                              -> neko_gemm_bf16
  + row/col position embeddings                 -> neko_patch_pos_add(/_bwd)
Patch-position indices are computed on the host exactly like the reference: eval = rounded interval
midpoint (embeddings.py:96-100); train = one ``torch.randint`` per row/col interval in the same call
order (embeddings.py:92-94), so a seeded reference run and a seeded neko_amd run draw identical positions.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops


class ResidualBlock_V2(nn.Module):
    """Parameter container of embeddings.py:111-125."""

    def __init__(self, mid_channels: int = 128, num_groups: int = 32):
        super().__init__()
        in_channels = 3
        self.mid_channels, self.num_groups = mid_channels, num_groups
        self.gn1 = nn.Identity()
        self.act1 = nn.GELU()
        self.conv1 = nn.Conv2d(in_channels, mid_channels, kernel_size=3, stride=1, padding=1)
        self.gn2 = nn.GroupNorm(num_groups, mid_channels)
        self.act2 = nn.GELU()
        self.conv2 = nn.Conv2d(mid_channels, in_channels, kernel_size=3, stride=1, padding=1)


class PatchPosEncoding(nn.Module):
    def __init__(self, position_vocab_size=128, embed_dim=768):
        super().__init__()
        self.position_vocab_size = position_vocab_size
        self.embed_dim = embed_dim
        self.height_pos_embedding = nn.Embedding(position_vocab_size, embed_dim)
        self.width_pos_embedding = nn.Embedding(position_vocab_size, embed_dim)

    def intervals(self, n: int) -> torch.Tensor:
        """embeddings.py:80-89: int32 [n,2] quantised (lo, hi) of each patch row/col."""
        lin = torch.linspace(0, 1, n + 1)
        iv = torch.stack([lin[:-1], lin[1:]]).T
        return (iv * self.position_vocab_size).to(dtype=torch.int32)

    def positions(self, n_height: int, n_width: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Host indices (int32 CPU tensors) for the n_height rows and n_width cols (embeddings.py:91-100)."""
        h_iv, w_iv = self.intervals(n_height), self.intervals(n_width)
        if self.training:
            h = torch.tensor([int(torch.randint(low=int(a), high=int(b), size=())) for a, b in h_iv], dtype=torch.int32)
            w = torch.tensor([int(torch.randint(low=int(a), high=int(b), size=())) for a, b in w_iv], dtype=torch.int32)
        else:
            h_iv[:, 1] -= 1
            w_iv[:, 1] -= 1
            h = h_iv.mean(dim=-1, dtype=torch.float32).round().to(dtype=torch.int32)
            w = w_iv.mean(dim=-1, dtype=torch.float32).round().to(dtype=torch.int32)
        return h, w


class _ImageEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "ImageEmbedding", images, hpos, wpos, sorted_rows, *params):
        f = mod._flat
        pre = mod._prefix
        need = any(ctx.needs_input_grad)
        f.ensure_shadow()
        pe = pre + "patch_embedding."
        w = lambda n: f.view(n)
        y16, xp, y16_all, stats = ops.patch_resblock_fwd(images, w(pe + "conv1.weight"), w(pe + "conv1.bias"), w(pe + "gn2.weight"),
                                                         w(pe + "gn2.bias"), w(pe + "conv2.weight"), w(pe + "conv2.bias"),
                                                         mod.patch_embedding.mid_channels, mod.patch_embedding.num_groups,
                                                         want_x=need, want_padded=True, want_stats=True)
        P = y16.shape[0]
        d = mod.embed_dim
        out = torch.empty(P, d, dtype=torch.float32, device=y16.device)
        Wp = f.sview(pre + "post_embedding_projection.weight")     # (d, 768) k-contiguous
        ops.gemm(y16, Wp, P, d, 768, bias=f.view(pre + "post_embedding_projection.bias"), out_f32=out)
        if mod.use_pos_encoding:
            ops.patch_pos_add(out, hpos, wpos, f.view(pre + "patch_pos_encoding.height_pos_embedding.weight"),
                              f.view(pre + "patch_pos_encoding.width_pos_embedding.weight"))
        ctx.mod, ctx.y16_all, ctx.xp, ctx.hpos, ctx.wpos, ctx.gn_stats, ctx.sorted_rows = mod, y16_all, xp, hpos, wpos, stats, sorted_rows
        return out

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        f, pre = mod._flat, mod._prefix
        names = mod.flat_param_names(pre)
        f.prepare_backward(names)
        g = g.contiguous().to(torch.float32)
        P, d = g.shape
        if mod.use_pos_encoding:
            ops.patch_pos_add_bwd(g, ctx.hpos, ctx.wpos, f.gview(pre + "patch_pos_encoding.height_pos_embedding.weight"),
                                  f.gview(pre + "patch_pos_encoding.width_pos_embedding.weight"), sorted_rows=ctx.sorted_rows)
        # rows P .. Ppad of both operands of the weight gradient are zero (ops.patch_resblock_fwd pads y16's storage the same way)
        y16_all = ctx.y16_all                       # the zero-padded storage the forward's y16 is a row prefix of
        Ppad = y16_all.shape[0]
        assert y16_all.shape == (Ppad, 768) and Ppad % 128 == 0 and Ppad >= P
        g16_all = torch.empty(Ppad, d, dtype=torch.bfloat16, device=g.device)
        if Ppad > P:
            g16_all[P:].zero_()
        g16 = g16_all[:P]
        ops.cast_f32_bf16(g, g16)
        ops.colsum_bf16(g16, P, d, f.gview(pre + "post_embedding_projection.bias"))
        # dW[d,768] += g^T @ y ;  dy[P,768] = g @ W
        sk, kps = ops.pick_splitk(d, 768, Ppad)
        ops.gemm(g16_all, y16_all, d, 768, Ppad, a_kstrided=True, b_kstrided=True, lda=d, ldb=768,
                 out_f32=f.gview(pre + "post_embedding_projection.weight"), ldcf=768, accumulate=True,   # also with split-K: one call per image-shape group
                 splitk=sk, k_per_split=kps)
        dy = torch.empty(P, 768, dtype=torch.float32, device=g.device)
        ops.gemm(g16, f.sview(pre + "post_embedding_projection.weight"), P, 768, d, b_kstrided=True, ldb=768,
                 out_f32=dy)
        pe = pre + "patch_embedding."
        ops.patch_resblock_bwd(ctx.xp, dy, f.view(pe + "conv1.weight"), f.view(pe + "conv1.bias"),
                               f.view(pe + "gn2.weight"), f.view(pe + "gn2.bias"), f.view(pe + "conv2.weight"),
                               f.view(pe + "conv2.bias"), mod.patch_embedding.mid_channels,
                               mod.patch_embedding.num_groups, f.gview(pe + "conv1.weight"), f.gview(pe + "conv1.bias"),
                               f.gview(pe + "gn2.weight"), f.gview(pe + "gn2.bias"), f.gview(pe + "conv2.weight"),
                               f.gview(pe + "conv2.bias"), stats=ctx.gn_stats)
        f.attach_grads(mod.used_param_names(pre))
        if mod._on_grads_ready is not None:
            mod._on_grads_ready()
        return (None,) * len(ctx.needs_input_grad)


class ImageEmbedding(nn.Module):
    def __init__(self, embed_dim=768, patch_size=16, resid_mid_channels=128, num_groups=32, position_vocab_size=128,
                 use_pos_encoding=True):
        super().__init__()
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.patch_embedding = ResidualBlock_V2(mid_channels=resid_mid_channels, num_groups=num_groups)
        self.post_embedding_projection = nn.Linear(patch_size * patch_size * 3, embed_dim)
        self.use_pos_encoding = use_pos_encoding
        self.patch_pos_encoding = PatchPosEncoding(position_vocab_size=position_vocab_size, embed_dim=embed_dim)
        self._flat = None
        self._prefix = ""
        self._on_grads_ready = None

    _NAMES = ("patch_embedding.conv1.weight", "patch_embedding.conv1.bias", "patch_embedding.gn2.weight",
              "patch_embedding.gn2.bias", "patch_embedding.conv2.weight", "patch_embedding.conv2.bias",
              "post_embedding_projection.weight", "post_embedding_projection.bias")
    _POS = ("patch_pos_encoding.height_pos_embedding.weight", "patch_pos_encoding.width_pos_embedding.weight")

    def flat_param_names(self, prefix=""):
        return [prefix + n for n in self._NAMES + self._POS]

    def used_param_names(self, prefix=""):
        return [prefix + n for n in (self._NAMES + (self._POS if self.use_pos_encoding else ()))]

    def attach_flat(self, flat, prefix):
        self._flat, self._prefix = flat, prefix

    def forward(self, x, normalize=True, positions: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """embeddings.py:28-61: (n,3,H,W) in 0..255 -> (n, n_h*n_w, embed_dim)."""
        if self.patch_size != 16:
            raise NotImplementedError("the HIP patch kernel is specialised for patch_size=16")
        if not normalize:
            raise NotImplementedError("normalize=False is not used on the hot path")
        if self._flat is None:
            raise RuntimeError("ImageEmbedding must be owned by a GatoPolicy (flat parameter storage)")
        dev = self._flat.device
        if x.dtype not in (torch.float32, torch.uint8):
            x = x.to(torch.float32)
        x = x.to(dev)
        n, c, H, W = x.shape
        assert H % self.patch_size == 0 and W % self.patch_size == 0, "Image dimensions must be divisible by patch size"
        nh, nw = H // self.patch_size, W // self.patch_size
        hp, wp = positions if positions is not None else self.patch_pos_encoding.positions(nh, nw)
        # per-patch index arrays in (b, n_h, n_w) order
        hpos = hp.to(torch.int32).view(1, nh, 1).expand(n, nh, nw).reshape(-1).contiguous().to(dev, non_blocking=True)
        wpos = wp.to(torch.int32).view(1, 1, nw).expand(n, nh, nw).reshape(-1).contiguous().to(dev, non_blocking=True)
        params = [self._flat.param_of[nm] for nm in self.used_param_names(self._prefix)]
        out = _ImageEmbedFn.apply(self, x, hpos, wpos, None, *params)
        return out.view(n, nh * nw, self.embed_dim)

    def _upload(self, t: torch.Tensor, dev) -> torch.Tensor:
        """Asynchronous pinned H2D copy (neko_amd.utils.utils.HostStager); the policy shares its stager."""
        if getattr(self, "_stager", None) is None:
            from ..utils.utils import HostStager
            self._stager = HostStager()
        return self._stager.upload(t, dev)

    def prepare_many(self, xs):
        """Host half of forward_many: patch positions drawn per example (reference RNG order), examples of equal image
        shape grouped, every group's images and position indices on the device.  Returns a list of groups
        (X (n,3,H,W) device, pos int32 [2, P] device, example indices, patches per example)."""
        if self._flat is None:
            raise RuntimeError("ImageEmbedding must be owned by a GatoPolicy (flat parameter storage)")
        dev = self._flat.device
        prepared, groups = [], {}
        for i, x in enumerate(xs):
            if x.dtype not in (torch.float32, torch.uint8):
                x = x.to(torch.float32)
            n, c, H, W = x.shape
            assert H % self.patch_size == 0 and W % self.patch_size == 0, "Image dimensions must be divisible by patch size"
            nh, nw = H // self.patch_size, W // self.patch_size
            hp, wp = self.patch_pos_encoding.positions(nh, nw)
            # per-patch index arrays in (image, row, col) order; numpy: tiny CPU torch ops pay a fork/join per call
            hpos = np.broadcast_to(hp.numpy().astype(np.int32).reshape(1, nh, 1), (n, nh, nw)).reshape(-1)
            wpos = np.broadcast_to(wp.numpy().astype(np.int32).reshape(1, 1, nw), (n, nh, nw)).reshape(-1)
            prepared.append((x, hpos, wpos, n * nh * nw))
            groups.setdefault((H, W, x.dtype), []).append(i)
        out = []
        for idxs in groups.values():
            on_dev = [prepared[i][0] for i in idxs if prepared[i][0].is_cuda]
            on_cpu = [prepared[i][0] for i in idxs if not prepared[i][0].is_cuda]
            if on_cpu and on_dev:       # keep example order: bring the stragglers over first (mixed residency is rare)
                X = torch.cat([self._upload(prepared[i][0], dev) if not prepared[i][0].is_cuda else prepared[i][0]
                               for i in idxs], dim=0)
            elif on_cpu:
                # a pageable .to(device) blocks the host until the stream drains (13 ms/step on the m-mix batch):
                # concatenate on the host, stage through a cached pinned buffer, copy asynchronously
                X = self._upload(on_cpu[0] if len(on_cpu) == 1 else torch.cat(on_cpu, dim=0), dev)
            else:
                X = on_dev[0] if len(on_dev) == 1 else torch.cat(on_dev, dim=0)
            hp_all, wp_all = np.concatenate([prepared[i][1] for i in idxs]), np.concatenate([prepared[i][2] for i in idxs])
            rows = [hp_all, wp_all]
            if torch.is_grad_enabled() and ops.SORTED_SCATTER:
                # rows 2..5: the position indices sorted on the host with the patches they belong to -- the backward's table
                # gradients are then fixed-order segment sums instead of atomics (ops.patch_pos_add_bwd)
                rows += [*ops.sorted_pairs(hp_all), *ops.sorted_pairs(wp_all)]
            pos = np.stack(rows)
            pos = self._upload(torch.from_numpy(np.ascontiguousarray(pos.astype(np.int32))), dev)
            out.append((X, pos, list(idxs), [prepared[i][3] for i in idxs]))
        return out

    def embed_groups(self, groups, n_examples: int):
        """Device half of forward_many: one launch chain per image-shape group.  Returns one (n_i * n_h * n_w, embed_dim)
        tensor per example, in example order."""
        params = [self._flat.param_of[nm] for nm in self.used_param_names(self._prefix)]
        outs = [None] * n_examples
        for X, pos, idxs, counts in groups:
            out = _ImageEmbedFn.apply(self, X, pos[0], pos[1], (pos[2:6] if pos.shape[0] >= 6 else None), *params)
            for i, o in zip(idxs, torch.split(out, counts, dim=0)):
                outs[i] = o
        return outs

    def forward_many(self, xs):
        """The reference embeds one example's images per call (gato_policy.py:221-233), drawing the patch positions
        once per call.  Same semantics here -- positions are drawn per example, in example order, so the host RNG
        sequence is the reference's -- but examples whose images share (H, W, dtype) go through the kernels as ONE
        batch (one launch chain and one partial-gradient reduce instead of one per example).
        Returns a list of (n_i * n_h * n_w, embed_dim) tensors, one per input."""
        return self.embed_groups(self.prepare_many(xs), len(xs))
