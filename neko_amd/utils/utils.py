"""save_model -- same on-disk format as gato/utils/utils.py:19-32: args.json once + <name>.pt = state_dict."""
import dataclasses
import json
import os

import torch


class DotDict(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def save_model(model, save_dir, save_name, config_args, optimizer=None, scheduler=None):
    """gato/utils/utils.py:19-32: args.json + <save_name>.pt (the model state_dict, reference keys).  With `optimizer`
    (and `scheduler`) a second file <save_name>.opt.pt holds what a true resume needs -- the reference never saves it."""
    os.makedirs(save_dir, exist_ok=True)
    args_path = os.path.join(save_dir, "args.json")
    if not os.path.exists(args_path):
        with open(args_path, "w") as f:
            cfg = dataclasses.asdict(config_args) if dataclasses.is_dataclass(config_args) else dict(vars(config_args))
            json.dump(cfg, f)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(save_dir, save_name + ".pt"))
    if optimizer is not None:
        torch.save({"optimizer": optimizer.state_dict(),
                    "scheduler": scheduler.state_dict() if scheduler is not None else None},
                   os.path.join(save_dir, save_name + ".opt.pt"))


class HostStager:
    """CPU tensor -> device without blocking the host.  `tensor.pin_memory()` allocates pinned memory on every call
    (~7 ms each on this stack) and a pageable `.to(device)` blocks until the stream drains; here uploads go through RINGS
    of cached pinned buffers, each buffer guarded by an event so it is not rewritten while its previous copy is in flight,
    and the H2D copy is asynchronous on the current stream.  A ring grows (up to `max_depth`) whenever its oldest buffer is
    still in flight: with ONE buffer the host sat in `Event.synchronize()` until the device reached the previous copy --
    15 ms of a 31 ms host step at B = 64.

    Rings are keyed by (dtype, capacity rounded up to a power of two), not by the exact shape: real variable-length batches
    produce value / index buffers of a different shape on almost every step, and a ring per shape pinned up to 16 fresh
    buffers for each of them.  The upload is a slice of the ring buffer viewed in the tensor's shape.  The total
    of pinned bytes is capped (`max_bytes`, NEKO_STAGER_MAX_MB, default 1 GiB): beyond it the least recently used idle ring
    is dropped."""

    def __init__(self, max_depth: int = 0, max_bytes: int = 0):
        self._slots = {}          # (dtype, capacity) -> [ring of [pinned flat buffer, event], next index, last use tick]
        self._max_depth = max_depth or int(os.environ.get("NEKO_STAGER_DEPTH", "16"))    # 1 = the old single buffer (A/B)
        self._max_bytes = max_bytes or int(os.environ.get("NEKO_STAGER_MAX_MB", "1024")) << 20
        self._bytes = 0
        self._tick = 0

    @staticmethod
    def _capacity(n: int) -> int:
        return 1 if n <= 1 else 1 << (n - 1).bit_length()

    def _evict(self, need: int, keep) -> None:
        """Drop idle rings (every copy retired), least recently used first, until `need` more bytes fit."""
        while self._bytes + need > self._max_bytes:
            idle = [(slot[2], key) for key, slot in self._slots.items()
                    if key != keep and all(ev is None or ev.query() for _, ev in slot[0])]
            if not idle:
                return                         # everything else is in flight: exceed the cap rather than stall
            _, key = min(idle)
            ring = self._slots.pop(key)[0]
            self._bytes -= sum(b.numel() * b.element_size() for b, _ in ring)

    def upload(self, t, dev):
        import torch
        n = t.numel()
        key = (t.dtype, self._capacity(n))
        slot = self._slots.get(key)
        if slot is None:
            slot = self._slots[key] = [[], 0, 0]
        ring, nxt = slot[0], slot[1]
        self._tick += 1
        slot[2] = self._tick
        if not ring or (ring[nxt][1] is not None and not ring[nxt][1].query() and len(ring) < self._max_depth):
            nbytes = key[1] * t.element_size()
            self._evict(nbytes, key)
            ring.insert(nxt, [torch.empty(key[1], dtype=t.dtype).pin_memory(), None])   # takes the busy buffer's turn
            self._bytes += nbytes
        ent = ring[nxt]
        buf, ev = ent
        if ev is not None:
            ev.synchronize()
        slot[1] = (nxt + 1) % len(ring)
        view = buf[:n].view(t.shape)
        view.copy_(t)
        out = view.to(dev, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record()
        return out
