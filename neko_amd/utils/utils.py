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
    (~7 ms each on this stack) and a pageable `.to(device)` blocks until the stream drains; here every (shape, dtype)
    gets a RING of cached pinned buffers, each guarded by an event so it is not rewritten while its previous copy is in
    flight, and the H2D copy is asynchronous on the current stream.  The ring grows (up to `max_depth`) whenever its
    oldest buffer is still in flight: with ONE buffer per key the host sat in `Event.synchronize()` until the device
    reached the previous copy of that shape -- 15 ms of a 31 ms host step at B = 64 (several uploads of one shape per
    step), i.e. it could never queue more than a step ahead; now it only waits once `max_depth` copies of one shape are
    pending."""

    def __init__(self, max_depth: int = 0):
        self._slots = {}
        self._max_depth = max_depth or int(os.environ.get("NEKO_STAGER_DEPTH", "16"))    # 1 = the old single buffer (A/B)

    def upload(self, t, dev):
        import torch
        key = (tuple(t.shape), t.dtype)
        slot = self._slots.get(key)
        if slot is None:
            slot = self._slots[key] = [[], 0]
        ring, nxt = slot
        if not ring or (ring[nxt][1] is not None and not ring[nxt][1].query() and len(ring) < self._max_depth):
            ring.insert(nxt, [torch.empty(t.shape, dtype=t.dtype).pin_memory(), None])   # takes the busy buffer's turn
        ent = ring[nxt]
        buf, ev = ent
        if ev is not None:
            ev.synchronize()
        slot[1] = (nxt + 1) % len(ring)
        buf.copy_(t)
        out = buf.to(dev, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record()
        return out
