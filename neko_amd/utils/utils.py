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
    gets ONE cached pinned buffer, guarded by an event so it is not rewritten while its previous copy is in flight,
    and the H2D copy is asynchronous on the current stream."""

    def __init__(self):
        self._slots = {}

    def upload(self, t, dev):
        import torch
        key = (tuple(t.shape), t.dtype)
        slot = self._slots.get(key)
        if slot is None:
            slot = self._slots[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory(), None]
        buf, ev = slot
        if ev is not None:
            ev.synchronize()
        buf.copy_(t)
        out = buf.to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out
