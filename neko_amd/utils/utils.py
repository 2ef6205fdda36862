"""save_model -- same on-disk format as gato/utils/utils.py:19-32: args.json once + <name>.pt = state_dict."""
import dataclasses
import json
import os

import torch


class DotDict(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def save_model(model, save_dir, save_name, config_args):
    os.makedirs(save_dir, exist_ok=True)
    args_path = os.path.join(save_dir, "args.json")
    if not os.path.exists(args_path):
        with open(args_path, "w") as f:
            cfg = dataclasses.asdict(config_args) if dataclasses.is_dataclass(config_args) else dict(vars(config_args))
            json.dump(cfg, f)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(save_dir, save_name + ".pt"))
