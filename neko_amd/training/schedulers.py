"""LR schedule -- mirror of gato/training/schedulers.py:8-32 (linear warm-up then cosine decay, returned
as a ratio to base_lr and wrapped in torch's LambdaLR)."""
from functools import partial

import numpy as np
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


def get_linear_warmup_cosine_decay_scheduler(optimizer: Optimizer, num_warmup_steps: int, num_training_steps: int,
                                             base_lr: float, init_lr: float, min_lr: float,
                                             cosine_decay: bool = True, last_epoch=-1):
    lr_lambda = partial(_linear_warmup_cosine_decay, num_warmup_steps=num_warmup_steps,
                        num_training_steps=num_training_steps, base_lr=base_lr, init_lr=init_lr, min_lr=min_lr,
                        cosine_decay=cosine_decay)
    return LambdaLR(optimizer, lr_lambda, last_epoch)


def _linear_warmup_cosine_decay(current_step: int, *, num_warmup_steps: int, num_training_steps: int, base_lr: float,
                                init_lr: float, min_lr: float, cosine_decay: bool):
    if current_step <= num_warmup_steps:
        lr = init_lr + (base_lr - init_lr) * current_step / num_warmup_steps
    elif cosine_decay:
        progress = (current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        lr = min_lr + 0.5 * (base_lr - min_lr) * (1 + np.cos(np.pi * progress))
    else:
        lr = base_lr
    return lr / base_lr
