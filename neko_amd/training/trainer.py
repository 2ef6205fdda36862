"""Trainer -- mirror of gato/training/trainer.py:14-247 for the HIP path.

``train_step`` keeps the reference's step semantics (trainer.py:127-188): mix the batch by
text/caption/vqa/control proportions, forward (``model.forward(inputs=..., compute_loss=True)``),
backward, clip to ``grad_norm_clip``, optimiser step, scheduler step, zero_grad.  Differences, all
MI355X-motivated: no HF Accelerate (the data-parallel reduction is neko_amd.dp.GradReducer over
RCCL), the (B,T,V) logits the reference discards are not materialised, and ``loss.item()`` (a host
sync every step, trainer.py:188) is deferred: losses stay on the device until the logs are built.
Evaluation (env rollouts) is out of scope; tasks that define ``evaluate`` are called.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch


class Trainer:
    def __init__(self, model, optimizer, accelerator, scheduler, tasks, exp_name, args, dp=None):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.accelerator = accelerator      # unused (kept for signature parity); may be None
        self.dp = dp
        self.tasks = tasks
        self.args = args
        self.print_logs = True
        self.device = torch.device(args.device)
        self.min_lr = self.args.learning_rate / self.args.min_factor
        self.deterministic = self.args.eval_mode == "deterministic"
        self.exp_name = exp_name
        self.exp_dir = os.path.join(self.args.save_dir, self.exp_name)
        self.steps = 0
        # gradient accumulation (train.py:37, trainer.py:176 `accelerator.accumulate`): every call of train_step is one
        # micro-batch whose loss is scaled by 1/k in backward; clip / optimiser / scheduler / zero_grad -- and the
        # data-parallel all-reduce -- only run on every k-th call
        self.accum_steps = max(1, int(getattr(args, "gradient_accumulation_steps", 1) or 1))
        self.captured = None
        if getattr(args, "capture_step", False):
            if dp is not None or self.accum_steps > 1:
                raise SystemExit("--capture_step: single rank, no gradient accumulation (the captured graph holds one whole step)")
            from .captured import CapturedTrainStep
            self.captured = CapturedTrainStep(model, optimizer, scheduler,
                                              None if args.disable_grad_clip else args.grad_norm_clip)
        self._micro = 0
        self.start_time = None
        self.is_main = (not torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0

    def train(self):
        self.start_time = time.time()
        iters = self.args.training_steps // self.args.log_eval_freq
        logs = {}
        for i in range(iters):
            logs = self.train_iteration(self.args.log_eval_freq, i)
        if self.captured is not None:      # back to plain eager state (dropout salt unregistered, lr from the param group)
            self.captured.close()
            self.captured = None
        if self.args.save_model and self.args.save_mode == "last" and self.is_main:
            from ..utils.utils import save_model
            save_model(self.model, self.exp_dir, f"checkpoint_{self.steps}", self.args, self.optimizer, self.scheduler)
        return logs

    def train_iteration(self, num_steps, iter):
        logs = {}
        train_start = time.time()
        losses = []
        self.model.train()
        step_logs = {}
        for _ in range(num_steps):
            self.steps += 1
            loss, step_logs = self.train_step()
            losses.append(loss)
        logs.update(step_logs)
        train_losses = torch.stack(losses).float().cpu().numpy()     # the only host sync of the iteration
        logs["time/training"] = time.time() - train_start
        eval_start = time.time()
        self.model.eval()
        with torch.no_grad():
            for task in self.tasks:
                if hasattr(task, "evaluate") and getattr(getattr(task, "env", None), "can_rollout", True):
                    if hasattr(task, "sample_batch_configurable"):       # rollouts (trainer.py:100-107 arguments)
                        res = task.evaluate(self.model, n_iterations=self.args.eval_episodes,
                                            deterministic=self.args.eval_mode == "deterministic",
                                            promptless_eval=bool(self.args.promptless_eval))
                    elif getattr(task, "kind", "") == "text" and hasattr(task, "text_dataset"):      # trainer.py:108-112
                        res = task.evaluate(self.model, num_examples_to_test=self.args.eval_text_num_examples,
                                            deterministic=self.args.eval_mode == "deterministic",
                                            log_examples_to_output=self.args.eval_text_log_examples)
                    elif getattr(task, "kind", "") in ("caption", "vqa") and hasattr(task, "dataset"):   # trainer.py:113-122
                        k = task.kind
                        res = task.evaluate(self.model, num_examples_to_test=getattr(self.args, f"eval_{k}_num_examples"),
                                            deterministic=self.args.eval_mode == "deterministic",
                                            log_examples_to_output=getattr(self.args, f"eval_{k}_log_examples"))
                    else:
                        res = task.evaluate(self.model)
                    for k, v in res.items():
                        logs[f"evaluation/{task.name}/{k}"] = v
        logs["time/total"] = time.time() - self.start_time
        logs["time/evaluation"] = time.time() - eval_start
        logs["training/train_loss_mean"] = float(np.mean(train_losses))
        logs["training/train_loss_std"] = float(np.std(train_losses))
        if self.is_main and self.print_logs:
            print("=" * 80)
            print(f"Iteration {iter}")
            for k, v in logs.items():
                print(f"{k}: {v}")
            print("=" * 80)
        if self.args.save_model and self.args.save_mode == "checkpoint" and self.is_main:
            from ..utils.utils import save_model
            save_model(self.model, self.exp_dir, f"checkpoint_{self.steps}", self.args, self.optimizer, self.scheduler)
        return logs

    def sample_batch(self):
        """Batch mix by proportions (trainer.py:134-172)."""
        a = self.args
        control_prop = 1 - a.text_prop - a.caption_prop - a.vqa_prop
        sizes = [int(a.text_prop * a.batch_size), int(a.caption_prop * a.batch_size), int(a.vqa_prop * a.batch_size),
                 int(control_prop * a.batch_size)]
        remainder = a.batch_size - sum(sizes)
        if remainder > 0:
            residuals = [a.text_prop * a.batch_size - sizes[0], a.caption_prop * a.batch_size - sizes[1],
                         a.vqa_prop * a.batch_size - sizes[2], control_prop * a.batch_size - sizes[3]]
            idx = torch.multinomial(torch.tensor(residuals), num_samples=1).item()
            sizes[idx] += remainder
        kinds = ("text", "caption", "vqa", "control")
        dicts = []
        for kind, n in zip(kinds, sizes):
            if n <= 0:
                continue
            tasks = [t for t in self.tasks if getattr(t, "kind", "control") == kind]
            if kind == "control" and tasks and all(hasattr(t, "sample_batch_configurable") for t in tasks):
                # episode-store tasks (neko_amd.tasks.control_task.ControlTask): the reference's prompted sampling
                from ..tasks.control_task import sample_control_batch
                dicts.extend(sample_control_batch(tasks, n, a.prompt_ep_proportion, self.model.device, a.sequence_length))
            elif kind == "control":
                # round-robin over control tasks, like trainer.py:223-228 without prompting (synthetic data)
                for j in range(n):
                    dicts.extend(tasks[j % len(tasks)].sample_batch(1))
            else:
                for t in tasks:
                    dicts.extend(t.sample_batch(n))
        return dicts

    def train_step(self):
        logs = {"training/learning_rate": self.scheduler.get_last_lr()[0]}
        t0 = time.time()
        batch = self.sample_batch()
        logs["time/sample_batch"] = time.time() - t0
        if self.captured is not None:          # forward, backward, clip, AdamW, scheduler step, zero_grad: one graph replay
            loss, _ = self.captured.step(batch)
            return loss, logs
        _, loss = self.model.forward(inputs=batch, compute_loss=True, return_logits=False)
        self._micro += 1
        sync = self._micro % self.accum_steps == 0
        if self.dp is not None:
            self.dp.sync = sync                   # no gradient all-reduce on the accumulating micro-steps (DDP no_sync)
        (loss / self.accum_steps if self.accum_steps > 1 else loss).backward()
        if sync:
            if self.dp is not None:
                self.dp.flush()
                self.dp.finish()
            if not self.args.disable_grad_clip:
                self.optimizer.clip_grad_norm_(self.args.grad_norm_clip)
            self.optimizer.step()
            self.scheduler.step()
            self.optimizer.zero_grad()
        return loss.detach(), logs
