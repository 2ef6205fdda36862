#!/usr/bin/env python
"""train.py -- same flags and wiring as ManifoldRG/NEKO's train.py:25-173, on the neko_amd HIP path with
synthetic tasks (the reference's Minari / HF-datasets loaders are out of scope and not installable here).

    python train.py --embed_dim 768 --layers 6 --heads 24 --batch_size 32 --sequence_length 1024 \
        --training_steps 100 --log_eval_freq 50 --warmup_steps 10 --dropout 0 --text_prop 0.5 --caption_prop 0.25
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...      # data parallel
"""
from __future__ import annotations

import os
import sys
from datetime import datetime

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # see bench.py: before the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL: read when the runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")           # compute + side + communication streams on distinct hardware queues (bench.py)

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from neko_amd.dp import GradReducer  # noqa: E402
from neko_amd.policy.gato_policy import GatoPolicy  # noqa: E402
from neko_amd.tasks import synthetic as S  # noqa: E402
from neko_amd.training.arguments import TrainingArgs, parse_args  # noqa: E402
from neko_amd.training.optim import NekoAdamW  # noqa: E402
from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler  # noqa: E402
from neko_amd.training.trainer import Trainer  # noqa: E402


def main(args: TrainingArgs):
    import os as _os
    torch.set_num_threads(min(8, _os.cpu_count() or 1))   # host side = many tiny tensor ops (see bench.py)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)
    args.device = str(dev)
    exp_name = f"neko-gato_{datetime.now().strftime('%y-%m-%d_%H-%M-%S')}"

    seed = args.seed + rank            # every rank draws its own batches (reference semantics)
    ts = args.sequence_length
    tasks = []
    names = args.control_datasets or ["halfcheetah"]
    for i, n in enumerate(names):
        if n.endswith(".npz"):
            # real episodes: an EpisodeStore file (neko_amd.tasks.control_task; stands in for a Minari dataset) sampled
            # with the reference's window / prompt sampler (control_task.py:178-325).  All control tasks must then be
            # episode files: the Trainer switches to Trainer.sample_control_batch semantics (trainer.py:211-250)
            from neko_amd.tasks.control_task import ControlTask, EpisodeStore, SpacesOnlyEnv
            store = EpisodeStore.from_npz(n, seed=seed + i)
            tasks.append(ControlTask(os.path.basename(n)[:-4], SpacesOnlyEnv(*store.spaces()), store, ts, args,
                                     training_prompt_len_proportion=args.prompt_len_proportion,
                                     share_prompt_episodes=not args.unique_prompt_episodes, top_k_prompting=args.top_k,
                                     host_batches=True))
            continue
        if "breakout" in n.lower() or "atari" in n.lower():
            tasks.append(S.SyntheticAtariTask(max(1, ts // 38), 96, 96, name=n, seed=seed + i, device=dev))
        elif "hopper" in n.lower():
            tasks.append(S.SyntheticControlTask(11, 3, max(1, ts // 15), name=n, seed=seed + i, device=dev))
        else:
            tasks.append(S.SyntheticControlTask(17, 6, max(1, ts // 24), name=n, seed=seed + i, device=dev))
    if args.text_prop > 0:
        tasks.append(S.SyntheticTextTask(ts - 1, args.text_vocab_size, seed=seed + 50, device=dev))
    if args.caption_prop > 0:
        tasks.append(S.SyntheticCaptionTask(ts - 257, args.text_vocab_size, seed=seed + 60, device=dev))
    assert args.vqa_prop == 0, "no synthetic VQA task (same dict format as caption)"

    if args.lora:
        raise SystemExit("--lora needs --pretrained_lm (train.py:109-112): downloaded weights are outside this build")
    try:
        tok = None if args.text_vocab_size <= 0 else args.text_vocab_size
        model = GatoPolicy(device=dev, embed_dim=args.embed_dim, layers=args.layers, heads=args.heads,
                           dropout=args.dropout, mu=args.mu, M=args.M, patch_size=args.patch_size,
                           resid_mid_channels=args.resid_mid_channels, continuous_tokens=args.continuous_tokens,
                           discrete_tokens=args.discrete_tokens, context_len=args.sequence_length,
                           use_patch_pos_encoding=not args.disable_patch_pos_encoding,
                           use_pos_encoding=not args.disable_inner_pos_encoding, activation_fn=args.activation_fn,
                           pretrained_lm=args.pretrained_lm, flash=args.flash,
                           tokenizer_model_name=args.tokenizer_model_name, pad_seq=args.pad_seq, text_tokenizer=tok)
    except NotImplementedError as e:
        raise SystemExit(f"unsupported configuration on the HIP path: {e}")
    if args.dropout == 0:
        model.transformer.drop.p = 0.0     # the reference keeps embd_pdrop = 0.1 regardless of --dropout
    if args.ragged_groups > 0:
        model.ragged_groups = args.ragged_groups
    if args.init_checkpoint is not None:
        model.load_state_dict(torch.load(args.init_checkpoint, map_location=dev))
    params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    if rank == 0:
        print("Trainable Parameters:", "{}M".format(params / 1e6))

    optimizer = NekoAdamW(model, lr=args.learning_rate, betas=(args.beta_1, args.beta_2), eps=args.adam_eps,
                          weight_decay=args.weight_decay)
    scheduler = get_linear_warmup_cosine_decay_scheduler(
        optimizer, args.warmup_steps, args.training_steps, base_lr=args.learning_rate, init_lr=args.init_lr,
        min_lr=args.learning_rate / args.min_factor, cosine_decay=not args.disable_cosine_decay)
    if args.init_checkpoint is not None:        # true resume when the optimiser file sits next to the checkpoint
        opt_path = args.init_checkpoint[:-3] + ".opt.pt" if args.init_checkpoint.endswith(".pt") else None
        if opt_path and os.path.exists(opt_path):
            st = torch.load(opt_path, map_location="cpu", weights_only=False)
            optimizer.load_state_dict(st["optimizer"])
            if st.get("scheduler") is not None:
                scheduler.load_state_dict(st["scheduler"])
            if rank == 0:
                print("resumed optimizer / scheduler state from", opt_path)
    dp = None
    if world > 1:
        dp = GradReducer(model._flat)
        dp.broadcast_parameters()
        dp.attach(model, optimizer)
        if args.text_prop == 0 and args.caption_prop == 0 and args.vqa_prop == 0:
            # control-only run (same arguments on every rank): the text rows of the token embedding never see a
            # gradient, so the one reduction that cannot hide behind backward shrinks from 160 MB to 6 MB at 768d
            dp.declare_unused_rows("embed_token.weight", 0, model.text_tokens)
            dp.no_text_declared = True
    trainer = Trainer(model=model, optimizer=optimizer, accelerator=None, scheduler=scheduler, tasks=tasks,
                      exp_name=exp_name, args=args, dp=dp)
    trainer.train()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    a = parse_args()
    assert a.training_steps % a.log_eval_freq == 0, "training_steps must be divisible by eval_freq"
    assert a.training_steps > a.warmup_steps, "training_steps must be greater than warmup_steps"
    assert a.learning_rate > a.init_lr, "learning_rate must be greater than init_lr"
    main(a)
