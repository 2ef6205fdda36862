"""TrainingArgs -- the flag names and defaults of gato/training/arguments.py:20-138 that concern the hot path
(the flag names are API surface).  Dataset / environment / W&B / LoRA flags of the reference
are accepted and ignored by the synthetic-data `train.py` (their subsystems are out of scope).
A small argparse front-end is generated from the dataclass (the reference's 423-line HF-style parser is not
re-implemented)."""
from __future__ import annotations

import argparse
import dataclasses
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class TrainingArgs:
    cpu: bool = False
    device: str = "cuda"
    mixed_precision: str = "bf16"          # the HIP path is always bf16-operand / fp32-accumulate

    # input & tokenization
    sequence_length: int = 1024
    patch_size: int = 16
    resid_mid_channels: int = 128
    num_groups: int = 32
    patch_position_vocab_size: int = 128
    disable_patch_pos_encoding: bool = False
    disable_inner_pos_encoding: bool = False
    mu: int = 100
    M: int = 256
    continuous_tokens: int = 1024
    discrete_tokens: int = 1024

    # architecture
    tokenizer_model_name: str = "gpt2"
    pretrained_lm: Optional[str] = None
    flash: bool = False
    init_checkpoint: Optional[str] = None
    embed_dim: int = 768
    layers: int = 8
    heads: int = 24
    activation_fn: str = "gelu"
    # not in the reference: > 0 packs every training batch into at most this many length buckets instead of
    # left-padding all examples to the longest one (same loss and gradients, fewer padded positions)
    ragged_groups: int = 0

    # training
    text_prop: float = 0.0
    caption_prop: float = 0.0
    vqa_prop: float = 0.0
    gradient_accumulation_steps: int = 1
    batch_size: int = 512
    dropout: float = 0.1
    beta_1: float = 0.9
    beta_2: float = 0.95
    adam_eps: float = 1e-8
    weight_decay: float = 0.1
    grad_norm_clip: float = 1.0
    disable_grad_clip: bool = False
    warmup_steps: int = 15000
    init_lr: float = 1e-7
    learning_rate: float = 1e-4
    min_factor: float = 10.0
    disable_cosine_decay: bool = False
    training_steps: int = 1_000_000
    log_eval_freq: int = 100_000
    pad_seq: bool = False

    # evaluation (kept for CLI compatibility; rollouts are out of scope)
    eval_episodes: int = 10
    eval_mode: str = "deterministic"
    promptless_eval: bool = False

    # control-task prompting (arguments.py:126-129)
    prompt_ep_proportion: float = 0.25
    prompt_len_proportion: float = 0.5
    unique_prompt_episodes: bool = False
    top_k: Optional[int] = None

    # datasets (synthetic stand-ins: names select the synthetic generator shapes)
    control_datasets: List[str] = field(default_factory=list)
    text_datasets: List[str] = field(default_factory=list)
    text_datasets_paths: List[str] = field(default_factory=list)
    caption_dataset: str = ""
    # the remaining dataset / LoRA / evaluation flags of arguments.py:58-61,95-123 are accepted so that existing command
    # lines parse; the readers behind them (webdataset tar shards, VQA json, HF hub) are outside this build
    caption_train_data: List[str] = field(default_factory=list)
    caption_test_data: List[str] = field(default_factory=list)
    test_data_prop: float = 0.1
    vqa_dataset: str = ""
    vqa_train_data: List[str] = field(default_factory=list)
    vqa_test_data: List[str] = field(default_factory=list)
    train_img_name_prefix: List[str] = field(default_factory=list)
    train_img_file_name_len: List[str] = field(default_factory=list)
    test_img_name_prefix: List[str] = field(default_factory=list)
    test_img_file_name_len: List[str] = field(default_factory=list)
    questions_file: str = "questions.json"
    annotations_file: str = "annotations.json"
    eval_text_num_examples: int = 100
    eval_text_log_examples: bool = False
    eval_caption_num_examples: int = 100
    eval_caption_log_examples: bool = False
    eval_vqa_num_examples: int = 100
    eval_vqa_log_examples: bool = False
    lora: bool = False                     # needs --pretrained_lm (train.py:109-112): rejected with it
    lora_r: int = 8
    lora_alpha: int = 32
    lora_dropout: float = 0.1

    # logging / saving
    use_wandb: bool = False
    wandb_project: str = "gato-control"
    save_model: bool = False
    save_mode: str = "last"
    save_dir: str = "models"

    # neko_amd extras
    capture_step: bool = False             # one replayed HIP graph per batch structure (training/captured.py; single rank)
    text_vocab_size: int = 50257           # used when the gpt2 tokenizer cannot be downloaded
    seed: int = 1234


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="neko_amd train.py (flag names of ManifoldRG/NEKO's train.py)")
    for f in dataclasses.fields(TrainingArgs):
        name = "--" + f.name
        default = f.default if f.default is not dataclasses.MISSING else f.default_factory()
        if f.type in ("bool", bool):
            p.add_argument(name, action="store_true", default=default)
        elif str(f.type).startswith("List"):
            p.add_argument(name, nargs="+", default=default)
        elif "Optional" in str(f.type):      # Optional[int] -> int, Optional[str] -> str (a bare default=None would leave '5' a string)
            p.add_argument(name, type=int if "int" in str(f.type) else str, default=default)
        else:
            p.add_argument(name, type=type(default), default=default)
    return p


def parse_args(argv=None) -> TrainingArgs:
    ns = build_parser().parse_args(argv)
    args = TrainingArgs(**vars(ns))
    check_supported(args)
    return args


def check_supported(args: TrainingArgs) -> None:
    """The HIP path computes on the MI355X with bf16 MFMA operands and fp32 accumulation, nothing else: flags that ask
    for another device or precision (reference: `--cpu` train.py:41, `--mixed_precision no|fp16` arguments.py:22) are
    refused loudly instead of being accepted and ignored."""
    if args.cpu or str(args.device).startswith("cpu"):
        raise SystemExit("neko_amd: --cpu / --device cpu is not supported -- the HIP path has no CPU fallback "
                         "(the reference's CPU run is what oracle/ restates for the parity tests)")
    if args.mixed_precision != "bf16":
        raise SystemExit(f"neko_amd: --mixed_precision {args.mixed_precision} is not supported -- the kernels use bf16 MFMA "
                         "operands with fp32 accumulation, statistics, residual stream, gradients and optimiser state "
                         "(= the reference's `--mixed_precision bf16`); pass --mixed_precision bf16 or drop the flag")
