#!/usr/bin/env python
"""Benchmark of the hot path: full training steps (pack -> fwd -> loss -> bwd -> [RCCL all-reduce] -> clip ->
AdamW) of the 768d x 6L x 24H Gato policy on fixed-shape synthetic multimodal sequences (T = 1024).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  metric = BASELINE.json's
"multimodal tokens/sec (fwd+bwd)", value = whole-job tokens/s with inputs resident on the device,
T counts every position of the padded sequence.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# kernel arguments in device memory: launches of the ~400 kernels of a step start ~0.6 us sooner each (C2 step -4 %,
# metric step -1 %); read by the HIP runtime when it initialises, so it has to be in the environment before torch loads
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, L, H, T, V_TEXT = 768, 6, 24, 1024, 50257
V = V_TEXT + 2048
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def flops_per_token_fwd(d=D, layers=L, t=T, v=V, lm_rows_frac=1.0):
    """linear layers 24 d^2, causal attention at its useful half 2 d (T+1), LM head 2 d V --
    the LM head counted only for the fraction of positions it is actually evaluated at (the build runs it on
    the loss positions only; FLOPs that are not executed are not claimed)."""
    return layers * (24 * d * d + 2 * d * (t + 1)) + 2 * d * v * lm_rows_frac


def make_batch(workload: str, B: int, seed: int, device):
    from neko_amd.tasks import synthetic as S
    if workload == "m-mix":
        return S.metric_mix_batch(B, seed, device)
    if workload == "m-text":
        return S.SyntheticTextTask(1023, V_TEXT, seed=seed, device=device).sample_batch(B)
    if workload == "c2":   # halfcheetah-shaped, T = 240
        return S.SyntheticControlTask(17, 6, 10, seed=seed, device=device).sample_batch(B)
    if workload == "c3":   # 3-task MuJoCo mix (BASELINE configs[2]): halfcheetah 10 x 24, hopper 16 x 15, walker2d 10 x 24 = 240 each
        tasks = [S.SyntheticControlTask(17, 6, 10, seed=seed, device=device), S.SyntheticControlTask(11, 3, 16, seed=seed + 1, device=device),
                 S.SyntheticControlTask(17, 6, 10, seed=seed + 2, device=device)]
        return [tasks[i % 3].sample_batch(1)[0] for i in range(B)]
    if workload == "c4":   # Atari Breakout-shaped (BASELINE configs[3]): 13 x (36 patches + SEP + 1 action) = 494
        return S.SyntheticAtariTask(13, 96, 96, seed=seed, device=device).sample_batch(B)
    if workload == "c5-mix":   # text 1024 / Atari 494 / caption 289 / halfcheetah 240 in one batch (ragged lengths)
        return S.ragged_mix_batch(B, seed, device)
    raise ValueError(workload)


def cpu_baseline(seconds_budget: float = 30.0):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference path, pinned to reference fixtures)
    timed on this box's host cores on a bounded sample of the same workload: ONE text sequence of
    T=1024 (1023 ids + SEP) through the same 768d x 6L x 24H model, forward + backward."""
    from oracle import neko_oracle as O
    cores = min(os.cpu_count() or 1, 32)      # more intra-op threads than this only oversubscribe the small GEMMs
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(embed_dim=D, layers=L, heads=H, text_tokens=V_TEXT, context_len=T)
    sd = O.init_state_dict(cfg, 0)
    g = torch.Generator().manual_seed(1234)
    batch = [{"text": torch.randint(0, V_TEXT, (T - 1,), generator=g).tolist()} for _ in range(2)]
    t0 = time.time()
    n = 0
    while True:
        O.loss_and_grads(sd, cfg, batch)
        n += 1
        el = time.time() - t0
        if n >= 2 or el > seconds_budget * 0.5:
            break
    return {"value": n * 2 * T / el, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} x (B=2, T=1024 text) fwd+bwd of the 768d/6L/24H model, fp32, torch CPU threads={cores} "
                      f"of {os.cpu_count()} hardware threads"}


def time_dominant_kernel(model, rows: int, iters: int = 10):
    """The LM-head logits GEMM (rows x 768 @ 768 x 52305, bf16 MFMA) is the largest single kernel of the step.
    Timed with HIP events on the stream it is launched on (torch's current stream)."""
    from neko_amd import ops
    hp = model._head_params()
    a = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
    out = torch.empty(rows, hp.Vpad, dtype=torch.bfloat16, device="cuda")    # same call as engine.lm_head_loss
    for _ in range(2):
        ops.gemm(a, hp.w, rows, hp.Vpad, D, ldb=D, out_bf16=out, ldcb=hp.Vpad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(a, hp.w, rows, hp.Vpad, D, ldb=D, out_bf16=out, ldcb=hp.Vpad)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * rows * D * hp.V          # useful columns only (the call computes Vpad = V + 47 of them)
    return {"kernel": "gemm_glds_kernel<A k-contig, B k-contig> (LM head logits)", "shape": [rows, hp.V, D],
            "ms": ms, "tflops": flops / ms / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU per step")
    ap.add_argument("--workload", default="m-mix", choices=["m-mix", "m-text", "c2", "c3", "c4", "c5-mix"])
    ap.add_argument("--model", default="768d", choices=["768d", "gato-1.2b"],
                    help="768d = the metric's 768d x 6L x 24H (hd=32); gato-1.2b = BASELINE configs[4], 2048d x 24L x 16H "
                         "(hd=128, streaming attention kernels) -- a side measurement, not the metric's config")
    ap.add_argument("--ragged-groups", type=int, default=0,
                    help="> 0: length-bucketed layout (GatoPolicy.ragged_groups) instead of padding to the longest example")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.1, help="attention/residual dropout (reference default 0.1)")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd(+all-reduce) only")
    args = ap.parse_args()
    global D, L, H
    if args.model == "gato-1.2b":
        D, L, H = 2048, 24, 16
        args.no_cpu_baseline = True        # the fp32 CPU oracle of a 1.4 B-parameter model does not fit the time budget

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # NEKO_BENCH_BACKEND=gloo + NEKO_BENCH_ONE_DEVICE=1: dry run of the multi-rank flow on a 1-GPU box (every rank on
    # cuda:0, gradients reduced through the host) -- exercises the DP hooks, barriers and max-over-ranks timing, not
    # a performance configuration
    backend = os.environ.get("NEKO_BENCH_BACKEND", "nccl")
    if os.environ.get("NEKO_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    from neko_amd.dp import GradReducer
    from neko_amd.policy.gato_policy import GatoPolicy
    from neko_amd.training.optim import NekoAdamW

    # the host side of a step is a few hundred tiny tensor ops: with torch's default of one intra-op thread per
    # hardware thread (256 here) every small CPU op pays a fork/join of milliseconds
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    torch.manual_seed(0)
    dropout = args.dropout   # reference default 0.1 (arguments.py:69); embd_pdrop is 0.1 regardless
    model = GatoPolicy(dev, D, L, H, dropout, resid_mid_channels=128, context_len=T, text_tokenizer=V_TEXT)
    if dropout == 0:
        model.transformer.drop.p = 0.0
    model.ragged_groups = args.ragged_groups
    model.train()
    opt = NekoAdamW(model, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    dp = None
    if world > 1:
        dp = GradReducer(model._flat)
        dp.broadcast_parameters()
        dp.attach(model, opt)
        if args.workload in ("c2", "c3", "c4"):      # control-only workloads: no rank ever touches the text rows (see dp.py)
            dp.declare_unused_rows("embed_token.weight", 0, model.text_tokens)
            dp.no_text_declared = True

    B = args.batch
    batches = [make_batch(args.workload, B, 1234 + rank + 100 * i, dev) for i in range(2)]
    Tlen = {"c2": 240, "c3": 240, "c4": 494}.get(args.workload, T)

    def step(i):
        _, loss = model.forward(inputs=batches[i % len(batches)], compute_loss=True, return_logits=False)
        loss.backward()
        if dp is not None:
            dp.flush()
            dp.finish()
        if not args.no_optimizer:
            opt.clip_grad_norm_(1.0)
            opt.step()
        opt.zero_grad()
        return loss

    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t)

    # the reference's metric is forward+backward (+ all-reduce); `value` above includes clip + AdamW, the
    # fwd+bwd(+all-reduce)-only rate is timed separately over the same number of steps and reported next to it
    el_fb = None
    if not args.no_optimizer:
        def step_fb(i):
            _, l = model.forward(inputs=batches[i % len(batches)], compute_loss=True, return_logits=False)
            l.backward()
            if dp is not None:
                dp.flush()
                dp.finish()
            opt.zero_grad()
        nfb = max(1, min(args.steps, 20))
        step_fb(0)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nfb):
            step_fb(i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        el_fb = (time.perf_counter() - t1) / nfb
        if world > 1:
            t = torch.tensor([el_fb], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el_fb = float(t)

    if rank == 0:
        tokens = world * B * Tlen * args.steps
        value = tokens / el
        lm_frac = (model._loss_rows[3] / float(B * Tlen)) if (model._loss_rows and model.lm_head_selected_rows) else 1.0
        fpt = 3 * flops_per_token_fwd(d=D, layers=L, t=Tlen, lm_rows_frac=lm_frac)
        dom = time_dominant_kernel(model, min(4096, B * Tlen))
        # HBM bytes per launch of that kernel: PMC counters cannot be collected from inside the timed process, so the
        # number is the committed rocprofv3 --pmc measurement of the same call (tools/pmc_lmhead.sh), null if absent
        traffic, traffic_src = None, None
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_lmhead_traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            if tj.get("shape_MNK") == dom["shape"]:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/r01_lmhead_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, calibrated)"
        # positions that are not padding (the metric counts every position of the padded (B, T) batch; for the
        # ragged c5-mix workload the useful rate is the one over real tokens) and rows that went through the stack
        from neko_amd.policy.gato_policy import K_PAD, build_layout
        lay = build_layout(batches[0], True, T, False, ragged_groups=args.ragged_groups)
        real_tokens = int((lay.desc[:, 0] != K_PAD).sum())
        rows = int(lay.desc.shape[0])
        # MFMA-pipe utilisation of the same kernel from the committed SQ counter pass (tools/pmc_gemm.sh): context for
        # `frac`, which is priced against the nominal 2.5 PFLOP/s at 2.4 GHz while the chip sustains ~1.5-1.7 GHz here
        mfma_busy = None
        cp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_step19_gemm_counters.txt")
        if os.path.exists(cp) and D == 768:
            import re
            mm = re.search(r"lmlogit16: .*?MFMA pipe busy ([0-9.]+) %", open(cp).read())
            mfma_busy = float(mm.group(1)) / 100 if mm else None
        out = {
            "metric": "multimodal tokens/sec (fwd+bwd+optimizer, whole job)", "value": value, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {D}d x {L}L x {H}H (hd={D // H}), V=52305, T={Tlen}, "
                                   f"{B} sequences/GPU/step, dropout {dropout}, "
                                   f"{'fwd+bwd only' if args.no_optimizer else 'fwd+bwd+clip+AdamW'}",
                       "global_batch": world * B, "seq_len": Tlen, "parallelism": f"dp{world}"},
            "tokens_per_sec_per_gpu": value / world,
            "fwd_bwd_only_tokens_per_sec": (world * B * Tlen / el_fb) if el_fb else None,
            "final_loss": float(loss.detach()),
            "step_mfma_frac": value / world * fpt / (MFMA_PEAK_TFLOPS * 1e12),
            "flops_per_token_fwd_bwd": fpt,
            "lm_head_rows_fraction": lm_frac,
            "ragged_groups": args.ragged_groups, "rows_per_step_per_gpu": rows, "real_tokens_per_step_per_gpu": real_tokens,
            "real_tokens_per_sec": world * real_tokens * args.steps / el,
            "roofline": {"bound": "mfma", "achieved": dom["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": dom["tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "kernel": dom["kernel"],
                         "shape_MNK": dom["shape"], "ms_per_launch": dom["ms"],
                         "mfma_pipe_busy_pmc": mfma_busy,
                         "mfma_pipe_busy_source": "profiles/r01_step19_gemm_counters.txt (SQ_VALU_MFMA_BUSY_CYCLES)" if mfma_busy else None},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
