#!/usr/bin/env python
"""Benchmark of the hot path: full training steps (pack -> fwd -> loss -> bwd -> [RCCL all-reduce] -> clip ->
AdamW) of the 768d x 6L x 24H Gato policy on fixed-shape synthetic multimodal sequences (T = 1024).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...            # N > 1 without a launcher: spawns N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  metric = BASELINE.json's metric, value = whole-job
tokens/s with inputs resident on the device, T counts every position of the padded sequence.
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
# dmabuf IPC for RCCL / cross-process device memory: also read once, when the runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP streams share this many hardware queues (default 4), and two streams on one queue serialise: with the compute stream, the
# weight-gradient side stream, c10d's communication stream and RCCL's own streams, four were not enough at README batch sizes
# (c3 with the reducer attached: 6.6 ms per step with 4 queues, 6.0-6.2 with 8; profiles/r05_hwq_matrix.txt)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, L, H, T, V_TEXT = 768, 6, 24, 1024, 50257
V = V_TEXT + 2048
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"] if os.path.exists(os.path.join(ROOT, "BASELINE.json")) \
    else "multimodal tokens/sec/GPU (fwd+bwd), 768d\u00d76L seq_len=1024, at 1/2/4/8 MI355X"


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run and hand back their exit code.  Called BEFORE anything in this process has touched the GPU;
    the children are new processes (never an exec of one that initialised HIP).  Rank 0's JSON line goes to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


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


def cpu_baseline():
    """the CPU oracle (plain PyTorch fp32 restatement of the reference path, pinned to reference
    fixtures) on the seeded M-text batch B = 2 x (1023 ids + SEP), 768d x 6L x 24H, dropout 0.1 (explicit Bernoulli masks
    at the reference's four dropout sites), 1 warm-up + 3 timed forward+backward iterations on this box's host cores."""
    from oracle import neko_oracle as O
    hw = os.cpu_count() or 1
    # intra-op threads: BASELINE.md section 3 says every hardware thread, but on the GPU box's 256 hardware threads that
    # is the SLOWEST setting by far (torch's fork/join per small op: 75.9 s per iteration = 27 tokens/s with 256 threads,
    # measured r02; sweep in profiles/r02_cpu_baseline_threads.txt), so the baseline runs at the fastest measured count
    cores = int(os.environ.get("NEKO_CPU_BASELINE_THREADS", min(hw, 32)))
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(embed_dim=D, layers=L, heads=H, text_tokens=V_TEXT, context_len=T)
    sd = O.init_state_dict(cfg, 0)
    g = torch.Generator().manual_seed(1234)
    batch = [{"text": torch.randint(0, V_TEXT, (T - 1,), generator=g).tolist()} for _ in range(2)]

    def masks():
        keep = lambda *shape: (torch.rand(*shape, generator=g) >= 0.1).to(torch.float32) / 0.9
        dm = {"embd": keep(2, T, D)}
        for i in range(L):
            dm[("attn", i)] = keep(2, H, T, T)
            dm[("resid_attn", i)] = keep(2, T, D)
            dm[("resid_mlp", i)] = keep(2, T, D)
        return dm

    times = []
    for it in range(4):
        t0 = time.perf_counter()
        dm = masks()                               # mask generation is part of the reference's step (31 % of it)
        O.loss_and_grads(sd, cfg, batch, drop_masks=dm)
        times.append(time.perf_counter() - t0)
    el = sum(times[1:]) / 3
    return {"value": 2 * T / el, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"1 warm-up + 3 timed x (B=2, T=1024 text) fwd+bwd of the 768d/6L/24H model, fp32, dropout 0.1, "
                      f"torch intra-op threads={cores} of {hw} hardware threads; {el:.2f} s per iteration"}


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
    flops = 2.0 * rows * D * hp.V          # useful columns only (the call computes Vpad = V + 175 of them)
    # which main loop neko_gemm_bf16 took for this call: asked from the library (neko_gemm_last_mainloop, ABI v19), not re-derived here
    loop = {0: "gemm_glds_kernel", 1: "gemm_a16_kernel", 2: "gemm_b16_kernel", 3: "gemm_glds64_kernel", 4: "gemm_bf16_kernel",
            5: "gemm_p16_kernel"}.get(ops.gemm_last_mainloop(), "?")
    return {"kernel": loop + "<A k-contig, B k-contig> (LM head logits)",
            "shape": [rows, hp.V, D], "ms": ms, "tflops": flops / ms / 1e9}


def _time_events(fn, iters: int = 10, warm: int = 2) -> float:
    """Average ms per call, HIP events on the stream the kernels are launched on (torch's current stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_more_kernels(B: int, Tlen: int, dropout: float):
    """The kernels the step's time actually sits in (the LM-head GEMM is 5 % of it), timed live at the
    workload's shapes: attention forward / backward of one layer (useful causal FLOPs: 2 B T^2 d forward, 2.5x that
    backward) and the three K = 768 / K = 3072 GEMM shapes of a block."""
    from neko_amd import engine, ops
    M, d, hd = B * Tlen, D, D // H
    out = []
    qkv = (torch.randn(M, 3 * d, device="cuda") * 0.5).to(torch.bfloat16)
    kb, ks = ops.mask_bias(torch.ones(B, Tlen, device="cuda"))
    drop = ops.Drop(dropout, 12345) if dropout > 0 else None
    o, lse, dmask = ops.attn_fwd(qkv, kb, ks, B, Tlen, H, hd, drop=drop, want_mask=True)      # the training call
    do = torch.randn_like(o)
    ms = _time_events(lambda: ops.attn_fwd(qkv, kb, ks, B, Tlen, H, hd, drop=drop, want_mask=True))
    fl = 2.0 * B * Tlen * Tlen * d
    out.append({"kernel": "attention forward (one layer)", "ms_per_launch": ms, "achieved": fl / ms / 1e9, "unit": "TFLOP/s",
                "frac": fl / ms / 1e9 / MFMA_PEAK_TFLOPS, "bound": "mfma (hd=32: VALU-issue limited, DESIGN 4)"})
    ms = _time_events(lambda: ops.attn_bwd(qkv, o, do, kb, ks, lse, B, Tlen, H, hd, drop=drop, mask=dmask))
    out.append({"kernel": "attention backward (one layer)", "ms_per_launch": ms, "achieved": 2.5 * fl / ms / 1e9,
                "unit": "TFLOP/s", "frac": 2.5 * fl / ms / 1e9 / MFMA_PEAK_TFLOPS, "bound": "mfma"})
    a = torch.randn(M, d, device="cuda").to(torch.bfloat16)
    # the calls stack_forward makes: both projections with their bias + residual dropout + fp32 residual in / out
    resid = torch.randn(M, d, device="cuda")
    rdrop = ops.Drop(dropout, 54321) if dropout > 0 else None
    for name, N, K in (("c_attn forward", 3 * d, d), ("attn c_proj forward (bias, dropout, fp32 residual)", d, d),
                       ("c_fc forward + GELU", 4 * d, d), ("mlp c_proj forward (bias, dropout, fp32 residual)", d, 4 * d)):
        x = a if K == d else torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(K, N, device="cuda") * 0.02).to(torch.bfloat16)
        bias = torch.zeros(N, device="cuda")
        if "c_proj" in name:
            y32 = torch.empty(M, N, device="cuda")
            ms = _time_events(lambda: ops.gemm(x, w, M, N, K, b_kstrided=True, bias=bias, resid=resid, out_f32=y32, drop=rdrop))
        else:
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if "GELU" in name else None
            act = (3 if engine.GELU_FACTOR else 1) if pre is not None else 0      # the call stack_forward makes
            ms = _time_events(lambda: ops.gemm(x, w, M, N, K, b_kstrided=True, bias=bias, act=act, pre_out=pre, out_bf16=y))
        fl = 2.0 * M * N * K
        out.append({"kernel": f"gemm {name}", "shape_MNK": [M, N, K], "ms_per_launch": ms, "achieved": fl / ms / 1e9,
                    "unit": "TFLOP/s", "frac": fl / ms / 1e9 / MFMA_PEAK_TFLOPS, "bound": "mfma"})
    # the backward's two dgrads through the MLP (the top GEMM family by time): d_pre = (g . Wpr^T) * gelu'
    # with the c_fc bias gradient folded in, and d_a2 = d_pre . Wfc^T in the dtype the LayerNorm backward is fed
    g16 = torch.randn(M, d, device="cuda").to(torch.bfloat16)
    w_pr = (torch.randn(4 * d, d, device="cuda") * 0.02).to(torch.bfloat16)         # (in = 4d, out = d): k-contiguous for the dgrad
    fac = torch.rand(M, 4 * d, device="cuda").to(torch.bfloat16)
    d_pre = torch.empty(M, 4 * d, dtype=torch.bfloat16, device="cuda")
    gb = torch.zeros(4 * d, device="cuda")
    ms = _time_events(lambda: ops.gemm_dgrad_gelu_colsum(g16, w_pr, M, 4 * d, d, fac, d_pre, gb, ldb=d,
                                                         act_in_is_factor=engine.GELU_FACTOR))
    fl = 2.0 * M * 4 * d * d
    out.append({"kernel": "gemm dgrad mlp c_proj * gelu' (+ c_fc bias gradient)", "shape_MNK": [M, 4 * d, d], "ms_per_launch": ms,
                "achieved": fl / ms / 1e9, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / MFMA_PEAK_TFLOPS, "bound": "mfma"})
    w_fc = (torch.randn(d, 4 * d, device="cuda") * 0.02).to(torch.bfloat16)         # (in = d, out = 4d)
    ms = _time_events(lambda: engine._dgrad_to_ln(d_pre, w_fc, M, d, 4 * d, 4 * d))
    out.append({"kernel": f"gemm dgrad c_fc ({'bf16' if engine.LN_DY_DTYPE == torch.bfloat16 else 'f32'} out)", "shape_MNK": [M, d, 4 * d],
                "ms_per_launch": ms, "achieved": fl / ms / 1e9, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / MFMA_PEAK_TFLOPS,
                "bound": "mfma"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="sequences per GPU per step; default 64 for the T = 1024 workloads, 32 (the README's) for c2 / c3 / c4.  64 x 1024 tokens = the reference's default global batch of 512 "
                         "(arguments.py:62) spread over the 8 GPUs of a node; 18 GB of the 288 GB at 768d.  Round 1 ran 32: "
                         "the optimiser tail, the weight-gradient split-K reductions and the launch tails amortise over twice "
                         "the tokens (+5 % tokens/s; sweep 32 / 64 / 96 / 128 in DESIGN.md section 5)")
    ap.add_argument("--workload", default="m-mix", choices=["m-mix", "m-text", "c2", "c3", "c4", "c5-mix"])
    ap.add_argument("--model", default="768d", choices=["768d", "gato-1.2b"],
                    help="768d = the metric's 768d x 6L x 24H (hd=32); gato-1.2b = BASELINE configs[4], 2048d x 24L x 16H "
                         "(hd=128, streaming attention kernels) -- a side measurement, not the metric's config")
    ap.add_argument("--ragged-groups", type=int, default=0,
                    help="> 0: length-bucketed layout (GatoPolicy.ragged_groups) instead of padding to the longest example")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.1, help="attention/residual dropout (reference default 0.1)")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd(+all-reduce) only")
    ap.add_argument("--force-dp", action="store_true",
                    help="one rank, but with the data-parallel reducer attached and every collective issued through RCCL "
                         "(world of one: sums are identities) -- the only way to execute the RCCL path on a one-GPU box")
    ap.add_argument("--capture", action="store_true",
                    help="run the step as ONE replayed HIP graph per batch structure (neko_amd/training/captured.py; "
                         "single rank): the lever for the README batch sizes (c2 / c3 / c4), where the host enqueue is the step")
    args = ap.parse_args()
    global D, L, H
    if args.model == "gato-1.2b":
        D, L, H = 2048, 24, 16
        args.no_cpu_baseline = True        # the fp32 CPU oracle of a 1.4 B-parameter model does not fit the time budget

    # ---- ranks: one process per GPU.  `--gpus N` without a launcher spawns the N ranks here, before any GPU call ----
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NEKO_BENCH_LAUNCH_CHECK") == "1":
        # launcher self-test (tests/test_dp_cpu.py, no GPU): rendezvous + one real all-reduce over gloo, rank 0 reports
        torch.distributed.init_process_group("gloo")
        t = torch.ones(1)
        torch.distributed.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": args.gpus, "ranks_seen": int(t.item()),
                              "world_size": torch.distributed.get_world_size()}), flush=True)
        torch.distributed.destroy_process_group()
        return
    # NEKO_BENCH_BACKEND=gloo + NEKO_BENCH_ONE_DEVICE=1: dry run of the multi-rank flow on a 1-GPU box (every rank on
    # cuda:0, gradients reduced through the host) -- exercises the DP hooks, barriers and max-over-ranks timing, not
    # a performance configuration
    backend = os.environ.get("NEKO_BENCH_BACKEND", "nccl")
    # stdout carries exactly ONE line, rank 0's JSON: whatever else writes to fd 1 during the run (RCCL prints a version banner
    # through C stdio when the first communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("NEKO_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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
    ranks_seen = 1
    if world > 1 or args.force_dp:
        dp = GradReducer(model._flat, payload=os.environ.get("NEKO_DP_PAYLOAD", "fp32"), force_collectives=args.force_dp)
        dp.broadcast_parameters()
        t = torch.ones(1, device=dev)
        torch.distributed.all_reduce(t)              # a real collective: how many ranks RCCL actually connected
        ranks_seen = int(t.item())
        dp.attach(model, opt)
        if args.workload in ("c2", "c3", "c4"):      # control-only workloads: no rank ever touches the text rows (see dp.py)
            dp.declare_unused_rows("embed_token.weight", 0, model.text_tokens)
            dp.no_text_declared = True

    B = args.batch if args.batch is not None else (32 if args.workload in ("c2", "c3", "c4") else 64)
    batches = [make_batch(args.workload, B, 1234 + rank + 100 * i, dev) for i in range(2)]
    Tlen = {"c2": 240, "c3": 240, "c4": 494}.get(args.workload, T)

    # exposed communication: the time the compute stream stalls between "backward enqueued" and "every reduction done"
    comm_ev = []

    def step(i):
        _, loss = model.forward(inputs=batches[i % len(batches)], compute_loss=True, return_logits=False)
        loss.backward()
        if dp is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dp.flush()
            dp.finish()
            e1.record()
            comm_ev.append((e0, e1))
        if not args.no_optimizer:
            opt.clip_grad_norm_(1.0)
            opt.step()
        opt.zero_grad()
        return loss

    cap = None
    if args.capture:
        assert world == 1 and not args.no_optimizer, "--capture: single rank, full step"
        from neko_amd.training.captured import CapturedTrainStep
        cap = CapturedTrainStep(model, opt, None, grad_norm_clip=1.0)
        eager_step = step

        def step(i):                               # noqa: F811  (the captured step replaces the eager one)
            return cap.step(batches[i % len(batches)])[0]
    trace_mode = os.environ.get("NEKO_BENCH_STEP_TIMES", "0")       # diagnosis only: "1" per-step wall times (a sync per step), "2" host enqueue times (no sync)
    trace_steps = trace_mode == "1"
    step_ms = []
    for i in range(max(args.warmup, 3 if cap else 0)):
        ts = time.perf_counter()
        loss = step(i)
        if trace_steps:
            torch.cuda.synchronize()
        if trace_mode != "0":
            step_ms.append(1e3 * (time.perf_counter() - ts))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    comm_ev.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        loss = step(i)
        if trace_steps:
            torch.cuda.synchronize()
        if trace_mode != "0":
            step_ms.append(1e3 * (time.perf_counter() - ts))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if trace_mode != "0" and rank == 0:
        print("step wall times (ms, warmup then timed): " + " ".join(f"{x:.1f}" for x in step_ms), file=sys.stderr)
    exposed_comm_ms = (sum(a.elapsed_time(b) for a, b in comm_ev) / len(comm_ev)) if comm_ev else 0.0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t)

    # the reference's metric is forward+backward (+ all-reduce); `value` above includes clip + AdamW, the
    # fwd+bwd(+all-reduce)-only rate is timed separately over the same number of steps and reported next to it
    el_fb = None
    if cap is not None:
        step = eager_step
        cap_stats = {"graphs": len(cap.entries), "replays": cap.replays, "eager_steps": cap.eager_steps}
        cap.close()
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
        lp = model.last_pack
        lm_frac = (lp.n_loss / float(B * Tlen)) if (lp is not None and lp.n_loss > 0 and model.lm_head_selected_rows) else 1.0
        fpt = 3 * flops_per_token_fwd(d=D, layers=L, t=Tlen, lm_rows_frac=lm_frac)
        if lp is not None and lp.segments:
            # length-grouped (ragged) layout: the stack runs the packed rows only and attention sees every group at its own length --
            # the executed FLOPs per step, expressed per position of the padded (B, T) batch that `value` counts
            ex = sum(bk * tk * L * (24 * D * D + 2 * D * (tk + 1)) for (_, bk, tk) in lp.segments) + 2 * D * V * lp.n_loss
            fpt = 3 * ex / float(B * Tlen)
        # the launch the step itself makes: every loss row of the batch in one logits GEMM, up to lm_head_chunk_rows
        # (same padding rule as engine.lm_head_loss_selected: whole 256-row tiles once the batch is large)
        pad_rows = lambda n: max(64, (n + 63) // 64 * 64) if n < 2048 else (n + 255) // 256 * 256
        n_rows = (pad_rows(lp.n_loss) if (lp is not None and lp.n_loss > 0 and model.lm_head_selected_rows) else B * Tlen)
        dom = time_dominant_kernel(model, min(model.lm_head_chunk_rows, n_rows))
        # HBM bytes per launch of that kernel: PMC counters cannot be collected from inside the timed process, so the
        # number is the committed rocprofv3 --pmc measurement of the same call (tools/pmc_lmhead.sh), null if absent
        traffic, traffic_src = None, None
        import glob
        # only files named exactly r<NN>_lmhead_traffic.json count (the measurement of the SHIPPED kernel, written by tools/pmc_lmhead.sh +
        # tools/pmc_lmhead_summarise.py for the final code of a round); probe builds keep other names.  Newest round first.
        import re as _re
        cands = [tp for tp in glob.glob(os.path.join(ROOT, "profiles", "r*_lmhead_traffic.json"))
                 if _re.fullmatch(r"r\d+_lmhead_traffic\.json", os.path.basename(tp))]
        for tp in sorted(cands, key=lambda tp: int(_re.match(r"r(\d+)_", os.path.basename(tp)).group(1)), reverse=True):
            tj = json.load(open(tp))
            if tj.get("build", "default") != "default":
                continue
            if tj.get("shape_MNK") == dom["shape"]:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{os.path.basename(tp)} (committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of this call, calibrated; not re-measured in this run)"
                break
        # positions that are not padding (the metric counts every position of the padded (B, T) batch; for the
        # ragged c5-mix workload the useful rate is the one over real tokens) and rows that went through the stack
        from neko_amd.policy.gato_policy import K_PAD, build_layout
        lay = build_layout(batches[0], True, T, False, ragged_groups=args.ragged_groups)
        real_tokens = int((lay.desc[:, 0] != K_PAD).sum())
        rows = int(lay.desc.shape[0])
        # MFMA-pipe utilisation of the same kernel from the committed SQ counter pass (tools/pmc_gemm.sh): context for
        # `frac`, which is priced against the nominal 2.5 PFLOP/s at 2.4 GHz while the chip sustains ~1.5-1.7 GHz here
        mfma_busy = None
        # newest round first (file names start with the round: r03_... > r02_... > r01_...); the first file that holds a
        # summarised lmlogit16 line wins, and the bench line names the file it quotes
        cp = None
        if D == 768:
            import re
            for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_counters*.txt")), reverse=True):
                mm = re.search(r"lmlogit16: .*?MFMA pipe busy ([0-9.]+) %", open(cand).read())
                if mm:
                    cp, mfma_busy = cand, float(mm.group(1)) / 100
                    break
        out = {
            "metric": METRIC, "value": value, "unit": "tokens/s",
            # the driver's bench contract: `value` is the WHOLE-JOB aggregate over all N GPUs (it divides by N itself for the scaling
            # curve); the metric's per-GPU figure is in `tokens_per_sec_per_gpu` / `value_per_gpu` (equal to `value` at N = 1)
            "value_scope": "whole job: sum over all n_gpus ranks",
            "value_per_gpu": value / world,
            "value_includes": "fwd + bwd + gradient all-reduce + clip + AdamW (whole job, all GPUs)",
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
                         "mfma_pipe_busy_source": f"profiles/{os.path.basename(cp)} (SQ_VALU_MFMA_BUSY_CYCLES)" if mfma_busy else None},
            "roofline_step": {"bound": "mfma", "achieved": value / world * fpt / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": value / world * fpt / (MFMA_PEAK_TFLOPS * 1e12),
                              "what": "executed model FLOPs of the whole step (fwd+bwd, LM head on loss rows only) / step time, per GPU"},
            "roofline_more": time_more_kernels(B, Tlen, dropout) if D == 768 else None,
            "rccl_ranks_seen": ranks_seen,
            "exposed_comm_ms_per_step": exposed_comm_ms,
            "dp_payload": (dp.payload if dp is not None else None),
            "dp_collective": (dp.collective if dp is not None else None),
            "captured_step": (cap_stats if args.capture else None),
        }
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (the other ranks would idle behind it)
            out["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
