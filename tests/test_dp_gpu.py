"""Data-parallel step on the GPU with two ranks (both on cuda:0, backend gloo -- the pool has one GPU per box, so
RCCL itself cannot be exercised here): the full neko_amd.dp path -- parameter broadcast, per-range all-reduce
launched from inside backward, deferred ranges, MAX-reduced activity flags read by the optimiser kernel, 1/world
gradient scale, clip on the averaged gradient -- must give every rank exactly the parameters a single process gets
from the two batches with gradient averaging.  Rank 1's batch has no image, rank 0's has one (different parameter
ranges touched per rank: the collective order must still match)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches():
    g = torch.Generator().manual_seed(5)
    b0 = [{"images": torch.floor(torch.rand(2, 3, 32, 32, generator=g) * 256), "discrete_actions": torch.randint(0, 4, (2, 1), generator=g).to(torch.int32)},
          {"text": torch.randint(0, 128, (30,), generator=g).tolist()}]
    b1 = [{"continuous_obs": torch.randn(4, 5, generator=g), "continuous_actions": torch.rand(4, 2, generator=g) * 2 - 1},
          {"text": torch.randint(0, 128, (41,), generator=g).tolist()}]
    return b0, b1


def _to_dev(b):
    return [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ex.items()} for ex in b]


def _make():
    from neko_amd.policy.gato_policy import GatoPolicy
    from oracle import neko_oracle as O
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=128, context_len=96)
    m = GatoPolicy("cuda:0", 64, 2, 2, 0.0, resid_mid_channels=128, context_len=96, text_tokenizer=128)
    m.transformer.drop.p = 0.0
    m.load_state_dict(O.init_state_dict(cfg, 21))
    m.eval()            # deterministic patch positions; gradients flow regardless of mode
    return m


def _batches_rank1_without_loss():
    """Rank 1's batch is a single image-only example: no position of it is a target (gato_policy.py:176-183 selects
    nothing), its loss is 0 and its gradients are zero -- but it must issue exactly the collectives rank 0 issues."""
    g = torch.Generator().manual_seed(9)
    b1 = [{"images": torch.floor(torch.rand(1, 3, 32, 32, generator=g) * 256)}]
    return _batches()[0], b1


def _worker(rank, world, port, out, batches_fn=None, det=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if det:
        from neko_amd import ops
        ops.SCATTER_DET = True
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neko_amd.dp import GradReducer
        from neko_amd.training.optim import NekoAdamW
        m = _make()
        if rank == 1:                       # prove the broadcast: rank 1 starts from different weights
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01)
        opt = NekoAdamW(m, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        dp = GradReducer(m._flat, bucket_bytes=32 * 1024)
        dp.broadcast_parameters()
        dp.attach(m, opt)
        batch = _to_dev((batches_fn or _batches)()[rank])
        for it in range(2):
            _, loss = m.forward(inputs=batch, compute_loss=True, return_logits=False)
            loss.backward()
            dp.flush()
            dp.finish()
            gn = opt.clip_grad_norm_(0.5)
            opt.step()
            opt.zero_grad()
            if it == 0:          # what the single-process reference is compared with (see the test)
                torch.cuda.synchronize()
                out[f"first{rank}"] = {k: v.detach().cpu() for k, v in m.state_dict().items()
                                       if v.dtype == torch.float32 and v.numel() < 70000}
                out[f"gn_first{rank}"] = float(gn)
        torch.cuda.synchronize()
        out[rank] = {k: v.detach().cpu() for k, v in m.state_dict().items() if v.dtype == torch.float32 and v.numel() < 70000}
        out[f"gn{rank}"] = float(gn)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("det", [False, True], ids=["atomic-scatters", "deterministic-scatters"])
@pytest.mark.parametrize("batches_fn", [_batches, _batches_rank1_without_loss], ids=["both-ranks-have-targets", "rank1-has-no-loss-position"])
def test_dp_two_ranks_match_single_process_average(batches_fn, det):
    """Second case: a rank whose batch has no loss position still runs the whole backward (zero
    dlogits) and therefore issues the same per-range collectives in the same order; a mismatch would hang this test."""
    from neko_amd.training.optim import NekoAdamW
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out, batches_fn, det)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0, f"worker exit code {p.exitcode}"
        r0, r1, gn0, gn1 = out[0], out[1], out["gn0"], out["gn1"]
        f0, gnf0 = out["first0"], out["gn_first0"]
    for k in r0:                              # after two steps the ranks hold the very same bits
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}"
    assert gn0 == gn1
    # single-process reference: the two batches one after the other, gradients averaged with equal weight per rank.  Compared
    # after ONE step: the first Adam update is lr * g / |g| per element, so an element whose gradient is rounding noise around
    # zero moves by +-lr with a sign that depends on the summation order (gradient accumulation here, all-reduce there) -- from
    # the second step on the two runs are different trajectories at the 1e-4 level (tools/determinism_probe.py).
    from neko_amd import ops
    prev_det = ops.SCATTER_DET
    ops.SCATTER_DET = bool(det)
    try:
        m = _make()
        opt = NekoAdamW(m, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        opt.grad_scale = torch.full((1,), 0.5, device="cuda")
        b0, b1 = (_to_dev(b) for b in batches_fn())
        for b in (b0, b1):
            _, loss = m.forward(inputs=b, compute_loss=True, return_logits=False)
            loss.backward()                      # accumulates into the flat gradient
        gn = opt.clip_grad_norm_(0.5)
        opt.step()
        opt.zero_grad()
        ref = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    finally:
        ops.SCATTER_DET = prev_det
    assert abs(float(gn) - gnf0) < 1e-5 * gnf0, (float(gn), gnf0)
    if det:
        # with the table gradients summed in a fixed order (NEKO_DETERMINISTIC, ABI v15) every gradient of a rank is a
        # deterministic function of its batch, and fl(g0 + g1) is what both the all-reduce of two ranks and the accumulating
        # kernels of one process compute: the first update must then agree entry by entry, no noise allowance
        worst = max(float((v - ref[k]).abs().max()) for k, v in f0.items())
        nbad = sum(int((~torch.isclose(v, ref[k], rtol=2e-4, atol=2e-6)).sum()) for k, v in f0.items())
        assert nbad == 0 and worst <= 2e-5, (nbad, worst)
        return
    bad = 0
    for k, v in f0.items():
        close_ = torch.isclose(v, ref[k], rtol=2e-4, atol=2e-6)
        # the +-lr sign flips of noise-level gradient elements: a handful of entries may differ by up to 2 lr
        assert float((v - ref[k]).abs().max()) <= 2.1e-2, (k, float((v - ref[k]).abs().max()))
        bad += int((~close_).sum())
    total = sum(v.numel() for v in f0.values())
    assert bad <= 2e-3 * total, (bad, total)


def _rccl_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from neko_amd.dp import GradReducer
        from neko_amd.training.optim import NekoAdamW
        res = {}
        for payload in ("fp32", "bf16", "fp32+rs_ag", None):   # None: no reducer at all (the reference run)
            m = _make()
            opt = NekoAdamW(m, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
            dp = None
            if payload is not None:
                # "+rs_ag" (round 6): every message as RCCL reduce-scatter + all-gather instead of one all-reduce
                dp = GradReducer(m._flat, bucket_bytes=32 * 1024, payload=payload.split("+")[0], force_collectives=True,
                                 collective="rs_ag" if payload.endswith("rs_ag") else "allreduce")
                dp.broadcast_parameters()
                dp.attach(m, opt)
            losses = []
            for b in (_to_dev(_batches()[0]), _to_dev(_batches()[1])):
                _, loss = m.forward(inputs=b, compute_loss=True, return_logits=False)
                loss.backward()
                if dp is not None:
                    dp.flush()
                    dp.finish()
                opt.clip_grad_norm_(0.5)
                opt.step()
                opt.zero_grad()
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            res[str(payload)] = (losses, {k: v.detach().cpu() for k, v in m.state_dict().items()
                                          if v.dtype == torch.float32 and 4096 <= v.numel() < 70000})
            if payload is not None and payload.endswith("rs_ag"):
                res["rs_ag_ran"] = dp._rs_ag_ok
        out["res"] = res
        out["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_reducer_collectives_execute_on_rccl_in_a_world_of_one():
    """The pool has one GPU per box, so RCCL cannot reduce across ranks here -- but it can RUN: with
    force_collectives=True the reducer issues its broadcast, its per-range all-reduces from inside backward (fp32 and
    bf16 payload), the deferred ranges and the flag reduction through the nccl (= RCCL) backend on its communication
    stream.  A sum over one rank is the identity, so training must equal the run without a reducer (exactly for fp32,
    to bf16 rounding of the gradients for the bf16 payload)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        p = ctx.Process(target=_rccl_worker, args=(port, out))
        p.start()
        p.join(timeout=600)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
        res, backend = out["res"], out["backend"]
    assert backend == "nccl"
    l32, w32 = res["fp32"]
    l16, w16 = res["bf16"]
    lref, wref = res["None"]
    lrs, wrs = res["fp32+rs_ag"]
    assert res["rs_ag_ran"] is True and lrs == lref          # RCCL's reduce_scatter_tensor / all_gather_into_tensor really ran
    for k in wref:
        assert torch.allclose(wrs[k], wref[k], rtol=1e-6, atol=1e-7), k
    assert l32 == lref
    for k in wref:
        assert torch.allclose(w32[k], wref[k], rtol=1e-6, atol=1e-7), k      # (atomics in the embedding scatter)
        assert float((w16[k] - wref[k]).norm()) <= 2e-2 * float(wref[k].norm()) + 1e-6, k
    assert abs(l16[1] - lref[1]) < 1e-3 * abs(lref[1])
