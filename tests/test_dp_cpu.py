"""world_size-2 tests of the data-parallel path on CPU (backend gloo): parameter broadcast, per-range
gradient all-reduce in backward order + deferred ranges, identical collective order on ranks whose batches
touch different parameter ranges, the reduced "range active" flags and the 1/world gradient scale."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results, declare=False, payload="fp32", collective="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neko_amd.dp import GradReducer
        from neko_amd.policy.gato_policy import GatoPolicy
        torch.manual_seed(100 + rank)            # different initial weights per rank
        m = GatoPolicy("cpu", 64, 2, 2, 0.0, resid_mid_channels=128, context_len=32, text_tokenizer=64)
        flat = m._flat
        dp = GradReducer(flat, bucket_bytes=64 * 1024, payload=payload, collective=collective)      # small buckets: several slices per range
        dp.broadcast_parameters()
        if declare:                 # control-only run: the 64 text rows of the token embedding never see a gradient
            dp.declare_unused_rows("embed_token.weight", 0, m.text_tokens)
        w0 = flat.data.clone()
        # rank-specific gradients; rank 1 has "no images" (its image range stays zero and it never signals it)
        g = torch.Generator().manual_seed(7 + rank)
        flat.grad.copy_(torch.randn(flat.total, generator=g))
        a, b = flat.group_ranges["image"]
        if rank == 1:
            flat.grad[a:b] = 0
        local = flat.grad.clone()
        # backward order: head, ln_f, layer1, layer0 ; rank 0 additionally reports the image range (ignored: deferred)
        for name in ["head", "lnf", "layer1", "layer0"]:
            dp.group_ready(name)
        if rank == 0:
            dp.group_ready("image")
        dp.group_ready("frontend")
        dp.flush()
        dp.finish()
        flags = torch.tensor([1, 1 if rank == 0 else 0, 1], dtype=torch.int32)
        dp.reduce_flags(flags)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        if payload == "bf16":       # each rank's gradient is rounded to bf16, summed in bf16, widened back
            expect = sum(t.to(torch.bfloat16) for t in gathered).to(torch.float32)
            expect[flat.group_ranges["never"][0]:] = local[flat.group_ranges["never"][0]:]
        else:
            expect = sum(gathered)
        na, nb = flat.group_ranges["never"]
        if declare:                 # the declared slice is left alone (local values), everything around it is summed
            o, n, shape = flat.offsets["embed_token.weight"]
            z0, z1 = o, o + m.text_tokens * shape[1]
            assert z1 - z0 == 64 * 64 and z1 < o + n
            skipped_untouched = torch.equal(flat.grad[z0:z1], local[z0:z1])
            expect[z0:z1] = local[z0:z1]
        else:
            skipped_untouched = True
        ok_sum = torch.allclose(flat.grad[:na], expect[:na], rtol=1e-6, atol=1e-6) and skipped_untouched
        ok_never = torch.equal(flat.grad[na:nb], local[na:nb])       # the never-used range is not reduced
        w_all = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(w_all, w0)
        results[rank] = dict(ok_sum=bool(ok_sum), ok_never=bool(ok_never), same_w=bool(torch.equal(w_all[0], w_all[1])),
                             flags=flags.tolist(), scale=float(dp.grad_scale), rs_ag=dp._rs_ag_ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("declare,payload,collective", [(False, "fp32", "allreduce"), (True, "fp32", "allreduce"), (False, "bf16", "allreduce"),
                                                        (True, "bf16", "allreduce"), (True, "fp32", "rs_ag"), (False, "bf16", "rs_ag")])
def test_grad_reducer_world2_gloo(declare, payload, collective):
    """collective = "rs_ag" (round 6): every message as reduce-scatter + all-gather of 1 / world shards, slices that do not
    divide by the world size finish with a small all-reduce -- the same sums as the all-reduce form."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, results, declare, payload, collective)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
            assert p.exitcode == 0, f"worker exit code {p.exitcode}"
        for r in range(world):
            res = results[r]
            assert res["ok_sum"] and res["ok_never"] and res["same_w"], res
            assert res["flags"] == [1, 1, 1]          # union over ranks of "range took part in this step"
            assert abs(res["scale"] - 0.5) < 1e-12    # averaging folded into the optimiser's gradient scale
            assert res["rs_ag"] is (True if collective == "rs_ag" else None)      # the split collectives really ran (gloo has them)


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must start 2 rank processes itself (VERDICT r01:
    the flag used to be parsed and ignored), rendezvous on 127.0.0.1 and run a real collective; `--gpus` that
    disagrees with an existing WORLD_SIZE is an error.  NEKO_BENCH_LAUNCH_CHECK=1 stops after the collective (gloo, no GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NEKO_BENCH_LAUNCH_CHECK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["world_size"] == 2
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
