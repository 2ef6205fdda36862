"""Policy-level parity on the GPU: neko_amd.GatoPolicy (HIP path, through the C ABI) against the golden
fixtures captured from the imported reference (tests/golden/*.pt) and against the CPU oracle on
seeded inputs.

Tolerances (bf16 MFMA operands, fp32 accumulation/statistics; SURVEY.md 8(d) parity gates):
  tokens / masks: bit-exact;  fp32 gathers: 1e-6;  hidden states / logits: 2e-2 of the tensor scale;
  loss: 1e-3 relative (one batch), 100-step trace: see test;  per-parameter grad norms: 2e-2 relative.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


def make_policy(cfg: O.OracleConfig, seed: int, train: bool = False):
    from neko_amd.policy.gato_policy import GatoPolicy
    m = GatoPolicy(DEV, cfg.embed_dim, cfg.layers, cfg.heads, 0.0, resid_mid_channels=128,
                   context_len=cfg.context_len, continuous_tokens=cfg.continuous_tokens,
                   discrete_tokens=cfg.discrete_tokens, text_tokenizer=cfg.text_tokens)
    m.transformer.drop.p = 0.0
    sd = O.init_state_dict(cfg, seed)
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    m.train(train)
    return m, sd


def to_dev(batch):
    out = []
    for ex in batch:
        out.append({k: (v.to(DEV) if torch.is_tensor(v) and k not in ("text",) and v.dtype != torch.uint8 else v)
                    for k, v in ex.items()})
    return out


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def test_state_dict_roundtrip(golden):
    cfg = O.OracleConfig(**golden("g3_pack")["cfg"])
    m, sd = make_policy(cfg, 1234)
    out = m.state_dict()
    assert list(out.keys()) == list(sd.keys()) or set(out.keys()) == set(sd.keys())
    for k in sd:
        assert torch.equal(out[k].cpu(), sd[k]), k


def test_g3_pack(golden):
    f = golden("g3_pack")
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"])
    with torch.no_grad():
        e, t, tg, pm = m.tokenize_input_dicts(to_dev(f["batch"]))
    assert torch.equal(t.cpu(), f["tokens"])
    assert torch.equal(tg.cpu(), f["target_masks"])
    assert torch.equal(pm.cpu(), f["pad_masks"])
    ref = f["embeddings"]
    is_img = torch.zeros_like(f["pad_masks"], dtype=torch.bool)
    # image-patch positions: token 0, not target, not pad, and not a separator -> compare loosely (bf16 projection)
    err = (e.cpu() - ref).abs().amax(dim=-1)
    exact = err <= 1e-5 * (1 + ref.abs().amax(dim=-1))
    loose = err <= 3e-2 * ref.abs().max()
    assert bool(loose.all()), float(err.max())
    # everything that is a table gather must be exact: count how many positions are only loosely equal
    n_img = sum(ex["images"].shape[0] * (ex["images"].shape[2] // 16) * (ex["images"].shape[3] // 16)
                for ex in f["batch"] if ex.get("images") is not None)
    assert int((~exact).sum()) <= n_img


def test_g4_image_embedding(golden):
    f = golden("g4_image")
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"])
    with torch.no_grad():
        out = m.image_embedding(f["images"].to(DEV))
    assert out.shape == f["out"].shape
    assert relerr(out, f["out"]) < 2e-2


@pytest.mark.parametrize("name", ["g5_hidden", "g5b_hidden"])
def test_g5_transformer(golden, name):
    f = golden(name)
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"])
    with torch.no_grad():
        out = m.transformer(inputs_embeds=f["x"].to(DEV), attention_mask=f["mask"].to(DEV))["last_hidden_state"]
    ref = f["last_hidden_state"]
    assert out.shape == ref.shape
    valid = f["mask"].bool()
    assert relerr(out.cpu()[valid], ref[valid]) < 2e-2
    # padded query rows follow the reference's finite -1e4 semantics too
    assert relerr(out, ref) < 5e-2


def test_g6_logits_loss_grads(golden):
    f = golden("g6_policy")
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"])
    logits, loss = m(to_dev(f["batch"]), compute_loss=True)
    assert tuple(logits.shape) == f["logits_shape"]
    assert relerr(logits[:, ::f["row_stride"], :], f["logits_rows"]) < 2e-2
    assert abs(float(loss) - f["loss"]) < 1e-3 * abs(f["loss"]), (float(loss), f["loss"])
    loss.backward()
    named = dict(m.named_parameters())
    sq = 0.0
    for k, n in f["grad_norms"].items():
        g = named[k].grad
        if n is None:
            assert g is None, k
            continue
        assert g is not None, k
        gn = float(g.float().norm())
        sq += gn * gn
        if k.endswith("c_attn.bias"):
            continue        # its K third has a mathematically zero gradient (rounding noise only)
        assert abs(gn - n) < 2e-2 * n + 1e-6, (k, gn, n)
    assert abs(math.sqrt(sq) - f["total_grad_norm"]) < 5e-3 * f["total_grad_norm"]
    for k, gref in f["small_grads"].items():
        if k.endswith("c_attn.bias"):
            continue
        assert relerr(named[k].grad, gref) < 4e-2, (k, relerr(named[k].grad, gref))


def test_forward_kwargs_form_and_no_logits(golden):
    f = golden("g6_policy")
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"])
    with torch.no_grad():
        e, t, tg, pm = m.tokenize_input_dicts(to_dev(f["batch"]))
        logits, loss = m(token_embeddings=e, tokens=t, token_target_masks=tg, token_masks=pm, compute_loss=True)
        none_logits, loss2 = m(to_dev(f["batch"]), compute_loss=True, return_logits=False)
        logits3, loss3 = m(token_embeddings=e, tokens=None, token_target_masks=None, token_masks=pm)
    assert none_logits is None and loss3 is None
    assert abs(float(loss) - f["loss"]) < 1e-3 * abs(f["loss"])
    # the kwargs form derives the loss rows from the masks it is given and runs the LM head on every row, the dict form
    # runs it on the host-known loss rows only: same rows, different GEMM tiling -> equal to fp32 summation noise
    assert abs(float(loss2) - float(loss)) < 2e-6 * abs(float(loss))
    assert torch.equal(logits3, logits)


def test_bf16_emulating_oracle_is_tighter():
    """Against the oracle run with the kernels' rounding points (bf16 operands) the agreement is ~10x tighter
    than against the fp32 reference: the residual error is rounding, not logic."""
    cfg = O.OracleConfig(embed_dim=128, layers=2, heads=4, text_tokens=128, context_len=128)
    m, sd = make_policy(cfg, 5)
    g = torch.Generator().manual_seed(0)
    B, T = 3, 96
    x = torch.randn(B, T, 128, generator=g)
    mask = torch.ones(B, T); mask[1, :17] = 0
    with torch.no_grad():
        out = m.transformer(inputs_embeds=x.to(DEV), attention_mask=mask.to(DEV))["last_hidden_state"]
    ref16 = O.transformer_forward(sd, cfg, x, mask, bf16=True)
    ref32 = O.transformer_forward(sd, cfg, x, mask, bf16=False)
    v = mask.bool()
    e16, e32 = relerr(out.cpu()[v], ref16[v]), relerr(out.cpu()[v], ref32[v])
    assert e16 < 1e-2 and e32 < 2e-2, (e16, e32)


def test_g7_training_trace(golden):
    """100 optimisation steps with the fused HIP optimiser vs the reference's trace (torch AdamW, LambdaLR,
    clip 1.0).  north_star: loss within 1e-3 rel of the CPU reference over 100 steps."""
    from neko_amd.training.optim import NekoAdamW
    from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler
    f = golden("g7_trace")
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"], train=True)
    opt = NekoAdamW(m, lr=f["lr"], betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, f["warmup"], f["total_steps"], base_lr=f["lr"],
                                                   init_lr=f["init_lr"], min_lr=f["min_lr"])
    tr = f["trace"]
    batches = [to_dev(b) for b in f["batches"]]
    losses, norms = [], []
    for step in range(f["total_steps"]):
        assert abs(sch.get_last_lr()[0] - tr["lr"][step]) < 1e-12 + 1e-9 * tr["lr"][step]
        torch.manual_seed(1000 + step)        # same patch-position draws as the fixture run
        _, loss = m.forward(inputs=batches[step % len(batches)], compute_loss=True, return_logits=False)
        loss.backward()
        norms.append(opt.clip_grad_norm_(1.0))
        opt.step()
        sch.step()
        opt.zero_grad()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().tolist()
    norms = torch.stack(norms).reshape(-1).cpu().tolist()
    rel = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(losses, tr["loss"])]
    reln = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(norms, tr["grad_norm"])]
    print("max rel loss dev", max(rel), "at", rel.index(max(rel)), "| max rel norm dev", max(reln))
    print("loss head", losses[:3], tr["loss"][:3], "tail", losses[-3:], tr["loss"][-3:])
    # lr 3e-3 on a 64-d model (loss 7.8 -> 0.2 in 100 steps) is chaotic: bf16 rounding moves the trajectory by a
    # few % mid-run.  Tight on the first steps, tracking afterwards; the 1e-3 gate is held on G7b below.
    assert max(rel[:10]) < 1e-3, rel[:10]
    assert sorted(rel)[len(rel) // 2] < 1e-2, sorted(rel)[len(rel) // 2]
    assert max(rel) < 1e-1, (max(rel), rel.index(max(rel)))
    assert abs(losses[-1] - tr["loss"][-1]) < 0.05 and losses[-1] < 0.6        # it trains, to the same place


def test_g7b_training_trace_1e3():
    """north_star gate: loss within 1e-3 relative of the CPU reference over 100 steps, in the regime of the
    reference's recipe (lr 1e-4 with warm-up, betas .9/.95, wd .1, clip 1.0); d=128, 3 layers, hd=32."""
    import os
    from neko_amd.training.optim import NekoAdamW
    from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler
    f = torch.load(os.path.join(os.path.dirname(__file__), "golden", "g7b_trace.pt"), weights_only=False)
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"], train=True)
    opt = NekoAdamW(m, lr=f["lr"], betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, f["warmup"], f["total_steps"], base_lr=f["lr"],
                                                   init_lr=f["init_lr"], min_lr=f["min_lr"])
    tr = f["trace"]
    batches = [to_dev(b) for b in f["batches"]]
    losses, norms = [], []
    for step in range(f["total_steps"]):
        _, loss = m.forward(inputs=batches[step % len(batches)], compute_loss=True, return_logits=False)
        loss.backward()
        norms.append(opt.clip_grad_norm_(1.0))
        opt.step()
        sch.step()
        opt.zero_grad()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().tolist()
    norms = torch.stack(norms).reshape(-1).cpu().tolist()
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, tr["loss"])]
    reln = [abs(a - b) / abs(b) for a, b in zip(norms, tr["grad_norm"])]
    print("g7b max rel loss dev", max(rel), "| max rel grad-norm dev", max(reln), "| loss", losses[0], "->", losses[-1])
    assert max(rel) < 1e-3, (max(rel), rel.index(max(rel)))
    # The gradient norm of a single step is a much noisier observable than the loss: steps 0-1 agree with the reference
    # to 1e-3..2e-3 and are bit-reproducible run to run, but after ~60 steps the bf16 trajectory has drifted enough from
    # the fp32 one (and, through fp32-atomic summation order, from its own previous run) that a late step's norm is
    # 2..5 % off (tools/g7b_probe.py: max 0.025..0.048 over 12 runs, always at steps 65-92).  Gate the typical
    # deviation tightly and the worst step loosely; the loss above is the north-star gate (measured max 1e-4).
    assert max(reln[:5]) < 5e-3, reln[:5]
    assert sorted(reln)[len(reln) // 2] < 1.5e-2, sorted(reln)[len(reln) // 2]
    assert max(reln) < 1.2e-1, (max(reln), reln.index(max(reln)))


def test_gradient_accumulation_two_micro_batches():
    """trainer.py:176 (`accelerator.accumulate`): two backward calls with losses scaled by 1/2 before one optimiser step
    must leave (gA + gB) / 2 in the gradients -- every gradient kernel adds to what is there."""
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=128, context_len=48)
    m, sd = make_policy(cfg, 5)
    m.eval()                                            # deterministic patch positions on both sides
    g = torch.Generator().manual_seed(11)
    bA = [{"text": torch.randint(0, 128, (20,), generator=g).tolist()},
          {"continuous_obs": torch.randn(3, 4, generator=g), "continuous_actions": torch.rand(3, 2, generator=g) * 2 - 1},
          {"images": torch.floor(torch.rand(1, 3, 32, 32, generator=g) * 256),
           "discrete_actions": torch.randint(0, 4, (1, 1), generator=g).to(torch.int32)}]
    bB = [{"text": torch.randint(0, 128, (33,), generator=g).tolist()},
          {"images": torch.floor(torch.rand(2, 3, 32, 48, generator=g) * 256),
           "discrete_actions": torch.randint(0, 4, (2, 1), generator=g).to(torch.int32)}]
    ref = {}
    for b in (bA, bB):
        _, _, grads = O.loss_and_grads(sd, cfg, b)
        for k, v in grads.items():
            if v is not None:
                ref[k] = ref.get(k, 0) + 0.5 * v
    for b in (bA, bB):
        _, loss = m(to_dev(b), compute_loss=True)
        (loss / 2).backward()
    named = dict(m.named_parameters())
    for k, gref in ref.items():
        if k.endswith("c_attn.bias"):
            continue
        assert named[k].grad is not None, k
        assert relerr(named[k].grad, gref) < 8e-2, (k, relerr(named[k].grad, gref))


def test_training_step_is_bit_reproducible_and_matches_the_atomic_scatter():
    """With the embedding-table gradients summed in sorted order (ABI v15, NEKO_DETERMINISTIC=1) nothing in a step depends on
    execution order: repeated forward + backward passes of the same batch give bit-identical gradients for EVERY parameter, and the
    default atomic form of the two scatters agrees with them to fp32 summation noise."""
    from neko_amd import ops
    f = torch.load(os.path.join(os.path.dirname(__file__), "golden", "g7_trace.pt"), weights_only=False)
    cfg = O.OracleConfig(**f["cfg"])
    m, _ = make_policy(cfg, f["seed"], train=False)
    batch = to_dev(f["batches"][0])

    def grads():
        m.zero_grad(set_to_none=True)
        m._flat.zero_grad()
        _, loss = m(batch, compute_loss=True, return_logits=False)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    prev, ops.SCATTER_DET = ops.SCATTER_DET, True           # what NEKO_DETERMINISTIC=1 selects
    try:
        l0, g0 = grads()
        for _ in range(3):
            l1, g1 = grads()
            assert l1 == l0
            for k in g0:
                assert torch.equal(g0[k], g1[k]), k
        ops.SCATTER_DET = False
        _, ga = grads()
    finally:
        ops.SCATTER_DET = prev
    for k in g0:
        sc = float(g0[k].abs().max())
        assert float((ga[k] - g0[k]).abs().max()) <= 2e-6 * sc + 1e-12, k


def test_layout_memo_reuses_descriptors_without_changing_a_bit():
    """GatoPolicy._prepare remembers the uploaded descriptor table, its sorted tail and the loss rows of a batch STRUCTURE (round 5:
    at README batch sizes the host enqueue is the step).  A second batch of the same structure with other values must hit the memo and
    give the very loss, logits and gradients a policy with the memo off gives; a batch of another structure must miss; with
    deterministic scatters everything is compared bit for bit.  Text ids on the device take part (K_DEVID rows), ragged groups too."""
    from neko_amd import ops
    from neko_amd.policy import gato_policy as gp
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=100, context_len=64)
    m, _ = make_policy(cfg, 7, train=False)

    def batch(seed, wide=False):
        g = torch.Generator().manual_seed(seed)
        b = []
        for i in range(5):
            if i % 3 == 0:
                b.append({"continuous_obs": torch.randn(5, 6 if wide else 4, generator=g).to(DEV), "continuous_actions": torch.randn(5, 2, generator=g).to(DEV)})
            elif i % 3 == 1:
                b.append({"images": torch.randint(0, 255, (2, 3, 32, 32), generator=g, dtype=torch.uint8),
                          "discrete_actions": torch.randint(0, 8, (2, 1), generator=g, dtype=torch.int32).to(DEV)})
            else:
                b.append({"text": torch.randint(0, 100, (1, 11), generator=g).to(DEV)})
        return b

    def run(b, rg):
        m.ragged_groups = rg
        m.zero_grad(set_to_none=True)
        m._flat.zero_grad()
        m.image_embedding.eval()                       # patch positions without the random draw
        logits, loss = m(b, compute_loss=True, return_logits=(rg == 0))
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), (logits.detach().clone() if rg == 0 else None), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    prev_det, ops.SCATTER_DET = ops.SCATTER_DET, True
    prev_n = gp.LAYOUT_CACHE
    try:
        for rg in (0, 2):
            gp.LAYOUT_CACHE = 0
            m._layout_cache.clear()
            ref = [run(batch(s), rg) for s in (1, 2)]
            assert len(m._layout_cache) == 0
            gp.LAYOUT_CACHE = 8
            got = [run(batch(s), rg) for s in (1, 2)]
            assert len(m._layout_cache) == 1                                   # second batch: same structure, a hit
            run(batch(3, wide=True), rg)
            assert len(m._layout_cache) == 2                                   # other structure: a miss
            for (l0, lg0, g0), (l1, lg1, g1) in zip(ref, got):
                assert l0 == l1
                assert lg0 is None or torch.equal(lg0, lg1)
                assert g0.keys() == g1.keys() and all(torch.equal(g0[k], g1[k]) for k in g0)
    finally:
        ops.SCATTER_DET, gp.LAYOUT_CACHE = prev_det, prev_n
        m.ragged_groups = 0


def test_batch_prepared_without_grad_still_gives_position_and_separator_gradients():
    """a batch prepared under no_grad (prefetch thread, evaluation helper) carries a placeholder instead of the
    host-sorted (key, row) tail; differentiating it later must not run the sorted segment sums over that placeholder (the position and
    separator gradients were silently zero) -- the flag travels with the _Prepared, it is not inferred from the tensor's size."""
    from neko_amd.policy import gato_policy as gp
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=100, context_len=64)
    m, _ = make_policy(cfg, 11, train=False)
    g = torch.Generator().manual_seed(4)
    batch = [{"continuous_obs": torch.randn(5, 4, generator=g).to(DEV), "continuous_actions": torch.randn(5, 2, generator=g).to(DEV)},
             {"images": torch.randint(0, 255, (2, 3, 32, 32), generator=g, dtype=torch.uint8),
              "discrete_actions": torch.randint(0, 8, (2, 1), generator=g, dtype=torch.int32).to(DEV)}]
    m.image_embedding.eval()
    prev_n, gp.LAYOUT_CACHE = gp.LAYOUT_CACHE, 0
    try:
        def grads(prepare_without_grad):
            m.zero_grad(set_to_none=True)
            m._flat.zero_grad()
            if prepare_without_grad:
                with torch.no_grad():
                    pr = m._prepare(batch, 0)
                assert not pr.sorted_tail
            else:
                pr = m._prepare(batch, 0)
                assert pr.sorted_tail
            m._loss_from_prepared(pr).backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        ref, got = grads(False), grads(True)
    finally:
        gp.LAYOUT_CACHE = prev_n
    for k in ("pos_embed_observation.weight", "separator_token"):
        assert float(ref[k].abs().max()) > 0
        sc = float(ref[k].abs().max())
        assert float((got[k] - ref[k]).abs().max()) <= 2e-5 * sc, k
    for k in ref:
        sc = float(ref[k].abs().max()) + 1e-12
        assert float((got[k] - ref[k]).abs().max()) <= 1e-4 * sc, k
