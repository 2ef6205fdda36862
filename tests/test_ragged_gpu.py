"""Length-bucketed training layout (GatoPolicy.ragged_groups): instead of left-padding every
example of a batch to the longest one (gato_policy.py:408-416) the batch is packed into a few length buckets; every
kernel but attention runs on the concatenated rows, attention once per bucket.  Padding never reaches a real row
(causal mask + additive -1e4 key bias underflow to exactly 0 in fp32), so loss and gradients must be those of the
reference layout: checked against the padded HIP path, the CPU oracle and the reference's G7b training trace.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


def make_policy(cfg, seed, train=False, dropout=0.0):
    from neko_amd.policy.gato_policy import GatoPolicy
    m = GatoPolicy(DEV, cfg.embed_dim, cfg.layers, cfg.heads, dropout, resid_mid_channels=128,
                   context_len=cfg.context_len, continuous_tokens=cfg.continuous_tokens,
                   discrete_tokens=cfg.discrete_tokens, text_tokenizer=cfg.text_tokens)
    if dropout == 0:
        m.transformer.drop.p = 0.0
    m.load_state_dict(O.init_state_dict(cfg, seed), strict=True)
    m.train(train)
    return m


def to_dev(batch):
    return [{k: (v.to(DEV) if torch.is_tensor(v) and v.dtype != torch.uint8 else v) for k, v in ex.items()}
            for ex in batch]


def ragged_batch(seed=0, vocab=128):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    ids = lambda n: torch.randint(0, vocab, (n,), generator=g).tolist()
    return [
        {"text": ids(90)},
        {"continuous_obs": torch.randn(2, 5, generator=g), "continuous_actions": r(2, 2) * 2 - 1},            # 16 tokens
        {"images": torch.floor(r(2, 3, 32, 32) * 256), "discrete_actions": torch.randint(0, 4, (2, 1), generator=g).to(torch.int32)},  # 12
        {"text": ids(88)},
        {"continuous_obs": torch.randn(5, 6, generator=g), "continuous_actions": r(5, 3) * 2 - 1},            # 50 tokens
        {"images": torch.floor(r(1, 3, 32, 48) * 256).to(torch.uint8), "text": ids(9)},                       # 6 + 9 + 1
        {"discrete_obs": torch.randint(0, 7, (3, 4), generator=g).to(torch.int32), "continuous_actions": r(3, 1) * 2 - 1},   # 18
    ]


def grads_of(m):
    return {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("hd", [32, 64, 128], ids=["hd32-head-resident", "hd64-dma-ring", "hd128-dma-ring"])
@pytest.mark.parametrize("groups", [1, 2, 3, 8])
def test_ragged_loss_and_grads_equal_padded_layout_and_oracle(groups, hd):
    """hd = 64 / 128 (configs[4]'s head width): since ABI v16 the packed layout runs ONE launch of the DMA-ring attention kernels
    too (before, one launch per length bucket)."""
    from neko_amd import ops
    assert ops.attn_varlen_supported(96, hd)
    cfg = O.OracleConfig(embed_dim=2 * hd, layers=2, heads=2, text_tokens=128, context_len=96)
    batch = ragged_batch()
    m = make_policy(cfg, 11)                     # eval mode: deterministic patch positions in both layouts
    _, loss_pad = m(to_dev(batch), compute_loss=True, return_logits=False)
    loss_pad.backward()
    g_pad = grads_of(m)
    m.zero_grad()
    m.ragged_groups = groups
    _, loss_rag = m(to_dev(batch), compute_loss=True, return_logits=False)
    if groups == 1:          # one bucket removes no padding: the policy keeps the reference layout (< 10 % rows saved)
        assert m.last_pack.segments is None
    else:
        assert m.last_pack.segments is not None and len(m.last_pack.segments) == min(groups, 6)
        assert sum(b * t for _, b, t in m.last_pack.segments) < 0.9 * len(batch) * 91
    loss_rag.backward()
    g_rag = grads_of(m)
    # (hd = 64 / 128: the DMA-ring kernels walk 64-key tiles from the first row of a sequence INCLUDING its padding, so the two layouts
    # add the same probabilities up in other groupings and a few bf16 roundings of the attention output fall differently: 2.0e-6 measured)
    assert abs(float(loss_rag) - float(loss_pad)) < (2e-6 if hd == 32 else 1e-5) * abs(float(loss_pad)), (float(loss_rag), float(loss_pad))
    assert g_rag.keys() == g_pad.keys()
    for k in g_pad:
        scale = float(g_pad[k].abs().max())
        # bf16 rounding points sit differently in the two layouts (attention walks 32-row blocks from the first row of
        # a sequence INCLUDING its padding, so a real row lands in another block position): rounding-level agreement
        assert float((g_rag[k] - g_pad[k]).abs().max()) <= 1e-2 * scale + 1e-7, k
    # and the oracle (fp32 CPU restatement of the reference) on the same batch
    loss_ref, _, grads_ref = O.loss_and_grads(O.init_state_dict(cfg, 11), cfg, batch)
    assert abs(float(loss_rag) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    for k, gr in grads_ref.items():
        if gr is None or k.endswith("c_attn.bias"):
            continue
        n = float(gr.norm())
        assert abs(float(g_rag[k].norm()) - n) < 5e-2 * n + 1e-6, k


def test_ragged_is_only_used_when_no_logits_are_returned():
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=128, context_len=96)
    m = make_policy(cfg, 11)
    m.ragged_groups = 4
    batch = to_dev(ragged_batch())
    with torch.no_grad():
        logits, loss = m(batch, compute_loss=True)                    # logits requested: the reference (B, T, V) layout
        assert m.last_pack.segments is None and tuple(logits.shape)[:2] == (len(batch), 91)
        _, loss2 = m(batch, compute_loss=True, return_logits=False)
        assert m.last_pack.segments is not None
        e, t, tg, pm = m.tokenize_input_dicts(batch)                  # public packing call keeps the reference's shapes
        assert tuple(t.shape) == (len(batch), 91)
    assert abs(float(loss) - float(loss2)) < 2e-6 * abs(float(loss))
    m.pad_seq = True
    with torch.no_grad():
        m(batch, compute_loss=True, return_logits=False)
    assert m.last_pack.segments is None                                          # pad_seq asks for context_len-wide rows


def test_ragged_with_dropout_trains_and_masks_differ_per_bucket():
    """Training mode (dropout 0.1 at every site): finite loss close to the dropout-free one, gradients for every
    parameter, and the attention dropout key differs between buckets."""
    from neko_amd import engine, ops
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=128, context_len=96)
    m = make_policy(cfg, 11, train=True, dropout=0.1)
    m.ragged_groups = 3
    batch = [ex for ex in to_dev(ragged_batch()) if ex.get("images") is None]
    _, loss = m(batch, compute_loss=True, return_logits=False)
    loss.backward()
    m.eval()
    with torch.no_grad():
        _, loss0 = m(batch, compute_loss=True, return_logits=False)
    assert torch.isfinite(loss) and abs(float(loss) - float(loss0)) < 0.2 * float(loss0)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters()
               if "image_embedding" not in n and "wte" not in n)
    d = ops.Drop(0.1, 1234)
    keys = {engine._seg_drop(d, i).key for i in range(4)}
    assert len(keys) == 4 and all(engine._seg_drop(d, i).thr == d.thr for i in range(4))


def test_g7b_trace_in_ragged_layout(golden):
    """The reference's 100-step training trace (fixture G7b: text of 100..109 tokens next to 75-token control
    episodes) reproduced with the bucketed layout: north_star gate 1e-3 relative on the loss."""
    from neko_amd.training.optim import NekoAdamW
    from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler
    f = golden("g7b_trace")
    cfg = O.OracleConfig(**f["cfg"])
    m = make_policy(cfg, f["seed"], train=True)
    m.ragged_groups = 2
    opt = NekoAdamW(m, lr=f["lr"], betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, f["warmup"], f["total_steps"], base_lr=f["lr"],
                                                   init_lr=f["init_lr"], min_lr=f["min_lr"])
    batches = [to_dev(b) for b in f["batches"]]
    losses = []
    for step in range(f["total_steps"]):
        _, loss = m.forward(inputs=batches[step % len(batches)], compute_loss=True, return_logits=False)
        assert m.last_pack.segments is not None and len(m.last_pack.segments) == 2
        loss.backward()
        opt.clip_grad_norm_(1.0)
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().tolist()
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, f["trace"]["loss"])]
    assert max(rel) < 1e-3, (max(rel), rel.index(max(rel)))


def test_ragged_full_size_mix_matches_padded_loss():
    """Metric-size model on a C5-like length mix (1024 / 494 / 289 / 240-token examples): same loss, 2x fewer rows."""
    import bench
    from neko_amd.policy.gato_policy import GatoPolicy
    torch.manual_seed(0)
    m = GatoPolicy(DEV, bench.D, bench.L, bench.H, 0.0, resid_mid_channels=128, context_len=bench.T,
                   text_tokenizer=bench.V_TEXT)
    m.transformer.drop.p = 0.0
    m.eval()
    batch = bench.make_batch("c5-mix", 16, 77, DEV)
    with torch.no_grad():
        _, l_pad = m(batch, compute_loss=True, return_logits=False)
        m.ragged_groups = 4
        _, l_rag = m(batch, compute_loss=True, return_logits=False)
    rows = sum(b * t for _, b, t in m.last_pack.segments)
    assert rows < 0.6 * 16 * 1024
    assert abs(float(l_rag) - float(l_pad)) < 1e-5 * abs(float(l_pad)), (float(l_rag), float(l_pad))
