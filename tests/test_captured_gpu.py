"""Captured training step (neko_amd/training/captured.py): one HIP graph per batch structure must train exactly like the
eager path (Trainer.train_step semantics, trainer.py:176-186), with the learning-rate schedule and the per-step dropout
variation coming from device memory."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


def _policy(dropout, seed=7):
    from neko_amd.policy.gato_policy import GatoPolicy
    cfg = O.OracleConfig(embed_dim=128, layers=2, heads=4, text_tokens=256, context_len=128)
    m = GatoPolicy(DEV, 128, 2, 4, dropout, resid_mid_channels=128, context_len=128, text_tokenizer=256)
    if dropout == 0:
        m.transformer.drop.p = 0.0
    m.load_state_dict(O.init_state_dict(cfg, seed))
    m.train()
    return m


def _batches():
    from neko_amd.tasks import synthetic as S
    ctl = [S.SyntheticControlTask(5, 2, 9, seed=s, device=DEV).sample_batch(6) for s in (1, 2, 3)]
    txt = [S.SyntheticTextTask(90, 256, seed=s, device=DEV).sample_batch(4) for s in (4, 5)]
    img = [S.SyntheticAtariTask(3, 32, 32, seed=s, device=DEV).sample_batch(3) for s in (6, 7)]
    # three structures, revisited: control, text, control, images, text, control, images, ...
    return [ctl[0], txt[0], ctl[1], img[0], txt[1], ctl[2], img[1], ctl[0], txt[0], img[0]]


def _opt(m):
    from neko_amd.training.optim import NekoAdamW
    from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler
    opt = NekoAdamW(m, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, 4, 40, base_lr=1e-3, init_lr=1e-5, min_lr=1e-4)
    return opt, sch


@pytest.mark.parametrize("det", [True, False])
def test_captured_steps_train_like_eager_steps(det):
    """det: the embedding-table gradients by sorted sums and the attention backward as its two kernels (what NEKO_DETERMINISTIC=1
    selects): eager and captured steps are then the same arithmetic in the same order, and the tight gates hold for EVERY step
    (ADVICE r03); with atomics the trajectories may separate from step 4 on (see below) and only the first steps are gated tightly."""
    from neko_amd import ops
    prev_det, ops.SCATTER_DET = ops.SCATTER_DET, bool(det)
    try:
        _captured_vs_eager(det)
    finally:
        ops.SCATTER_DET = prev_det


def _captured_vs_eager(det):
    from neko_amd.training.captured import CapturedTrainStep
    batches = _batches()
    m0 = _policy(0.0)
    m0.eval(); m0.train()
    opt0, sch0 = _opt(m0)
    torch.manual_seed(3)
    ref, refn = [], []
    for b in batches:
        _, loss = m0.forward(inputs=b, compute_loss=True, return_logits=False)
        loss.backward()
        refn.append(opt0.clip_grad_norm_(1.0))
        opt0.step(); sch0.step(); opt0.zero_grad()
        ref.append(loss.detach())
    m1 = _policy(0.0)
    opt1, sch1 = _opt(m1)
    cap = CapturedTrainStep(m1, opt1, sch1, grad_norm_clip=1.0)
    torch.manual_seed(3)                     # same patch-position draws (train mode) as the eager run
    got, gotn = [], []
    try:
        for b in batches:
            loss, gn = cap.step(b)
            got.append(loss); gotn.append(gn)
        torch.cuda.synchronize()
        assert cap.eager_steps == 3 and cap.replays == len(batches) - 3 and len(cap.entries) == 3
    finally:
        cap.close()
    ref, got = torch.stack(ref).cpu(), torch.stack(got).cpu()
    refn, gotn = torch.stack(refn).reshape(-1).cpu(), torch.stack(gotn).reshape(-1).cpu()
    # same kernels, same order; only the fp32 atomics of the embedding / patch-position scatters differ in their last bits,
    # and Adam normalises that noise: an entry whose gradient IS rounding noise moves by +-lr with a noise-dependent sign.
    # Two EAGER runs of this very recipe are either bit-identical or, from step 4 on, on a second trajectory (relative loss
    # differences 6e-8, 8e-6, 3.3e-4, 5e-5, 1e-7, 4.9e-4 at steps 4..9; tools/trajectory_noise_probe.py, 8 runs: 4 of each) --
    # so the tight gate covers the steps before any run can separate (incl. the first replay) and the loose one the rest.
    tight = len(batches) if det else 4
    assert torch.allclose(got[:tight], ref[:tight], rtol=2e-5, atol=0), (got, ref)
    assert torch.allclose(got, ref, rtol=5e-3, atol=0), (got, ref)
    assert torch.allclose(gotn[:tight], refn[:tight], rtol=2e-4, atol=0), (gotn, refn)
    assert torch.allclose(gotn, refn, rtol=3e-2, atol=0), (gotn, refn)
    for (k, a), (_, b2) in zip(m1.state_dict().items(), m0.state_dict().items()):
        # (small tensors whose gradient is rounding noise random-walk under Adam: only the weight matrices are compared)
        if a.dtype == torch.float32 and a.numel() >= 4096:
            assert float((a - b2).norm()) <= 2e-3 * float(b2.norm()) + 1e-6, k
    assert abs(sch1.get_last_lr()[0] - sch0.get_last_lr()[0]) < 1e-12


def test_captured_dropout_masks_change_between_replays_and_training_converges():
    from neko_amd.tasks import synthetic as S
    from neko_amd.training.captured import CapturedTrainStep
    from neko_amd.training.optim import NekoAdamW
    m = _policy(0.1)
    opt = NekoAdamW(m, lr=0.0, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0)     # lr 0: weights frozen, only the masks move
    cap = CapturedTrainStep(m, opt, None, grad_norm_clip=1.0)
    b = S.SyntheticTextTask(100, 256, seed=2, device=DEV).sample_batch(8)
    try:
        losses = torch.stack([cap.step(b)[0] for _ in range(6)]).cpu()
        assert cap.replays == 5
        assert len(set(losses.tolist())) == 6, losses          # every replay drew other masks (device-side salt)
        assert float(losses.std()) < 0.05 * float(losses.mean())
        opt.param_groups[0]["lr"] = 2e-3                        # the kernel reads lr from device memory: takes effect at once
        losses = torch.stack([cap.step(b)[0] for _ in range(30)]).cpu()
        assert cap.replays == 35
        assert torch.isfinite(losses).all() and float(losses[-5:].mean()) < float(losses[:5].mean()) - 0.3, losses
    finally:
        cap.close()


def test_eager_path_is_untouched_after_close():
    """close() unregisters the salt and the device-side lr: a following eager step behaves as before (same loss as a model
    that never saw a CapturedTrainStep)."""
    from neko_amd.tasks import synthetic as S
    from neko_amd.training.captured import CapturedTrainStep
    from neko_amd.training.optim import NekoAdamW
    b = S.SyntheticTextTask(60, 256, seed=9, device=DEV).sample_batch(4)
    m = _policy(0.0)
    opt = NekoAdamW(m, lr=1e-3)
    cap = CapturedTrainStep(m, opt, None)
    cap.step(b); cap.step(b); cap.step(b)
    cap.close()
    assert opt.lr_dev is None
    _, l1 = m.forward(inputs=b, compute_loss=True, return_logits=False)
    m2 = _policy(0.0)
    m2.load_state_dict(m.state_dict())
    _, l2 = m2.forward(inputs=b, compute_loss=True, return_logits=False)
    assert abs(float(l1.detach()) - float(l2.detach())) < 2e-6 * abs(float(l2.detach()))


def test_train_py_with_capture_step(tmp_path, monkeypatch):
    """train.py --capture_step end to end: synthetic halfcheetah-shaped control + text, 12 steps, three batch structures at
    most; the trainer's loop runs on replayed graphs and leaves the process in eager mode when it is done."""
    import train
    from neko_amd import ops
    from neko_amd.training.arguments import parse_args
    monkeypatch.chdir(tmp_path)
    a = parse_args(["--embed_dim", "64", "--layers", "2", "--heads", "2", "--sequence_length", "96", "--batch_size", "6",
                    "--training_steps", "12", "--log_eval_freq", "6", "--warmup_steps", "2", "--text_prop", "0.5",
                    "--text_vocab_size", "128", "--resid_mid_channels", "128", "--capture_step", "--dropout", "0.1"])
    try:
        train.main(a)
    finally:
        ops.set_drop_salt(None)


def test_captured_ragged_batch_replays_like_eager_steps():
    """A length-bucketed batch at hd = 32 runs ONE packed attention launch (neko_attn_*_varlen) whose row / mask offset arrays are
    built on the host.  They must be ready device tensors (or copies from a live pinned buffer) when the step is captured
    (a copy from a temporary pageable tensor inside the capture would replay from freed host memory): eager steps
    and replays of the same ragged step agree, and every replay of the captured step keeps using the packed launch."""
    from neko_amd import engine
    from neko_amd.tasks import synthetic as S
    from neko_amd.training.captured import CapturedTrainStep
    batch = (S.SyntheticTextTask(90, 256, seed=4, device=DEV).sample_batch(3)
             + S.SyntheticControlTask(5, 2, 9, seed=1, device=DEV).sample_batch(4)
             + S.SyntheticTextTask(40, 256, seed=5, device=DEV).sample_batch(2))
    assert engine.ATTN_VARLEN
    m0 = _policy(0.0)
    m0.ragged_groups = 3
    opt0, sch0 = _opt(m0)
    ref = []
    for _ in range(6):
        _, loss = m0.forward(inputs=batch, compute_loss=True, return_logits=False)
        loss.backward()
        opt0.clip_grad_norm_(1.0)
        opt0.step(); sch0.step(); opt0.zero_grad()
        ref.append(loss.detach())
    m1 = _policy(0.0)
    opt1, sch1 = _opt(m1)
    cap = CapturedTrainStep(m1, opt1, sch1, grad_norm_clip=1.0)
    got = []
    try:
        for _ in range(6):
            got.append(cap.step(batch, ragged_groups=3)[0])
        torch.cuda.synchronize()
        assert cap.replays == 5 and len(cap.entries) == 1
    finally:
        cap.close()
    ref, got = torch.stack(ref).cpu(), torch.stack(got).cpu()
    assert torch.allclose(got[:4], ref[:4], rtol=2e-5, atol=0), (got, ref)
    assert torch.allclose(got, ref, rtol=2e-3, atol=0), (got, ref)
