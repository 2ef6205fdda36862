"""Control-task rollouts through the HIP policy (neko_amd.tasks.control_task.ControlTask.evaluate, the restatement
of gato/tasks/control_task.py:104-176): prompt from the end of a stored episode, one predict_control call per
environment step (KV-cached), context trimmed to whole timesteps.  The sampler itself is pinned bit-exactly against
the reference on the CPU (tests/test_host_cpu.py, fixture G9); here the loop runs on the GPU with toy environments."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neko_oracle as O  # noqa: E402

DEV = "cuda"


class ToyEnv:
    """Deterministic dynamics; records the actions it receives."""

    def __init__(self, obs_space, act_space, horizon):
        self.observation_space, self.action_space, self.horizon = obs_space, act_space, horizon
        self.actions, self.t = [], 0

    def _obs(self):
        if len(self.observation_space.shape) == 1:
            return np.sin(np.arange(self.observation_space.shape[0], dtype=np.float32) + self.t).astype(np.float32)
        g = np.arange(np.prod(self.observation_space.shape)).reshape(self.observation_space.shape)
        return ((g * 7 + 13 * self.t) % 256).astype(np.uint8)

    def reset(self):
        self.t, self.actions = 0, []
        return self._obs(), {}

    def step(self, a):
        self.actions.append(np.array(a, copy=True))
        self.t += 1
        return self._obs(), 1.5, self.t >= self.horizon, False, {}


def _policy(ctx):
    from neko_amd.policy.gato_policy import GatoPolicy
    cfg = O.OracleConfig(embed_dim=64, layers=2, heads=2, text_tokens=128, context_len=ctx)
    m = GatoPolicy(DEV, 64, 2, 2, 0.0, resid_mid_channels=128, context_len=ctx, text_tokenizer=128)
    m.transformer.drop.p = 0.0
    m.load_state_dict(O.init_state_dict(cfg, 9))
    m.eval()
    return m


def test_continuous_rollout_with_prompt_and_window_trim():
    from neko_amd.tasks.control_task import BoxSpace, ControlTask, Episode, EpisodeStore
    rng = np.random.default_rng(0)
    env = ToyEnv(BoxSpace((5,)), BoxSpace((2,)), horizon=7)
    eps = [Episode(rng.standard_normal((T, 5)).astype(np.float32), (rng.random((T, 2)) * 2 - 1).astype(np.float32),
                   rng.standard_normal(T).astype(np.float32), id=i) for i, T in enumerate((12, 30, 9))]
    task = ControlTask("toy-continuous", env, EpisodeStore(eps, seed=3), 40, types.SimpleNamespace(patch_size=16),
                       top_k_prompting=2)
    assert task.tokens_per_timestep == 8                       # 5 obs + SEP + 2 actions: 5 timesteps fit 40 positions
    m = _policy(40)
    with torch.no_grad():
        res = task.evaluate(m, n_iterations=2, deterministic=True)
    assert res == {"mean_return": pytest.approx(1.5 * 7), "mean_episode_len": 7.0}
    acts = np.stack(env.actions)
    assert acts.shape == (7, 2) and np.isfinite(acts).all() and (np.abs(acts) <= 1.0).all()
    # the captured-graph decode and the eager cached decode pick the same action tokens (both are pinned to the
    # reference's predict_control by fixture G12, tests/test_decode_gpu.py)
    np.random.seed(1)
    d = task.sample_batch_configurable(1, DEV, [1.0], ["end"], max_tokens=40, ep_ids=task.top_ids)[0]
    with torch.no_grad():
        a_graph = m.predict_control(d, task=task, deterministic=True)
        os.environ["NEKO_DECODE_GRAPH"] = "0"
        try:
            a_eager = m.predict_control(d, task=task, deterministic=True)
        finally:
            os.environ.pop("NEKO_DECODE_GRAPH")
    assert torch.equal(a_graph, a_eager)
    with torch.no_grad():
        res0 = task.evaluate(m, n_iterations=1, promptless_eval=True)
    assert res0["mean_episode_len"] == 7.0


def test_atari_like_rollout_grayscale_discrete():
    from neko_amd.tasks.control_task import BoxSpace, ControlTask, DiscreteSpace, Episode, EpisodeStore
    rng = np.random.default_rng(1)
    env = ToyEnv(BoxSpace((20, 28), dtype=np.uint8), DiscreteSpace(4), horizon=4)
    eps = [Episode(rng.integers(0, 256, (T, 20, 28)).astype(np.uint8), rng.integers(0, 4, (T,)), id=i)
           for i, T in enumerate((6, 11))]
    task = ControlTask("ALE/toy", env, EpisodeStore(eps, seed=5), 48, types.SimpleNamespace(patch_size=16))
    assert task.obs_str == "images" and task.observation_tokens == 4 and task.tokens_per_timestep == 6     # 32x32 padded
    m = _policy(48)
    with torch.no_grad():
        res = task.evaluate(m, n_iterations=1, deterministic=True)
    assert res["mean_episode_len"] == 4.0 and res["mean_clipped_return"] == pytest.approx(4.0)
    assert all(0 <= int(a) < 4 for a in env.actions)


def test_trainer_samples_prompted_control_batches_and_trains():
    from neko_amd.tasks.control_task import BoxSpace, ControlTask, Episode, EpisodeStore, sample_control_batch
    rng = np.random.default_rng(2)
    tasks = []
    for name, n_obs, n_act in (("a", 5, 2), ("b", 3, 1)):
        eps = [Episode(rng.standard_normal((T, n_obs)).astype(np.float32), (rng.random((T, n_act)) * 2 - 1).astype(np.float32),
                       id=i) for i, T in enumerate((40, 25, 33, 60))]
        tasks.append(ControlTask(name, ToyEnv(BoxSpace((n_obs,)), BoxSpace((n_act,)), 3), EpisodeStore(eps, seed=1), 64,
                                 types.SimpleNamespace(patch_size=16)))
    np.random.seed(0)
    batch = sample_control_batch(tasks, 6, 0.5, DEV, 64)
    assert len(batch) == 6 and all(d["continuous_obs"].is_cuda for d in batch)
    lens = sorted(d["continuous_obs"].shape[0] * (d["continuous_obs"].shape[1] + 1 + d["continuous_actions"].shape[1]) for d in batch)
    assert lens[-1] <= 64
    m = _policy(64)
    m.train()
    m.ragged_groups = 2
    _, loss = m(batch, compute_loss=True, return_logits=False)
    loss.backward()
    assert torch.isfinite(loss)


def test_text_task_evaluate_on_the_hip_policy():
    from neko_amd.tasks.text_task import TokenTextTask
    rng = np.random.default_rng(3)
    corpus = {"train": [rng.integers(0, 128, 70).tolist() for _ in range(6)],
              "test": [rng.integers(0, 128, n).tolist() for n in (20, 31, 9)]}     # (a 33-token document would leave a 1-token chunk: the reference raises on it)
    task = TokenTextTask(corpus, 32)
    m = _policy(40)
    np.random.seed(5)
    with torch.no_grad():
        res = task.evaluate(m, num_examples_to_test=3)
    assert np.isfinite(res["loss"]) and abs(res["perplexity"] - float(np.exp(res["loss"]))) < 1e-3 * res["perplexity"]
    batch = task.sample_batch(4)
    m.train()
    _, loss = m(batch, compute_loss=True, return_logits=False)
    assert torch.isfinite(loss)


def test_train_py_on_an_episode_file(tmp_path, monkeypatch):
    """train.py end to end on real-format data: an EpisodeStore .npz sampled with the reference's prompted sampler,
    mixed with synthetic text, bucketed layout, 4 optimisation steps."""
    import train
    from neko_amd.tasks.control_task import Episode, EpisodeStore
    from neko_amd.training.arguments import parse_args
    rng = np.random.default_rng(4)
    eps = [Episode(rng.standard_normal((T, 6)).astype(np.float32), (rng.random((T, 2)) * 2 - 1).astype(np.float32),
                   rng.standard_normal(T), id=i) for i, T in enumerate((50, 70, 35, 90))]
    path = str(tmp_path / "toy.npz")
    EpisodeStore(eps).save_npz(path)
    monkeypatch.chdir(tmp_path)
    a = parse_args(["--embed_dim", "64", "--layers", "2", "--heads", "2", "--sequence_length", "90", "--batch_size", "6",
                    "--training_steps", "4", "--log_eval_freq", "4", "--warmup_steps", "2", "--text_prop", "0.34",
                    "--text_vocab_size", "128", "--control_datasets", path, "--prompt_ep_proportion", "0.5",
                    "--ragged_groups", "2", "--resid_mid_channels", "128"])
    train.main(a)


def test_host_batches_give_the_same_loss_as_device_batches():
    """ControlTask(host_batches=True) hands CPU tensors to the policy (one pinned upload instead of two device tensors
    per example): identical packing, identical loss."""
    from neko_amd.tasks.control_task import BoxSpace, ControlTask, Episode, EpisodeStore
    rng = np.random.default_rng(6)
    eps = [Episode(rng.standard_normal((T, 5)).astype(np.float32), (rng.random((T, 2)) * 2 - 1).astype(np.float32), id=i)
           for i, T in enumerate((30, 45, 12))]
    m = _policy(64)
    losses = []
    for host in (False, True):
        task = ControlTask("toy", ToyEnv(BoxSpace((5,)), BoxSpace((2,)), 3), EpisodeStore(eps, seed=2), 64,
                           types.SimpleNamespace(patch_size=16), host_batches=host)
        np.random.seed(9)
        batch = task.sample_batch(2, {"end": 1}, DEV, max_tokens=64)
        assert all(t.is_cuda != host for d in batch for t in d.values())
        with torch.no_grad():
            losses.append(float(m(batch, compute_loss=True, return_logits=False)[1]))
    assert losses[0] == losses[1]


def test_text_is_refused_when_the_reducer_was_told_there_is_none():
    """GradReducer.declare_unused_rows("embed_token.weight", 0, text_tokens) skips reducing the text rows; a batch that
    contains text would then silently diverge across ranks, so the policy refuses it."""
    m = _policy(64)
    m._dp = types.SimpleNamespace(no_text_declared=True, group_ready=lambda *_: None)
    ctl = [{"continuous_obs": torch.randn(3, 4).to(DEV), "continuous_actions": (torch.rand(3, 2) * 2 - 1).to(DEV)}]
    _, loss = m(ctl, compute_loss=True, return_logits=False)
    assert torch.isfinite(loss)
    with pytest.raises(RuntimeError, match="declare_unused_rows"):
        m(ctl + [{"text": [1, 2, 3]}], compute_loss=True, return_logits=False)
    m._dp = None


def test_caption_and_vqa_tasks_on_the_hip_policy():
    import random
    from neko_amd.tasks.caption_task import TokenCaptionTask, TokenVqaTask
    rng = np.random.default_rng(8)
    img = lambda: torch.tensor(rng.integers(0, 256, (1, 3, 32, 32)).astype(np.uint8))
    ids = lambda n: rng.integers(0, 128, n).tolist()
    cap = {p: [{"image": img(), "text": ids(6)} for _ in range(3)] for p in ("train", "test")}
    vqa = {p: [{"image": img(), "question": ids(4), "answers": [ids(2), ids(3)]} for _ in range(3)] for p in ("train", "test")}
    m = _policy(48)
    random.seed(1)
    for task in (TokenCaptionTask(cap), TokenVqaTask(vqa)):
        batch = task.sample_batch(3)
        m.train()
        _, loss = m(batch, compute_loss=True, return_logits=False)
        assert torch.isfinite(loss)
        m.eval()
        with torch.no_grad():
            res = task.evaluate(m, num_examples_to_test=2)
        assert np.isfinite(res["loss"]) and res["perplexity"] > 1.0
