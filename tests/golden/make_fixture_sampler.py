"""Generate tests/golden/g9_sampler.pt by running the REFERENCE's control-task sampler.

Build container only.  ``gato/tasks/control_task.py`` and ``gato/training/trainer.py`` are imported unmodified; the
third-party packages they import at module level and this image lacks (gymnasium, minari, wandb, webdataset, peft ...)
are replaced by empty stub modules that carry just the names the reference binds (``gym.spaces.Box`` / ``Discrete``,
``minari.dataset.minari_dataset.EpisodeData``).  The dataset handed to the reference is a plain object with the three
private members its ``sample_episodes`` touches (control_task.py:327-340).

    python tests/golden/make_fixture_sampler.py

G9 = for several (env geometry, prompt mix, seed) cases: the episode arrays, the seeds and the list of dicts the
reference's ``ControlTask.sample_batch`` / ``sample_batch_configurable`` / ``Trainer.sample_control_batch`` returned.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REFERENCE_ROOT = os.environ.get("NEKO_REFERENCE_ROOT", "/root/reference")


class Box:
    def __init__(self, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(shape), dtype

    def sample(self):
        return np.zeros(self.shape, dtype=self.dtype)


class Discrete:
    def __init__(self, n):
        self.n, self.shape = n, ()

    def sample(self):
        return np.int64(0)


class EpisodeData:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def install_stubs():
    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    spaces = mod("gymnasium.spaces", Box=Box, Discrete=Discrete)
    mod("gymnasium", spaces=spaces, Env=object)
    mod("minari", MinariDataset=object)
    mod("minari.dataset")
    mod("minari.dataset.minari_dataset", EpisodeData=EpisodeData)

    class _Any(types.ModuleType):          # any attribute is a placeholder class (wandb.init, peft.LoraConfig, ...)
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})

    import accelerate  # noqa: F401  -- probe their optional back-ends (wandb, ...) BEFORE the stubs exist
    import transformers  # noqa: F401
    import ref_shims
    ref_shims.install(128)               # transformers 5.x / tokenizer shims for gato.policy (imported by the trainer)
    for name in ("wandb", "webdataset", "peft", "gdown"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Any(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class FakeData:
    def __init__(self, eps):
        self.eps = eps

    def get_episodes(self, indices):
        return [dict(id=int(i), total_timesteps=len(self.eps[int(i)]["actions"]), **self.eps[int(i)]) for i in indices]


class FakeDataset:
    def __init__(self, eps, seed):
        self._data = FakeData(eps)
        self._episode_indices = np.arange(len(eps))
        self._generator = np.random.default_rng(seed)
        self.total_episodes = len(eps)

    def __iter__(self):
        return (EpisodeData(**d) for d in self._data.get_episodes(self._episode_indices))


class FakeEnv:
    def __init__(self, obs_space, act_space):
        self.observation_space, self.action_space = obs_space, act_space


def make_episodes(rng, n_eps, obs_shape, act_kind, n_act, lens, discrete_obs=False):
    eps = []
    for e in range(n_eps):
        T = int(lens[e % len(lens)])
        if discrete_obs:
            obs = rng.integers(0, 9, size=(T,)).astype(np.int64)
        elif len(obs_shape) == 1:
            obs = rng.standard_normal((T,) + tuple(obs_shape)).astype(np.float32)
        else:
            obs = rng.integers(0, 256, size=(T,) + tuple(obs_shape)).astype(np.uint8)
        if act_kind == "box":
            act = (rng.random((T, n_act)) * 2 - 1).astype(np.float32)
        else:
            act = rng.integers(0, 4, size=(T,)).astype(np.int64)
        eps.append({"observations": obs, "actions": act, "rewards": rng.standard_normal(T).astype(np.float32)})
    return eps


CASES = [
    # name, obs space, act space, episode lengths, context/max_tokens
    dict(name="halfcheetah-like", obs=("box", (17,)), act=("box", 6), lens=[60, 9, 33, 10, 200], max_tokens=240),
    dict(name="hopper-like", obs=("box", (11,)), act=("box", 3), lens=[40, 16, 7, 90], max_tokens=240),
    dict(name="ALE/breakout-like", obs=("box", (10, 12)), act=("disc", 4), lens=[30, 50, 8], max_tokens=60),
    # (channel-last RGB cannot be constructed in the reference: its ctor permutes an un-batched sample, :77,386)
    dict(name="rgb-first", obs=("box", (3, 6, 5)), act=("disc", 4), lens=[12, 25], max_tokens=40),
    dict(name="discrete-obs", obs=("disc", 9), act=("box", 2), lens=[15, 40, 5], max_tokens=48),
]


def build_case(c, ControlTask, seed):
    rng = np.random.default_rng(1000 + seed)
    discrete_obs = c["obs"][0] == "disc"
    eps = make_episodes(rng, 7, () if discrete_obs else c["obs"][1], c["act"][0], c["act"][1] if c["act"][0] == "box" else 1,
                        c["lens"], discrete_obs)
    osp = Discrete(c["obs"][1]) if discrete_obs else Box(c["obs"][1], np.float32 if len(c["obs"][1]) == 1 else np.uint8)
    asp = Box((c["act"][1],)) if c["act"][0] == "box" else Discrete(c["act"][1])
    args = types.SimpleNamespace(patch_size=4 if c["name"] == "rgb-first" else 16)
    if c["name"] == "ALE/breakout-like":
        args.patch_size = 8
    task = ControlTask(c["name"], FakeEnv(osp, asp), FakeDataset(eps, 77 + seed), c["max_tokens"], args,
                       training_prompt_len_proportion=0.5, share_prompt_episodes=(seed % 2 == 0),
                       top_k_prompting=3 if seed == 1 else None)
    return task, eps, args


def cpu(dicts):
    """Detached copies; image observations (float32 holding exact 0..255 integers) are stored as uint8."""
    out = []
    for d in dicts:
        o = {}
        for k, v in d.items():
            v = v.clone()
            if k == "images":
                assert v.dtype == torch.float32 and bool((v == v.round()).all()) and 0 <= float(v.min()) and float(v.max()) <= 255
                v = v.to(torch.uint8)
            o[k] = v
        out.append(o)
    return out


def main():
    install_stubs()
    from gato.tasks.control_task import ControlTask
    from gato.training.trainer import Trainer

    out = {"cases": []}
    for ci, c in enumerate(CASES):
        for seed in (0, 1):
            task, eps, args = build_case(c, ControlTask, seed)
            rec = {"case": c, "seed": seed, "episodes": eps, "patch_size": args.patch_size,
                   "tokens_per_timestep": task.tokens_per_timestep, "obs_str": task.obs_str,
                   "action_str": task.action_str, "top_ids": None if task.top_ids is None else task.top_ids.copy(),
                   "calls": []}
            np.random.seed(500 + 10 * ci + seed)
            for (vanilla, prompted) in [(3, {}), (1, {"end": 2, "uniform": 1}), (0, {"start": 2}), (2, {"uniform": 2})]:
                if sum(prompted.values()) + vanilla > 7:
                    continue
                try:
                    r = task.sample_batch(vanilla, dict(prompted), "cpu", max_tokens=c["max_tokens"])
                    rec["calls"].append({"kind": "sample_batch", "vanilla": vanilla, "prompted": prompted, "out": cpu(r)})
                except ValueError as e:        # np.random.randint(0, <=0): the reference raises on too-short episodes
                    rec["calls"].append({"kind": "sample_batch", "vanilla": vanilla, "prompted": prompted,
                                         "raises": "ValueError"})
            r = task.sample_batch_configurable(1, "cpu", [1.0], ["end"], max_tokens=c["max_tokens"],
                                               share_prompt_episodes=True, ep_ids=task.top_ids)
            rec["calls"].append({"kind": "eval_prompt", "out": cpu(r)})
            out["cases"].append(rec)

    # Trainer.sample_control_batch over three tasks (trainer.py:211-250)
    tasks, recs = [], []
    for c in CASES[:3]:
        t, eps, args = build_case(c, ControlTask, 0)
        tasks.append(t)
        recs.append({"case": c, "episodes": eps, "patch_size": args.patch_size})
    fake_self = types.SimpleNamespace(tasks=tasks, device="cpu",
                                      args=types.SimpleNamespace(prompt_ep_proportion=0.25, sequence_length=60))
    np.random.seed(4242)
    batches = [cpu(Trainer.sample_control_batch(fake_self, bs)) for bs in (5, 8, 3)]
    out["trainer"] = {"tasks": recs, "np_seed": 4242, "prompt_ep_proportion": 0.25, "sequence_length": 60,
                      "batch_sizes": [5, 8, 3], "batches": batches}
    path = os.path.join(HERE, "g9_sampler.pt")
    torch.save(out, path)
    n_calls = sum(len(r["calls"]) for r in out["cases"])
    print(f"g9_sampler: {os.path.getsize(path) / 1024:.1f} KiB, {len(out['cases'])} task instances, {n_calls} calls, "
          f"{sum(len(b) for b in batches)} trainer-sampled episodes")


if __name__ == "__main__":
    main()
