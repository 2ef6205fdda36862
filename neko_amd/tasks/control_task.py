"""Control-task data side of the hot path: the episode window / prompt sampler and the
rollout loop of ``gato/tasks/control_task.py`` on top of an in-memory episode store.

The reference's ``ControlTask`` (control_task.py:28-340) is built on a gymnasium ``Env`` and a Minari HDF5 dataset;
neither package (nor h5py) exists in this image, so the two third-party objects are replaced by small protocols with
the attributes the reference actually reads:

  * ``env``: ``observation_space`` / ``action_space`` (``BoxSpace`` / ``DiscreteSpace`` below, or gymnasium's own
    classes -- only ``type(space).__name__`` and ``.shape`` are used), ``reset() -> (obs, info)``,
    ``step(a) -> (obs, reward, terminated, truncated, info)``;
  * ``dataset``: ``EpisodeStore`` -- episodes as numpy arrays with Minari's ``sample_episodes`` semantics
    (``generator.choice(indices, size=n, replace=False)``, control_task.py:327-340).

What is kept literally (same numpy RNG call sequence, same slicing expressions, including the reference's quirks:
an episode shorter than the window loses its last timestep, :253-256; the 'end' prompt start may go negative and
wraps, :284-286): ``sample_batch`` (:178-207), ``sample_batch_configurable`` (:209-325), ``ControlImageTransform``
(:345-389), ``evaluate`` (:104-176, with the KV-cached ``predict_control`` of the HIP policy).  Pinned bit-exactly
against the imported reference by fixture G9 (tests/golden/make_fixture_sampler.py, tests/test_host_cpu.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


class BoxSpace:
    """Stand-in for gymnasium.spaces.Box: shape (+ bounds, unused by the path)."""

    def __init__(self, shape, low=-np.inf, high=np.inf, dtype=np.float32):
        self.shape, self.low, self.high, self.dtype = tuple(shape), low, high, dtype

    def sample(self):
        return np.zeros(self.shape, dtype=self.dtype)


class DiscreteSpace:
    """Stand-in for gymnasium.spaces.Discrete."""

    def __init__(self, n: int):
        self.n, self.shape = int(n), ()

    def sample(self):
        return np.int64(0)


def _is_box(space) -> bool:
    return type(space).__name__ in ("Box", "BoxSpace")


def _is_discrete(space) -> bool:
    return type(space).__name__ in ("Discrete", "DiscreteSpace")


def tokens_per_space(space) -> int:
    """control_task.py:19-25."""
    if _is_box(space):
        return space.shape[0]
    if _is_discrete(space):
        return 1
    raise NotImplementedError(f"Unsupported space: {space}")


class Episode:
    """The fields of minari's EpisodeData the reference reads (:252,257-258,99)."""

    def __init__(self, observations, actions, rewards=None, id: int = 0):
        self.id = id
        self.observations = np.asarray(observations)
        self.actions = np.asarray(actions)
        self.rewards = np.zeros(len(self.actions), dtype=np.float32) if rewards is None else np.asarray(rewards)
        self.total_timesteps = len(self.actions)


class EpisodeStore:
    """In-memory replacement of ``minari.MinariDataset`` for the sampler: ``total_episodes``, iteration, and
    ``sample_episodes`` drawing WITHOUT replacement from a numpy Generator (minari's ``_generator``)."""

    def __init__(self, episodes: Sequence[Episode], seed: Optional[int] = None):
        self.episodes = list(episodes)
        self.total_episodes = len(self.episodes)
        self.episode_indices = np.arange(self.total_episodes)
        self.generator = np.random.default_rng(seed)

    def __iter__(self):
        return iter(self.episodes)

    # ---- on-disk form (stands in for Minari's main_data.hdf5: h5py is not in this image) --------------------------
    # one .npz: observations / actions / rewards concatenated over episodes along axis 0 + episode_lengths
    def save_npz(self, path: str) -> None:
        np.savez_compressed(path,
                            observations=np.concatenate([e.observations for e in self.episodes], axis=0),
                            actions=np.concatenate([e.actions for e in self.episodes], axis=0),
                            rewards=np.concatenate([e.rewards for e in self.episodes], axis=0),
                            episode_lengths=np.array([e.total_timesteps for e in self.episodes], dtype=np.int64))

    @classmethod
    def from_npz(cls, path: str, seed: Optional[int] = None) -> "EpisodeStore":
        z = np.load(path)
        ends = np.cumsum(z["episode_lengths"])
        eps = [Episode(z["observations"][e - n:e], z["actions"][e - n:e], z["rewards"][e - n:e], id=i)
               for i, (n, e) in enumerate(zip(z["episode_lengths"], ends))]
        return cls(eps, seed=seed)

    def spaces(self):
        """(observation_space, action_space) inferred from the stored arrays: float -> Box, integer -> Discrete."""
        o, a = self.episodes[0].observations, self.episodes[0].actions
        if np.issubdtype(o.dtype, np.integer) and o.ndim == 1:
            osp = DiscreteSpace(int(max(e.observations.max() for e in self.episodes)) + 1)
        else:
            osp = BoxSpace(o.shape[1:], dtype=o.dtype)
        if np.issubdtype(a.dtype, np.integer):
            asp = DiscreteSpace(int(max(e.actions.max() for e in self.episodes)) + 1)
        else:
            asp = BoxSpace(a.shape[1:] if a.ndim > 1 else (1,), dtype=a.dtype)
        return osp, asp


    def sample_episodes(self, n_episodes: int, episode_indices=None) -> List[Episode]:
        """control_task.py:327-340."""
        if episode_indices is None:
            episode_indices = self.episode_indices
        idx = self.generator.choice(episode_indices, size=n_episodes, replace=False)
        return [self.episodes[int(i)] for i in idx]


class SpacesOnlyEnv:
    """An ``env`` for training without rollouts: only the two spaces (evaluate() needs a real reset/step)."""
    can_rollout = False

    def __init__(self, observation_space, action_space):
        self.observation_space, self.action_space = observation_space, action_space

    def reset(self):
        raise RuntimeError("SpacesOnlyEnv has no simulator: pass a real environment to evaluate")

    step = reset


class ControlImageTransform:
    """control_task.py:345-389: grayscale -> 3 channels, channel-last -> channel-first, zero-pad bottom/right to a
    multiple of the patch size (Atari 84x84 -> 96x96 = 36 patches)."""

    def __init__(self, env, patch_size: int = 16):
        self.env, self.patch_size = env, patch_size
        space = env.observation_space
        assert _is_box(space), "Only supports Box observation space"
        assert len(space.shape) in (2, 3), "Only supports 2D or 3D observation space"
        self.channel_first, self.grayscale = None, False
        if len(space.shape) == 3:
            assert space.shape[0] == 3 or space.shape[-1] == 3, "3 channel first or channel last"
            self.channel_first = space.shape[0] == 3
            self.height, self.width = (space.shape[1], space.shape[2]) if self.channel_first else (space.shape[0], space.shape[1])
        else:
            self.grayscale = True
            self.height, self.width = space.shape[0], space.shape[1]
        self.padding_h = (patch_size - self.height % patch_size) % patch_size
        self.padding_w = (patch_size - self.width % patch_size) % patch_size

    def transform(self, images: torch.Tensor) -> torch.Tensor:
        if self.grayscale:
            images = images.reshape(-1, 1, self.height, self.width).repeat(1, 3, 1, 1)
        elif not self.channel_first:
            images = images.permute(0, 3, 1, 2)
        return torch.nn.functional.pad(images, (0, self.padding_w, 0, self.padding_h), value=0)


class ControlTask:
    """Same constructor, attributes and methods as the reference's ControlTask (control_task.py:28-340)."""
    kind = "control"
    prompt_types = ["start", "end", "uniform"]

    def __init__(self, env_name: str, env, dataset: EpisodeStore, context_len: int, args,
                 training_prompt_len_proportion: float = 0.5, share_prompt_episodes: bool = True,
                 top_k_prompting: Optional[int] = None, host_batches: bool = False):
        #: True: sample_batch returns CPU tensors whatever `device` says.  The reference builds two device tensors per
        #: example (:304,314) -- 64 small pageable H2D copies per batch of 32; the HIP policy instead concatenates
        #: host-resident sources and uploads them in ONE pinned asynchronous copy (GatoPolicy._gather_values).
        self.host_batches = host_batches
        self.name = env_name
        self.is_atari = "ALE" in env_name
        self.env, self.dataset, self.args = env, dataset, args
        osp, asp = env.observation_space, env.action_space
        self.action_type, self.observation_type = type(asp), type(osp)        # read by GatoPolicy.predict_control (:564)
        assert _is_box(asp) or _is_discrete(asp), f"Unsupported action space: {asp}"
        assert _is_box(osp) or _is_discrete(osp), f"Unsupported observation space: {osp}"
        if _is_box(osp):
            self.obs_str = "images" if len(osp.shape) in (2, 3) else "continuous_obs"
        else:
            self.obs_str = "discrete_obs"
        patch = getattr(args, "patch_size", 16)
        self.image_transform = ControlImageTransform(env, patch) if self.obs_str == "images" else None
        self.action_str = "continuous_actions" if _is_box(asp) else "discrete_actions"
        self.action_tokens = tokens_per_space(asp)
        if self.obs_str == "images":
            shp = self.image_transform.transform(torch.as_tensor(osp.sample())).shape
            self.observation_tokens = shp[-1] // patch * shp[-2] // patch
        else:
            self.observation_tokens = tokens_per_space(osp)
        self.tokens_per_timestep = self.action_tokens + self.observation_tokens + 1          # + separator
        assert context_len >= self.tokens_per_timestep, \
            f"Context length must be at least {self.tokens_per_timestep} for env {env_name}"
        assert 0 <= training_prompt_len_proportion <= 1
        self.training_prompt_len_proportion = training_prompt_len_proportion
        self.share_prompt_episodes = share_prompt_episodes
        self.top_k_prompting = top_k_prompting
        if top_k_prompting is not None:
            assert 0 < top_k_prompting <= dataset.total_episodes, "top k must be between 0 and total episodes for all datasets"
            ep_returns = np.array([ep.rewards.sum() for ep in dataset])
            self.top_ids = np.argsort(ep_returns)[-top_k_prompting:]
        else:
            self.top_ids = None

    # ---- sampling ------------------------------------------------------------------------------------------
    def sample_batch(self, vanilla_batch_size: int, prompted_batch_sizes: Optional[Dict[str, int]] = None,
                     device="cpu", max_tokens: int = 1024) -> List[dict]:
        """control_task.py:178-207."""
        prompted_batch_sizes = prompted_batch_sizes or {}
        proportions: List[float] = [0] * vanilla_batch_size
        types: List[Optional[str]] = [None] * vanilla_batch_size
        for prompt_type, n in prompted_batch_sizes.items():
            assert prompt_type in self.prompt_types
            proportions += [self.training_prompt_len_proportion] * n
            types += [prompt_type] * n
        return self.sample_batch_configurable(len(proportions), device, proportions, types, max_tokens=max_tokens,
                                              share_prompt_episodes=self.share_prompt_episodes)

    def sample_batch_configurable(self, batch_size: int, device, prompt_proportions: list, prompt_types: list,
                                  max_tokens: int = 1024, share_prompt_episodes: bool = True, ep_ids=None) -> List[dict]:
        """control_task.py:209-325: a main window of round(n*(1-p)) timesteps from each sampled episode, optionally
        preceded by a prompt of the remaining timesteps taken from the start / the end / a uniform position of the
        same (or the next) episode."""
        num_timesteps = max_tokens // self.tokens_per_timestep
        obs_l, act_l = [], []
        all_episodes = self.dataset.sample_episodes(batch_size, ep_ids)
        main_episodes = all_episodes
        prompt_episodes = all_episodes if share_prompt_episodes else all_episodes[1:] + all_episodes[:1]
        t_prompts = []
        for i, ep in enumerate(main_episodes):
            t_main = round(num_timesteps * (1 - prompt_proportions[i]))
            t_prompts.append(num_timesteps - t_main)
            ep_len = ep.total_timesteps
            if t_main >= ep_len:
                start, end = 0, ep_len - 1
            else:
                start = np.random.randint(0, ep_len - t_main)
                end = start + t_main
            obs_l.append(ep.observations[start:end, ])
            act_l.append(ep.actions[start:end, ])
        for i, ep in enumerate(prompt_episodes):
            ep_len, t_prompt, ptype = ep.total_timesteps, t_prompts[i], prompt_types[i]
            if t_prompt > 0:
                assert ptype in self.prompt_types, "Invalid prompt type"
                if t_prompt >= ep_len:
                    p_start, p_end = 0, ep_len - 1
                if ptype == "start":
                    p_start, p_end = 0, t_prompt - 1
                elif ptype == "end":
                    p_end = ep_len - 1
                    p_start = p_end - t_prompt + 1
                elif ptype == "uniform":
                    p_start = np.random.randint(0, ep_len - t_prompt)
                    p_end = p_start + t_prompt - 1
                obs_l[i] = np.concatenate([ep.observations[p_start:(p_end + 1), ], obs_l[i]], axis=0)
                act_l[i] = np.concatenate([ep.actions[p_start:(p_end + 1), ], act_l[i]], axis=0)
        out = []
        osp, asp = self.env.observation_space, self.env.action_space
        if self.host_batches:
            device = "cpu"
        for i in range(batch_size):
            obs = torch.tensor(obs_l[i], dtype=torch.float32 if _is_box(osp) else torch.int32, device=device)
            if self.image_transform is not None:
                obs = self.image_transform.transform(obs)
            act = torch.tensor(act_l[i], dtype=torch.float32 if _is_box(asp) else torch.int32, device=device)
            act = act.reshape(act.shape[0], self.action_tokens)
            out.append({self.action_str: act, self.obs_str: obs})
        return out

    # ---- rollouts ------------------------------------------------------------------------------------------
    def evaluate(self, model, n_iterations: int = 1, deterministic: bool = True, promptless_eval: bool = False) -> dict:
        """control_task.py:104-176: serial rollouts, prompted by the end of a (top-k) episode; the per-step action
        comes from ``predict_control`` (KV-cached on the HIP policy)."""
        returns, clipped, lens = [], [], []
        pol = getattr(model, "module", model)
        context_timesteps = pol.context_len // self.tokens_per_timestep
        for _ in range(n_iterations):
            observation, _info = self.env.reset()
            input_dict = self.sample_batch_configurable(1, pol.device, [1.0], ["end"], max_tokens=pol.context_len,
                                                        share_prompt_episodes=True, ep_ids=self.top_ids)[0]
            action_type = input_dict[self.action_str].dtype
            if promptless_eval:
                input_dict = None
            done, ep_return, ep_clipped, ep_len = False, 0.0, 0.0, 0
            while not done:
                new_obs = torch.as_tensor(observation, device=pol.device).unsqueeze(0)
                if self.image_transform is not None:
                    new_obs = self.image_transform.transform(new_obs)
                pad = torch.zeros(1, self.action_tokens, device=pol.device, dtype=action_type)
                if input_dict is not None:
                    new_obs = new_obs.to(input_dict[self.obs_str].dtype)
                    input_dict[self.obs_str] = torch.cat([input_dict[self.obs_str], new_obs], dim=0)
                    input_dict[self.action_str] = torch.cat([input_dict[self.action_str], pad], dim=0)
                else:
                    input_dict = {self.obs_str: new_obs, self.action_str: pad}
                input_dict[self.obs_str] = input_dict[self.obs_str][-context_timesteps:, ]
                input_dict[self.action_str] = input_dict[self.action_str][-context_timesteps:, ]
                action = pol.predict_control(input_dict, task=self, deterministic=deterministic)
                input_dict[self.action_str][-1, ] = action
                observation, reward, terminated, truncated, _info = self.env.step(action.cpu().numpy())
                done = terminated or truncated
                ep_return += reward
                ep_clipped += float(np.clip(reward, -1.0, 1.0))
                ep_len += 1
            returns.append(ep_return); clipped.append(ep_clipped); lens.append(ep_len)
        metrics = {"mean_return": float(np.mean(returns)), "mean_episode_len": float(np.mean(lens))}
        if self.is_atari:
            metrics["mean_clipped_return"] = float(np.mean(clipped))
        return metrics


def sample_control_batch(control_tasks: Sequence[ControlTask], batch_size: int, prompt_ep_proportion: float, device,
                         max_tokens: int) -> List[dict]:
    """Trainer.sample_control_batch (gato/training/trainer.py:211-250): tasks drawn round by round without
    replacement, a ``prompt_ep_proportion`` share of the episodes prompted, half of those from the episode end and
    half from a uniform position; same numpy RNG call sequence."""
    n_tasks = len(control_tasks)
    sampled: List[int] = []
    while len(sampled) < batch_size:
        max_n = min(n_tasks, batch_size - len(sampled))
        sampled.extend(np.random.choice(np.arange(n_tasks), size=max_n, replace=False).tolist())
    n_prompted = round(batch_size * prompt_ep_proportion)
    prompt_indices = np.random.choice(batch_size, size=n_prompted, replace=False).tolist()
    end_indices = np.random.choice(prompt_indices, size=round(len(prompt_indices) / 2), replace=False).tolist()
    uniform_indices = [i for i in prompt_indices if i not in end_indices]
    dicts: List[dict] = []
    for i, task in enumerate(control_tasks):
        total = vanilla = 0
        prompted: Dict[str, int] = {}
        for type_index, task_index in enumerate(sampled):
            if task_index == i:
                total += 1
                if type_index in end_indices:
                    prompted["end"] = prompted.get("end", 0) + 1
                elif type_index in uniform_indices:
                    prompted["uniform"] = prompted.get("uniform", 0) + 1
                else:
                    vanilla += 1
        if total > 0:
            dicts.extend(task.sample_batch(vanilla, prompted, device, max_tokens=max_tokens))
    return dicts
