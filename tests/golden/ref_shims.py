"""In-process shims that let the *unmodified* reference (/root/reference) import in this
container (transformers 5.x instead of the pinned 4.30.2, no gymnasium, no network).

Used ONLY by tests/golden/make_fixtures.py (fixture generation, this container only) and by
tests that are skipped when /root/reference is absent.  Nothing here is reference source: it
only patches third-party modules so the reference's own files run as they are.

Shim list follows SURVEY.md section 8(c):
  1. transformers.modeling_utils.Conv1D & friends moved/removed in transformers 5.x
     (imported at gato/transformers/trajectory_gpt2.py:37-45)
  2. gymnasium stub (gato/policy/gato_policy.py:6; only spaces.Box/Discrete identities used :564-567)
  3. AutoTokenizer.from_pretrained -> object with .vocab_size (gato_policy.py:57,60)
  4. GPT2Model.init_weights -> apply(_init_weights)   (4.30 semantics, trajectory_gpt2.py:545)
  5. GPT2Model.get_head_mask -> [None]*n              (removed in 5.x, trajectory_gpt2.py:696)
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("NEKO_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gato"))


class _FakeTokenizer:
    def __init__(self, vocab_size):
        self.vocab_size = vocab_size

    def encode(self, s):  # never used for fixtures
        raise NotImplementedError

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def install(text_vocab_size: int = 50257):
    """Apply the shims and put the reference on sys.path. Returns the reference GatoPolicy class."""
    import transformers
    import transformers.modeling_utils as mu
    from transformers.pytorch_utils import Conv1D

    # 1
    mu.Conv1D = Conv1D
    for name in ("SequenceSummary", "find_pruneable_heads_and_indices", "prune_conv1d_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, None)
    if "transformers.utils.model_parallel_utils" not in sys.modules:
        m = types.ModuleType("transformers.utils.model_parallel_utils")
        m.assert_device_map = None
        m.get_device_map = None
        sys.modules["transformers.utils.model_parallel_utils"] = m
    # 2
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")

        class Box:  # noqa: D401 - placeholder
            pass

        class Discrete:
            pass

        spaces.Box, spaces.Discrete = Box, Discrete
        gym.spaces = spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    # 3
    transformers.AutoTokenizer.from_pretrained = staticmethod(
        lambda *a, **k: _FakeTokenizer(text_vocab_size))

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import gato.transformers.trajectory_gpt2 as tg
    # 4, 5
    tg.GPT2Model.init_weights = lambda self: self.apply(self._init_weights)
    tg.GPT2Model.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    import gato.policy.gato_policy as gp
    # gato_policy imported AutoTokenizer by name before/after the patch: patch its binding too
    gp.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: _FakeTokenizer(text_vocab_size))
    return gp.GatoPolicy
