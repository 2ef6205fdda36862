import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # same runtime options as bench.py / train.py (before torch loads)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need the MI355X AND the in-tree HIP library: on a box without them a plain `pytest` run skips them
    (with the reason) instead of failing.  On a GPU box a missing library is an error, not a skip: the product path
    must not pass by doing nothing."""
    if not any("gpu" in it.keywords for it in items):
        return
    import torch
    if torch.cuda.is_available():
        lib = os.path.join(ROOT, "neko_amd", "csrc", "libneko_hip.so")
        if not os.path.exists(lib):
            raise pytest.UsageError(f"{lib} is missing on a GPU box: run `python -c 'import __graft_entry__ as g; g.build()'`")
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)
    return load
