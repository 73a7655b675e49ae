import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# several tests flip the library's OVO_* tuning knobs (monkeypatch.setenv) between launches of this one process: the library reads them once
# per process unless this is set before its first launch (ovo_amd/csrc/common.h: ovo_knobs_dynamic)
os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a GPU-less host runs the CPU suite and SKIPS the `gpu`-marked tests (they need a real MI355X)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack(bits: np.ndarray, w: int) -> np.ndarray:
    return np.unpackbits(bits, axis=-1)[..., :w].astype(bool)


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
