import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instruct-video-to-video_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The CPU oracle's torch kernels stop scaling at ~16 threads and regress badly at the 256 the GPU boxes' hosts offer
    (tools/cpu_threads_probe.py: one UNet branch 8 s at 16 threads, 317 s at 256); round 5's GPU suite spent 800 of its 1 040 s in three
    oracle-bound optical-flow tests for that reason."""
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return {k: v for k, v in np.load(os.path.join(GOLDEN, name + ".npz")).items()}
    return load
