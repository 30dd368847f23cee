import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    so = os.path.join(ROOT, "ray-tracer-engine_amd", "csrc", "librt_engine.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.dirname(so), "-j8"], check=True)
    import oracle_py
    oracle_py.build()


@pytest.fixture(scope="session")
def rt():
    _ensure_built()
    import rt_amd
    return rt_amd.load()


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    import oracle_py
    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def gpu(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no GPU is visible (there is no CPU fallback to fall back on)")
    return torch.device("cuda:0")
