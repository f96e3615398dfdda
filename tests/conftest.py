import os
import sys

import pytest

# GEMM auto-tuning (PPO_Args.use_tuned_gemms) would tune every small test shape: keep the suite fast and deterministic
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "0")

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, os.path.join(REPO, "oracle"), REPO, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import pyoracle
    pyoracle.build()
    return pyoracle
