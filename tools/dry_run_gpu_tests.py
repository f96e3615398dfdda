#!/usr/bin/env python3
"""Run `-m gpu` simulator tests WITHOUT a GPU: `go1sim_host.Go1Sim` is replaced by the SIMT emulator of tests/emu (the
unmodified kernel sources compiled for the host), device buffers stay on the CPU.  For checking the LOGIC of new GPU tests when
no hardware is at hand — the emulator computes in plain fp32 without the HIP build's contraction / approximate divides, so a
pass here does not replace the run on the MI355X.

    python tools/dry_run_gpu_tests.py tests/test_gpu_parity.py::test_ragged_env_counts_match_oracle [more node ids / -k ...]
"""
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "emu")]


def install():
    import conftest  # noqa: F401  (paths)
    import torch
    import emu_sim
    import go1sim_host as H
    H.Go1Sim = lambda S, B, dev=0: emu_sim.EmuSim(S, B)
    to = H.SimBuffers.clone_to
    H.SimBuffers.clone_to = lambda self, dev: to(self, "cpu")
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None


if __name__ == "__main__":
    os.environ["GO1_DRY_RUN_GPU_TESTS"] = "1"
    import pytest

    class Plugin:
        def pytest_configure(self, config):
            install()
    os.chdir(REPO)
    sys.exit(pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider"] + sys.argv[1:], plugins=[Plugin()]))
