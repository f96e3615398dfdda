"""`EmuSim`: the C-ABI of include/go1sim.h served by tests/emu/_build/libgo1sim_emu.so — the product's kernel sources
compiled for the host against the SIMT emulator — on CPU `SimBuffers`.  TEST INFRASTRUCTURE (see tests/emu/hip/hip_runtime.h)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build as emu_build  # noqa: E402
import go1sim_host as H  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = H.bind_library(ctypes.CDLL(emu_build.build()))
    return _lib


_study_libs = {}


def study_lib(defines, tag):
    """a study build of the kernel sources (tests/emu/build.py): e.g. (("GO1_PROFILE",), "_prof")"""
    if tag not in _study_libs:
        _study_libs[tag] = H.bind_library(ctypes.CDLL(emu_build.build(defines=defines, tag=tag)))
    return _study_libs[tag]


class EmuSim(H.Go1Sim):
    def __init__(self, S, buffers, device_index=0, library=None):
        assert buffers.device.type == "cpu"
        super().__init__(S, buffers, device_index, lib=library if library is not None else lib())

    def _stream(self):
        return ctypes.c_void_p(0)

    def step(self, actions):
        actions = actions.detach().float().contiguous()
        self._keep_actions = actions
        self._check(self.lib.go1sim_step(self.handle, ctypes.c_void_p(actions.data_ptr()), self._stream()), "go1sim_step")
