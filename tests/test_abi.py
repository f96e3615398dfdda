"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/go1sim.h declares
(no compute calls without a GPU), and the ctypes mirror matches the C struct sizes."""
import ctypes
import os
import re

import pytest

import go1sim_abi as abi
import go1sim_host as H

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "go1sim.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(go1sim_\w+)\s*\(", src)))


def test_header_functions_are_bound():
    assert declared_functions() == sorted(H.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build_hip()
    lib = ctypes.CDLL(H.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.go1sim_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.go1sim_version()


def test_struct_sizes_match_oracle_build(oracle_lib):
    L = oracle_lib.lib()
    assert L.go1_oracle_sizeof_config() == ctypes.sizeof(abi.Go1SimConfig)
    assert L.go1_oracle_sizeof_buffers() == ctypes.sizeof(abi.Go1SimBuffers)


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(H, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(H, "_lib", None)
    with pytest.raises(H.Go1SimLibraryMissing):
        H.load_library()


def test_bad_config_is_rejected_without_gpu():
    import __graft_entry__ as g
    g.build_hip()
    lib = ctypes.CDLL(H.LIB_PATH)
    cfg = abi.Go1SimConfig()
    cfg.abi_version = 999
    handle = ctypes.c_void_p()
    lib.go1sim_create.restype = ctypes.c_int
    assert lib.go1sim_create(ctypes.byref(cfg), ctypes.byref(abi.Go1SimBuffers()), 0, ctypes.byref(handle)) == -2
