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
    # stale-binary guard: the library carries the hash of the sources it was built from, and build() keys on it (content, not mtimes)
    want = g.source_hash(g.sim_sources(), g.SIM_FLAGS)
    assert g.library_stamp(H.LIB_PATH) == want and want.encode() in lib.go1sim_version()


def test_build_rebuilds_a_library_whose_stamp_does_not_match_the_sources(tmp_path, monkeypatch):
    """build_ppo_hip() calls the compiler for a library built from other sources (here: a copy of the shipped one with its stamp overwritten)
    and does not for one whose stamp matches; the stamp reader refuses an absent, an unstamped and an ambiguous file"""
    import shutil
    import subprocess
    import __graft_entry__ as g
    g.build_ppo_hip()
    good = os.path.join(g.CSRC, "libgo1ppo.so")
    want = g.source_hash(g.ppo_sources(), g.PPO_FLAGS)
    assert g.library_stamp(good) == want
    # a scratch csrc directory: the sources as they are + the library under test
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for f in g.ppo_sources():
        if os.path.dirname(f) == g.CSRC:
            shutil.copy(f, csrc / os.path.basename(f))
    lib = csrc / "libgo1ppo.so"
    blob = open(good, "rb").read()
    lib.write_bytes(blob.replace(g.STAMP + want.encode(), g.STAMP + b"0" * 16))
    assert g.library_stamp(str(lib)) == "0" * 16 != want
    calls = []

    def fake_compiler(cmd, cwd=None):
        calls.append(cmd)
        out = cmd[cmd.index("-o") + 1]
        assert ('-DGO1_SOURCE_HASH="%s"' % want) in cmd
        shutil.copy(good, out)                        # "compiles" the current sources

    patched = [str(csrc / os.path.basename(f)) if os.path.dirname(f) == g.CSRC else f for f in g.ppo_sources()]
    monkeypatch.setattr(g, "ppo_sources", lambda: patched)
    monkeypatch.setattr(g, "CSRC", str(csrc))
    monkeypatch.setattr(subprocess, "check_call", fake_compiler)
    assert g.source_hash(g.ppo_sources(), g.PPO_FLAGS) == want          # same bytes, same names: same hash
    g.build_ppo_hip()
    assert len(calls) == 1 and g.library_stamp(str(lib)) == want         # the stale library was rebuilt ...
    g.build_ppo_hip()
    assert len(calls) == 1                                               # ... and an up-to-date one is left alone
    # the reader: absent / unstamped / two different stamps in one file
    assert g.library_stamp(str(tmp_path / "absent.so")) is None
    (tmp_path / "unstamped.so").write_bytes(b"xx" + g.STAMP + b"unstamped" + b"yy")
    assert g.library_stamp(str(tmp_path / "unstamped.so")) is None
    (tmp_path / "two.so").write_bytes(g.STAMP + b"0" * 16 + b" ... " + g.STAMP + want.encode())
    assert g.library_stamp(str(tmp_path / "two.so")) is None
    (tmp_path / "twice_the_same.so").write_bytes(g.STAMP + want.encode() + b" ... " + g.STAMP + want.encode())
    assert g.library_stamp(str(tmp_path / "twice_the_same.so")) == want


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


# ---- libgo1ppo.so (include/go1ppo.h): fused PPO-update kernels ------------------------------------------------
PPO_HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "go1ppo.h")


def ppo_declared_functions():
    src = re.sub(r"/\*.*?\*/", "", open(PPO_HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(go1ppo_\w+)\s*\(", src)))


def test_ppo_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from go1_gym_learn.ppo_cse import fused
    assert ppo_declared_functions() == sorted(fused.EXPORTED_SYMBOLS)
    lib = fused.load_library(g.build_ppo_hip())
    for name in ppo_declared_functions():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.go1ppo_version()
    # argument validation happens before any launch: callable without a GPU
    assert lib.go1ppo_wgrad(None, 0, None, 0, 0, 0, 0, None, 0, None, None) == -1
    assert lib.go1ppo_elu_fwd(None, 0, 0, 0, None, 0, 0, None, 0, 0, None) == -1
    assert lib.go1ppo_grad_reduce(None, None, 0, None) == -1
    assert lib.go1ppo_opt_prestep_pieces(None, 0, None, 0, 1.0, None, None, None, None, 1.0, 0.01, 1e-5, 1e-2, None) == -1


def test_ppo_loss_args_mirror_matches_header():
    """field order of the ctypes mirror == the C struct (names in declaration order)."""
    from go1_gym_learn.ppo_cse import fused
    src = re.sub(r"/\*.*?\*/", "", open(PPO_HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct \{(.*?)\} Go1PpoLossArgs;", src, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip().lstrip("*") for n in re.sub(r"^(const\s+)?\w+\s*\*?", "", decl, count=1).split(",")]
    assert names == [f[0] for f in fused.LossArgs._fields_]


def test_ppo_wgrad_problem_mirror_matches_header():
    import ctypes as C
    from go1_gym_learn.ppo_cse import fused
    src = re.sub(r"/\*.*?\*/", "", open(PPO_HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct \{([^}]*?)\} Go1PpoWgradProblem;", src, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip().lstrip("*") for n in re.sub(r"^(const\s+)?\w+\s*\*?", "", decl, count=1).split(",")]
    assert names == [f[0] for f in fused.WgradProblem._fields_]
    assert C.sizeof(fused.WgradProblem) == 96


@pytest.mark.parametrize("cname,mirror,size", [("Go1PpoGemmArgs", "GemmArgs", 96), ("Go1PpoMlp2Fwd", "Mlp2Fwd", 80), ("Go1PpoMlp2Bwd", "Mlp2Bwd", 88),
                                                ("Go1PpoGradPiece", "GradPiece", 56)])
def test_ppo_new_struct_mirrors_match_header(cname, mirror, size):
    """field order and size of the ctypes mirrors of the GEMM / LDS-resident MLP argument structs."""
    import ctypes as C
    from go1_gym_learn.ppo_cse import fused
    src = re.sub(r"/\*.*?\*/", "", open(PPO_HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct(?: \w+)? \{([^}]*?)\} %s;" % cname, src, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip().lstrip("*") for n in re.sub(r"^(const\s+)?\w+\s*\*?", "", decl, count=1).split(",")]
    assert names == [f[0] for f in getattr(fused, mirror)._fields_]
    assert C.sizeof(getattr(fused, mirror)) == size


def test_ppo_adam_extras_mirror_size():
    """Go1PpoAdamExtras (transposed weight copies kept by the optimiser step): the ctypes mirror has the size the library asserts at
    compile time (csrc/go1ppo.hip static_assert: 56 bytes)."""
    import ctypes as C
    from go1_gym_learn.ppo_cse import fused
    assert C.sizeof(fused.AdamExtras) == 56 and C.sizeof(fused._AdamTranspose) == 24
    assert [f[0] for f in fused.AdamExtras._fields_] == ["num_transposes", "_pad", "transpose"]


def test_fused_update_fails_loudly_without_library(tmp_path):
    from go1_gym_learn.ppo_cse import fused
    with pytest.raises(fused.Go1PpoLibraryMissing):
        fused.load_library(str(tmp_path / "missing.so"))


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: no Python source of the product package imports it, loads its libraries or names its build directory
    (comments may mention it); the only users outside tests/ are __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
    import ast
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "walk-these-ways_amd")
    offenders = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(root, f)
            tree = ast.parse(open(path).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                elif isinstance(node, ast.Constant) and isinstance(node.value, str) and not isinstance(getattr(node, "parent", None), ast.Expr):
                    if "pyoracle" in node.value or "oracle/_build" in node.value or "libgo1oracle" in node.value:
                        offenders.append((path, node.lineno, node.value[:60]))
                if any(n.split(".")[0] in ("oracle", "pyoracle") for n in names):
                    offenders.append((path, node.lineno, names))
    assert not offenders, offenders
