"""Build tests/emu/_build/libgo1sim_emu.so: the product's kernel sources (walk-these-ways_amd/csrc/go1sim.hip and its headers,
unmodified) compiled for the HOST against the SIMT emulator in this directory.  TEST INFRASTRUCTURE: lets the parity tests
exercise the real device code without a GPU; the product never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(REPO, "walk-these-ways_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libgo1sim_emu.so")
CLANG = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def content_hash(files, flags):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:16]


def build(force=False, defines=(), tag=""):
    """defines / tag: a study build of the same sources (e.g. ("GO1_PROFILE",), "_prof") next to the product's"""
    out = OUT.replace(".so", tag + ".so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    deps += [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "emu_runtime.cpp"), os.path.join(REPO, "include", "go1sim.h")]
    flags = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wno-everything"] + ["-D" + d for d in defines]
    want = content_hash(sorted(deps), flags)          # content, not mtimes (as __graft_entry__.build_hip): a stale emulator cannot pass for the sources
    stamp = out + ".stamp"
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [CLANG] + flags + ["-I", HERE, "-o", out, os.path.join(CSRC, "go1sim.hip"), os.path.join(HERE, "emu_runtime.cpp")]
    subprocess.check_call(cmd, cwd=CSRC)
    with open(stamp, "w") as fh:
        fh.write(want)
    return out


if __name__ == "__main__":
    print(build(force=True))
