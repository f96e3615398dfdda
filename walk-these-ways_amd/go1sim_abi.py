"""ctypes mirror of include/go1sim.h, generated at import time by parsing the header.

The structs (`Go1SimConfig`, `Go1SimBuffers`) and the enum constants are read from the one
authoritative C header so that the Python host side can never drift from the C-ABI.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, "..", "include", "go1sim.h")

_CT = {
    "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64, "float": ctypes.c_float, "int": ctypes.c_int,
    "uint8_t": ctypes.c_uint8, "int16_t": ctypes.c_int16,
}


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _parse(header):
    src = _strip_comments(open(header).read())
    consts = {}
    for m in re.finditer(r"#define\s+(\w+)\s+(\d+)\s", src):
        consts[m.group(1)] = int(m.group(2))
    for body in re.findall(r"enum\s+\w+\s*\{(.*?)\}", src, flags=re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [s.strip() for s in item.split("=")]
                nxt = int(val)
            else:
                name = item
            consts[name] = nxt
            nxt += 1
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = decl.replace("const ", "")
            mm = re.match(r"(\w+)\s*(\*?)\s*(.*)", decl)
            ctype, star, rest = mm.group(1), mm.group(2), mm.group(3)
            for var in rest.split(","):
                var = var.strip()
                ptr = bool(star)
                if var.startswith("*"):
                    ptr, var = True, var[1:].strip()
                am = re.match(r"(\w+)\s*\[(\w+)\]", var)
                if ptr:
                    fields.append((var, ctypes.c_void_p))
                elif am:
                    n = am.group(2)
                    n = consts[n] if n in consts else int(n)
                    fields.append((am.group(1), _CT[ctype] * n))
                else:
                    fields.append((var, _CT[ctype]))
        structs[m.group(3)] = fields
    return consts, structs


CONSTS, _STRUCTS = _parse(HEADER)
globals().update(CONSTS)


class Go1SimConfig(ctypes.Structure):
    _fields_ = _STRUCTS["Go1SimConfig"]


class Go1SimBuffers(ctypes.Structure):
    _fields_ = _STRUCTS["Go1SimBuffers"]


BUFFER_FIELDS = [f[0] for f in _STRUCTS["Go1SimBuffers"]]

REWARD_IDS = {k[len("GO1_REW_"):].lower(): v for k, v in CONSTS.items() if k.startswith("GO1_REW_")}
PRIV_IDS = {k[len("GO1_PRIV_"):].lower(): v for k, v in CONSTS.items()
            if k.startswith("GO1_PRIV_") and k != "GO1_PRIV_COUNT"}
