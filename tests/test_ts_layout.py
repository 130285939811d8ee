"""The struct layout the TypeScript binding (ts/abiLayout.ts) and the ctypes mirror (spicey_amd/abi.py) assume
must be the C compiler's layout of include/spicey_hip.h."""
import ctypes as C
import os
import re
import subprocess
import sys

from conftest import REPO
from spicey_amd import abi

sys.path.insert(0, os.path.join(REPO, "tools"))


def _c_offsets(tmp_path, struct_name, fields):
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{REPO}/include/spicey_hip.h"', "int main(void){"]
    for f in fields:
        src.append(f'  printf("{f} %zu\\n", offsetof({struct_name}, {f}));')
    src.append(f'  printf("__size %zu\\n", sizeof({struct_name}));')
    src.append("  return 0; }")
    c = tmp_path / f"{struct_name}.c"
    c.write_text("\n".join(src))
    exe = tmp_path / struct_name
    subprocess.run(["gcc", "-o", str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    return {k: int(v) for k, v in (line.split() for line in out.strip().splitlines())}


def test_ctypes_mirror_matches_header(tmp_path):
    for struct in (abi.SpiceyDesc, abi.SpiceyOptions, abi.SpiceyInfo):
        names = [n for n, _ in struct._fields_]
        offs = _c_offsets(tmp_path, struct.__name__, names)
        assert offs.pop("__size") == C.sizeof(struct)
        assert offs == {n: getattr(struct, n).offset for n in names}, struct.__name__


def test_ts_layout_is_current():
    import gen_ts_layout
    with open(os.path.join(REPO, "ts", "abiLayout.ts")) as f:
        assert f.read() == gen_ts_layout.render()


def test_ts_binding_declares_only_real_symbols():
    from spicey_amd import lib
    text = open(os.path.join(REPO, "ts", "spiceyHip.ts")).read()
    block = text[text.index("dlopen(libPath"):text.index("})", text.index("dlopen(libPath"))]
    used = set(re.findall(r"^\s*(spicey_[a-z_]+):", block, re.M))
    assert used and used <= set(lib.EXPORTS)
