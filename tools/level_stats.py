#!/usr/bin/env python3
"""Elimination-tree level profile of a synthetic circuit (CPU only, through the test emulator's build of the symbolic
phase): pivots and task slices per level — where a large circuit's cross-workgroup barriers go."""
import argparse, ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from emul import pyemul

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rcd_mesh")
ap.add_argument("--n", type=int, default=100)
args = ap.parse_args()
L = pyemul.lib()
i32p = C.POINTER(C.c_int32)
L.spicey_emul_level_stats.restype = C.c_int32
L.spicey_emul_level_stats.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, i32p, i32p, i32p, i32p]
flat = abi.flatten(parseNetlist(getattr(synth, args.workload)(args.n)))
d = flat.desc()
cap = 4096
nl = C.c_int32(0)
u, b, pv = (np.zeros(cap, np.int32) for _ in range(3))
assert L.spicey_emul_level_stats(C.byref(d), cap, C.byref(nl), u.ctypes.data_as(i32p), b.ctypes.data_as(i32p), pv.ctypes.data_as(i32p)) == 0
n = nl.value
print(f"{args.workload}({args.n}): {flat.n_var} unknowns, {n} levels, {int(u[:n].sum()) * 64} factor task slots")
print("level  pivots  factor_slices  backward_slices")
for l in range(n):
    if l < 12 or l >= n - 6 or l % max(1, n // 24) == 0:
        print(f"{l:5d} {pv[l]:7d} {u[l]:14d} {b[l]:16d}")
