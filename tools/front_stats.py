#!/usr/bin/env python3
"""Front tree of a synthetic circuit above a level cut (CPU only, through the test emulator's build of the symbolic
phase): sizes, workspace, and the critical path of a G-workgroup proportional-mapping schedule."""
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
ap.add_argument("--cut", type=int, default=12)
ap.add_argument("--G", type=int, default=16)
args = ap.parse_args()
L = pyemul.lib()
i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
L.spicey_emul_front_stats.restype = C.c_int32
L.spicey_emul_front_stats.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.c_int32, C.c_int32, i32p, i64p] + [i32p] * 6
flat = abi.flatten(parseNetlist(getattr(synth, args.workload)(args.n)))
d = flat.desc()
cap = 1 << 16
meta = np.zeros(4, np.int32); ws = C.c_int64(0)
arr = [np.zeros(cap, np.int32) for _ in range(6)]
rc = L.spicey_emul_front_stats(C.byref(d), args.cut, args.G, cap, meta.ctypes.data_as(i32p), C.byref(ws), *[a.ctypes.data_as(i32p) for a in arr])
assert rc == 0, rc
nf = int(meta[0])
k0, p, q, parent, owner, seq = [a[:nf] for a in arr]
print(f"{args.workload}({args.n}) cut {meta[1]} of {meta[3]} levels: {nf} fronts, {int(p.sum())} pivots above the cut, max Mp {meta[2]}, "
      f"workspace {ws.value * 8 / 1e6:.1f} MB")
if nf:
    panels = (p + 15) // 16
    work = np.array([sum((pp + qq - i) ** 2 for i in range(pp)) for pp, qq in zip(p, q)], dtype=np.float64)
    print("p histogram:", {k: int(((p >= lo) & (p < hi)).sum()) for k, (lo, hi) in {"1": (1, 2), "2-4": (2, 5), "5-16": (5, 17), "17-64": (17, 65), "65+": (65, 10**9)}.items()})
    print(f"panels total {int(panels.sum())}, multiply-adds {work.sum():.3g}")
    per = np.bincount(owner, minlength=args.G)
    pan = np.bincount(owner, weights=panels, minlength=args.G)
    wk = np.bincount(owner, weights=work, minlength=args.G)
    print("fronts per workgroup:", per.tolist())
    print("panels per workgroup:", pan.astype(int).tolist())
    print("Mflop per workgroup:", (wk / 1e6).round(2).tolist())
    # root path
    f = int(np.argmax(k0 + p)); path = []
    # deepest chain by panels
    depth = np.zeros(nf)
    for i in range(nf):
        pass
    chain = np.zeros(nf)
    for i in range(nf):  # children precede parents
        chain[i] += panels[i]
        if parent[i] >= 0:
            chain[parent[i]] = max(chain[parent[i]], chain[i])
    print("critical path (panels):", int(chain.max()), " fronts on the biggest:", [(int(p[i]), int(q[i])) for i in np.argsort(-(p + q))[:8]])
