#!/usr/bin/env python3
"""Forward sweep of the front tree under the G-workgroup schedule, SIMULATED on the CPU with the per-front / per-panel
costs the section timers measured (tools/mesh_probe.py --profile): which chain of fronts ends last, and how much of it
is hand-over waits, serialisation behind other fronts of the same workgroup, panels, staged (non-LDS) fronts."""
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
ap.add_argument("--cut", type=int, default=10)
ap.add_argument("--G", type=int, default=128)
ap.add_argument("--panel", type=float, default=7.2, help="us per 16-pivot panel of an LDS-resident front")
ap.add_argument("--panel-staged", type=float, default=23.0, help="us per panel of a front staged through the workspace")
ap.add_argument("--own", type=float, default=4.7, help="us: zero + own entries (runs before the wait for foreign children)")
ap.add_argument("--child", type=float, default=1.8, help="us per child's extend-add")
ap.add_argument("--store", type=float, default=2.1, help="us: store + post")
ap.add_argument("--handover", type=float, default=2.0, help="us between a post and the waiter seeing it")
ap.add_argument("--lds-doubles", type=int, default=19456)
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
assert L.spicey_emul_front_stats(C.byref(d), args.cut, args.G, cap, meta.ctypes.data_as(i32p), C.byref(ws), *[a.ctypes.data_as(i32p) for a in arr]) == 0
nf = int(meta[0])
k0, p, q, parent, owner, seq = [a[:nf].astype(int) for a in arr]
Pp = (p + 15) // 16 * 16
Mp = Pp + q
staged = Mp * (Mp + 17) + 512 > args.lds_doubles
kids = [[] for _ in range(nf)]
for f in range(nf):
    if parent[f] >= 0: kids[parent[f]].append(f)
lists = [[] for _ in range(args.G)]
for f in np.argsort(owner * (nf + 1) + seq): lists[owner[f]].append(int(f))
end = np.full(nf, -1.0); start = np.zeros(nf); waited = np.zeros(nf); why = [None] * nf
t_wg = np.zeros(args.G); pos = [0] * args.G
done = 0
while done < nf:
    progressed = False
    for g in range(args.G):
        while pos[g] < len(lists[g]):
            f = lists[g][pos[g]]
            if any(end[c] < 0 for c in kids[f]): break
            t = t_wg[g]; start[f] = t
            t += args.own
            last = None
            for c in kids[f]:  # in order; a foreign child is waited for right before its turn
                if owner[c] != g:
                    ready = end[c] + args.handover
                    if ready > t: waited[f] += ready - t; t = ready; last = c
                t += args.child
            why[f] = last
            t += (Pp[f] // 16) * (args.panel_staged if staged[f] else args.panel) + args.store
            end[f] = t; t_wg[g] = t; pos[g] += 1; done += 1; progressed = True
    assert progressed, "schedule deadlock"
root = int(np.argmax(end))
print(f"{args.workload}({args.n}) cut {meta[1]}, G {args.G}: {nf} fronts, {int(staged.sum())} staged; simulated forward sweep {end.max():.0f} us")
# walk the chain that determined the end time
f = root; chain = []
while f is not None:
    chain.append(f)
    if why[f] is not None: f = why[f]
    else:
        # started right after the previous front of the same workgroup (or at 0)
        g = owner[f]; i = lists[g].index(f)
        f = lists[g][i - 1] if i > 0 else None
print("chain (root first): front, workgroup, p, q, panels, staged, start, end, waited")
for f in chain:
    print(f"  {f:4d} wg {owner[f]:3d}  p {p[f]:3d} q {q[f]:3d}  panels {Pp[f] // 16}  {'STAGED' if staged[f] else 'lds   '}  {start[f]:6.1f} -> {end[f]:6.1f}  waited {waited[f]:5.1f}")
own_chain = [f for f in chain]
print(f"on the chain: {sum(Pp[f] // 16 for f in chain)} panels, {sum(1 for f in chain if staged[f])} staged fronts "
      f"({sum((Pp[f] // 16) * args.panel_staged for f in chain if staged[f]):.0f} us), {len(chain)} fronts")
