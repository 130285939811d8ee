#!/usr/bin/env python3
"""MEASURED timeline of the front tree of one large instance (GPU; SpiceyOptions.profile): per front the times at which
its children were assembled / it was done (forward) and its parent's unknowns arrived / it was solved (backward), averaged
over the solves, and the chain that ends each sweep.  The CPU replay with a cost model is tools/front_timeline.py."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import Handle

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--wgs", type=int, default=0)
ap.add_argument("--cut", type=int, default=0)
args = ap.parse_args()
ckt = parseNetlist(synth.rcd_mesh(args.rows, seed=3, tran=f".tran 1e-6 {args.steps * 1e-6!r}"))
flat = abi.flatten(ckt)
src = abi.source_table(ckt, 1e-6, args.steps)
h = Handle(flat, wgs_per_inst=args.wgs, front_cut=args.cut, profile=True)
r = h.run(args.steps, 1e-6, src, want_currents=True)
assert r["status"] == 0, r["detail"]
ticks, meta = h.front_ticks(0)
solves = r["solves"]
us = ticks.astype(np.float64) * 0.01 / solves  # 100 MHz ticks -> us, per solve
p, q, parent, owner = meta[:, 0], meta[:, 1], meta[:, 2], meta[:, 3]
nf = len(p)
print(f"rcd_mesh({args.rows}): {nf} fronts, G {h.info()['wgs_per_inst']}, {r['kernel_ms'] / (args.steps + 1) * 1000:.0f} us per step; "
      f"forward sweep ends {us[:, 1].max():.0f} us, backward {us[:, 3].max():.0f} us after the forward sweep began")
kids = [[] for _ in range(nf)]
for f in range(nf):
    if parent[f] >= 0: kids[parent[f]].append(f)
print("forward: the chain that ends last (root first): front wg p q | assembled -> done | own time | gap to the child it waited for")
f = int(np.argmax(us[:, 1]))
while True:
    c = max(kids[f], key=lambda k: us[k, 1]) if kids[f] else None
    gap = us[f, 0] - us[c, 1] if c is not None else float("nan")
    print(f"  {f:4d} wg {owner[f]:3d} p {p[f]:3d} q {q[f]:3d} | {us[f, 0]:6.1f} -> {us[f, 1]:6.1f} | {us[f, 1] - us[f, 0]:5.1f} | {gap:5.1f}")
    if c is None: break
    f = c
print("backward: the chain that ends last (leaf first): front wg p q | parent there -> solved | own time | gap to the parent's end")
f = int(np.argmax(us[:, 3]))
while f >= 0:
    pa = parent[f]
    gap = us[f, 2] - us[pa, 3] if pa >= 0 else float("nan")
    print(f"  {f:4d} wg {owner[f]:3d} p {p[f]:3d} q {q[f]:3d} | {us[f, 2]:6.1f} -> {us[f, 3]:6.1f} | {us[f, 3] - us[f, 2]:5.1f} | {gap:5.1f}")
    f = pa
h.close()
