#!/usr/bin/env python3
"""One mesh through chosen group sizes / front cuts (diagnostics): python tools/repro_mesh.py rows cols seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import HipBackend
rows, cols, seed = (int(x) for x in sys.argv[1:4])
ckt = parseNetlist(synth.rcd_mesh(rows, cols, seed=seed, tran=".tran 1e-6 6e-6"))
dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
flat = abi.flatten(ckt); src = abi.source_table(ckt, dt, steps)
base = None
for G in (1, 4, 7, 16):
    for cut in (3, 5, 7):
        be = HipBackend(force_global=True, wgs_per_inst=G, front_cut=cut)
        got = be.run(flat, steps, dt, src)
        same = None
        if got["status"] == 0:
            if base is None: base = {}
            same = np.array_equal(base.setdefault(cut, got["out_v"]), got["out_v"])
        print(f"G={G} cut={cut} status={got['status']} detail={got['detail']!r} fronts={be.info['n_fronts']} same_as_G1={same}", flush=True)
