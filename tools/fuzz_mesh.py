#!/usr/bin/env python3
"""GPU soak run: R / C / diode meshes of many shapes through the large-instance paths (global workspace, cooperating
workgroups, dense fronts at several cuts) against the oracle and against each other (bit-identical for every group size)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import HipBackend
from oracle.pyoracle import OracleBackend

ob = OracleBackend()
rng = random.Random(9)
worst, t0 = 0.0, time.time()
for it in range(int(os.environ.get("FUZZ_N", "14"))):
    rows, cols = rng.choice([(8, 8), (12, 7), (16, 16), (20, 11), (24, 24), (30, 17), (33, 33)])
    ckt = parseNetlist(synth.rcd_mesh(rows, cols, seed=rng.randrange(1, 10000), tran=".tran 1e-6 6e-6"))
    dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
    flat = abi.flatten(ckt); src = abi.source_table(ckt, dt, steps)
    ref = ob.run(flat, steps, dt, src)
    first = None
    for kw in (dict(force_global=True), dict(force_global=True, wgs_per_inst=4), dict(force_global=True, wgs_per_inst=16, front_cut=rng.choice([3, 5, 7])),
               dict(force_global=True, wgs_per_inst=7, front_cut=4), dict()):
        be = HipBackend(**kw); got = be.run(flat, steps, dt, src)
        assert got["status"] == ref["status"] == 0, (rows, cols, kw, got["detail"])
        scale = max(1.0, float(np.abs(ref["out_v"]).max()))
        e = float((np.abs(got["out_v"] - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12 * scale)).max())
        worst = max(worst, e)
        assert e <= 1.0, (rows, cols, kw, e)
        if kw.get("front_cut"):
            if first is None: first = (kw["front_cut"], got["out_v"])
        # group sizes without fronts share one summation order
        if kw == dict(force_global=True): plain = got["out_v"]
        if kw == dict(force_global=True, wgs_per_inst=4): assert np.array_equal(plain, got["out_v"]), (rows, cols, "G=4 differs from G=1")
    print(it, f"{rows}x{cols}", "ok  worst %.3g  t=%.0fs" % (worst, time.time() - t0), flush=True)
print("MESH FUZZ DONE worst err/tol %.3g" % worst)
