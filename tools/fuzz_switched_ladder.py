#!/usr/bin/env python3
"""GPU soak run: long ladders with series switches and shunt diodes (switch iterations + tridiagonal top + row records) through
several kernel variants, against the oracle: status, iteration counts, switch states, values."""
import sys, numpy as np, random
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from spicey_amd import abi
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import HipBackend
from oracle.pyoracle import OracleBackend
ob = OracleBackend(); rng = random.Random(3); worst = 0
for it in range(12):
    n = rng.choice([80, 200, 520, 1000])
    L = ["* switched ladder", ".model SW SW(Ron=1 Roff=1e6 Vt=2.5 Vh=0.2)", ".model DM D(Is=1e-14 N=1)",
         "V1 n1 0 PULSE(0 5 0 1e-6 1e-6 4e-6 1e-5)", "VC ctl 0 PULSE(0 5 2e-6 1e-6 1e-6 3e-6 8e-6)"]
    for k in range(1, n):
        if k % rng.choice([37, 50, 97]) == 0: L.append(f"S{k} n{k} n{k+1} ctl 0 SW")
        else: L.append(f"R{k} n{k} n{k+1} {100*(1+0.1*rng.random()):.6g}")
        L.append(f"C{k} n{k+1} 0 {1e-9*(1+0.1*rng.random()):.6g}")
        if rng.random() < 0.5: L.append(f"D{k} n{k+1} 0 DM")
    L += [".tran 1e-6 1.2e-5", ".end", ""]
    ckt = parseNetlist("\n".join(L))
    dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
    flat = abi.flatten(ckt); src = abi.source_table(ckt, dt, steps)
    ref = ob.run(flat, steps, dt, src)
    for kw in (dict(), dict(no_rows=True), dict(no_pcr=True), dict(threads=256)):
        be = HipBackend(**kw); got = be.run(flat, steps, dt, src)
        assert got["status"] == ref["status"] == 0, (got["detail"], ref.get("detail"))
        assert np.array_equal(got["iters"], ref["iters"]) and np.array_equal(got["state"]["S_ison"], ref["state"]["S_ison"])
        scale = max(1.0, float(np.abs(ref["out_v"]).max()))
        e = float((np.abs(got["out_v"] - ref["out_v"]) / (1e-9*np.abs(ref["out_v"]) + 1e-12*scale)).max())
        worst = max(worst, e); assert e <= 1.0, (n, kw, e)
    print(it, n, "switches", flat.nS, "max iters", int(ref["iters"].max()), "pcr", be.info["pcr_rows"], "ok worst %.3g" % worst, flush=True)
print("SWITCH FUZZ DONE worst", worst)
