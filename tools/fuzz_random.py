#!/usr/bin/env python3
"""GPU soak run: the random netlists of tests/random_circuits.py beyond the seeds the test-suite uses, against the oracle;
differences above the budget are arbitrated by the 80-bit replay like in the tests.  Not a test."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from random_circuits import random_netlist
from spicey_amd import abi
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import HipBackend
from oracle.pyoracle import OracleBackend
import hp_reference

ob = OracleBackend()
lo, hi = int(os.environ.get("FUZZ_LO", "100")), int(os.environ.get("FUZZ_HI", "400"))
ran = skipped = arbitrated = bad = marginal = 0
t0 = time.time()
for fl in (False, True):
    for seed in range(lo, hi):
        ckt = parseNetlist(random_netlist(seed, floating_sources=fl))
        dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
        flat = abi.flatten(ckt); src = abi.source_table(ckt, dt, steps)
        ref = ob.run(flat, steps, dt, src)
        if ref["status"] == 0 and ref["iters"].max() >= 20: skipped += 1; continue
        got = HipBackend().run(flat, steps, dt, src)
        ran += 1
        if got["status"] != ref["status"]: bad += 1; print("STATUS", fl, seed, got["status"], got["detail"], ref["status"], flush=True); continue
        if ref["status"] != 0: continue
        if not np.array_equal(got["iters"], ref["iters"]):
            # a control voltage within rounding of a switch threshold: the 80-bit replay decides whether the double-precision
            # ORACLE itself is on the fence there (its iteration counts differ from the replay's too) — then nobody is wrong
            _, hp_it = hp_reference.run(flat, steps, dt, src)
            if np.array_equal(np.asarray(hp_it).ravel(), ref["iters"].ravel()): bad += 1; print("ITERS", fl, seed, flush=True)
            else: marginal += 1; print("MARGINAL switch threshold (oracle and 80-bit replay disagree as well)", fl, seed, flush=True)
            continue
        scale = max(1.0, float(np.nanmax(np.abs(ref["out_v"]))))
        e = (np.abs(got["out_v"] - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12 * scale)).max()
        if e > 1.0:
            hp, _ = hp_reference.run(flat, steps, dt, src)
            tol = 1e-9 * np.abs(hp) + 1e-12 * scale
            e_ref = (np.abs(ref["out_v"][0] - hp) / tol).max(); e_dev = (np.abs(got["out_v"][0] - hp) / tol).max()
            arbitrated += 1
            if e_dev > max(1.0, 4.0 * e_ref): bad += 1; print("ACCURACY", fl, seed, "device %.3g reference %.3g" % (e_dev, e_ref), flush=True)
print("RANDOM FUZZ DONE ran", ran, "skipped", skipped, "arbitrated", arbitrated, "marginal", marginal, "bad", bad, "t=%.0fs" % (time.time() - t0))
