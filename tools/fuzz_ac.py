#!/usr/bin/env python3
"""GPU fuzz of the AC sweep: random R / L / C ladders and bridges, frequencies on, next to and away from the resonances of
their L - C pairs, against the oracle (status and values).  Not a test: a soak run for the dense partial-pivoting fallback."""
import math, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spicey_amd import abi
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import AcHandle
from oracle.pyoracle import OracleBackend

ob = OracleBackend()
rng = random.Random(11)
worst, dense, t0 = 0.0, 0, time.time()
for it in range(int(os.environ.get("FUZZ_N", "60"))):
    n = rng.choice([3, 5, 8, 13, 21, 40, 80, 150])
    lines = ["* random RLC", "V1 n0 0 AC 1"]
    res = []
    for k in range(n):
        a, b = f"n{k}", f"n{k+1}"
        kind = rng.choice("RLCRLC")
        val = {"R": 10 ** rng.uniform(0, 4), "L": 10 ** rng.uniform(-6, -2), "C": 10 ** rng.uniform(-9, -5)}[kind]
        lines.append(f"{kind}s{k} {a} {b} {val!r}")
        kind2 = rng.choice("RLC")
        val2 = {"R": 10 ** rng.uniform(1, 5), "L": 10 ** rng.uniform(-6, -2), "C": 10 ** rng.uniform(-9, -5)}[kind2]
        lines.append(f"{kind2}g{k} {b} 0 {val2!r}")
        if kind == "L" and kind2 == "C": res.append(1 / (2 * math.pi * math.sqrt(val * val2)))
        if kind == "C" and kind2 == "L": res.append(1 / (2 * math.pi * math.sqrt(val * val2)))
    lines += [".ac lin 2 1 2", ".end", ""]
    flat = abi.flatten(parseNetlist("\n".join(lines)))
    fs = [10 ** rng.uniform(1, 7) for _ in range(4)]
    for f0 in res[:3]: fs += [f0, f0 * (1 + 1e-9), f0 * (1 - 1e-12), f0 * (1 + 1e-5)]
    freqs = np.array(fs)
    vph = np.ones(flat.nV, np.complex128)
    ref = ob.run_ac(flat, freqs, vph)
    h = AcHandle(flat)
    got = h.run(freqs, vph)
    dense += h.info()["tail_levels"]
    h.close()
    if got["status"] != ref["status"]:
        print("STATUS", it, n, got["status"], got["detail"], ref["status"], flush=True); continue
    if ref["status"] == 0:
        e = float((np.abs(got["out_v"] - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12)).max())
        worst = max(worst, e)
        if e > 0.3: print("LARGE", it, n, "%.3g" % e, "dense", dense, flush=True)
print("AC FUZZ DONE circuits 60 dense solves", dense, "worst err/tol %.3g" % worst, "t=%.0fs" % (time.time() - t0))
