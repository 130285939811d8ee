#!/usr/bin/env python3
"""GPU soak run: batches of parameter-swept ladders / chains (BASELINE config 4's shape) of many sizes and instance counts,
every geometry the handle picks, second run continuing from the first one's state; sampled instances against the oracle."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.lib import Handle
from oracle.pyoracle import OracleBackend

ob = OracleBackend()
rng = random.Random(5)
worst, t0 = 0.0, time.time()
for it in range(int(os.environ.get("FUZZ_N", "24"))):
    kind = rng.choice(["rc_ladder", "diode_chain"])
    n = rng.choice([48, 64, 100, 129, 300, 512, 777, 1000, 1024])
    ni = rng.choice([1, 2, 3, 7, 64, 300, 511, 512, 513, 700])
    steps = rng.choice([5, 9, 17])
    seeds = [rng.randrange(1, 100000) for _ in range(ni)]
    flat, dt, _, src = synth.chain_batch(kind, n, seeds, tran=f".tran 1e-6 {steps * 1e-6!r}")
    h = Handle(flat)
    info = h.info()
    a = h.run(steps, dt, src)
    b = h.run(steps, dt, src)            # continues from the state a left
    st = h.state()
    h.close()
    assert a["status"] == 0 and b["status"] == 0, (kind, n, ni, a["detail"], b["detail"])
    for k in sorted(set([0, ni - 1, ni // 2, rng.randrange(ni)])):
        one, _, _, _ = synth.chain_batch(kind, n, [seeds[k]], tran=f".tran 1e-6 {steps * 1e-6!r}")
        r1 = ob.run(one, steps, dt, src)
        for key, arr in (("C_vprev", one.C_vprev), ("L_iprev", one.L_iprev), ("D_vdprev", one.D_vdprev)):
            arr[...] = r1["state"][key]
        r2 = ob.run(one, steps, dt, src)
        for got, ref in ((a, r1), (b, r2)):
            scale = max(1.0, float(np.abs(ref["out_v"]).max()))
            e = float((np.abs(got["out_v"][k] - ref["out_v"][0]) / (1e-9 * np.abs(ref["out_v"][0]) + 1e-12 * scale)).max())
            fin = np.isfinite(ref["out_i"][0])
            isc = max(1.0, float(np.abs(ref["out_i"][0][fin]).max()))
            ei = float((np.abs(got["out_i"][k][fin] - ref["out_i"][0][fin]) / (1e-9 * np.abs(ref["out_i"][0][fin]) + 1e-12 * isc)).max())
            worst = max(worst, e, ei)
            assert e <= 1.0 and ei <= 1.0, (kind, n, ni, k, e, ei)
        assert np.allclose(st["C_vprev"][k], r2["state"]["C_vprev"][0], rtol=1e-9, atol=1e-12)
    print(it, kind, n, "x", ni, "steps", steps, "geom", info["geometry"], "T", info["threads"], "pcr", info["pcr_rows"], "ok  worst %.3g  t=%.0fs" % (worst, time.time() - t0), flush=True)
print("BATCH FUZZ DONE worst err/tol %.3g" % worst)
