import sys, time, numpy as np, random
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from random_circuits import series_diode_chain
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import HipBackend
from oracle.pyoracle import OracleBackend
ob = OracleBackend()
rng = random.Random(7)
worst = 0.0
t0 = time.time()
for it in range(36):
    n = rng.choice([33, 64, 65, 70, 97, 128, 129, 200, 257, 400, 511, 512, 513, 700, 1000, 1023, 1024, 1025, 1200, 1500])
    kind = rng.choice(["series", "diode_chain", "rc_ladder"])
    if kind == "series":
        text = series_diode_chain(rng.randrange(10000), n)
        ckt = parseNetlist(text)
    else:
        ckt = parseNetlist(getattr(synth, kind)(n, seed=rng.randrange(1, 10000), tran=".tran 1e-6 8e-6"))
    dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
    steps = min(steps, 8)
    flat = abi.flatten(ckt); src = abi.source_table(ckt, dt, steps)
    ref = ob.run(flat, steps, dt, src)
    for kw in (dict(), dict(geometry=2) if n <= 1024 else dict(threads=512), dict(no_rows=True), dict(no_pcr=True)):
        try:
            be = HipBackend(**kw); got = be.run(flat, steps, dt, src)
        except Exception as e:
            print("EXC", kind, n, kw, str(e)[:100]); continue
        assert got["status"] == ref["status"], (kind, n, kw, got["status"], ref["status"])
        if ref["status"] != 0: continue
        scale = max(1.0, float(np.nanmax(np.abs(ref["out_v"]))))
        e = float((np.abs(got["out_v"] - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12 * scale)).max())
        fin = np.isfinite(ref["out_i"])
        isc = max(1.0, float(np.abs(ref["out_i"][fin]).max())) if fin.any() else 1.0
        ei = float((np.abs(got["out_i"][fin] - ref["out_i"][fin]) / (1e-9 * np.abs(ref["out_i"][fin]) + 1e-12 * isc)).max()) if fin.any() else 0.0
        worst = max(worst, e, ei)
        if e > 0.5 or ei > 0.5: print("LARGE", kind, n, kw, e, ei, be.info["pcr_rows"], flush=True)
    print(it, kind, n, "ok  worst so far %.3g  t=%.0fs" % (worst, time.time() - t0), flush=True)
print("FUZZ DONE worst err/tol", worst)
