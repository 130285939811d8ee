#!/usr/bin/env python3
"""AC sweep probe: n_inst swept instances x n_freq frequencies of a synthetic ladder / mesh on the GPU."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spicey_amd import abi, synth
from spicey_amd import ac as sac
from spicey_amd.lib import AcHandle

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--inst", type=int, nargs="+", default=[1, 64])
ap.add_argument("--freqs", type=int, default=201)
ap.add_argument("--threads", type=int, nargs="+", default=[0])
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--no-resident", action="store_true", help="one workgroup per (instance, frequency) even for large batches")
args = ap.parse_args()
freqs = np.array(sac.logspace(1e3, 1e8, (args.freqs - 1) / 5.0))[: args.freqs]
for ni in args.inst:
    flat, _, _, _ = synth.chain_batch("rc_ladder", args.n, range(1, ni + 1), tran=".tran 1e-6 3e-5")
    for T in args.threads:
        h = AcHandle(flat, threads=T, no_resident=args.no_resident)
        best = None
        for rep in range(3):
            t0 = time.time()
            r = h.run(freqs, np.array([1.0 + 0j]), want_currents=True)
            wall = time.time() - t0
            assert r["status"] == 0, r["detail"]
            best = r["kernel_ms"] if best is None else min(best, r["kernel_ms"])
        info = h.info()
        rec = dict(n=args.n, inst=ni, freqs=len(freqs), T=info["threads"], mode="resident sweep" if info["interpreter"] == 2 else "one workgroup per solve",
                   resident_tasks=info["resident_tasks"], streamed_tasks=info["streamed_tasks"], lds=info["lds_bytes"], nnz_lu=info["nnz_lu"], levels=info["n_levels"],
                   kernel_ms=best, solves=ni * len(freqs), solves_per_s=ni * len(freqs) / (best * 1e-3), wall_ms=wall * 1e3)
        # roofline in the terms of SURVEY.md §8(d), complex: every matrix / factor entry is 16 bytes, 4 complex vectors of
        # Nvar, one complex value per recorded node and element; fill-free nnz(L+U) = nnzA + 2 like the transient figure
        nnz_a, nvar = info["nnz_a"], info["n_var"]
        ncur = flat.nR + flat.nC + flat.nL + flat.nV
        b_solve = 16 * (3 * nnz_a + 2 * (nnz_a + 2)) + 4 * (2 * nnz_a + 2) + 64 * nvar + 16 * (flat.n_nodes + ncur)
        rec["roofline"] = dict(bound="hbm", algorithmic_bytes_per_solve=b_solve, achieved=b_solve * rec["solves_per_s"] / 1e9, peak=8000.0, unit="GB/s",
                               frac=b_solve * rec["solves_per_s"] / 8e12,
                               note="formula bytes of a streaming implementation; the sweep keeps matrix and factors of a solve in LDS: real traffic is the "
                                    "16 B per recorded value it writes plus the program it re-reads from L2")
        if args.check and ni * len(freqs) * args.n <= 3e6 and args.n <= 300:
            from oracle.pyoracle import OracleBackend
            ref = OracleBackend().run_ac(flat, freqs, np.array([1.0 + 0j]))
            rec["err_over_tol"] = float((np.abs(r["out_v"] - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12)).max())
        print(json.dumps(rec), flush=True)
        h.close()
