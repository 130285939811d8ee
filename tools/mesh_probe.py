#!/usr/bin/env python3
"""Large-instance probe: rcd_mesh(rows) on the GPU (global workspace when L+U exceeds LDS), optionally with several
workgroups cooperating on the instance (--wgs)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import Handle

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[20, 32, 50])
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--wgs", type=int, nargs="+", default=[0])
ap.add_argument("--cuts", type=int, nargs="+", default=[0], help="front_cut values: 0 auto, -1 no fronts, > 0 explicit level")
ap.add_argument("--inst", type=int, default=1)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
for rows in args.rows:
    t0 = time.time()
    ckt = parseNetlist(synth.rcd_mesh(rows, seed=3, tran=f".tran 1e-6 {args.steps * 1e-6!r}"))
    steps = args.steps
    flat = abi.flatten(ckt)
    if args.inst > 1:
        flat = flat.replicate(args.inst)
    src = abi.source_table(ckt, 1e-6, steps)
    ref = None
    first_out = None
    for G, cut in [(g, c) for c in args.cuts for g in args.wgs]:
        t1 = time.time()
        h = Handle(flat, wgs_per_inst=G, front_cut=cut, profile=args.profile)
        t2 = time.time()
        info = h.info()
        r = h.run(steps, 1e-6, src, want_currents=True)
        rec = dict(rows=rows, n=info["n_var"], nnz_lu=info["nnz_lu"], levels=info["n_levels"], interp=info["interpreter"], T=info["threads"],
                   G=info["wgs_per_inst"], lds=info["lds_bytes"], program_MB=info["program_bytes"] / 1e6, create_s=t2 - t1, status=r["status"],
                   detail=r.get("detail"), ms_per_step=(r.get("kernel_ms") or 0) / (steps + 1), fronts=info["n_fronts"], cut=info["front_cut"],
                   max_front=info["max_front"], front_ws_MB=info["front_ws_bytes"] / 1e6, inst=args.inst)
        if r["status"] == 0:
            if first_out is None:
                first_out = r["out_v"]
            else:  # against the first configuration of this mesh (e.g. --cuts -1 12: task lists vs fronts)
                e = np.abs(r["out_v"] - first_out) / (1e-9 * np.abs(first_out) + 1e-12)
                rec["vs_first_config_over_tol"] = float(e.max())
        if args.check and r["status"] == 0 and info["n_var"] <= 2600:
            if ref is None:
                from oracle.pyoracle import OracleBackend
                s2 = min(steps, 5)
                ref = OracleBackend().run(flat, s2, 1e-6, src[: s2 + 1], want_currents=False)
            e = np.abs(r["out_v"][:, : ref["out_v"].shape[1]] - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12)
            rec["err_over_tol"] = float(e.max())
        print(json.dumps(rec), flush=True)
        if args.profile and r["status"] == 0:
            names = {1: "B", 8: "factor<cut", 9: "fronts_fwd", 10: "fronts_bwd", 11: "publish+sync", 40: "backward", 4: "Z",
                     12: "f.wait", 13: "f.asm_children(lds)", 14: "f.factor_lds_rest", 15: "f.store", 16: "f.asm(glob)", 17: "f.factor(glob)",
                     18: "b.wait", 19: "b.solve", 20: "f.post", 21: "l.pre", 22: "l.diag", 23: "l.trsm", 24: "l.trail", 25: "a.zero+asm",
                     26: "bins.factor", 27: "bins.sync", 28: "bk.interface", 29: "bk.bins", 30: "factor.interface.work", 31: "bk.interface.work",
                     32: "g.stage_in", 33: "g.diag", 34: "g.trsm", 35: "g.update", 36: "g.store", 37: "g.cb"}
            for wgi in range(info["wgs_per_inst"]):
                tk = h.section_ticks(wgi)
                per = {names[k]: round(tk[k] * 10.0 / 1000.0 / (steps + 1), 1) for k in names}  # us per step
                print(f"  wg {wgi:2d} us/step:", per, flush=True)
        h.close()
