#!/usr/bin/env python3
"""GPU perf probe: sweep (workload, batch, K, T) and print solves/s (device-resident buffers)."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from spicey_amd import synth  # noqa: E402
from spicey_amd.lib import Handle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="diode_chain")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--configs", default="1:1:512,256:1:512,512:1:512,512:2:512")  # B:K:T
    ap.add_argument("--currents", type=int, default=1)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--global", dest="force_global", type=int, default=0)
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--interp", type=int, default=0)
    ap.add_argument("--no-tail", type=int, default=0)
    ap.add_argument("--empty", type=int, default=0)
    ap.add_argument("--packed", type=int, default=0)
    ap.add_argument("--no-rows", type=int, default=0, help="1: wide factor levels keep one task per target entry (A/B against the row records)")
    ap.add_argument("--no-pcr", type=int, default=0, help="1: no tridiagonal top")
    ap.add_argument("--csr", type=int, default=0, help="1: plain CSR numbering of the L+U entries (A/B against the bank-aware one)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for cfg in args.configs.split(","):
        B, K, T = (int(x) for x in cfg.split(":"))
        flat, dt, _, _ = synth.chain_batch(args.workload, args.n, range(1, B + 1), tran=f".tran 1e-6 {args.steps * 1e-6!r}")
        from spicey_amd import abi
        from spicey_amd.netlist import parseNetlist
        ckt = parseNetlist(getattr(synth, args.workload)(args.n, seed=1, tran=".tran 1e-6 1e-2"))
        steps = args.steps
        src = torch.tensor(abi.source_table(ckt, 1e-6, steps), device=dev)
        h = Handle(flat, threads=T, inst_per_wg=K, force_global=bool(args.force_global), profile=bool(args.profile), interpreter=args.interp, no_tail=bool(args.no_tail), debug_empty_phases=args.empty, geometry=2 if args.packed else 0, csr_numbering=bool(args.csr), no_rows=bool(args.no_rows), no_pcr=bool(args.no_pcr))
        info = h.info()
        out_v = torch.empty((B, steps + 1, info["n_out"]), dtype=torch.float64, device=dev)
        out_i = torch.empty((B, steps + 1, info["n_cur"]), dtype=torch.float64, device=dev) if args.currents else None
        best = None
        for rep in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h.run_device(steps, 1e-6, src.data_ptr(), out_v.data_ptr(), out_i.data_ptr() if out_i is not None else 0)
            rc = h.sync()
            t1 = time.perf_counter()
            assert rc == 0, h.error()
            ms = h.kernel_ms()
            best = ms if best is None else min(best, ms)
        solves = h.solves()
        rec = dict(interp=info['interpreter'], geom=info['geometry'], tail=info['tail_levels'], rslots=info['resident_slots'], rtasks=info['resident_tasks'], stasks=info['streamed_tasks'], workload=args.workload, n=args.n, B=B, K=info["inst_per_wg"], T=info["threads"], lds=info["lds_bytes"],
                   steps=steps, kernel_ms=best, solves=solves, solves_per_s=solves / (best * 1e-3), wall_ms=(t1 - t0) * 1e3,
                   us_per_step=best * 1e3 / (steps + 1), levels=info["n_levels"], nnz_lu=info["nnz_lu"], currents=args.currents,
                   finite=bool(torch.isfinite(out_v[:, -1]).all().item()))
        if args.profile:
            pc = h.phase_cycles()
            per = lambda v: round(v / (steps + 1))  # noqa: E731
            rec['cyc_per_step'] = dict(B=per(pc['B']), Z=per(pc['Z']), tail=per(pc['U'][31]), U=[per(x) for x in pc['U'][:info['n_levels']]],
                                       K=[per(x) for x in pc['K'][:info['n_levels']]], total=per(pc['B'] + pc['Z'] + sum(pc['U']) + sum(pc['K'])),
                                       gaps=per(pc['between_phases']), marks=[per(x) for x in pc['K'][16:31]], run=per(pc['run_cycles']), clock_GHz=round(pc['run_cycles'] / max(pc['run_wall_ticks_100MHz'], 1) * 0.1, 3))
        print(json.dumps(rec), flush=True)
        h.close()
        del out_v, out_i


if __name__ == "__main__":
    main()
