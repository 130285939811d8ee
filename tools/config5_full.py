#!/usr/bin/env python3
"""BASELINE config 5 exactly as written: the 10 000-node R/C/diode mesh (rcd_mesh(100), seed 3), `.tran 1e-6 0.1` =
100 001 timesteps, ONE instance on one GPU (16 workgroups cooperating), every node voltage and every element current
recorded into device buffers (8 GB + 25 GB).  Prints one JSON line; the first steps are checked against the oracle."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import Handle

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100)
ap.add_argument("--tstop", default="0.1")
ap.add_argument("--check-steps", type=int, default=3)
ap.add_argument("--steps", type=int, default=0, help="shorter run for the profiler passes: tstop = steps * 1e-6")
ap.add_argument("--inst", type=int, default=1, help="parameter-identical replicas (BASELINE: 8 k replicas on 8 GPUs), each with its own group of CUs")
ap.add_argument("--wgs", type=int, default=0)
ap.add_argument("--front-cut", type=int, default=0, help="0 auto, -1 no fronts (the round-1 path: one barrier-separated level per pivot)")
ap.add_argument("--no-currents", action="store_true")
args = ap.parse_args()
if args.steps > 0:
    args.tstop = repr(args.steps * 1e-6)
t0 = time.time()
ckt = parseNetlist(synth.rcd_mesh(args.rows, seed=3, tran=f".tran 1e-6 {args.tstop}"))
dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
flat = abi.flatten(ckt)
if args.inst > 1:
    flat = flat.replicate(args.inst)
src_np = abi.source_table(ckt, dt, steps)
t_host = time.time() - t0
dev = torch.device("cuda:0")
t0 = time.time()
h = Handle(flat, wgs_per_inst=args.wgs, front_cut=args.front_cut)
info = h.info()
t_create = time.time() - t0
src = torch.as_tensor(src_np, device=dev)
out_v = torch.empty((args.inst, steps + 1, info["n_out"]), dtype=torch.float64, device=dev)
out_i = None if args.no_currents else torch.empty((args.inst, steps + 1, info["n_cur"]), dtype=torch.float64, device=dev)
print(f"config 5: n_var {info['n_var']} nnz_lu {info['nnz_lu']} levels {info['n_levels']} steps {steps} G {info['wgs_per_inst']} "
      f"fronts {info['n_fronts']} cut {info['front_cut']} out {out_v.numel() * 8 / 1e9:.1f}+{(out_i.numel() if out_i is not None else 0) * 8 / 1e9:.1f} GB", file=sys.stderr, flush=True)
t0 = time.time()
h.run_device(steps, dt, src.data_ptr(), out_v.data_ptr(), out_i.data_ptr() if out_i is not None else 0)
rc = h.sync()
wall = time.time() - t0
assert rc == 0, h.error()
# group mode health (include/spicey_hip.h): no launch was repeated, no wait had to be ended by the read-modify-write poll
assert h.group_retries() == 0 and h.group_stale_polls() == 0, (h.group_retries(), h.group_stale_polls(), h.error())
sps = h.solves() / (h.kernel_ms() / 1e3)
algo = info["algorithmic_bytes_solve"]  # SURVEY.md §8(d) formula with this build's nnz(L+U): 9.4 MB per solve
rec = dict(config="BASELINE configs[4]: rcd_mesh(%d), %d timesteps, %d instance(s)" % (args.rows, steps, args.inst), n_var=info["n_var"], nnz_lu=info["nnz_lu"],
           levels=info["n_levels"], wgs_per_inst=info["wgs_per_inst"], threads=info["threads"], n_fronts=info["n_fronts"], front_cut=info["front_cut"],
           max_front=info["max_front"], front_ws_MB=info["front_ws_bytes"] / 1e6, kernel_s=h.kernel_ms() / 1e3, wall_s=wall,
           ms_per_step=h.kernel_ms() / (steps + 1), solves=h.solves(), solves_per_s=sps, group_retries=h.group_retries(), group_stale_polls=h.group_stale_polls(),
           roofline={"bound": "hbm", "achieved": algo * sps / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": algo * sps / 8e12,
                     "algorithmic_bytes_per_solve": algo, "roofline_solves_per_s": 8e12 / algo, "traffic": None,
                     "note": "a single instance is a serial recurrence over timesteps (SURVEY fact 4): the step time is the critical path of ONE "
                             "sparse LU (dependent fronts / levels), not bandwidth; replicas (--inst) scale the rate until the CUs are used"},
           host_prepare_s=t_host, create_s=t_create, finite=bool(torch.isfinite(out_v).all().item() and (out_i is None or torch.isfinite(out_i).all().item())),
           v_min=float(out_v.min().item()), v_max=float(out_v.max().item()))
if args.check_steps > 0:
    from oracle.pyoracle import OracleBackend
    k = args.check_steps
    t0 = time.time()
    ref = OracleBackend().run(flat, k, dt, src_np[: k + 1], want_currents=False)
    got = out_v[0, : k + 1].cpu().numpy()
    rec["oracle_check_steps"] = k + 1
    rec["oracle_check_s"] = time.time() - t0
    rec["err_over_tol"] = float((np.abs(got - ref["out_v"][0]) / (1e-9 * np.abs(ref["out_v"][0]) + 1e-12)).max())
print(json.dumps(rec), flush=True)
h.close()
