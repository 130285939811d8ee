#!/usr/bin/env python3
"""Batched large-instance probe: n_inst replicas of rcd_mesh(rows) (parameter sweep of R values), K instances per workgroup."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import Handle

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=50)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--configs", default="1:1,32:1,32:2,32:4,256:1,256:4")  # n_inst:K
args = ap.parse_args()
ckt = parseNetlist(synth.rcd_mesh(args.rows, seed=3, tran=f".tran 1e-6 {args.steps * 1e-6!r}"))
flat1 = abi.flatten(ckt)
src = abi.source_table(ckt, 1e-6, args.steps)
dev = torch.device("cuda:0")
d_src = torch.tensor(src, device=dev)
for cfg in args.configs.split(","):
    ni, K = (int(x) for x in cfg.split(":"))
    flat = flat1.replicate(ni)
    rng = np.random.default_rng(1)
    flat.R_val *= 1.0 + 0.05 * rng.random(flat.R_val.shape)
    h = Handle(flat, inst_per_wg=K)
    info = h.info()
    ov = torch.empty((ni, args.steps + 1, info["n_out"]), dtype=torch.float64, device=dev)
    best = None
    for rep in range(2):
        h.run_device(args.steps, 1e-6, d_src.data_ptr(), ov.data_ptr())
        assert h.sync() == 0, h.error()
        best = h.kernel_ms() if best is None else min(best, h.kernel_ms())
    print(json.dumps(dict(rows=args.rows, n=info["n_var"], n_inst=ni, K=info["inst_per_wg"], T=info["threads"], interp=info["interpreter"],
                          lds=info["lds_bytes"], ms_per_step=best / (args.steps + 1), solves_per_s=ni * (args.steps + 1) / (best * 1e-3),
                          finite=bool(torch.isfinite(ov[:, -1]).all().item()))), flush=True)
    h.close()
    del ov
