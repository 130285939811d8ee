#!/usr/bin/env python3
"""One long chain / ladder on the GPU: the automatic choice (LDS-resident up to ~2 270 nodes, the hybrid workspace beyond, cooperating
workgroups on a global workspace for what fits neither) against the 32-bit lists with 1 / 4 / 16 workgroups."""
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import Handle
for wl, n in (("diode_chain", 1000), ("diode_chain", 2000), ("diode_chain", 2300), ("diode_chain", 2600), ("diode_chain", 3000), ("diode_chain", 3200), ("diode_chain", 4000), ("diode_chain", 4600), ("diode_chain", 8000), ("rc_ladder", 3000), ("rc_ladder", 8000)):
    steps = 200
    ckt = parseNetlist(getattr(synth, wl)(n, tran=f".tran 1e-6 {steps*1e-6!r}"))
    flat = abi.flatten(ckt); src = abi.source_table(ckt, 1e-6, steps)
    for G, interp in ((0, 0), (1, 1), (4, 1), (16, 1)):
        try:
            h = Handle(flat, wgs_per_inst=G, interpreter=interp)
        except Exception as e:
            print(wl, n, G, "create failed", str(e)[:80]); continue
        r = h.run(steps, 1e-6, src); i = h.info()
        print(wl, n, "G req", G, "->", i["wgs_per_inst"], "interp", i["interpreter"], "T", i["threads"], "hybrid", i["hybrid_entries"], "lds", i["lds_bytes"], "levels", i["n_levels"], "nnz", i["nnz_lu"], "status", r["status"], "us/step %.1f" % (r["kernel_ms"]*1000/(steps+1)), flush=True)
        h.close()
