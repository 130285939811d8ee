#!/usr/bin/env python3
"""Experiment (build with -DSPICEY_DIAG_TIMING): shader cycles inside the 16 x 16 diagonal block, per call, on the workgroups of the root chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spicey_amd import abi, synth
from spicey_amd.netlist import parseNetlist
from spicey_amd.lib import Handle
steps = 200
ckt = parseNetlist(synth.rcd_mesh(100, seed=3, tran=f".tran 1e-6 {steps * 1e-6!r}"))
flat = abi.flatten(ckt)
src = abi.source_table(ckt, 1e-6, steps)
h = Handle(flat, profile=True)
r = h.run(steps, 1e-6, src, want_currents=True)
print("status", r["status"])
for wg in (0, 64, 104, 1):
    tk = h.section_ticks(wg)
    n = max(1, tk[63])
    print(f"wg {wg}: calls/step {tk[63] / (steps + 1):.1f}  load {tk[60] / n:.0f}  steps {tk[61] / n:.0f}  store {tk[62] / n:.0f} cycles per call; l.diag {tk[22] * 10 / 1000 / (steps + 1):.1f} us, g.diag {tk[33] * 10 / 1000 / (steps + 1):.1f} us per step")
