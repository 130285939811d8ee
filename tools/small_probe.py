#!/usr/bin/env python3
"""Latency of the public API on the reference's own (tiny) test netlists: where does the time go when the circuit has
3-6 unknowns?  (create = symbolic phase + device allocations; run = launch + kernel + copies)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from spicey_amd import abi
from spicey_amd.netlist import parseNetlist
from spicey_amd.simulate import simulateTRAN
from spicey_amd.lib import Handle

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "netlists")
simulateTRAN(parseNetlist(open(os.path.join(GOLD, "transient01.cir")).read()))  # HIP start-up outside the timings
for name in ("transient01", "two_probes", "switch_vt_vh", "vswitch_pwl", "diode_switch", "boost_probe", "case_insensitive"):
    text = open(os.path.join(GOLD, name + ".cir")).read()
    t0 = time.perf_counter(); ckt = parseNetlist(text); t_parse = time.perf_counter() - t0
    dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
    flat = abi.flatten(ckt); src = abi.source_table(ckt, dt, steps)
    t0 = time.perf_counter(); h = Handle(flat); t_create = time.perf_counter() - t0
    t0 = time.perf_counter(); r = h.run(steps, dt, src); t_run = time.perf_counter() - t0
    t0 = time.perf_counter(); h.close(); t_close = time.perf_counter() - t0
    t0 = time.perf_counter(); simulateTRAN(parseNetlist(text)); t_all = time.perf_counter() - t0
    print(json.dumps(dict(name=name, n_var=flat.n_var, points=steps + 1, parse_ms=t_parse * 1e3, create_ms=t_create * 1e3, run_ms=t_run * 1e3,
                          kernel_ms=r["kernel_ms"], close_ms=t_close * 1e3, simulateTRAN_ms=t_all * 1e3, us_per_step=r["kernel_ms"] * 1e3 / (steps + 1))), flush=True)
