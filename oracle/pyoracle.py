"""ctypes front-end of oracle/spicey_ref.c.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Exposes the same `run(flat, steps, dt, src, want_currents)` backend interface as
spicey_amd.lib.HipBackend so host-side flatten/re-key logic can be checked on CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from spicey_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "_ref", "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_ref", "liboracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(os.path.join(_HERE, f))
                                                                     for f in ("spicey_ref.c", "spicey_ref_ac.c")):
            build()
        L = C.CDLL(path)
        f64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.spicey_ref_run.restype = C.c_int32
        L.spicey_ref_run.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.c_int64, C.c_double, f64p, f64p, f64p, i32p,
                                     f64p, f64p, f64p, i32p, C.POINTER(C.c_int64), i32p]
        L.spicey_ref_set_knobs.restype = None
        L.spicey_ref_set_knobs.argtypes = [C.c_int32, f64p]
        L.spicey_ref_get_skips.restype = None
        L.spicey_ref_get_skips.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.spicey_ref_timestep.restype = None
        L.spicey_ref_timestep.argtypes = [C.c_double, C.c_double, f64p, C.POINTER(C.c_int64)]
        L.spicey_ref_ac.restype = C.c_int32
        L.spicey_ref_ac.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.c_int64, f64p, f64p, f64p, f64p, i32p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class OracleBackend:
    """Single-threaded reference-algorithm backend (dense GE in the reference's operation order).

    skip_off=True runs solveReal WITHOUT its `|f| < EPS` row-update skip (solveReal.ts:45) — a test knob, not the reference's
    behaviour.  Every run also reports `skipped` (nonzero multipliers the reference's skip dropped, per instance),
    `skip_solves` (solves with at least one) and `lin_err` [n_inst][steps+1] (the one-shot linearisation error per step)."""

    def __init__(self, skip_off: bool = False):
        self.skip_off = bool(skip_off)

    def run(self, flat: abi.FlatCircuit, steps: int, dt: float, src: np.ndarray, want_currents: bool = True,
            want_iters: bool = True) -> dict:
        L = lib()
        d = flat.desc()
        ni = flat.n_inst
        src = np.ascontiguousarray(src, dtype=np.float64)
        assert src.shape == (steps + 1, flat.nV)
        out_v = np.zeros((ni, steps + 1, flat.n_out))
        out_i = np.zeros((ni, steps + 1, flat.n_cur)) if want_currents else None
        iters = np.zeros((ni, steps + 1), np.int32) if want_iters else None
        st = {"C_vprev": flat.C_vprev.copy(), "L_iprev": flat.L_iprev.copy(), "D_vdprev": flat.D_vdprev.copy(),
              "S_ison": flat.S_ison.copy()}
        status, detail = abi.OK, ""
        es, ei = C.c_int64(0), C.c_int32(0)
        lin_err = np.zeros((ni, steps + 1))
        skipped, skip_solves = np.zeros(ni, np.int64), np.zeros(ni, np.int64)
        for k in range(ni):
            L.spicey_ref_set_knobs(1 if self.skip_off else 0, _p(lin_err[k], C.c_double))
            rc = L.spicey_ref_run(C.byref(d), k, steps, dt, _p(src, C.c_double), _p(out_v[k], C.c_double),
                                  _p(out_i[k], C.c_double) if want_currents else None,
                                  _p(iters[k], C.c_int32) if want_iters else None,
                                  _p(st["C_vprev"][k], C.c_double), _p(st["L_iprev"][k], C.c_double),
                                  _p(st["D_vdprev"][k], C.c_double), _p(st["S_ison"][k], C.c_int32),
                                  C.byref(es), C.byref(ei))
            a, b2 = C.c_int64(0), C.c_int64(0)
            L.spicey_ref_get_skips(C.byref(a), C.byref(b2))
            skipped[k], skip_solves[k] = a.value, b2.value
            if rc != abi.OK and status == abi.OK:
                status, detail = rc, f"singular at inst {k} step {es.value} iter {ei.value}"
        L.spicey_ref_set_knobs(0, None)
        return {"status": status, "detail": detail, "out_v": out_v, "out_i": out_i, "iters": iters, "state": st,
                "skipped": skipped, "skip_solves": skip_solves, "lin_err": lin_err}


    def run_ac(self, flat: abi.FlatCircuit, freqs: np.ndarray, vph: np.ndarray, want_currents: bool = True) -> dict:
        """AC sweep (oracle/spicey_ref_ac.c).  vph: complex [nV]; out_v complex [n_inst][n_freq][n_nodes],
        out_i complex [n_inst][n_freq][nR+nC+nL+nV].  status 1 singular, 5 complex divide by ~0, 6 resistor <= 0."""
        L = lib()
        d = flat.desc()
        ni, nf = flat.n_inst, len(freqs)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        vph = np.ascontiguousarray(vph, dtype=np.complex128).reshape(flat.nV)
        ncur = flat.nR + flat.nC + flat.nL + flat.nV
        out_v = np.zeros((ni, nf, flat.n_nodes), np.complex128)
        out_i = np.zeros((ni, nf, ncur), np.complex128) if want_currents else None
        status, detail = abi.OK, ""
        ei = C.c_int32(-1)
        for k in range(ni):
            rc = L.spicey_ref_ac(C.byref(d), k, nf, _p(freqs, C.c_double), _p(vph.view(np.float64), C.c_double),
                                 _p(out_v[k].view(np.float64), C.c_double),
                                 _p(out_i[k].view(np.float64), C.c_double) if want_currents else None, C.byref(ei))
            if rc != abi.OK and status == abi.OK:
                status, detail = rc, {1: "Singular matrix (complex)", 5: "Complex divide by ~0", 6: f"resistor {ei.value} <= 0"}.get(rc, str(rc))
        return {"status": status, "detail": detail, "out_v": out_v, "out_i": out_i}


def timestep(dt_requested: float, tstop: float):
    dt, steps = C.c_double(0), C.c_int64(0)
    lib().spicey_ref_timestep(dt_requested, tstop, C.byref(dt), C.byref(steps))
    return dt.value, steps.value
