"""ctypes view of include/spicey_hip.h plus ParsedCircuit -> SpiceyDesc flattening.

The flattening is what the TypeScript layer does before its bun:ffi call:
AoS JS objects -> SoA int32/f64 arrays, node ids kept as the reference's ids (0 = ground).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np

from .netlist import EPS, ParsedCircuit

ABI_VERSION = 2
OK, ERR_SINGULAR, ERR_BAD_DESC, ERR_HIP, ERR_NO_DEVICE, ERR_COMPLEX_DIV = 0, 1, 2, 3, 4, 5

_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)


class SpiceyDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_nodes", C.c_int32), ("n_inst", C.c_int32),
        ("nR", C.c_int32), ("nC", C.c_int32), ("nL", C.c_int32), ("nV", C.c_int32), ("nS", C.c_int32), ("nD", C.c_int32),
        ("R_n1", _I32P), ("R_n2", _I32P), ("R_val", _F64P),
        ("C_n1", _I32P), ("C_n2", _I32P), ("C_val", _F64P), ("C_vprev", _F64P),
        ("L_n1", _I32P), ("L_n2", _I32P), ("L_val", _F64P), ("L_iprev", _F64P),
        ("V_n1", _I32P), ("V_n2", _I32P),
        ("S_n1", _I32P), ("S_n2", _I32P), ("S_cp", _I32P), ("S_cn", _I32P),
        ("S_ron", _F64P), ("S_roff", _F64P), ("S_von", _F64P), ("S_voff", _F64P), ("S_ison", _I32P),
        ("D_np", _I32P), ("D_nm", _I32P), ("D_is", _F64P), ("D_n", _F64P), ("D_vdprev", _F64P),
        ("n_out", C.c_int32), ("out_nodes", _I32P),
    ]


class SpiceyOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("threads", C.c_int32), ("inst_per_wg", C.c_int32),
                ("want_currents", C.c_int32), ("force_global", C.c_int32), ("profile", C.c_int32), ("interpreter", C.c_int32), ("geometry", C.c_int32), ("debug", C.c_int32), ("wgs_per_inst", C.c_int32), ("front_cut", C.c_int32),
                ("group_retry", C.c_int32), ("group_timeout_ms", C.c_int32), ("diagnostics", C.c_int32)]


class SpiceyInfo(C.Structure):
    _fields_ = [("n_var", C.c_int32), ("nnz_a", C.c_int32), ("nnz_lu", C.c_int32), ("n_levels", C.c_int32),
                ("threads", C.c_int32), ("inst_per_wg", C.c_int32), ("lds_bytes", C.c_int32), ("n_cur", C.c_int32),
                ("n_out", C.c_int32), ("n_workgroups", C.c_int32), ("interpreter", C.c_int32), ("geometry", C.c_int32),
                ("tail_levels", C.c_int32), ("wgs_per_inst", C.c_int32), ("resident_slots", C.c_int32),
                ("resident_tasks", C.c_int64), ("streamed_tasks", C.c_int64), ("program_bytes", C.c_int64),
                ("algorithmic_bytes_solve", C.c_int64), ("factor_reuse", C.c_int32), ("n_fronts", C.c_int32),
                ("front_cut", C.c_int32), ("max_front", C.c_int32), ("front_ws_bytes", C.c_int64),
                ("pcr_rows", C.c_int32), ("pcr_level", C.c_int32), ("hybrid_entries", C.c_int32)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class FlatCircuit:
    """Owns the numpy arrays a SpiceyDesc points into.

    ``n_inst`` instances share the topology; per-instance value arrays have shape [n_inst, n].
    """

    TOPO = ("R_n1", "R_n2", "C_n1", "C_n2", "L_n1", "L_n2", "V_n1", "V_n2", "S_n1", "S_n2", "S_cp", "S_cn", "D_np", "D_nm")
    VALS = ("R_val", "C_val", "C_vprev", "L_val", "L_iprev", "S_ron", "S_roff", "S_von", "S_voff", "D_is", "D_n", "D_vdprev")
    _KIND = {"R": "nR", "C": "nC", "L": "nL", "V": "nV", "S": "nS", "D": "nD"}

    def __init__(self, n_nodes: int, n_inst: int = 1, **arrays) -> None:
        self.n_nodes = int(n_nodes)
        self.n_inst = int(n_inst)
        self.out_nodes: Optional[np.ndarray] = None
        for k in self.TOPO:
            setattr(self, k, _i32(arrays.get(k, [])))
        self.nR, self.nC, self.nL = len(self.R_n1), len(self.C_n1), len(self.L_n1)
        self.nV, self.nS, self.nD = len(self.V_n1), len(self.S_n1), len(self.D_np)
        for k in self.VALS:
            n = getattr(self, self._KIND[k[0]])
            a = arrays.get(k)
            a = np.zeros((self.n_inst, n)) if a is None else _f64(a).reshape(self.n_inst, n)
            setattr(self, k, a)
        ison = arrays.get("S_ison")
        self.S_ison = np.zeros((self.n_inst, self.nS), np.int32) if ison is None else _i32(ison).reshape(self.n_inst, self.nS)
        if arrays.get("out_nodes") is not None:
            self.out_nodes = _i32(arrays["out_nodes"])

    @property
    def n_var(self) -> int:
        return self.n_nodes + self.nV

    @property
    def n_cur(self) -> int:
        return self.nR + self.nC + self.nL + self.nV + self.nS + self.nD

    @property
    def n_out(self) -> int:
        return len(self.out_nodes) if self.out_nodes is not None and len(self.out_nodes) else self.n_nodes

    def desc(self) -> SpiceyDesc:
        d = SpiceyDesc()
        d.abi_version = ABI_VERSION
        d.n_nodes, d.n_inst = self.n_nodes, self.n_inst
        d.nR, d.nC, d.nL, d.nV, d.nS, d.nD = self.nR, self.nC, self.nL, self.nV, self.nS, self.nD
        for k in self.TOPO + ("S_ison",):
            setattr(d, k, getattr(self, k).ctypes.data_as(_I32P))
        for k in self.VALS:
            setattr(d, k, getattr(self, k).ctypes.data_as(_F64P))
        if self.out_nodes is not None and len(self.out_nodes):
            d.n_out = len(self.out_nodes)
            d.out_nodes = self.out_nodes.ctypes.data_as(_I32P)
        else:
            d.n_out = 0
            d.out_nodes = None
        d._keepalive = self  # the arrays must outlive the struct
        return d

    def replicate(self, n_inst: int) -> "FlatCircuit":
        """Same topology, values of instance 0 copied to n_inst instances."""
        kw = {k: getattr(self, k) for k in self.TOPO}
        for k in self.VALS + ("S_ison",):
            kw[k] = np.repeat(getattr(self, k)[:1], n_inst, axis=0)
        kw["out_nodes"] = self.out_nodes
        return FlatCircuit(self.n_nodes, n_inst, **kw)


def flatten(ckt: ParsedCircuit, probe_filter: bool = False) -> FlatCircuit:
    """ParsedCircuit (one instance) -> FlatCircuit; state fields are the circuit's CURRENT state,
    so a second simulateTRAN continues where the first stopped."""
    out_nodes = None
    if probe_filter and len(ckt.probes["tran"]) > 0:
        upper = [p.upper() for p in ckt.probes["tran"]]
        out_nodes = [i for i in range(1, ckt.nodes.count()) if ckt.nodes.rev[i].upper() in upper]
    S = [s for s in ckt.S if s.model is not None]
    D = [d for d in ckt.D if d.model is not None]
    return FlatCircuit(
        ckt.nodes.count() - 1, 1,
        R_n1=[e.n1 for e in ckt.R], R_n2=[e.n2 for e in ckt.R], R_val=[e.R for e in ckt.R],
        C_n1=[e.n1 for e in ckt.C], C_n2=[e.n2 for e in ckt.C], C_val=[e.C for e in ckt.C], C_vprev=[e.vPrev for e in ckt.C],
        L_n1=[e.n1 for e in ckt.L], L_n2=[e.n2 for e in ckt.L], L_val=[e.L for e in ckt.L], L_iprev=[e.iPrev for e in ckt.L],
        V_n1=[e.n1 for e in ckt.V], V_n2=[e.n2 for e in ckt.V],
        S_n1=[e.n1 for e in S], S_n2=[e.n2 for e in S], S_cp=[e.ncPos for e in S], S_cn=[e.ncNeg for e in S],
        S_ron=[e.model.Ron for e in S], S_roff=[e.model.Roff for e in S],
        S_von=[e.model.Von for e in S], S_voff=[e.model.Voff for e in S], S_ison=[1 if e.isOn else 0 for e in S],
        D_np=[e.nPlus for e in D], D_nm=[e.nMinus for e in D], D_is=[e.model.Is for e in D], D_n=[e.model.N for e in D],
        D_vdprev=[e.vdPrev for e in D],
        out_nodes=out_nodes,
    )


def stack_instances(flats: Sequence[FlatCircuit]) -> FlatCircuit:
    """Batch circuits that share one topology (parameter sweeps, BASELINE config 4)."""
    f0 = flats[0]
    for f in flats[1:]:
        for k in FlatCircuit.TOPO:
            if not np.array_equal(getattr(f, k), getattr(f0, k)) or f.n_nodes != f0.n_nodes:
                raise ValueError("instances of one batch must share the topology (" + k + " differs)")
    kw = {k: getattr(f0, k) for k in FlatCircuit.TOPO}
    for k in FlatCircuit.VALS + ("S_ison",):
        kw[k] = np.concatenate([getattr(f, k) for f in flats], axis=0)
    kw["out_nodes"] = f0.out_nodes
    return FlatCircuit(f0.n_nodes, sum(f.n_inst for f in flats), **kw)


def computeEffectiveTimeStep(dt_requested: float, tstop: float):
    """simulateTRAN.ts:14-19, same double operations in the same order."""
    dt_eff = dt_requested if dt_requested > EPS else max(tstop / 1000, EPS)
    steps = max(1, math.ceil(tstop / max(dt_eff, EPS)))
    dt = tstop / steps if steps > 0 else tstop
    return dt, steps


def source_table(ckt: ParsedCircuit, dt: float, steps: int) -> np.ndarray:
    """Pre-evaluate `vs.waveform ? vs.waveform(t) : vs.dc || 0` (simulateTRAN.ts:67) at
    t = step*dt (:147) — closures cannot cross the FFI."""
    nv = len(ckt.V)
    tab = np.zeros((steps + 1, nv), dtype=np.float64)
    for k, vs in enumerate(ckt.V):
        if vs.waveform is not None:
            wf = vs.waveform
            tab[:, k] = [wf(step * dt) for step in range(steps + 1)]
        else:
            dc = vs.dc
            tab[:, k] = 0.0 if (dc == 0 or dc != dc) else dc
    return tab
