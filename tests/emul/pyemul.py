"""ctypes front-end of tests/emul/emul.cpp (CPU emulation of the device program; test infrastructure)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from spicey_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        # SPICEY_EMUL_ASAN=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so)): the address / UB sanitizer build
        asan = os.environ.get("SPICEY_EMUL_ASAN") == "1"
        import fcntl
        os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
        with open(os.path.join(_HERE, "_build", ".lock"), "w") as lk:  # (pytest-xdist workers: one build at a time)
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.run(["make", "-s", "-C", _HERE] + (["asan"] if asan else []), check=True, stderr=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_HERE, "_build", "libspicey_emul_asan.so" if asan else "libspicey_emul.so"))
        f64p, i32p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.spicey_emul_run.restype = C.c_int32
        L.spicey_emul_run.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.c_int32, C.c_int64, C.c_double, f64p, f64p, f64p,
                                      i32p, f64p, f64p, f64p, i32p, C.c_int32, C.POINTER(abi.SpiceyInfo), i32p, i64p, C.c_int32]
        L.spicey_emul_symbolic.restype = C.c_int32
        L.spicey_emul_symbolic.argtypes = [C.POINTER(abi.SpiceyDesc), i32p, i32p, i32p, C.POINTER(abi.SpiceyInfo), i64p]
        L.spicey_emul_resident.restype = C.c_int32
        L.spicey_emul_resident.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.c_int32, C.c_int32, i32p, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), i32p, C.c_int32]
        L.spicey_emul_bank_cost.restype = C.c_int32
        L.spicey_emul_bank_cost.argtypes = [C.POINTER(abi.SpiceyDesc), i64p]
        L.spicey_emul_ac.restype = C.c_int32
        L.spicey_emul_ac.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.c_int64, f64p, f64p, f64p, f64p, C.c_int32,
                                     C.POINTER(abi.SpiceyInfo)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class EmulBackend:
    def __init__(self, K=1, T=256, reverse=False, rmax=-1, no_reuse=False, chain=False, front_cut=0, stage_fronts=False, ac_resident=False, no_pcr=False, no_rows=False, ac_no_dense=False, virt_wgs=1, diagnostics=0, hybrid=False):
        """rmax < 0: v1 interpreter (sliced-ELL, 32-bit); rmax >= 0: v2 with `rmax` register-resident slots.
        no_reuse: refactor every step even when the circuit is linear."""
        self.K, self.T, self.reverse, self.rmax, self.no_reuse = K, T, reverse, rmax, no_reuse
        self.chain = chain  # v1 only: backward levels as a serial chain over half of the threads (group-mode emulation)
        self.ac_no_dense = ac_no_dense  # AC: a solve that trips a pivot guard reports the error instead of taking the dense fallback
        self.no_rows = no_rows  # v2: streamed factor phases keep one task per target entry (no row records)
        self.no_pcr = no_pcr  # v2: keep the task lists for the top levels even where their Schur complement is tridiagonal
        self.ac_resident = ac_resident  # AC: persistent workgroup per instance, task records and stamp parts in registers
        self.stage_fronts = stage_fronts  # dense fronts above 64 rows take the panel-staging (global workspace) path
        assert virt_wgs in (1, 3, 7, 21)
        self.virt_wgs = virt_wgs  # the subtree-local levels below a front cut are played as this many workgroups, one after the other
        self.hybrid = hybrid  # v2 only: the hybrid workspace layout (leaf-owned entries and element vectors outside the "LDS" array)
        self.diagnostics = diagnostics  # bit 0: skip-risk counters, bit 1: per-step linearisation error (SpiceyOptions.diagnostics)
        self.front_cut = front_cut  # v1 only: pivots of elimination-tree level >= front_cut are factored as dense fronts
        self.info = None
        self.solves = None

    def run(self, flat: abi.FlatCircuit, steps: int, dt: float, src: np.ndarray, want_currents: bool = True,
            want_iters: bool = True) -> dict:
        L = lib()
        d = flat.desc()
        ni = flat.n_inst
        src = np.ascontiguousarray(src, dtype=np.float64)
        out_v = np.zeros((ni, steps + 1, flat.n_out))
        out_i = np.zeros((ni, steps + 1, flat.n_cur)) if want_currents else None
        iters = np.zeros((ni, steps + 1), np.int32) if want_iters else None
        st = {"C_vprev": flat.C_vprev.copy(), "L_iprev": flat.L_iprev.copy(), "D_vdprev": flat.D_vdprev.copy(),
              "S_ison": flat.S_ison.copy()}
        info = abi.SpiceyInfo()
        err4 = np.zeros(4, np.int32)
        solves = C.c_int64(0)
        skip = np.zeros(ni, np.uint64) if self.diagnostics & 1 else None
        linerr = np.zeros((ni, steps + 1)) if self.diagnostics & 2 else None
        L.spicey_emul_set_hybrid.restype = None
        L.spicey_emul_set_hybrid.argtypes = [C.c_int32]
        L.spicey_emul_set_hybrid(1 if self.hybrid else 0)
        L.spicey_emul_set_diag.restype = None
        L.spicey_emul_set_diag.argtypes = [C.c_void_p, C.c_void_p]
        L.spicey_emul_set_diag(skip.ctypes.data if skip is not None else None, linerr.ctypes.data if linerr is not None else None)
        rc = L.spicey_emul_run(C.byref(d), self.K, self.T, steps, dt, _p(src, C.c_double), _p(out_v, C.c_double),
                               _p(out_i, C.c_double), _p(iters, C.c_int32), _p(st["C_vprev"], C.c_double),
                               _p(st["L_iprev"], C.c_double), _p(st["D_vdprev"], C.c_double), _p(st["S_ison"], C.c_int32),
                               (1 if self.reverse else 0) | (2 if self.no_reuse else 0) | (4 if self.chain else 0) | (8 if self.stage_fronts else 0) | (16 if self.no_pcr else 0) | (32 if self.no_rows else 0) | (64 if self.virt_wgs % 3 == 0 else 0) | (128 if self.virt_wgs % 7 == 0 else 0) | (int(self.front_cut) << 8), C.byref(info), _p(err4, C.c_int32), C.byref(solves), self.rmax)
        L.spicey_emul_set_hybrid(0)
        self.info = info.as_dict()
        self.solves = solves.value
        detail = f"singular at inst {err4[1]} step {err4[2]} iter {err4[3]}" if rc == abi.ERR_SINGULAR else ""
        res = {"status": rc, "detail": detail, "out_v": out_v, "out_i": out_i, "iters": iters, "state": st}
        if skip is not None:
            res["skip_risk"] = skip.astype(np.int64)
        if linerr is not None:
            res["lin_err"] = linerr
        return res


    def run_ac(self, flat: abi.FlatCircuit, freqs, vph, want_currents: bool = True) -> dict:
        """AC sweep through spicey_amd/csrc/ac_exec.h (same interface as HipBackend.run_ac)."""
        L = lib()
        d = flat.desc()
        ni, nf = flat.n_inst, len(freqs)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        ph = np.ascontiguousarray(np.broadcast_to(np.asarray(vph, np.complex128).reshape(-1, flat.nV), (ni, flat.nV)))
        out_v = np.zeros((ni, nf, flat.n_out), np.complex128)
        out_i = np.zeros((ni, nf, flat.nR + flat.nC + flat.nL + flat.nV), np.complex128) if want_currents else None
        info = abi.SpiceyInfo()
        rc = L.spicey_emul_ac(C.byref(d), self.T, nf, _p(freqs, C.c_double), _p(ph.view(np.float64), C.c_double),
                              _p(out_v.view(np.float64), C.c_double), _p(out_i.view(np.float64), C.c_double) if want_currents else None,
                              (1 if self.reverse else 0) | (2 if self.ac_resident else 0) | (4 if self.ac_no_dense else 0), C.byref(info))
        self.info = info.as_dict()
        detail = {1: "Singular matrix (complex)", 5: "Complex divide by ~0"}.get(rc, "")
        return {"status": rc, "detail": detail, "out_v": out_v, "out_i": out_i}


def symbolic(flat: abi.FlatCircuit):
    L = lib()
    d = flat.desc()
    n = flat.n_var
    cpos, rpos, level = (np.zeros(n, np.int32) for _ in range(3))
    info = abi.SpiceyInfo()
    prods = np.zeros(2, np.int64)
    rc = L.spicey_emul_symbolic(C.byref(d), _p(cpos, C.c_int32), _p(rpos, C.c_int32), _p(level, C.c_int32), C.byref(info),
                                _p(prods, C.c_int64))
    return rc, cpos, rpos, level, info.as_dict(), prods


def resident_layout(flat: abi.FlatCircuit, T: int, rmax: int, max_tail: int, pcr_top: bool = False, row_records: bool = False):
    L = lib()
    d = flat.desc()
    rc, _, _, level, info, _ = symbolic(flat)
    nph = 2 * info["n_levels"]
    res_phase = np.zeros((T // 64, rmax), np.int32)
    res_valid = np.zeros((rmax, T), np.uint32)
    ph_cnt = np.zeros(nph, np.uint32)
    st_cnt = np.zeros(nph, np.uint32)
    meta = np.zeros(6, np.int32)
    rc = L.spicey_emul_resident(C.byref(d), T, rmax, max_tail, _p(res_phase, C.c_int32), _p(res_valid, C.c_uint32), _p(ph_cnt, C.c_uint32),
                                _p(st_cnt, C.c_uint32), _p(meta, C.c_int32), (1 if pcr_top else 0) | (2 if row_records else 0))
    return rc, res_phase, res_valid, ph_cnt, st_cnt, meta


def row_record_counts(flat: abi.FlatCircuit):
    """Row records per factor level of the circuit's program (0 = the level has no row-record encoding)."""
    out = np.zeros(4096, np.uint32)
    d = flat.desc()
    n = lib().spicey_emul_row_records(C.byref(d), _p(out, C.c_uint32), len(out))
    assert n >= 0
    return [int(x) for x in out[:n]]


def bank_cost(flat: abi.FlatCircuit):
    """(cycles, ideal) of the records' operand reads without and with the bank-aware numbering pass."""
    out = np.zeros(4, np.int64)
    d = flat.desc()
    rc = lib().spicey_emul_bank_cost(C.byref(d), _p(out, C.c_int64))
    assert rc == 0
    return (int(out[0]), int(out[1])), (int(out[2]), int(out[3]))


def bin_stats(flat: abi.FlatCircuit, front_cut: int = -1) -> dict:
    """Subtree-local levels below a front cut (program.h, nBins): bins, cut, slice counts."""
    L = lib()
    L.spicey_emul_bin_stats.restype = C.c_int32
    L.spicey_emul_bin_stats.argtypes = [C.POINTER(abi.SpiceyDesc), C.c_int32, C.POINTER(C.c_int32)]
    out = (C.c_int32 * 6)()
    d = flat.desc()
    rc = L.spicey_emul_bin_stats(C.byref(d), front_cut, out)
    assert rc == 0, rc
    return dict(zip(("bins", "cut", "factor_slices", "interface_slices", "backward_slices", "widest_bin_slices"), list(out)))
