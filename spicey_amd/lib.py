"""ctypes binding of libspicey_hip.so — the only compute path of this package.

Loading fails loudly when the shared library is missing (run ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C spicey_amd/csrc``); creating a handle fails with
SPICEY_ERR_NO_DEVICE when there is no GPU.  Nothing here falls back to a CPU solver.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPICEY_HIP_LIB") or os.path.join(_HERE, "libspicey_hip.so")  # same override as ts/spiceyHip.ts
_LIB = None

EXPORTS = ["spicey_create", "spicey_run", "spicey_run_device", "spicey_sync", "spicey_get_state", "spicey_set_state", "spicey_reset_state",
           "spicey_last_solve_count", "spicey_group_retries", "spicey_group_stale_polls", "spicey_last_skip_risk", "spicey_get_lin_err",
           "spicey_last_kernel_ms", "spicey_get_info", "spicey_last_error", "spicey_destroy", "spicey_version",
           "spicey_debug_phase_cycles", "spicey_debug_phase_cycles_wg", "spicey_debug_front_ticks",
           "spicey_create_multi", "spicey_run_multi", "spicey_get_state_multi", "spicey_multi_get_shard", "spicey_multi_last_solve_count", "spicey_multi_group_retries", "spicey_multi_group_stale_polls",
           "spicey_multi_last_kernel_ms", "spicey_multi_last_error", "spicey_destroy_multi",
           "spicey_ac_create", "spicey_ac_run", "spicey_ac_get_info", "spicey_ac_last_kernel_ms", "spicey_ac_last_error", "spicey_ac_destroy",
           "spicey_format_tran", "spicey_to_precision6"]


class SpiceyNativeError(RuntimeError):
    pass


# Group mode health of this process: launches repeated after a bounded-wait abort and waits that only the read-modify-write
# poll saw satisfied, summed over every handle closed so far (include/spicey_hip.h: spicey_group_retries,
# spicey_group_stale_polls).  Both stay 0 in a healthy process; the GPU tests assert that around every test.
GROUP_TOTALS = {"retries": 0, "stale_polls": 0}


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SpiceyNativeError(f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                                "spicey_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    f64p, i32p, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    L.spicey_create.restype = C.c_int32
    L.spicey_create.argtypes = [C.POINTER(abi.SpiceyDesc), C.POINTER(abi.SpiceyOptions), C.POINTER(vp)]
    L.spicey_run.restype = C.c_int32
    L.spicey_run.argtypes = [vp, C.c_int64, C.c_double, f64p, f64p, f64p, i32p]
    L.spicey_run_device.restype = C.c_int32
    L.spicey_run_device.argtypes = [vp, C.c_int64, C.c_double, vp, vp, vp, vp, vp]
    L.spicey_sync.restype = C.c_int32
    L.spicey_sync.argtypes = [vp]
    L.spicey_group_retries.restype = C.c_int32
    L.spicey_group_retries.argtypes = [vp]
    L.spicey_group_stale_polls.restype = C.c_int64
    L.spicey_group_stale_polls.argtypes = [vp]
    L.spicey_last_skip_risk.restype = C.c_int64
    L.spicey_last_skip_risk.argtypes = [vp, C.POINTER(C.c_int64)]
    L.spicey_get_lin_err.restype = C.c_int32
    L.spicey_get_lin_err.argtypes = [vp, f64p]
    L.spicey_get_state.restype = C.c_int32
    L.spicey_get_state.argtypes = [vp, f64p, f64p, f64p, i32p]
    L.spicey_set_state.restype = C.c_int32
    L.spicey_set_state.argtypes = [vp, f64p, f64p, f64p, i32p]
    L.spicey_reset_state.restype = C.c_int32
    L.spicey_reset_state.argtypes = [vp, vp]
    L.spicey_last_solve_count.restype = C.c_int64
    L.spicey_last_solve_count.argtypes = [vp]
    L.spicey_last_kernel_ms.restype = C.c_double
    L.spicey_last_kernel_ms.argtypes = [vp]
    L.spicey_get_info.restype = C.c_int32
    L.spicey_get_info.argtypes = [vp, C.POINTER(abi.SpiceyInfo)]
    L.spicey_last_error.restype = C.c_char_p
    L.spicey_last_error.argtypes = [vp]
    L.spicey_destroy.restype = None
    L.spicey_destroy.argtypes = [vp]
    L.spicey_version.restype = C.c_char_p
    L.spicey_debug_phase_cycles.restype = C.c_int32
    L.spicey_debug_phase_cycles.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int32]
    L.spicey_debug_phase_cycles_wg.restype = C.c_int32
    L.spicey_debug_phase_cycles_wg.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64), C.c_int32]
    L.spicey_create_multi.restype = C.c_int32
    L.spicey_create_multi.argtypes = [C.POINTER(abi.SpiceyDesc), C.POINTER(abi.SpiceyOptions), i32p, C.c_int32, C.POINTER(vp)]
    L.spicey_run_multi.restype = C.c_int32
    L.spicey_run_multi.argtypes = [vp, C.c_int64, C.c_double, f64p, f64p, f64p, i32p]
    L.spicey_get_state_multi.restype = C.c_int32
    L.spicey_get_state_multi.argtypes = [vp, f64p, f64p, f64p, i32p]
    L.spicey_multi_get_shard.restype = C.c_int32
    L.spicey_multi_get_shard.argtypes = [vp, C.c_int32, C.POINTER(abi.SpiceyInfo), i32p, i32p, i32p]
    L.spicey_multi_last_solve_count.restype = C.c_int64
    L.spicey_multi_last_solve_count.argtypes = [vp]
    L.spicey_multi_group_retries.restype = C.c_int32
    L.spicey_multi_group_retries.argtypes = [vp]
    L.spicey_multi_group_stale_polls.restype = C.c_int64
    L.spicey_multi_group_stale_polls.argtypes = [vp]
    L.spicey_multi_last_kernel_ms.restype = C.c_double
    L.spicey_multi_last_kernel_ms.argtypes = [vp]
    L.spicey_multi_last_error.restype = C.c_char_p
    L.spicey_multi_last_error.argtypes = [vp]
    L.spicey_destroy_multi.restype = None
    L.spicey_destroy_multi.argtypes = [vp]
    L.spicey_ac_create.restype = C.c_int32
    L.spicey_ac_create.argtypes = [C.POINTER(abi.SpiceyDesc), C.POINTER(abi.SpiceyOptions), C.POINTER(vp)]
    L.spicey_ac_run.restype = C.c_int32
    L.spicey_ac_run.argtypes = [vp, C.c_int64, f64p, f64p, f64p, f64p]
    L.spicey_ac_get_info.restype = C.c_int32
    L.spicey_ac_get_info.argtypes = [vp, C.POINTER(abi.SpiceyInfo)]
    L.spicey_ac_last_kernel_ms.restype = C.c_double
    L.spicey_ac_last_kernel_ms.argtypes = [vp]
    L.spicey_ac_last_error.restype = C.c_char_p
    L.spicey_ac_last_error.argtypes = [vp]
    L.spicey_ac_destroy.restype = None
    L.spicey_ac_destroy.argtypes = [vp]
    L.spicey_format_tran.restype = C.c_int64
    L.spicey_format_tran.argtypes = [C.c_int64, C.c_int32, f64p, f64p, C.c_int64, i32p, C.c_char_p, C.c_char_p, C.c_int64]
    L.spicey_to_precision6.restype = C.c_int32
    L.spicey_to_precision6.argtypes = [C.c_double, C.c_char_p]
    _LIB = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class Handle:
    """Owns one SpiceyHandle (one topology, n_inst instances, one device)."""

    def __init__(self, flat: abi.FlatCircuit, device: int = 0, threads: int = 0, inst_per_wg: int = 0,
                 force_global: bool = False, profile: bool = False, interpreter: int = 0, geometry: int = 0, no_tail: bool = False, debug_empty_phases: int = 0, wgs_per_inst: int = 0,
                 no_reuse: bool = False, csr_numbering: bool = False, front_cut: int = 0, stage_fronts: bool = False, no_pcr: bool = False, no_rows: bool = False,
                 group_retry: bool = False, group_timeout_ms: int = 0, diagnostics: int = 0):
        self.L = load()
        self.flat = flat
        opt = abi.SpiceyOptions()
        opt.device, opt.threads, opt.inst_per_wg, opt.want_currents, opt.force_global = device, threads, inst_per_wg, 1, int(force_global)
        opt.profile = int(profile)
        opt.interpreter = int(interpreter)
        opt.geometry = int(geometry)
        opt.wgs_per_inst = int(wgs_per_inst)
        opt.front_cut = int(front_cut)
        opt.group_retry, opt.group_timeout_ms, opt.diagnostics = int(group_retry), int(group_timeout_ms), int(diagnostics)
        self.diagnostics = int(diagnostics)
        opt.debug = (1 if no_tail else 0) | (2 if no_reuse else 0) | (4 if csr_numbering else 0) | (8 if stage_fronts else 0) | (32 if no_pcr else 0) | (64 if no_rows else 0) | (int(debug_empty_phases) << 8)
        d = flat.desc()
        hp = C.c_void_p()
        rc = self.L.spicey_create(C.byref(d), C.byref(opt), C.byref(hp))
        if rc != abi.OK:
            msg = self.L.spicey_last_error(None)
            raise SpiceyNativeError(f"spicey_create failed ({rc}): {msg.decode() if msg else ''}")
        self.h = hp

    def info(self) -> dict:
        i = abi.SpiceyInfo()
        self.L.spicey_get_info(self.h, C.byref(i))
        return i.as_dict()

    def error(self) -> str:
        m = self.L.spicey_last_error(self.h)
        return m.decode() if m else ""

    def run(self, steps: int, dt: float, src: np.ndarray, want_currents: bool = True, want_iters: bool = True) -> dict:
        f = self.flat
        src = np.ascontiguousarray(src, dtype=np.float64)
        if src.shape != (steps + 1, f.nV):
            raise ValueError(f"src_table must be [steps+1][nV] = {(steps + 1, f.nV)}, got {src.shape}")
        out_v = np.empty((f.n_inst, steps + 1, f.n_out))
        out_i = np.empty((f.n_inst, steps + 1, f.n_cur)) if want_currents else None
        iters = np.zeros((f.n_inst, steps + 1), np.int32) if want_iters else None
        rc = self.L.spicey_run(self.h, steps, dt, _p(src, C.c_double), _p(out_v, C.c_double), _p(out_i, C.c_double),
                               _p(iters, C.c_int32))
        res = {"status": rc, "detail": self.error() if rc != abi.OK else "", "out_v": out_v, "out_i": out_i, "iters": iters}
        if rc == abi.OK:
            res["state"] = self.state()
            res["solves"] = self.L.spicey_last_solve_count(self.h)
            res["kernel_ms"] = self.L.spicey_last_kernel_ms(self.h)
            if self.diagnostics & 1:
                per = np.zeros(f.n_inst, np.int64)
                self.L.spicey_last_skip_risk(self.h, _p(per, C.c_int64))
                res["skip_risk"] = per
            if self.diagnostics & 2:
                le = np.zeros((f.n_inst, steps + 1))
                if self.L.spicey_get_lin_err(self.h, _p(le, C.c_double)) != abi.OK:
                    raise SpiceyNativeError(f"spicey_get_lin_err failed: {self.error()}")
                res["lin_err"] = le
        return res

    def run_device(self, steps: int, dt: float, d_src: int, d_out_v: int, d_out_i: int = 0, d_iters: int = 0, stream: int = 0) -> None:
        """Enqueue with raw device pointers (e.g. torch tensors' data_ptr()); no synchronisation."""
        rc = self.L.spicey_run_device(self.h, steps, dt, d_src, d_out_v, d_out_i or None, d_iters or None, stream or None)
        if rc != abi.OK:
            raise SpiceyNativeError(f"spicey_run_device failed ({rc}): {self.error()}")

    def sync(self) -> int:
        return self.L.spicey_sync(self.h)

    def solves(self) -> int:
        return self.L.spicey_last_solve_count(self.h)

    def front_ticks(self, grp: int = 0):
        """(ticks[nf][4] uint64 summed over the solves, meta[nf][4] = pivots, boundary, parent, owner) of the last run
        (profile=True, circuits with dense fronts); include/spicey_hip.h, spicey_debug_front_ticks."""
        nf = self.info()["n_fronts"]
        t = np.zeros((max(nf, 1), 4), np.uint64)
        m = np.zeros((max(nf, 1), 4), np.int32)
        self.L.spicey_debug_front_ticks.restype = C.c_int32
        self.L.spicey_debug_front_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_int32]
        got = self.L.spicey_debug_front_ticks(self.h, grp, t.ctypes.data_as(C.POINTER(C.c_uint64)), m.ctypes.data_as(C.POINTER(C.c_int32)), nf)
        return t[:got], m[:got]

    def group_retries(self) -> int:
        """Group-mode launches this handle repeated after a bounded-spin abort (include/spicey_hip.h, spicey_sync)."""
        return self.L.spicey_group_retries(self.h)

    def group_stale_polls(self) -> int:
        """Group mode: waits of this handle's launches that only the read-modify-write poll saw satisfied (0 when healthy)."""
        return self.L.spicey_group_stale_polls(self.h)

    def kernel_ms(self) -> float:
        return self.L.spicey_last_kernel_ms(self.h)

    def phase_cycles(self) -> dict:
        """Shader-clock cycles per phase kind of workgroup 0 in the last run (needs profile=True)."""
        buf = (C.c_uint64 * 72)()
        self.L.spicey_debug_phase_cycles(self.h, buf, 72)
        a = list(buf)
        return {"prologue": a[0], "B": a[1], "S": a[2], "A": a[3], "Z": a[4], "run_cycles": a[5], "run_wall_ticks_100MHz": a[6], "between_phases": a[7],
                "U": a[8:40], "K": a[40:72]}

    def section_ticks(self, wg: int) -> list:
        """Group mode, profile=True: 100 MHz wall ticks per section of launched workgroup `wg` (see spicey_hip.h)."""
        buf = (C.c_uint64 * 72)()
        self.L.spicey_debug_phase_cycles_wg(self.h, int(wg), buf, 72)
        return list(buf)

    def state(self) -> dict:
        f = self.flat
        st = {"C_vprev": np.zeros((f.n_inst, f.nC)), "L_iprev": np.zeros((f.n_inst, f.nL)), "D_vdprev": np.zeros((f.n_inst, f.nD)),
              "S_ison": np.zeros((f.n_inst, f.nS), np.int32)}
        rc = self.L.spicey_get_state(self.h, _p(st["C_vprev"], C.c_double), _p(st["L_iprev"], C.c_double),
                                     _p(st["D_vdprev"], C.c_double), _p(st["S_ison"], C.c_int32))
        if rc != abi.OK:
            raise SpiceyNativeError(f"spicey_get_state failed ({rc}): {self.error()}")
        return st

    def set_state(self, st: dict) -> None:
        """State entering the next run: any of C_vprev / L_iprev / D_vdprev [n_inst][n] float64, S_ison int32."""
        a = {k: (np.ascontiguousarray(st[k], dtype=np.int32 if k == "S_ison" else np.float64) if st.get(k) is not None else None)
             for k in ("C_vprev", "L_iprev", "D_vdprev", "S_ison")}
        rc = self.L.spicey_set_state(self.h, _p(a["C_vprev"], C.c_double), _p(a["L_iprev"], C.c_double), _p(a["D_vdprev"], C.c_double),
                                     _p(a["S_ison"], C.c_int32))
        if rc != abi.OK:
            raise SpiceyNativeError(f"spicey_set_state failed ({rc}): {self.error()}")

    def reset_state(self, stream: int = 0) -> None:
        """Back to the state the handle was created with (device-to-device, enqueued on `stream`)."""
        rc = self.L.spicey_reset_state(self.h, stream or None)
        if rc != abi.OK:
            raise SpiceyNativeError(f"spicey_reset_state failed ({rc}): {self.error()}")

    def close(self) -> None:
        if getattr(self, "h", None):
            GROUP_TOTALS["retries"] += self.L.spicey_group_retries(self.h)
            GROUP_TOTALS["stale_polls"] += self.L.spicey_group_stale_polls(self.h)
            self.L.spicey_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiHandle:
    """spicey_create_multi / spicey_run_multi: the instances of one FlatCircuit block-partitioned over several devices
    inside this process, results gathered into one host buffer (SURVEY.md §8(b) "device ordinal(s)", §8(e))."""

    def __init__(self, flat: abi.FlatCircuit, devices, **opts):
        self.L = load()
        self.flat = flat
        opt = abi.SpiceyOptions()
        opt.want_currents = 1
        for k, v in opts.items():
            setattr(opt, k, int(v))
        d = flat.desc()
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        hp = C.c_void_p()
        rc = self.L.spicey_create_multi(C.byref(d), C.byref(opt), _p(devs, C.c_int32) if len(devs) else None, len(devs), C.byref(hp))
        if rc != abi.OK:
            msg = self.L.spicey_multi_last_error(None)
            raise SpiceyNativeError(f"spicey_create_multi failed ({rc}): {msg.decode() if msg else ''}")
        self.h = hp

    def shards(self) -> list:
        out = []
        i = 0
        while True:
            info = abi.SpiceyInfo()
            dev, first, cnt = C.c_int32(), C.c_int32(), C.c_int32()
            if self.L.spicey_multi_get_shard(self.h, i, C.byref(info), C.byref(dev), C.byref(first), C.byref(cnt)) != abi.OK:
                return out
            out.append({"device": dev.value, "first_inst": first.value, "n_inst": cnt.value, "info": info.as_dict()})
            i += 1

    def run(self, steps: int, dt: float, src: np.ndarray, want_currents: bool = True, want_iters: bool = True) -> dict:
        f = self.flat
        src = np.ascontiguousarray(src, dtype=np.float64)
        out_v = np.empty((f.n_inst, steps + 1, f.n_out))
        out_i = np.empty((f.n_inst, steps + 1, f.n_cur)) if want_currents else None
        iters = np.zeros((f.n_inst, steps + 1), np.int32) if want_iters else None
        rc = self.L.spicey_run_multi(self.h, steps, dt, _p(src, C.c_double), _p(out_v, C.c_double), _p(out_i, C.c_double), _p(iters, C.c_int32))
        detail = self.L.spicey_multi_last_error(self.h).decode() if rc != abi.OK else ""
        res = {"status": rc, "detail": detail, "out_v": out_v, "out_i": out_i, "iters": iters}
        if rc == abi.OK:
            st = {"C_vprev": np.zeros((f.n_inst, f.nC)), "L_iprev": np.zeros((f.n_inst, f.nL)), "D_vdprev": np.zeros((f.n_inst, f.nD)),
                  "S_ison": np.zeros((f.n_inst, f.nS), np.int32)}
            self.L.spicey_get_state_multi(self.h, _p(st["C_vprev"], C.c_double), _p(st["L_iprev"], C.c_double), _p(st["D_vdprev"], C.c_double),
                                          _p(st["S_ison"], C.c_int32))
            res["state"] = st
            res["solves"] = self.L.spicey_multi_last_solve_count(self.h)
            res["kernel_ms"] = self.L.spicey_multi_last_kernel_ms(self.h)
        return res

    def group_retries(self) -> int:
        return self.L.spicey_multi_group_retries(self.h)

    def group_stale_polls(self) -> int:
        return self.L.spicey_multi_group_stale_polls(self.h)

    def close(self) -> None:
        if getattr(self, "h", None):
            GROUP_TOTALS["retries"] += self.L.spicey_multi_group_retries(self.h)
            GROUP_TOTALS["stale_polls"] += self.L.spicey_multi_group_stale_polls(self.h)
            self.L.spicey_destroy_multi(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_tran_native(times: np.ndarray, values: np.ndarray, cols, header: str) -> str:
    """spicey_format_tran: CSV text of formatTranResult from a [n_points][stride] matrix (host code, multi-threaded)."""
    L = load()
    times = np.ascontiguousarray(times, dtype=np.float64)
    values = np.ascontiguousarray(values, dtype=np.float64)
    if values.ndim != 2 or values.shape[0] != len(times):
        raise ValueError("values must be [n_points][stride]")
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    hdr = header.encode("utf-8")
    need = L.spicey_format_tran(len(times), len(cols), _p(times, C.c_double), _p(values, C.c_double), values.shape[1],
                                _p(cols, C.c_int32), hdr, None, 0)
    if need < 0:
        raise SpiceyNativeError("spicey_format_tran: bad arguments")
    buf = C.create_string_buffer(int(need) + 1)
    L.spicey_format_tran(len(times), len(cols), _p(times, C.c_double), _p(values, C.c_double), values.shape[1],
                         _p(cols, C.c_int32), hdr, buf, need)
    return buf.raw[:need].decode("utf-8")


def to_precision6_native(x: float) -> str:
    buf = C.create_string_buffer(40)
    n = load().spicey_to_precision6(float(x), buf)
    return buf.raw[:n].decode("ascii")


class AcHandle:
    """spicey_ac_* of include/spicey_hip.h: AC sweep of n_inst instances of one topology."""

    def __init__(self, flat: abi.FlatCircuit, device: int = 0, threads: int = 0, force_global: bool = False, no_resident: bool = False, no_dense: bool = False):
        self.L = load()
        self.flat = flat
        opt = abi.SpiceyOptions()
        opt.device, opt.threads, opt.force_global = int(device), int(threads), int(bool(force_global))
        # bit 4: one workgroup per (instance, frequency) even for large batches; bit 7: no dense partial-pivoting fallback
        opt.debug = (16 if no_resident else 0) | (128 if no_dense else 0)
        d = flat.desc()
        h = C.c_void_p()
        rc = self.L.spicey_ac_create(C.byref(d), C.byref(opt), C.byref(h))
        if rc != abi.OK:
            raise SpiceyNativeError(f"spicey_ac_create failed ({rc}): {self.L.spicey_ac_last_error(None).decode()}")
        self.h = h

    def info(self) -> dict:
        info = abi.SpiceyInfo()
        self.L.spicey_ac_get_info(self.h, C.byref(info))
        return info.as_dict()

    def run(self, freqs, vph, want_currents: bool = True) -> dict:
        f = self.flat
        ni, nf = f.n_inst, len(freqs)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        # one phasor set for every instance, or one per instance
        ph = np.ascontiguousarray(np.broadcast_to(np.asarray(vph, np.complex128).reshape(-1, f.nV), (ni, f.nV)))
        out_v = np.zeros((ni, nf, f.n_out), np.complex128)
        out_i = np.zeros((ni, nf, f.nR + f.nC + f.nL + f.nV), np.complex128) if want_currents else None
        rc = self.L.spicey_ac_run(self.h, nf, _p(freqs, C.c_double), _p(ph.view(np.float64), C.c_double),
                                  _p(out_v.view(np.float64), C.c_double), _p(out_i.view(np.float64), C.c_double) if want_currents else None)
        detail = self.L.spicey_ac_last_error(self.h).decode() if rc != abi.OK else ""
        return {"status": rc, "detail": detail, "out_v": out_v, "out_i": out_i, "kernel_ms": self.L.spicey_ac_last_kernel_ms(self.h)}

    def close(self) -> None:
        if getattr(self, "h", None):
            self.L.spicey_ac_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipBackend:
    """Backend interface used by spicey_amd.simulate: one handle per call (the reference API is stateless)."""

    def __init__(self, device: int = 0, threads: int = 0, inst_per_wg: int = 0, force_global: bool = False, interpreter: int = 0,
                 geometry: int = 0, wgs_per_inst: int = 0, no_reuse: bool = False, front_cut: int = 0, stage_fronts: bool = False, no_pcr: bool = False, no_rows: bool = False,
                 group_retry: bool = False, group_timeout_ms: int = 0, diagnostics: int = 0):
        self.kw = dict(device=device, threads=threads, inst_per_wg=inst_per_wg, force_global=force_global, interpreter=interpreter,
                       geometry=geometry, wgs_per_inst=wgs_per_inst, no_reuse=no_reuse, front_cut=front_cut, stage_fronts=stage_fronts, no_pcr=no_pcr, no_rows=no_rows,
                       group_retry=group_retry, group_timeout_ms=group_timeout_ms, diagnostics=diagnostics)
        self.info: Optional[dict] = None
        self.group_retries = 0      # summed over this backend's runs (group mode; 0 when healthy)
        self.group_stale_polls = 0

    def run(self, flat: abi.FlatCircuit, steps: int, dt: float, src: np.ndarray, want_currents: bool = True,
            want_iters: bool = True) -> dict:
        h = Handle(flat, **self.kw)
        try:
            self.info = h.info()
            res = h.run(steps, dt, src, want_currents, want_iters)
            self.group_retries += h.group_retries()
            self.group_stale_polls += h.group_stale_polls()
            res["group_retries"], res["group_stale_polls"] = h.group_retries(), h.group_stale_polls()
            return res
        finally:
            h.close()

    def run_ac(self, flat: abi.FlatCircuit, freqs, vph, want_currents: bool = True) -> dict:
        h = AcHandle(flat, device=self.kw["device"], threads=self.kw["threads"], force_global=self.kw["force_global"])
        try:
            self.info = h.info()
            return h.run(freqs, vph, want_currents)
        finally:
            h.close()
