"""simulateAC() / formatAcResult(): the reference's AC sweep API with the native solver underneath.

Mirrors /root/reference/lib/analysis/simulateAC.ts:64-130: same result shape
({freqs, nodeVoltages, elementCurrents} keyed by canonical node / element name, JS key order; values are Python
`complex` where the reference holds `Complex` objects), same Error messages.

Host side (what the TypeScript layer keeps, because Math.pow / cos / sin are engine-defined): the frequency list
(buildFrequencyArray :9-23, utils/logspace.ts:3-17), the source phasors (Complex.fromPolar, math/Complex.ts:16-19)
and the argument checks that throw before any arithmetic.  Everything between — one complex MNA solve per
(instance, frequency), all of them independent — is ONE blocking call into libspicey_hip.so
(spicey_ac_run, include/spicey_hip.h).  No CPU path here: without the HIP library or a GPU this raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from . import abi
from .netlist import EPS, ParsedCircuit, js_object_key_order
from .simulate import _to_precision6_array

ERR_COMPLEX_DIV = abi.ERR_COMPLEX_DIV  # "Complex divide by ~0" (Complex.ts:42,50)


def logspace(f1: float, f2: float, pointsPerDecade: float) -> List[float]:
    """utils/logspace.ts:3-17."""
    if f1 <= 0 or f2 <= 0:
        raise ValueError(".ac frequencies must be > 0")
    if f2 < f1:
        f1, f2 = f2, f1
    decades = math.log10(f2 / f1)
    n = max(1, math.ceil(decades * pointsPerDecade))
    arr = [f1 * math.pow(10, i / pointsPerDecade) for i in range(n + 1)]
    if arr[-1] < f2 * (1 - EPS):
        arr.append(f2)
    return arr


def buildFrequencyArray(mode: str, N: float, f1: float, f2: float) -> List[float]:
    """simulateAC.ts:9-23."""
    if mode == "dec":
        return logspace(f1, f2, N)
    npts = max(2, N)
    step = (f2 - f1) / (npts - 1)
    return [f1 + i * step for i in range(int(npts))]


def source_phasors(ckt: ParsedCircuit) -> np.ndarray:
    """Complex.fromPolar(vs.acMag || 0, vs.acPhaseDeg || 0) per source (simulateAC.ts:57)."""
    out = np.zeros(len(ckt.V), np.complex128)
    for k, vs in enumerate(ckt.V):
        mag, deg = vs.acMag or 0.0, vs.acPhaseDeg or 0.0
        ph = (deg * math.pi) / 180
        out[k] = complex(mag * math.cos(ph), mag * math.sin(ph))
    return out


def _host_checks(ckt: ParsedCircuit, freqs: List[float]) -> None:
    """Errors the reference throws while building the system, before any solve (simulateAC.ts:39,51-53)."""
    if not len(freqs):
        return
    for r in ckt.R:  # :39, thrown at the first frequency
        if r.R <= 0:
            raise ValueError(f"R {r.name} must be > 0")
    if ckt.L:  # Complex.from(1,0).div(denom) throws when |denom|^2 < EPS although |denom| >= EPS (Complex.ts:40-42)
        w = (2 * math.pi) * np.asarray(freqs, dtype=np.float64)[:, None] * np.array([ind.L for ind in ckt.L])[None, :]
        if np.any(~(np.abs(w) < EPS) & (w * w < EPS)):
            raise ZeroDivisionError("Complex divide by ~0")


def _default_backend():
    from .lib import HipBackend  # fails loudly if the extension is missing

    return HipBackend()


class SingularComplexMatrixError(RuntimeError):
    """The reference throws Error("Singular matrix (complex)") (solveComplex.ts:28)."""

    def __init__(self, detail: str = "") -> None:
        super().__init__("Singular matrix (complex)")
        self.detail = detail


def simulateAC(ckt: ParsedCircuit, backend=None, freqs: Optional[List[float]] = None) -> Optional[dict]:
    ac = ckt.analyses.get("ac")
    if not ac:
        return None
    if freqs is None:
        freqs = buildFrequencyArray(ac["mode"], ac["N"], ac["f1"], ac["f2"])
    _host_checks(ckt, freqs)
    flat = abi.flatten(ckt)
    vph = source_phasors(ckt)
    be = backend if backend is not None else _default_backend()
    res = be.run_ac(flat, np.asarray(freqs, dtype=np.float64), vph, want_currents=True)
    if res["status"] == abi.ERR_SINGULAR:
        raise SingularComplexMatrixError(res.get("detail", ""))
    if res["status"] == ERR_COMPLEX_DIV:
        raise ZeroDivisionError("Complex divide by ~0")
    if res["status"] != abi.OK:
        raise RuntimeError(res.get("detail", f"spicey native error {res['status']}"))
    out_v = res["out_v"][0]  # [n_freq][n_nodes] complex
    out_i = res["out_i"][0]  # [n_freq][nR+nC+nL+nV]

    names = ckt.nodes.rev
    order = js_object_key_order([names[i] for i in range(1, ckt.nodes.count())])
    col = {names[i]: i - 1 for i in range(1, ckt.nodes.count())}
    node_voltages: Dict[str, list] = {name: out_v[:, col[name]].tolist() for name in order}

    # element currents: R, C, L, V recording order (:95-125); duplicate names append to one array, interleaved per frequency
    elem_names = [e.name for e in ckt.R] + [e.name for e in ckt.C] + [e.name for e in ckt.L] + [e.name for e in ckt.V]
    groups: Dict[str, List[int]] = {}
    for j, nm in enumerate(elem_names):
        groups.setdefault(nm, []).append(j)
    element_currents: Dict[str, list] = {}
    for nm in js_object_key_order(elem_names):
        cols = groups[nm]
        element_currents[nm] = (out_i[:, cols[0]] if len(cols) == 1 else out_i[:, cols].reshape(-1)).tolist()
    return {"freqs": list(freqs), "nodeVoltages": node_voltages, "elementCurrents": element_currents}


def formatAcResult(ac: Optional[dict]) -> str:
    """/root/reference/lib/formatting/formatAcResult.ts:3-25: `f, |V|,phase(deg)` per node, toPrecision(6)."""
    if not ac:
        return "No AC analysis.\n"
    nodes = list(ac["nodeVoltages"].keys())
    n = len(ac["freqs"])
    lines = ["f(Hz), " + ", ".join(f"{nm}:|V|,∠V(deg)" for nm in nodes)]
    cols = [_to_precision6_array(np.asarray(ac["freqs"], dtype=np.float64))]
    for nm in nodes:
        z = np.asarray(ac["nodeVoltages"][nm], dtype=np.complex128)[:n]
        mag = _to_precision6_array(np.hypot(z.real, z.imag))
        ph = _to_precision6_array(np.arctan2(z.imag, z.real) * 180 / math.pi)
        cols.append(np.array([f"{a},{b}" for a, b in zip(mag, ph)], dtype=object))
    for k in range(n):
        lines.append(", ".join(c[k] for c in cols if k < len(c)))
    return "\n".join(lines)
