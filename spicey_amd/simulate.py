"""simulate() / simulateTRAN(): the reference's public API with the native solver underneath.

Mirrors /root/reference/lib/analysis/simulate.ts:5-10 and simulateTRAN.ts:130-252: same result
shapes ({times, nodeVoltages, elementCurrents} keyed by canonical node / element name, JS key
order), same in-place mutation of the circuit's state fields, same Error messages.

The work between "flatten" and "re-key" is ONE blocking call into libspicey_hip.so
(spicey_amd/lib.py).  There is no CPU path here: without the HIP library or a GPU this raises.
A backend object can be injected for tests (tests/ use the oracle through it to check the
flatten / re-key logic on CPU); the default is always the HIP backend.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import abi
from .netlist import ParsedCircuit, js_object_key_order, parseNetlist


class SingularMatrixError(RuntimeError):
    """The reference throws Error("Singular matrix (real)") (solveReal.ts:28)."""

    def __init__(self, detail: str = "") -> None:
        super().__init__("Singular matrix (real)")
        self.detail = detail


def _default_backend():
    from .lib import HipBackend  # fails loudly if the extension is missing

    # (diagnostics bit 0: the result says when the reference's row-update skip may have made ITS answer differ, `skipRisk`)
    return HipBackend(diagnostics=1)


def simulateTRAN(ckt: ParsedCircuit, backend=None, as_lists: bool = True) -> Optional[dict]:
    tran = ckt.analyses.get("tran")
    if not tran:
        return None
    dt, steps = abi.computeEffectiveTimeStep(tran["dt"], tran["tstop"])
    # .PRINT TRAN probes go DOWN to the device (SpiceyDesc.out_nodes): the reference computes every node and filters
    # afterwards (simulateTRAN.ts:240-249); here only the probed columns are written, moved and re-keyed
    flat = abi.flatten(ckt, probe_filter=True)
    src = abi.source_table(ckt, dt, steps)
    be = backend if backend is not None else _default_backend()
    res = be.run(flat, steps, dt, src, want_currents=True)
    if res["status"] == abi.ERR_SINGULAR:
        raise SingularMatrixError(res.get("detail", ""))
    if res["status"] != abi.OK:
        raise RuntimeError(res.get("detail", f"spicey native error {res['status']}"))

    out_v = res["out_v"][0]  # [steps+1][n_nodes]
    out_i = res["out_i"][0]  # [steps+1][n_cur]
    conv = (lambda a: a.tolist()) if as_lists else (lambda a: a)

    times = [step * dt for step in range(steps + 1)]  # t = step*dt (:147), first push is 0
    times[0] = 0.0

    names = ckt.nodes.rev
    node_voltages: Dict[str, object] = {}
    # JS: later duplicate names cannot occur (interned), order = js key order of insertion order
    # (probes present but none of them names a node — the reference's parser does not intern `.PRINT TRAN v(x)` names,
    # parseNetlist.ts:196-206 — : the reference's filter leaves nodeVoltages = {} (:240-249).  An empty out_nodes means "all
    # nodes" to the device, so that case is told apart here: the device records what it must, nothing of it is keyed)
    if len(ckt.probes["tran"]) > 0:
        recorded = [int(i) for i in flat.out_nodes] if flat.out_nodes is not None else []
    else:
        recorded = list(range(1, ckt.nodes.count()))
    order = js_object_key_order([names[i] for i in recorded])
    col = {names[i]: c for c, i in enumerate(recorded)}
    for name in order:
        node_voltages[name] = conv(out_v[:, col[name]])

    # element currents: R, C, L, V, S, D recording order (:173-219); duplicate names append to the
    # same JS array, interleaved per step
    elem_names: List[str] = ([e.name for e in ckt.R] + [e.name for e in ckt.C] + [e.name for e in ckt.L]
                             + [e.name for e in ckt.V] + [e.name for e in ckt.S if e.model is not None]
                             + [e.name for e in ckt.D if e.model is not None])
    element_currents: Dict[str, object] = {}
    groups: Dict[str, List[int]] = {}
    for j, nm in enumerate(elem_names):
        groups.setdefault(nm, []).append(j)
    for nm in js_object_key_order(elem_names):
        cols = groups[nm]
        if len(cols) == 1:
            element_currents[nm] = conv(out_i[:, cols[0]])
        else:
            element_currents[nm] = conv(out_i[:, cols].reshape(-1))

    # state write-back (:221-237, :122-124)
    st = res["state"]
    for i, c in enumerate(ckt.C):
        c.vPrev = float(st["C_vprev"][0, i])
    for i, l in enumerate(ckt.L):
        l.iPrev = float(st["L_iprev"][0, i])
    for i, d in enumerate([d for d in ckt.D if d.model is not None]):
        d.vdPrev = float(st["D_vdprev"][0, i])
    for i, s in enumerate([s for s in ckt.S if s.model is not None]):
        s.isOn = bool(st["S_ison"][0, i])

    # (probes, if any, were applied on the device: node_voltages holds exactly the probed nodes, in JS key order)
    # Beyond the reference's three keys: `iterations` (solves per step) and `skipRisk` — the number of (solve, column) pairs
    # in which the stamped matrix had a nonzero entry below 1e-15 x its column's largest, i.e. where the reference's
    # `if (Math.abs(f) < EPS) continue` (solveReal.ts:45) drops a row update that this solver performs (0: the two agree to
    # the 1e-9 bar; > 0: the reference's own numbers may differ, include/spicey_hip.h spicey_last_skip_risk)
    skip = res.get("skip_risk")
    return {"times": times, "nodeVoltages": node_voltages, "elementCurrents": element_currents,
            "iterations": res.get("iters"), "skipRisk": int(skip[0]) if skip is not None else 0}


def simulate(netlist_text: str, backend=None) -> dict:
    """simulate.ts:5-10: parse, AC sweep (if an .ac card is present), transient (if a .tran card is present)."""
    from .ac import simulateAC  # (ac.py imports this module's number formatter)

    circuit = parseNetlist(netlist_text)
    ac = simulateAC(circuit, backend=backend)
    tran = simulateTRAN(circuit, backend=backend)
    return {"circuit": circuit, "ac": ac, "tran": tran}


def formatTranResult(tran: Optional[dict]) -> str:
    """/root/reference/lib/formatting/formatTranResult.ts:1-23 (toPrecision(6) CSV), straight from the typed
    result arrays (SURVEY.md §8(f) rank 3 — after the solve is fast, per-value string formatting dominates end-to-end
    time for large runs).  Rectangular results go through the native formatter of libspicey_hip.so
    (spicey_format_tran: host code, multi-threaded, ~50x the numpy path below); ragged ones (the reference skips
    missing values) and builds without the library use the vectorised numpy path — both produce the same text."""
    if not tran:
        return "No TRAN analysis.\n"
    nodes = list(tran["nodeVoltages"].keys())
    n = len(tran["times"])
    header = ", ".join(["t(s)"] + [f"{nm}:V" for nm in nodes])
    series = [np.asarray(tran["nodeVoltages"][name], dtype=np.float64) for name in nodes]
    if n and all(len(a) >= n for a in series):
        try:
            from .lib import SpiceyNativeError, format_tran_native
            mat = np.stack([a[:n] for a in series], axis=1) if series else np.zeros((n, 0))
            return format_tran_native(np.asarray(tran["times"], dtype=np.float64), mat, np.arange(len(series)), header)
        except (ImportError, OSError, SpiceyNativeError):
            pass
    cols = [_to_precision6_array(np.asarray(tran["times"], dtype=np.float64))]
    for a in series:
        col = _to_precision6_array(a[:n])
        if len(col) < n:  # the reference skips missing values (`if (value == null) continue`)
            col = np.concatenate([col, np.full(n - len(col), None, dtype=object)])
        cols.append(col)
    lines = [header]
    for k in range(n):
        lines.append(", ".join(c[k] for c in cols if c[k] is not None))
    return "\n".join(lines)


def spiceyTranToVGraphs(tranResult: Optional[dict], ckt: ParsedCircuit, simulation_experiment_id: str) -> List[dict]:
    """/root/reference/lib/formatting/formatToVGraph.ts:11-39: one circuit-json
    `simulation_transient_voltage_graph` per recorded node.  `timestamps_ms` is computed once for all graphs (one
    vectorised multiply of the typed time axis; the reference maps the array again for every node)."""
    tran = ckt.analyses.get("tran")
    if not tranResult or not tran:
        return []
    ts_ms = (np.asarray(tranResult["times"], dtype=np.float64) * 1000).tolist()
    graphs = []
    for node_name, levels in tranResult["nodeVoltages"].items():
        graphs.append({
            "type": "simulation_transient_voltage_graph",
            "simulation_transient_voltage_graph_id": f"stvg_{simulation_experiment_id}_{node_name}",
            "simulation_experiment_id": simulation_experiment_id,
            "timestamps_ms": ts_ms,
            "voltage_levels": levels if isinstance(levels, list) else np.asarray(levels).tolist(),
            "time_per_step": tran["dt"] * 1000,
            "start_time_ms": 0,
            "end_time_ms": tran["tstop"] * 1000,
            "name": f"V({node_name})",
        })
    return graphs


def eecEngineTranToVGraphs(tranResult: dict, ckt: ParsedCircuit, simulation_experiment_id: str) -> List[dict]:
    """formatToVGraph.ts:41-65: the same graphs from an ngspice-style {time_s, voltages} result."""
    tran = ckt.analyses.get("tran")
    if not tran:
        return []
    ts_ms = (np.asarray(tranResult["time_s"], dtype=np.float64) * 1000).tolist()
    return [{
        "type": "simulation_transient_voltage_graph",
        "simulation_transient_voltage_graph_id": f"stvg_{simulation_experiment_id}_{node_name}_eec",
        "simulation_experiment_id": simulation_experiment_id,
        "timestamps_ms": ts_ms,
        "voltage_levels": levels,
        "time_per_step": tran["dt"] * 1000,
        "start_time_ms": 0,
        "end_time_ms": tran["tstop"] * 1000,
        "name": f"V({node_name}) (ngspice)",
    } for node_name, levels in js_ordered_items(tranResult["voltages"])]


def js_ordered_items(d: dict):
    """`for (const k in obj)`: integer-like keys first, ascending, then insertion order."""
    return [(k, d[k]) for k in js_object_key_order(list(d.keys()))]


def _to_precision6_array(x: np.ndarray) -> np.ndarray:
    """Number.prototype.toPrecision(6) for a whole array (object array of str).  printf("%.5e") is correctly rounded
    but breaks exact ties to even where ECMA-262 takes the larger digit string (100000.5 -> "100001"): candidates
    within 1e-5 of a rounding boundary are redone exactly."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty(x.shape, dtype=object)
    finite = np.isfinite(x)
    zero = finite & (x == 0)
    out[zero] = "0.00000"
    out[np.isnan(x)] = "NaN"
    out[np.isposinf(x)] = "Infinity"
    out[np.isneginf(x)] = "-Infinity"
    nz = finite & ~zero
    if nz.any():
        v = x[nz]
        sci = np.char.mod("%.5e", v)  # d.ddddde+XX, correctly rounded to 6 significant digits
        mant = np.char.partition(sci, "e")
        e = mant[:, 2].astype(np.int64)
        res = np.empty(v.shape, dtype=object)
        big = (e < -6) | (e >= 6)
        if big.any():
            sign = np.where(e[big] >= 0, "+", "-")
            res[big] = [f"{m}e{s}{abs(int(k))}" for m, s, k in zip(mant[big, 0], sign, e[big])]
        for dec in range(0, 12):  # fixed notation: 5 - e decimals
            sel = ~big & (np.maximum(0, 5 - e) == dec)
            if sel.any():
                res[sel] = np.char.mod(f"%.{dec}f", v[sel])
        with np.errstate(over="ignore", invalid="ignore"):
            scaled = np.abs(v).astype(np.longdouble) * np.power(np.longdouble(10), (5 - e).astype(np.longdouble))
            frac = scaled - np.floor(scaled)
        for i in np.nonzero(~(np.abs(frac - 0.5) > 1e-5))[0]:  # near a boundary (or not computable): exact route
            res[i] = _to_precision6(float(v[i]))
        out[nz] = res
    return out


def _to_precision6(x: float) -> str:
    """Number.prototype.toPrecision(6), exact: round half UP on the binary value (ECMA-262 picks the larger n)."""
    import decimal
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Infinity" if x > 0 else "-Infinity"
    if x == 0:
        return "0.00000"
    d = decimal.Decimal(abs(x))  # exact
    e = d.adjusted()
    with decimal.localcontext() as ctx:
        ctx.prec = 800
        n = int((d.scaleb(5 - e)).to_integral_value(rounding=decimal.ROUND_HALF_UP))
    if n == 1000000:
        n, e = 100000, e + 1
    digits = str(n)
    sign = "-" if x < 0 else ""
    if e < -6 or e >= 6:
        return f"{sign}{digits[0]}.{digits[1:]}e{'+' if e >= 0 else '-'}{abs(e)}"
    if e >= 0:
        return sign + digits[: e + 1] + ("." + digits[e + 1:] if e < 5 else "")
    return sign + "0." + "0" * (-e - 1) + digits
