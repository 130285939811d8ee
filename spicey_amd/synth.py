"""Synthetic benchmark netlists (BASELINE.json configs 2-5).

Pure text generators: the netlists go through the same parser as user input.  Values are drawn
from the LCG ``s <- (1664525*s + 1013904223) mod 2^32``, ``u = s / 2^32`` and printed with Python's
shortest round-trip ``repr`` (plain decimal or explicit exponent, never a unit suffix), so the
reference's parser reads back exactly the same doubles.
"""
from __future__ import annotations

from typing import List

PULSE = "PULSE(0 5 0 1e-9 1e-9 5e-3 1e-2)"


class _LCG:
    def __init__(self, seed: int) -> None:
        self.s = seed & 0xFFFFFFFF

    def u(self) -> float:
        self.s = (1664525 * self.s + 1013904223) & 0xFFFFFFFF
        return self.s / 4294967296.0


def _num(x: float) -> str:
    return repr(float(x))


def rc_ladder(n: int = 1000, seed: int = 1, tran: str = ".tran 1e-6 1e-2") -> str:
    """Config 2: V1 n1 0 PULSE; R{k} n{k} n{k+1}; C{k} n{k+1} 0.  Nvar = n + 1."""
    g = _LCG(seed)
    lines: List[str] = [f"* rc_ladder n={n} seed={seed}", f"V1 n1 0 {PULSE}"]
    for k in range(1, n):
        r = 10.0 * (1.0 + 0.1 * g.u())
        c = 1e-9 * (1.0 + 0.1 * g.u())
        lines.append(f"R{k} n{k} n{k+1} {_num(r)}")
        lines.append(f"C{k} n{k+1} 0 {_num(c)}")
    lines += [tran, ".end", ""]
    return "\n".join(lines)


def diode_chain(n: int = 1000, seed: int = 2, tran: str = ".tran 1e-6 1e-2") -> str:
    """Config 3: chain of diode-clamped RC stages (R{k}, D{k} to ground, C{k} to ground)."""
    g = _LCG(seed)
    lines: List[str] = [f"* diode_chain n={n} seed={seed}", ".model DM D(Is=1e-14 N=1)", f"V1 n1 0 {PULSE}"]
    for k in range(1, n):
        r = 100.0 * (1.0 + 0.1 * g.u())
        c = 1e-9 * (1.0 + 0.1 * g.u())
        lines.append(f"R{k} n{k} n{k+1} {_num(r)}")
        lines.append(f"D{k} n{k+1} 0 DM")
        lines.append(f"C{k} n{k+1} 0 {_num(c)}")
    lines += [tran, ".end", ""]
    return "\n".join(lines)


def rcd_mesh(rows: int = 100, cols: int | None = None, seed: int = 3, tran: str = ".tran 1e-6 0.1") -> str:
    """Config 5: rows x cols grid of resistors, a capacitor to ground at every node but the driven
    corner, and a diode to ground at ~10 % of the nodes."""
    cols = rows if cols is None else cols
    g = _LCG(seed)
    lines: List[str] = [f"* rcd_mesh {rows}x{cols} seed={seed}", ".model DM D(Is=1e-14 N=1)", f"V1 m0_0 0 {PULSE}"]
    for i in range(rows):
        for j in range(cols):
            if j + 1 < cols:
                lines.append(f"RH{i}_{j} m{i}_{j} m{i}_{j+1} {_num(10.0 * (1.0 + 0.1 * g.u()))}")
            if i + 1 < rows:
                lines.append(f"RV{i}_{j} m{i}_{j} m{i+1}_{j} {_num(10.0 * (1.0 + 0.1 * g.u()))}")
            if (i, j) != (0, 0):
                lines.append(f"C{i}_{j} m{i}_{j} 0 {_num(1e-9 * (1.0 + 0.1 * g.u()))}")
                if g.u() < 0.1:
                    lines.append(f"D{i}_{j} m{i}_{j} 0 DM")
    lines += [tran, ".end", ""]
    return "\n".join(lines)


def chain_values(n: int, seeds, r0: float, c0: float = 1e-9):
    """Vectorised LCG draws of rc_ladder / diode_chain for many seeds at once (BASELINE config 4:
    same topology, per-instance r_k, c_k).  Returns (R[len(seeds)][n-1], C[len(seeds)][n-1]) equal
    bit for bit to what the netlist text of the same seed parses to."""
    import numpy as np

    s = np.asarray(list(seeds), dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    R = np.empty((len(s), n - 1))
    C = np.empty((len(s), n - 1))
    a, c, m = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    for k in range(n - 1):
        s = (a * s + c) & m
        R[:, k] = r0 * (1.0 + 0.1 * (s.astype(np.float64) / 4294967296.0))
        s = (a * s + c) & m
        C[:, k] = c0 * (1.0 + 0.1 * (s.astype(np.float64) / 4294967296.0))
    return R, C


def chain_batch(kind: str, n: int, seeds, tran: str = ".tran 1e-6 1e-2"):
    """FlatCircuit batch of `kind` in {"rc_ladder", "diode_chain"} for the given seeds, plus
    (dt, steps, src_table).  Topology is parsed once from the netlist text of the first seed."""
    from . import abi
    from .netlist import parseNetlist

    seeds = list(seeds)
    gen = {"rc_ladder": rc_ladder, "diode_chain": diode_chain}[kind]
    ckt = parseNetlist(gen(n, seed=seeds[0], tran=tran))
    flat = abi.flatten(ckt).replicate(len(seeds))
    R, C = chain_values(n, seeds, 10.0 if kind == "rc_ladder" else 100.0)
    flat.R_val[:, :] = R
    flat.C_val[:, :] = C
    tr = ckt.analyses["tran"]
    dt, steps = abi.computeEffectiveTimeStep(tr["dt"], tr["tstop"])
    return flat, dt, steps, abi.source_table(ckt, dt, steps)
