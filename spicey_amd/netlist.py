"""Netlist text -> ParsedCircuit: the drop-in surface above the C-ABI.

Host-side mirror of the reference's parser (the reference is TypeScript and neither Bun nor a TS
transpiler exists in this image, so the host layer is Python; the TypeScript binding a maintainer
would add is in ``ts/`` and INTEGRATION.md).  Same names, argument meaning and error strings as

  * ``parseNetlist``            /root/reference/lib/parsing/parseNetlist.ts:123-481
  * ``NodeIndex``               /root/reference/lib/parsing/NodeIndex.ts:1-32
  * ``parseNumberWithUnits``    /root/reference/lib/parsing/parseNumberWithUnits.ts:1-30
  * ``parsePulseArgs``          /root/reference/lib/parsing/parsePulseArgs.ts:4-22
  * ``parsePwlArgs``            /root/reference/lib/parsing/parsePwlArgs.ts:3-19
  * ``pulseValue`` / ``pwlValue``  /root/reference/lib/parsing/pulseValue.ts:4-22, pwlValue.ts:3-16

This is a fresh implementation written against the behaviours listed in SURVEY.md Appendix D; it
runs once per netlist (µs–ms) and is not accelerated.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

EPS = 1e-15  # /root/reference/lib/constants/EPS.ts:1
VT_300K = 0.02585  # /root/reference/lib/constants/physics.ts:1

_A = re.ASCII


class NodeIndex:
    """Case-insensitive node interning; ``"0"`` is ground (NodeIndex.ts:1-32)."""

    def __init__(self) -> None:
        self._map: Dict[str, int] = {"0": 0}
        self.rev: List[str] = ["0"]

    def getOrCreate(self, name) -> int:
        orig = str(name)
        key = _js_upper(orig)
        if key in self._map:
            return self._map[key]
        idx = len(self.rev)
        self._map[key] = idx
        self.rev.append(orig)
        return idx

    def get(self, name) -> Optional[int]:
        return self._map.get(_js_upper(str(name)))

    def count(self) -> int:
        return len(self.rev)

    @staticmethod
    def matrixIndexOfNode(node_id: int) -> int:
        return -1 if node_id == 0 else node_id - 1


def _js_upper(s: str) -> str:
    return s.upper()


@dataclass
class ParsedResistor:
    name: str
    n1: int
    n2: int
    R: float


@dataclass
class ParsedCapacitor:
    name: str
    n1: int
    n2: int
    C: float
    vPrev: float = 0.0


@dataclass
class ParsedInductor:
    name: str
    n1: int
    n2: int
    L: float
    iPrev: float = 0.0


@dataclass
class ParsedDiodeModel:
    name: str
    Is: float = 1e-14
    N: float = 1.0


@dataclass
class ParsedVSwitchModel:
    name: str
    Ron: float = 1.0
    Roff: float = 1e12
    Von: float = 0.0
    Voff: float = 0.0


@dataclass
class ParsedVoltageSource:
    name: str
    n1: int
    n2: int
    dc: float = 0.0
    acMag: float = 0.0
    acPhaseDeg: float = 0.0
    waveform: Optional[Callable[[float], float]] = None
    index: int = -1


@dataclass
class ParsedDiode:
    name: str
    nPlus: int
    nMinus: int
    modelName: str
    model: Optional[ParsedDiodeModel] = None
    vdPrev: float = 0.0


@dataclass
class ParsedSwitch:
    name: str
    n1: int
    n2: int
    ncPos: int
    ncNeg: int
    modelName: str
    model: Optional[ParsedVSwitchModel] = None
    isOn: bool = False


@dataclass
class ParsedCircuit:
    nodes: NodeIndex = field(default_factory=NodeIndex)
    R: List[ParsedResistor] = field(default_factory=list)
    C: List[ParsedCapacitor] = field(default_factory=list)
    L: List[ParsedInductor] = field(default_factory=list)
    V: List[ParsedVoltageSource] = field(default_factory=list)
    S: List[ParsedSwitch] = field(default_factory=list)
    D: List[ParsedDiode] = field(default_factory=list)
    analyses: Dict[str, Optional[dict]] = field(default_factory=lambda: {"ac": None, "tran": None})
    probes: Dict[str, List[str]] = field(default_factory=lambda: {"tran": []})
    skipped: List[str] = field(default_factory=list)
    models: Dict[str, dict] = field(default_factory=lambda: {"vswitch": {}, "diode": {}})


# --------------------------------------------------------------------------------------------
# numbers

_PLAIN_NUM = re.compile(r"^[+-]?\d*\.?\d+(?:[eE][+-]?\d+)?$", _A)
_NUM_SUFFIX = re.compile(r"^([+-]?\d*\.?\d+(?:[eE][+-]?\d+)?)([a-zA-Z]+)$", _A)
_FLOAT_PREFIX = re.compile(r"^\s*([+-]?(?:Infinity|\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?))", _A)
_INT_PREFIX = re.compile(r"^\s*([+-]?\d+)", _A)
_UNIT_MUL = {"t": 1e12, "g": 1e9, "meg": 1e6, "k": 1e3, "m": 1e-3, "u": 1e-6, "n": 1e-9, "p": 1e-12, "f": 1e-15}


def _parse_float(s: str) -> float:
    """ECMAScript ``parseFloat``: longest numeric prefix, else NaN."""
    m = _FLOAT_PREFIX.match(s)
    if not m:
        return math.nan
    tok = m.group(1)
    if tok.endswith("Infinity"):
        return -math.inf if tok.startswith("-") else math.inf
    return float(tok)


def _parse_int10(s: str) -> float:
    m = _INT_PREFIX.match(s)
    return float(int(m.group(1))) if m else math.nan


def parseNumberWithUnits(raw) -> float:
    """parseNumberWithUnits.ts:1-30 — note the trailing ``ohm|v|a|s|h|f`` strip eats femto."""
    if raw is None:
        return math.nan
    s = str(raw).strip()
    if s == "":
        return math.nan
    if _PLAIN_NUM.match(s):
        return _parse_float(s)
    m = _NUM_SUFFIX.match(s)
    if not m:
        return _parse_float(s)
    val = _parse_float(m.group(1))
    suf = m.group(2).lower()
    suf = re.sub(r"(ohm|v|a|s|h|f)$", "", suf)
    if suf == "meg":
        return val * _UNIT_MUL["meg"]
    if len(suf) == 1 and suf in _UNIT_MUL:
        return val * _UNIT_MUL[suf]
    return val


# --------------------------------------------------------------------------------------------
# waveforms


def _split_args(token: str, kw: str) -> List[str]:
    clean = re.sub(r"^" + kw + r"\s*\(", "(", token.strip(), flags=re.I | _A)
    inside = re.sub(r"\)$", "", re.sub(r"^\(", "", clean)).strip()
    return [x for x in re.split(r"[\s,]+", inside) if len(x)]


def parsePulseArgs(token: str) -> dict:
    parts = _split_args(token, "pulse")
    if len(parts) < 7:
        raise ValueError("PULSE(...) requires 7 or 8 args")
    vals = [parseNumberWithUnits(v) for v in parts]
    if any(math.isnan(v) for v in vals):
        raise ValueError("Invalid PULSE() numeric value")
    return {
        "v1": vals[0], "v2": vals[1], "td": vals[2], "tr": vals[3], "tf": vals[4],
        "ton": vals[5], "period": vals[6], "ncycles": vals[7] if len(parts) > 7 else math.inf,
    }


def parsePwlArgs(token: str) -> List[dict]:
    parts = _split_args(token, "pwl")
    if len(parts) == 0 or len(parts) % 2 != 0:
        raise ValueError("PWL(...) requires an even number of time/value pairs")
    pairs = []
    for i in range(0, len(parts), 2):
        t = parseNumberWithUnits(parts[i])
        v = parseNumberWithUnits(parts[i + 1])
        if math.isnan(t) or math.isnan(v):
            raise ValueError("Invalid PWL() numeric value")
        pairs.append({"t": t, "v": v})
    return pairs


def _js_div(a: float, b: float) -> float:
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0 or math.isnan(a):
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def pulseValue(p: dict, t: float) -> float:
    """pulseValue.ts:4-22, same operation order (IEEE double)."""
    if t < p["td"]:
        return p["v1"]
    tt = t - p["td"]
    q = _js_div(tt, p["period"])
    cycles = math.floor(q) if math.isfinite(q) else q
    if cycles >= p["ncycles"]:
        return p["v1"]
    tc = tt - cycles * p["period"]
    if tc < p["tr"]:
        a = tc / max(p["tr"], EPS)
        return p["v1"] + (p["v2"] - p["v1"]) * a
    if tc < p["tr"] + p["ton"]:
        return p["v2"]
    if tc < p["tr"] + p["ton"] + p["tf"]:
        a = (tc - (p["tr"] + p["ton"])) / max(p["tf"], EPS)
        return p["v2"] + (p["v1"] - p["v2"]) * a
    return p["v1"]


def pwlValue(pairs: List[dict], t: float) -> float:
    """pwlValue.ts:3-16."""
    if len(pairs) == 0:
        return 0.0
    if t <= pairs[0]["t"]:
        return pairs[0]["v"]
    for i in range(1, len(pairs)):
        prev, curr = pairs[i - 1], pairs[i]
        if t <= curr["t"]:
            dt = max(curr["t"] - prev["t"], EPS)
            a = (t - prev["t"]) / dt
            return prev["v"] + (curr["v"] - prev["v"]) * a
    return pairs[-1]["v"]


# --------------------------------------------------------------------------------------------
# parser

_TOKEN_RE = re.compile(r'"[^"]*"|\w+\s*\([^)]*\)|\([^()]*\)|\S+', _A)
_ELEMENT_FIRST = re.compile(r"^[rclvgsmiqd]\w*$", re.I | _A)
_JS_WS = (" \t\n\r\v\f\u00a0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a"
          "\u2028\u2029\u202f\u205f\u3000\ufeff")


def smartTokens(line: str) -> List[str]:
    return _TOKEN_RE.findall(line)


def _require(tokens: List[str], index: int, context: str) -> str:
    if index >= len(tokens):
        raise ValueError(context)
    return tokens[index]


def _model_params(tokens: List[str]):
    type_tok = _require(tokens, 2, ".model missing type")
    typ = type_tok
    params = ""
    if "(" in typ:
        idx = typ.index("(")
        params = typ[idx + 1:]
        typ = typ[:idx]
    rest = " ".join(tokens[3:])
    if not params:
        params = re.sub(r"\)$", "", re.sub(r"^\(", "", rest))
    else:
        params = (params + " " + re.sub(r"\)$", "", rest)).strip(_JS_WS)
    params = re.sub(r"\)$", "", re.sub(r"^\(", "", params)).strip(_JS_WS)
    return typ, params


def _assignments(params: str):
    if len(params) == 0:
        return
    for a in [x for x in re.split(r"[\s,]+", params) if x]:
        bits = a.split("=")
        key_raw = bits[0]
        if not key_raw or len(bits) < 2:
            continue
        value = parseNumberWithUnits(bits[1])
        if math.isnan(value):
            continue
        yield key_raw.lower(), value


def parseNetlist(text: str) -> ParsedCircuit:
    ckt = ParsedCircuit()
    vswitch = ckt.models["vswitch"]
    diode = ckt.models["diode"]
    seen_title = False

    for raw in re.split(r"\r?\n", text):
        line = raw.strip(_JS_WS)
        if not line:
            continue
        if line.startswith("*"):
            continue
        if re.match(r"^\s*\.end\b", line, re.I | _A):
            break
        line = re.sub(r"//.*$", "", line)
        line = re.sub(r";.*$", "", line)

        tokens = smartTokens(line)
        if len(tokens) == 0:
            continue
        first = tokens[0]
        if len(first) == 0:
            continue

        if not seen_title and not _ELEMENT_FIRST.match(first) and not first.startswith("."):
            seen_title = True
            continue

        if first.startswith("."):
            d = first.lower()
            if d == ".ac":
                mode = _require(tokens, 1, ".ac missing mode").lower()
                if mode not in ("dec", "lin"):
                    raise ValueError(".ac supports 'dec' or 'lin'")
                n = _parse_int10(_require(tokens, 2, ".ac missing point count"))
                f1 = parseNumberWithUnits(_require(tokens, 3, ".ac missing start frequency"))
                f2 = parseNumberWithUnits(_require(tokens, 4, ".ac missing stop frequency"))
                ckt.analyses["ac"] = {"mode": mode, "N": n, "f1": f1, "f2": f2}
            elif d == ".tran":
                dt = parseNumberWithUnits(_require(tokens, 1, ".tran missing timestep"))
                tstop = parseNumberWithUnits(_require(tokens, 2, ".tran missing stop time"))
                ckt.analyses["tran"] = {"dt": dt, "tstop": tstop}
            elif d == ".print":
                kind = _require(tokens, 1, ".print missing analysis type").lower()
                if kind == "tran":
                    for tok in tokens[2:]:
                        m = re.match(r"^v\(([^)]+)\)$", tok, re.I)
                        if m and m.group(1):
                            name = m.group(1)
                            if not any(p.upper() == name.upper() for p in ckt.probes["tran"]):
                                ckt.probes["tran"].append(name)
                else:
                    ckt.skipped.append(line)
            elif d == ".model":
                name_tok = _require(tokens, 1, ".model missing name")
                typ, params = _model_params(tokens)
                tl = typ.lower()
                if tl in ("vswitch", "sw"):
                    model = ParsedVSwitchModel(name=name_tok)
                    vt = vh = None
                    for key, value in _assignments(params):
                        if key == "ron":
                            model.Ron = value
                        elif key == "roff":
                            model.Roff = value
                        elif key == "von":
                            model.Von = value
                        elif key == "voff":
                            model.Voff = value
                        elif key == "vt":
                            vt = value
                        elif key == "vh":
                            vh = value
                    if vt is not None:
                        h = vh if vh is not None else 0
                        model.Von = vt + h / 2
                        model.Voff = vt - h / 2
                    vswitch[name_tok.lower()] = model
                elif tl == "d":
                    dm = ParsedDiodeModel(name=name_tok)
                    for key, value in _assignments(params):
                        if key == "is":
                            dm.Is = value
                        elif key == "n":
                            dm.N = value
                    diode[name_tok.lower()] = dm
                else:
                    ckt.skipped.append(line)
            else:
                ckt.skipped.append(line)
            continue

        tc = first[0].lower()
        name = first
        try:
            if tc == "r":
                n1 = ckt.nodes.getOrCreate(_require(tokens, 1, "Resistor missing node"))
                n2 = ckt.nodes.getOrCreate(_require(tokens, 2, "Resistor missing node"))
                val = parseNumberWithUnits(_require(tokens, 3, "Resistor missing value"))
                ckt.R.append(ParsedResistor(name, n1, n2, val))
            elif tc == "c":
                n1 = ckt.nodes.getOrCreate(_require(tokens, 1, "Capacitor missing node"))
                n2 = ckt.nodes.getOrCreate(_require(tokens, 2, "Capacitor missing node"))
                val = parseNumberWithUnits(_require(tokens, 3, "Capacitor missing value"))
                ckt.C.append(ParsedCapacitor(name, n1, n2, val, 0.0))
            elif tc == "l":
                n1 = ckt.nodes.getOrCreate(_require(tokens, 1, "Inductor missing node"))
                n2 = ckt.nodes.getOrCreate(_require(tokens, 2, "Inductor missing node"))
                val = parseNumberWithUnits(_require(tokens, 3, "Inductor missing value"))
                ckt.L.append(ParsedInductor(name, n1, n2, val, 0.0))
            elif tc == "v":
                n1 = ckt.nodes.getOrCreate(_require(tokens, 1, "Voltage source missing node"))
                n2 = ckt.nodes.getOrCreate(_require(tokens, 2, "Voltage source missing node"))
                vs = ParsedVoltageSource(name, n1, n2)
                i = 3
                if i < len(tokens) and not re.match(r"^[a-zA-Z]", tokens[i]):
                    vs.dc = parseNumberWithUnits(tokens[i])
                    i += 1
                while i < len(tokens):
                    key = tokens[i].lower()
                    if key == "dc":
                        vs.dc = parseNumberWithUnits(_require(tokens, i + 1, "DC value missing"))
                        i += 2
                    elif key == "ac":
                        vs.acMag = parseNumberWithUnits(_require(tokens, i + 1, "AC magnitude missing"))
                        ph = tokens[i + 2] if i + 2 < len(tokens) else None
                        if ph is not None and re.match(r"^[+-]?\d", ph, _A):
                            vs.acPhaseDeg = parseNumberWithUnits(ph)
                            i += 3
                        else:
                            i += 2
                    elif key.startswith("pulse"):
                        arg = key if "(" in key else _require(tokens, i + 1, "PULSE() missing arguments")
                        if not arg or not re.search(r"\(.*\)", arg):
                            raise ValueError("Malformed PULSE() specification")
                        p = parsePulseArgs(arg)
                        vs.waveform = (lambda pp: (lambda t: pulseValue(pp, t)))(p)
                        vs.waveform.spec = ("pulse", p)  # type: ignore[attr-defined]
                        i += 1 if "(" in key else 2
                    elif key.startswith("pwl"):
                        arg = key if "(" in key else _require(tokens, i + 1, "PWL() missing arguments")
                        if not arg or not re.search(r"\(.*\)", arg):
                            raise ValueError("Malformed PWL() specification")
                        pairs = parsePwlArgs(arg)
                        vs.waveform = (lambda pp: (lambda t: pwlValue(pp, t)))(pairs)
                        vs.waveform.spec = ("pwl", pairs)  # type: ignore[attr-defined]
                        i += 1 if "(" in key else 2
                    else:
                        i += 1
                ckt.V.append(vs)
            elif tc == "s":
                n1 = ckt.nodes.getOrCreate(_require(tokens, 1, "Switch missing node"))
                n2 = ckt.nodes.getOrCreate(_require(tokens, 2, "Switch missing node"))
                cp = ckt.nodes.getOrCreate(_require(tokens, 3, "Switch missing control node"))
                cn = ckt.nodes.getOrCreate(_require(tokens, 4, "Switch missing control node"))
                mn = _require(tokens, 5, "Switch missing model")
                ckt.S.append(ParsedSwitch(name, n1, n2, cp, cn, mn.lower()))
            elif tc == "d":
                if len(tokens) == 4:
                    np_ = ckt.nodes.getOrCreate(_require(tokens, 1, "Diode missing node"))
                    nm_ = ckt.nodes.getOrCreate(_require(tokens, 2, "Diode missing node"))
                    mn = _require(tokens, 3, "Diode missing model")
                    ckt.D.append(ParsedDiode(name, np_, nm_, mn.lower()))
                else:
                    ckt.skipped.append(line)
            else:
                ckt.skipped.append(line)
        except ValueError as err:
            raise ValueError(f'Parse error on line: "{line}"\n{err}') from None

    n_nodes = ckt.nodes.count() - 1
    for i, vs in enumerate(ckt.V):
        vs.index = n_nodes + i
    for sw in ckt.S:
        model = vswitch.get(sw.modelName)
        if model is None:
            raise ValueError(f"Unknown .model {sw.modelName} referenced by switch {sw.name}")
        sw.model = model
        sw.isOn = False
    for d in ckt.D:
        model = diode.get(d.modelName)
        if model is None:
            raise ValueError(f"Unknown .model {d.modelName} referenced by diode {d.name}")
        d.model = model
    return ckt


def js_object_key_order(keys: List[str]) -> List[str]:
    """Order in which a JS object enumerates string keys: canonical array indices (0 … 2^32-2)
    ascending first, then the rest in insertion order (last bullet)."""
    ints, rest, seen = [], [], set()
    for k in keys:
        if k in seen:
            continue
        seen.add(k)
        if re.match(r"^(0|[1-9]\d*)$", k, _A) and int(k) < 2**32 - 1:
            ints.append(k)
        else:
            rest.append(k)
    ints.sort(key=int)
    return ints + rest
