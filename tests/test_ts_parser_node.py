"""ts/parseNetlist.ts — this build's own TypeScript parser (with ts/NodeIndex.ts, numbers.ts, waveforms.ts) EXECUTED under the
Node 12 of the image (type-erased by tools/node_shim/erase_own_ts.py; no GPU, no native library involved) and held to

  * the reference's own parser on 101 snippets (tests/golden/parser_cases.json, made by tools/js_oracle/make_golden_parse.py
    from the reference itself: structures, node order, waveform samples, every Error text) — identical, key for key;
  * the Python mirror (spicey_amd/netlist.py) on every netlist of tests/golden/netlists/ and on the BASELINE generators;
  * the reference-generated goldens' element tables for the reference's seven test netlists.

So the three parsers of this repository answer every pinned question the same way: TS == Python mirror == reference."""
import json
import math
import os
import shutil
import subprocess

import pytest

from conftest import GOLD, REPO, load_golden
from spicey_amd import synth
from spicey_amd.netlist import parseNetlist

NODE = ["node", "--harmony-nullish", "--harmony-optional-chaining"]
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node not available")
TS = [0, 1e-9, 5e-7, 1e-6, 2.5e-6, 1e-5, 3.3e-5, 1e-4, 1e-3, 0.0123, 1]


def _ts_parse(tmp_path, texts, ts=TS):
    erased = str(tmp_path / "erased")
    subprocess.run(["python3", os.path.join(REPO, "tools", "node_shim", "erase_own_ts.py"), erased], check=True)
    cj, oj = str(tmp_path / "cases.json"), str(tmp_path / "out.json")
    json.dump({"ts": ts, "cases": texts}, open(cj, "w"))
    r = subprocess.run(NODE + [os.path.join(REPO, "tests", "node", "run_parser.mjs"), erased, cj, oj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return json.load(open(oj))["results"]


def _num(x):
    return str(x).replace("inf", "Infinity").replace("nan", "NaN") if isinstance(x, float) and not math.isfinite(x) else x


def _py_parse(text, ts=TS):
    """The Python mirror's answer in the fixture's layout."""
    try:
        c = parseNetlist(text)
    except ValueError as e:
        return {"error": str(e)}
    return {
        "nodes": c.nodes.rev,
        "R": [[e.name, e.n1, e.n2, _num(e.R)] for e in c.R],
        "C": [[e.name, e.n1, e.n2, _num(e.C), _num(e.vPrev)] for e in c.C],
        "L": [[e.name, e.n1, e.n2, _num(e.L), _num(e.iPrev)] for e in c.L],
        "V": [[e.name, e.n1, e.n2, _num(e.dc), _num(e.acMag), _num(e.acPhaseDeg), e.index, [_num(e.waveform(t)) for t in ts] if e.waveform else None] for e in c.V],
        "S": [[e.name, e.n1, e.n2, e.ncPos, e.ncNeg, e.modelName, e.isOn, [e.model.name, _num(e.model.Ron), _num(e.model.Roff), _num(e.model.Von), _num(e.model.Voff)] if e.model else None] for e in c.S],
        "D": [[e.name, e.nPlus, e.nMinus, e.modelName, _num(e.vdPrev), [e.model.name, _num(e.model.Is), _num(e.model.N)] if e.model else None] for e in c.D],
        "analyses": json.loads(json.dumps(c.analyses)), "probes": c.probes, "skipped": c.skipped,
    }


def _same(a, b):
    """JSON values equal, numbers bit for bit (1 == 1.0 across the two JSON writers; -0 and 0 differ)."""
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, list) and isinstance(b, list):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool) and not isinstance(b, bool):
        return float(a) == float(b) and math.copysign(1.0, float(a)) == math.copysign(1.0, float(b))
    return a == b


def test_ts_parser_answers_like_the_reference_on_the_parser_fixture(tmp_path):
    g = json.load(open(os.path.join(GOLD, "parser_cases.json")))
    assert len(g["cases"]) == 101 and g["ts"] == TS
    got = _ts_parse(tmp_path, g["cases"])
    nerr = 0
    for text, mine, ref in zip(g["cases"], got, g["results"]):
        extra = {k: mine.pop(k) for k in ("count", "ground", "row") if k in mine}
        assert _same(mine, ref), (text, mine, ref)
        if "error" in ref:
            nerr += 1
        else:
            assert extra == {"count": len(ref["nodes"]), "ground": 0, "row": len(ref["nodes"]) - 2}
        assert _same(_py_parse(text), ref), text  # and so does the Python mirror
    assert nerr >= 15  # (the fixture holds every Error text of the parser)


def test_ts_parser_matches_the_python_mirror_on_every_netlist_of_the_suite(tmp_path):
    names = sorted(f for f in os.listdir(os.path.join(GOLD, "netlists")) if f.endswith(".cir"))
    texts = [open(os.path.join(GOLD, "netlists", f)).read() for f in names]
    texts += [synth.rc_ladder(40, seed=1), synth.diode_chain(40, seed=2), synth.rcd_mesh(6, seed=3), synth.rc_ladder(1000, seed=1), synth.diode_chain(1000, seed=2)]
    names += ["rc_ladder(40)", "diode_chain(40)", "rcd_mesh(6)", "rc_ladder(1000)", "diode_chain(1000)"]
    assert len(texts) >= 45
    got = _ts_parse(tmp_path, texts)
    for name, text, mine in zip(names, texts, got):
        for k in ("count", "ground", "row"):
            mine.pop(k, None)
        assert _same(mine, _py_parse(text)), name
    # the reference's own seven test netlists: element tables as the reference parsed them (goldens made by the reference)
    for name in ("two_probes", "transient01", "case_insensitive", "switch_vt_vh", "vswitch_pwl", "diode_switch", "boost_probe"):
        g = load_golden(name)
        mine = got[names.index(name + ".cir")]
        assert mine["nodes"] == g["nodes"] and mine["probes"]["tran"] == g["probes"] and mine["skipped"] == g["skipped"]
        assert [[r[0], r[1], r[2], r[3]] for r in mine["R"]] == g["elements"]["R"]
        assert [[r[0], r[1], r[2], r[3]] for r in mine["C"]] == g["elements"]["C"]
        assert [[r[0], r[1], r[2], r[3]] for r in mine["L"]] == g["elements"]["L"]
        assert [[s[0], s[1], s[2], s[3], s[4]] + s[7][1:] for s in mine["S"]] == g["elements"]["S"]
        assert [[d[0], d[1], d[2]] + d[5][1:] for d in mine["D"]] == g["elements"]["D"]
        assert mine["analyses"]["tran"] == g["tranSpec"]


def test_ts_package_is_self_contained():
    """ts/ imports nothing from the reference tree (`../lib/...` imports made the layer unusable on its own)."""
    import re
    for f in sorted(os.listdir(os.path.join(REPO, "ts"))):
        if not f.endswith(".ts"):
            continue
        src = open(os.path.join(REPO, "ts", f)).read()
        for spec in re.findall(r'from "([^"]+)"', src):
            assert spec == "bun:ffi" or (spec.startswith("./") and os.path.exists(os.path.join(REPO, "ts", spec[2:] + ".ts"))), (f, spec)
