"""Pin the oracle (oracle/spicey_ref.c + the Python parser/flatten mirror) bit-for-bit against
outputs of the reference's own TRAN path (tests/golden/*.json, made by
tools/js_oracle/make_golden.py) and against the reference's decoded SVG snapshots."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import (GOLD, LARGE_GOLDENS, PROBE_GOLDENS, QUIRK_GOLDENS, SINGULAR_GOLDENS, SKIP_CASES, SMALL_GOLDENS, bits_equal, farr, fnum, golden_netlist, load_golden)
from spicey_amd import abi
from spicey_amd.netlist import parseNetlist
from spicey_amd.simulate import SingularMatrixError, formatTranResult, simulateTRAN


@pytest.mark.parametrize("name", SMALL_GOLDENS + sorted(set(QUIRK_GOLDENS) | set(SKIP_CASES)) + PROBE_GOLDENS)
def test_small_goldens_bit_exact(name, oracle_backend):
    g = load_golden(name)
    ckt = parseNetlist(golden_netlist(g))
    assert ckt.nodes.rev == g["nodes"]
    assert ckt.probes["tran"] == g["probes"]
    assert ckt.skipped == g["skipped"]
    assert {k: len(getattr(ckt, k)) for k in "RCLVSD"} == g["counts"]
    for kind in "RCL":
        assert [[e.name, e.n1, e.n2, getattr(e, kind)] for e in getattr(ckt, kind)] == g["elements"][kind]
    assert [[s.name, s.n1, s.n2, s.ncPos, s.ncNeg, s.model.Ron, s.model.Roff, s.model.Von, s.model.Voff] for s in ckt.S] == g["elements"]["S"]
    assert [[d.name, d.nPlus, d.nMinus, d.model.Is, d.model.N] for d in ckt.D] == g["elements"]["D"]
    for ri, run in enumerate(g["runs"]):  # run 2 continues from run 1's end state
        res = simulateTRAN(ckt, backend=oracle_backend)
        assert list(res["nodeVoltages"]) == run["keysV"]
        assert list(res["elementCurrents"]) == run["keysI"]
        assert res["times"] == run["times"]
        for k in run["keysV"]:
            assert bits_equal(res["nodeVoltages"][k], farr(run["V"][k])).all(), (name, ri, k)
        for k in run["keysI"]:
            assert bits_equal(res["elementCurrents"][k], farr(run["I"][k])).all(), (name, ri, k)
        assert [c.vPrev for c in ckt.C] == run["state"]["C_vPrev"]
        assert [l.iPrev for l in ckt.L] == run["state"]["L_iPrev"]
        assert [d.vdPrev for d in ckt.D] == run["state"]["D_vdPrev"]
        assert [int(s.isOn) for s in ckt.S] == run["state"]["S_isOn"]
        if ri == 0 and "formatted_head" in run:
            assert formatTranResult(res).split("\n")[:4] == run["formatted_head"]


def test_unmatched_probes_give_no_node_voltages(oracle_backend):
    """`.PRINT TRAN v(zz)` with names that match no node: the reference's filter (simulateTRAN.ts:240-249) leaves
    nodeVoltages = {} (its parser does not intern probe names, parseNetlist.ts:196-206) — pinned by a reference-generated
    golden above; here the two neighbouring cases through the same host code (an empty device-side column list
    means "all nodes" and must not be taken for this case)."""
    g = load_golden("probe_unmatched")
    assert g["runs"][0]["keysV"] == [] and g["probes"] == ["zz", "out"]
    text = golden_netlist(g)
    res = simulateTRAN(parseNetlist(text), backend=oracle_backend)
    assert res["nodeVoltages"] == {} and list(res["elementCurrents"]) == g["runs"][0]["keysI"]
    res = simulateTRAN(parseNetlist(text.replace("V(zz) V(out)", "V(zz) V(2)")), backend=oracle_backend)
    assert list(res["nodeVoltages"]) == ["2"]
    res = simulateTRAN(parseNetlist(text.replace(".PRINT TRAN V(zz) V(out)\n", "")), backend=oracle_backend)
    assert list(res["nodeVoltages"]) == ["1", "2"]


@pytest.mark.parametrize("name", sorted(SKIP_CASES))
def test_oracle_counts_what_the_reference_skips(name, oracle_backend):
    """The checker's knobs around `if (Math.abs(f) < EPS) continue` (solveReal.ts:45): the count of NONZERO multipliers the
    reference's own algorithm dropped on each skip case, and the same algorithm with the line switched off (which is what
    the product is compared with where the reference skips: tests/test_program_emul.py, tests/test_gpu_parity.py)."""
    from oracle.pyoracle import OracleBackend
    g = load_golden(name)
    ckt = parseNetlist(golden_netlist(g))
    dt, steps = abi.computeEffectiveTimeStep(ckt.analyses["tran"]["dt"], ckt.analyses["tran"]["tstop"])
    flat, src = abi.flatten(ckt), abi.source_table(ckt, dt, steps)
    ref = oracle_backend.run(flat, steps, dt, src)
    assert (ref["skipped"][0] > 0) == SKIP_CASES[name][0]
    nos = OracleBackend(skip_off=True).run(flat, steps, dt, src)
    assert nos["status"] == 0 and np.array_equal(nos["iters"], ref["iters"])
    if ref["skipped"][0] == 0:
        assert bits_equal(nos["out_v"], ref["out_v"]).all()  # the knob changes nothing where the reference never skips


@pytest.mark.parametrize("name", LARGE_GOLDENS)
def test_large_goldens_sha256(name, oracle_backend):
    """1000-unknown prefixes of BASELINE configs 2/3 (SURVEY.md Appendix C spot values + sha256)."""
    g = load_golden(name)
    ckt = parseNetlist(golden_netlist(g))
    res = simulateTRAN(ckt, backend=oracle_backend, as_lists=False)
    assert list(res["nodeVoltages"]) == g["keysV"] and list(res["elementCurrents"]) == g["keysI"]
    V = np.stack([res["nodeVoltages"][k] for k in g["keysV"]], axis=1)
    I = np.stack([res["elementCurrents"][k] for k in g["keysI"]], axis=1)
    assert V.shape[0] == g["npoints"]
    assert hashlib.sha256(np.ascontiguousarray(V).tobytes()).hexdigest() == g["sha256_V"]
    assert hashlib.sha256(np.ascontiguousarray(I).tobytes()).hexdigest() == g["sha256_I"]
    for k, series in g["V_nodes"].items():
        assert bits_equal(res["nodeVoltages"][k], farr(series)).all()


def test_appendix_c_spot_values(oracle_backend):
    g = load_golden("rc1000_200")
    assert g["sha256_V"] == "54438181f9e09096127a9163c41f81e142c606eb6e6c55f5176fb1e27b1b1ea8"
    assert g["V_nodes"]["n2"][1] == 4.510237884227233 and g["V_nodes"]["n1000"][200] == 2.303629467015315e-06
    g = load_golden("dchain1000_200")
    assert g["sha256_V"] == "908061ae7a0cb84f93b64747a763f629b659e28d14fe78b74e8b152c5cfb76cd"
    assert g["V_nodes"]["n2"][1] == 3.613634986544149 and g["V_nodes"]["n2"][200] == 0.7504015734687133


@pytest.mark.parametrize("name", SINGULAR_GOLDENS)
def test_singular_errors(name, oracle_backend):
    g = load_golden(name)
    assert g["error"] == "Singular matrix (real)"
    ckt = parseNetlist(golden_netlist(g))
    with pytest.raises(SingularMatrixError, match=r"Singular matrix \(real\)"):
        simulateTRAN(ckt, backend=oracle_backend)


def test_readme_rc_config1(oracle_backend):
    """BASELINE config 1: README netlist + .tran 1us 10ms -> 10000 steps, Nvar 3, all zero."""
    g = load_golden("readme_rc")
    ckt = parseNetlist(golden_netlist(g))
    res = simulateTRAN(ckt, backend=oracle_backend, as_lists=False)
    run = g["runs"][0]
    assert len(res["times"]) == run["npoints"] == 10001
    assert [res["times"][0], res["times"][1], res["times"][-1]] == run["times_first_last"]
    assert list(res["nodeVoltages"]) == run["keysV"] and list(res["elementCurrents"]) == run["keysI"]
    assert all(not np.any(v) for v in res["nodeVoltages"].values())


def test_svg_snapshots(oracle_backend):
    """Oracle-independent cross-check: the reference's five committed SVG snapshots (11 series)."""
    svg = json.load(open(os.path.join(GOLD, "svg_series.json")))
    n_series = 0
    for snap, rec in svg.items():
        g = load_golden(rec["netlist"])
        ckt = parseNetlist(golden_netlist(g))
        res = simulateTRAN(ckt, backend=oracle_backend, as_lists=False)
        byupper = {k.upper(): v for k, v in res["nodeVoltages"].items()}
        for label, series in rec["series"].items():
            node = label[2:-1].upper()
            got = byupper[node]
            assert len(got) == len(series)
            # y printed to 6 decimals: half a quantum + slack
            assert np.max(np.abs(got - np.array(series))) <= 0.6 * rec["quantum"], (snap, label)
            n_series += 1
    assert n_series == 11


def test_reference_test_thresholds(oracle_backend):
    """Numeric pins of switch-vt-vh.test.ts:33-34,61-70 and vswitch-pwl.test.ts:58-76."""
    g = load_golden("switch_vt_vh")
    ckt = parseNetlist(golden_netlist(g))
    m = ckt.S[0].model
    assert abs(m.Von - 2.55) < 5e-3 and abs(m.Voff - 2.45) < 5e-3
    res = simulateTRAN(ckt, backend=oracle_backend, as_lists=False)
    t = np.array(res["times"])
    v = res["nodeVoltages"]["N2"]
    s = lambda x: v[np.argmin(np.abs(t - x))]  # noqa: E731
    assert s(0.0002) > 4.9 and s(0.0007) < 0.1 and s(0.0012) > 4.9 and s(0.0017) < 0.1
    g = load_golden("vswitch_pwl")
    ckt = parseNetlist(golden_netlist(g))
    res = simulateTRAN(ckt, backend=oracle_backend, as_lists=False)
    assert len(res["times"]) == 1001  # dt defaulted -> 1000 steps


def test_timestep_rounding():
    """same double ops in the same order as simulateTRAN.ts:14-19."""
    from oracle import pyoracle
    for dt_req, tstop, want in [(1e-6, 0.1, 100001), (1e-6, 1e-2, 10000), (0.0, 0.01, 1000), (3e-6, 1e-6, 1),
                                (0.3e-6, 1e-6, 4), (1e-7, 2e-5, 201), (0.1 * 1e-6, 20 * 1e-6, 200)]:
        dt, steps = abi.computeEffectiveTimeStep(dt_req, tstop)
        assert steps == want, (dt_req, tstop, steps)
        assert pyoracle.timestep(dt_req, tstop) == (dt, steps)


def test_format_tran_result_vectorised_matches_scalar():
    """formatTranResult from typed arrays (vectorised toPrecision(6)) equals the scalar definition value by value."""
    from spicey_amd.simulate import _to_precision6, _to_precision6_array
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(2000) * 10.0 ** rng.integers(-12, 12, 2000), [0.0, -0.0, 1e-7, 9.999995, 999999.5, 1e21,
                        123456.5, 0.000001, 5e-324, np.inf, -np.inf, np.nan, 4.9999995e-5, 1e5, 99999.95]])
    got = _to_precision6_array(x)
    for v, g in zip(x, got):
        assert g == _to_precision6(float(v)), (v, g, _to_precision6(float(v)))


@pytest.mark.parametrize("name", ["two_probes", "switch_vt_vh"])
def test_vgraph_formatter_matches_reference(name, oracle_backend):
    """spiceyTranToVGraphs / eecEngineTranToVGraphs (formatToVGraph.ts:11-65) against the reference's own output."""
    from spicey_amd.simulate import eecEngineTranToVGraphs, spiceyTranToVGraphs
    g = load_golden("vgraph_" + name)
    ckt = parseNetlist(golden_netlist(g))
    res = simulateTRAN(ckt, backend=oracle_backend)
    graphs = spiceyTranToVGraphs(res, ckt, "exp_1")
    assert json.loads(json.dumps(graphs)) == g["graphs"]  # keys, ids, names, bit-identical doubles (shortest round-trip)
    assert [list(a) for a in graphs] == [list(b) for b in g["graphs"]]  # and the reference's key order
    eec = eecEngineTranToVGraphs({"time_s": [0, 1e-3, 2.5e-3], "voltages": {"out": [0, 1.5, 3.25], "2": [1, 2, 3]}}, ckt, "exp_2")
    assert json.loads(json.dumps(eec)) == g["eec"] and [e["name"] for e in eec] == [e["name"] for e in g["eec"]]
    assert spiceyTranToVGraphs(None, ckt, "x") == []


def test_to_precision6_matches_js_engine():
    """Number.prototype.toPrecision(6) of V8 (tests/golden/toprecision6.json: 9 701 values incl. exact ties, the
    1e-7 / 1e6 notation switches, denormals, +-0, NaN, Infinity): the scalar and vectorised Python formatters and the
    native one of libspicey_hip.so (host code; loads without a GPU) all reproduce it character for character."""
    from spicey_amd.simulate import _to_precision6, _to_precision6_array
    g = load_golden("toprecision6")
    vals = np.array([float(v) for v in g["values"]])
    assert [_to_precision6(float(v)) for v in vals] == g["strings"]
    assert list(_to_precision6_array(vals)) == g["strings"]
    from spicey_amd.lib import format_tran_native, to_precision6_native
    assert [to_precision6_native(v) for v in vals] == g["strings"]
    # and the table formatter (threads, column selection, header) against the line-by-line composition
    n = 1500
    t = vals[:n]
    mat = np.stack([vals[100:100 + n], vals[2000:2000 + n], vals[4000:4000 + n]], axis=1)
    text = format_tran_native(t, mat, [2, 0], "t(s), b:V, a:V")
    want = ["t(s), b:V, a:V"] + [", ".join([g["strings"][k], g["strings"][4000 + k], g["strings"][100 + k]]) for k in range(n)]
    assert text == "\n".join(want)
    big = np.tile(mat, (200, 1))  # 300 000 x 3: the multi-threaded path
    tb = np.tile(t, 200)
    assert format_tran_native(tb, big, [0, 1, 2], "h").split("\n")[1:] == [", ".join([g["strings"][k % n], g["strings"][100 + k % n], g["strings"][2000 + k % n], g["strings"][4000 + k % n]]) for k in range(len(tb))]


def test_parser_matches_reference_on_tricky_netlists():
    """spicey_amd/netlist.py against the reference's own parseNetlist (tests/golden/parser_cases.json, 101 snippets:
    unit suffixes, titles / comments / continuations, source specifications incl. waveform samples, models, analysis
    cards, and the text of every Error)."""
    g = load_golden("parser_cases")
    ts = g["ts"]
    enc = lambda x: (("Infinity" if x > 0 else "-Infinity") if isinstance(x, float) and x in (float("inf"), float("-inf")) else
                     ("NaN" if isinstance(x, float) and x != x else x))
    bad = []
    for idx, (text, want) in enumerate(zip(g["cases"], g["results"])):
        try:
            c = parseNetlist(text)
            got = {
                "nodes": c.nodes.rev,
                "R": [[e.name, e.n1, e.n2, enc(e.R)] for e in c.R],
                "C": [[e.name, e.n1, e.n2, enc(e.C), enc(e.vPrev)] for e in c.C],
                "L": [[e.name, e.n1, e.n2, enc(e.L), enc(e.iPrev)] for e in c.L],
                "V": [[e.name, e.n1, e.n2, enc(e.dc), enc(e.acMag), enc(e.acPhaseDeg), e.index,
                       [enc(e.waveform(t)) for t in ts] if e.waveform else None] for e in c.V],
                "S": [[e.name, e.n1, e.n2, e.ncPos, e.ncNeg, e.modelName, e.isOn,
                       [e.model.name, enc(e.model.Ron), enc(e.model.Roff), enc(e.model.Von), enc(e.model.Voff)] if e.model else None] for e in c.S],
                "D": [[e.name, e.nPlus, e.nMinus, e.modelName, enc(e.vdPrev), [e.model.name, enc(e.model.Is), enc(e.model.N)] if e.model else None] for e in c.D],
                "analyses": c.analyses, "probes": c.probes, "skipped": c.skipped,
            }
        except Exception as e:  # noqa: BLE001 — the reference throws plain Errors; the message is what is pinned
            got = {"error": str(e)}
        if json.loads(json.dumps(got)) != want:
            bad.append((idx, text, got, want))
    assert not bad, "\n\n".join(f"case {i}: {t!r}\n got  {a}\n want {w}" for i, t, a, w in bad[:6]) + f"\n... {len(bad)} mismatches"
