"""The reference's own test-suite, assertion for assertion, against this implementation's public API
(spicey_amd.api = lib/index.ts): tests/basics/basics01.test.ts and tests/transient/*.test.ts.  The netlists are the
test inputs those files embed (tests/golden/netlists/); SVG snapshot comparisons are covered by test_oracle.py's decoded
series, the ngspice comparison of boost-converter-probe needs `eecircuit-engine` (absent offline).

Every test runs twice: with the oracle backend on the CPU (host-layer logic) and, marked `gpu`, with the default HIP
backend — the way a user of the reference would run it after switching."""
import os

import numpy as np
import pytest

from conftest import GOLD
from spicey_amd.api import formatAcResult, formatTranResult, parseNetlist, simulate, spiceyTranToVGraphs


def _net(name):
    return open(os.path.join(GOLD, "netlists", name + ".cir")).read()


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, oracle_backend):
    return oracle_backend if request.param == "oracle" else None  # None = the default (HIP) backend


def _sampler(times, *series):
    t = np.asarray(times)

    def sample(target):
        i = int(np.argmin(np.abs(t - target)))  # first minimum, like the reference's `diff < bestDiff`
        return [s[i] for s in series]
    return sample


def test_basics01(backend):
    """tests/basics/basics01.test.ts:15-219 (inline snapshot; also README.md:21-33)."""
    from conftest import load_golden
    result = simulate(_net("ac_readme"), backend=backend)
    text = formatAcResult(result["ac"])
    assert text == load_golden("ac_readme")["formatted"]
    assert text.split("\n")[1] == "1.00000, 1.00000,0.00000, 0.999822,-1.07987"
    assert text.split("\n")[-1] == "100.000, 1.00000,0.00000, 0.468650,-62.0533"


def test_case_insensitive_nodes(backend):
    """tests/transient/case-insensitive-nodes.test.ts:23-41."""
    r = simulate(_net("case_insensitive"), backend=backend)
    circuit, tran = r["circuit"], r["tran"]
    assert circuit.nodes.count() == 3 and circuit.nodes.rev == ["0", "nOdE1", "nOde2"]
    assert sorted(circuit.probes["tran"]) == sorted(["NODE2", "node1"])
    assert tran is not None
    nv = tran["nodeVoltages"]
    assert sorted(nv) == sorted(["nOde2", "nOdE1"])
    assert len(nv["nOdE1"]) > 10 and len(nv["nOde2"]) > 10
    text = formatTranResult(tran)
    assert "nOdE1:V" in text and "nOde2:V" in text


def test_two_probes(backend):
    """tests/transient/two-probes.test.ts:25-52."""
    r = simulate(_net("two_probes"), backend=backend)
    circuit, tran = r["circuit"], r["tran"]
    assert circuit.probes["tran"] == ["1", "2"] and tran is not None
    nv = tran["nodeVoltages"]
    assert sorted(nv) == ["1", "2"] and len(nv["1"]) > 10 and len(nv["2"]) > 10
    assert abs(nv["1"][0]) < 0.005 and abs(nv["2"][0]) < 0.005  # toBeCloseTo(0)
    assert "t(s), 1:V, 2:V" in formatTranResult(tran)
    assert len(spiceyTranToVGraphs(tran, circuit, "two_probes_test")) == 2


def test_transient01(backend):
    """tests/transient/transient01.test.ts: runs, yields one graph per node (the SVG itself: test_oracle.py)."""
    r = simulate(_net("transient01"), backend=backend)
    assert r["tran"] is not None
    graphs = spiceyTranToVGraphs(r["tran"], r["circuit"], "rc_pulse")
    assert len(graphs) == len(r["tran"]["nodeVoltages"]) and all(len(g["timestamps_ms"]) == len(g["voltage_levels"]) for g in graphs)


def test_switch_vt_vh(backend):
    """tests/transient/switch-vt-vh.test.ts:33-36,61-70."""
    r = simulate(_net("switch_vt_vh"), backend=backend)
    circuit, tran = r["circuit"], r["tran"]
    m = circuit.S[0].model
    assert abs(m.Von - 2.55) < 5e-3 and abs(m.Voff - 2.45) < 5e-3
    assert tran is not None
    sample = _sampler(tran["times"], tran["nodeVoltages"]["N2"])
    assert sample(0.0002)[0] > 4.9   # control high, switch ON
    assert sample(0.0007)[0] < 0.1   # control low, switch OFF
    assert sample(0.0012)[0] > 4.9
    assert sample(0.0017)[0] < 0.1


def test_vswitch_pwl(backend):
    """tests/transient/vswitch-pwl.test.ts:28-76."""
    r = simulate(_net("vswitch_pwl"), backend=backend)
    circuit, tran = r["circuit"], r["tran"]
    assert len(circuit.S) == 1
    m = circuit.S[0].model
    assert abs(m.Ron - 1) < 5e-7 and abs(m.Roff - 1e9) < 0.5 and abs(m.Von - 2) < 5e-7 and abs(m.Voff - 1) < 5e-7
    assert tran is not None
    nv = tran["nodeVoltages"]
    assert "OUT" in nv and "CTRL" in nv
    sample = _sampler(tran["times"], nv["OUT"], nv["CTRL"])
    out, ctrl = sample(0.0005)
    assert ctrl > 2 and abs(out) < 0.02           # earlyOn
    out, ctrl = sample(0.0035)
    assert ctrl < 1 and out > 2                   # afterOff
    out, ctrl = sample(0.0045)
    assert ctrl < 2 and out > 4                   # stillOffBeforeReon
    out, ctrl = sample(0.0085)
    assert ctrl > 1 and abs(out) < 0.02           # onAgain
    out, ctrl = sample(0.0095)
    assert abs(ctrl) < 5e-10 and out > 2          # finalRecharge: toBeCloseTo(0, 9)


def test_diode_switch(backend):
    """tests/transient/diode-switch.test.ts:22-41."""
    r = simulate(_net("diode_switch"), backend=backend)
    circuit, tran = r["circuit"], r["tran"]
    assert len(circuit.D) == 1 and len(circuit.S) == 1
    assert "d" in circuit.models["diode"] and "swmod" in circuit.models["vswitch"]
    assert circuit.models["diode"]["d"].Is == 1e-14      # default
    assert circuit.models["vswitch"]["swmod"].Ron == 1   # default
    assert tran is not None
    text = formatTranResult(tran)
    assert "t(s)," in text and len(text.split("\n")) > 10


def test_boost_converter_probe(backend):
    """tests/transient/boost-converter-probe.test.ts:34 (+ the probe set; the ngspice statistics need eecircuit-engine)."""
    r = simulate(_net("boost_probe"), backend=backend)
    assert r["tran"] is not None
    probes = [p.upper() for p in r["circuit"].probes["tran"]]
    assert probes and all(k.upper() in probes for k in r["tran"]["nodeVoltages"])
    v = np.concatenate([np.asarray(s) for s in r["tran"]["nodeVoltages"].values()])
    assert np.all(np.isfinite(v))
