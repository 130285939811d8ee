"""Pin the AC oracle (oracle/spicey_ref_ac.c + the Python simulateAC / formatAcResult mirror) bit-for-bit against
outputs of the reference's own AC path (tests/golden/ac_*.json, made by tools/js_oracle/make_golden_ac.py) and
against the reference's inline snapshot of tests/basics/basics01.test.ts."""
import hashlib
import struct

import numpy as np
import pytest

from conftest import bits_equal, golden_netlist, load_golden
from spicey_amd import abi, ac as sac
from spicey_amd.netlist import parseNetlist
from spicey_amd.simulate import simulate

AC_SMALL = ["ac_readme", "ac_rlc", "ac_two_src", "ac_fv", "ac_ladder30", "ac_mesh6"]


def ac_golden_netlist(g):
    if "netlist_file" in g:
        return golden_netlist(g)
    from spicey_amd import synth
    gen, kw, _ = g["generator"]  # the generator's V1 line gets an `ac 1` phasor (make_golden_ac.ac_netlist)
    return "\n".join(ln + " ac 1" if ln.startswith("V1 ") else ln for ln in getattr(synth, gen)(**kw).split("\n"))


def cplx(pairs):
    a = np.asarray(pairs, dtype=np.float64).reshape(-1, 2)
    return a[:, 0] + 1j * a[:, 1]


def cbits(a, b):
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    # "+ 0.0" folds -0 into +0: JSON.stringify(-0) is "0", so the goldens cannot carry the sign of a zero
    return bits_equal(a.real + 0.0, b.real + 0.0).all() and bits_equal(a.imag + 0.0, b.imag + 0.0).all()


@pytest.mark.parametrize("name", AC_SMALL)
def test_ac_goldens_bit_exact(name, oracle_backend):
    g = load_golden(name)
    ckt = parseNetlist(ac_golden_netlist(g))
    assert ckt.nodes.rev == g["nodes"]
    assert {k: len(getattr(ckt, k)) for k in "RCLVSD"} == g["counts"]
    assert ckt.analyses["ac"] == g["acSpec"]
    # host-side pieces: engine-defined Math.pow / cos / sin agree with this host's libm to the last bits
    freqs = sac.buildFrequencyArray(**{k: g["acSpec"][k] for k in ("mode", "N", "f1", "f2")})
    assert len(freqs) == len(g["freqs"]) and np.allclose(freqs, g["freqs"], rtol=4e-16, atol=0)
    assert np.allclose(sac.source_phasors(ckt), cplx(g["vph"]), rtol=0, atol=1e-16)
    # the solve, on the reference's own frequency list and phasors: bit for bit
    flat = abi.flatten(ckt)
    raw = oracle_backend.run_ac(flat, np.array(g["freqs"]), cplx(g["vph"]))
    assert raw["status"] == 0
    res = sac.simulateAC(ckt, backend=_Fixed(oracle_backend, cplx(g["vph"])), freqs=g["freqs"])
    assert list(res["nodeVoltages"]) == g["keysV"] and list(res["elementCurrents"]) == g["keysI"]
    for k in g["keysV"]:
        assert cbits(res["nodeVoltages"][k], cplx(g["V"][k])), (name, k)
    for k in g["keysI"]:
        assert cbits(res["elementCurrents"][k], cplx(g["I"][k])), (name, k)
    assert sac.formatAcResult(res) == g["formatted"]


class _Fixed:
    """Backend wrapper that substitutes the golden's phasors (the JS engine's cos / sin) for the host's."""

    def __init__(self, be, vph):
        self.be, self.vph = be, vph

    def run_ac(self, flat, freqs, vph, want_currents=True):
        return self.be.run_ac(flat, freqs, self.vph, want_currents)


def test_ac_reference_inline_snapshot(oracle_backend):
    """/root/reference/tests/basics/basics01.test.ts:15-219 (also README.md:21-33): spot lines of the snapshot, through
    the public simulate() with this host's own frequency list."""
    g = load_golden("ac_readme")
    out = simulate(golden_netlist(g), backend=oracle_backend)
    text = sac.formatAcResult(out["ac"]).split("\n")
    assert len(text) == 202 and out["tran"] is None
    assert text[0] == "f(Hz), 1:|V|,∠V(deg), 2:|V|,∠V(deg)"
    assert text[1] == "1.00000, 1.00000,0.00000, 0.999822,-1.07987"
    assert text[2] == "1.02329, 1.00000,0.00000, 0.999814,-1.10502"
    assert text[101] == "10.0000, 1.00000,0.00000, 0.982695,-10.6747"
    assert text[201] == "100.000, 1.00000,0.00000, 0.468650,-62.0533"
    assert "\n".join(text) == g["formatted"]


def test_ac_large_golden_sha256(oracle_backend):
    """BASELINE-sized topology (1001 unknowns) at 16 frequencies: sha256 over every node voltage / element current."""
    g = load_golden("ac_rc1000")
    ckt = parseNetlist(ac_golden_netlist(g))
    flat = abi.flatten(ckt)
    raw = oracle_backend.run_ac(flat, np.array(g["freqs"]), cplx(g["vph"]))
    assert raw["status"] == 0
    res = sac.simulateAC(ckt, backend=_Fixed(oracle_backend, cplx(g["vph"])), freqs=g["freqs"])
    assert len(res["nodeVoltages"]) == g["nkeysV"] and len(res["elementCurrents"]) == g["nkeysI"]
    assert list(res["nodeVoltages"])[:5] == g["keysV_head"] and list(res["elementCurrents"])[:5] == g["keysI_head"]
    for k, v in g["V"].items():
        assert cbits(res["nodeVoltages"][k], cplx(v)), k
    for k, v in g["I"].items():
        assert cbits(res["elementCurrents"][k], cplx(v)), k
    for key, series in (("sha256_V", res["nodeVoltages"]), ("sha256_I", res["elementCurrents"])):
        h = hashlib.sha256()
        for fi in range(len(g["freqs"])):
            for name in series:
                z = series[name][fi]
                h.update(struct.pack("<2d", z.real, z.imag))
        assert h.hexdigest() == g[key]


def test_ac_errors_and_absent_card(oracle_backend):
    g = load_golden("ac_err_r0")
    with pytest.raises(ValueError, match="R R1 must be > 0"):
        sac.simulateAC(parseNetlist(golden_netlist(g)), backend=oracle_backend)
    assert g["error"] == "R R1 must be > 0"
    g = load_golden("ac_err_float")
    with pytest.raises(sac.SingularComplexMatrixError, match=r"Singular matrix \(complex\)"):
        sac.simulateAC(parseNetlist(golden_netlist(g)), backend=oracle_backend)
    assert g["error"] == "Singular matrix (complex)"
    g = load_golden("ac_none")
    assert g.get("none") and sac.simulateAC(parseNetlist(golden_netlist(g)), backend=oracle_backend) is None
    assert sac.formatAcResult(None) == "No AC analysis.\n"


def test_frequency_arrays():
    assert sac.buildFrequencyArray("lin", 5, 10.0, 50.0) == [10.0, 20.0, 30.0, 40.0, 50.0]
    assert sac.buildFrequencyArray("lin", 1, 10.0, 50.0) == [10.0, 50.0]  # npts = max(2, N)
    assert len(sac.logspace(1, 100, 100)) == 201 and sac.logspace(100, 1, 10)[0] == 1  # swapped bounds
    assert sac.logspace(1, 50, 1) == [1.0, 10.0, 100.0]  # the decade grid overshoots the stop frequency (logspace.ts:8-12)
    with pytest.raises(ValueError, match="frequencies must be > 0"):
        sac.logspace(0, 10, 10)
