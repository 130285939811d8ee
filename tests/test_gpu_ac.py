"""GPU parity tests of the AC sweep: the HIP path (spicey_ac_* through libspicey_hip.so) against the oracle on the
same inputs and against the reference-generated goldens.  Tolerance: |z - z_ref| <= 1e-9 |z_ref| + 1e-12 on every
complex node voltage and element current (the transient bar of SURVEY.md §8(d) applied to complex magnitudes)."""
import numpy as np
import pytest

from conftest import golden_netlist, load_golden
from spicey_amd import abi, synth
from spicey_amd import ac as sac
from spicey_amd.netlist import parseNetlist
from test_oracle_ac import ac_golden_netlist, cplx

pytestmark = pytest.mark.gpu


def cratio(got, ref, rtol=1e-9, atol=1e-12):
    return np.abs(got - ref) / (rtol * np.abs(ref) + atol)


@pytest.mark.parametrize("name", ["ac_readme", "ac_rlc", "ac_two_src", "ac_fv", "ac_ladder30", "ac_mesh6"])
def test_ac_hip_vs_oracle_and_golden(name, oracle_backend):
    from spicey_amd.lib import HipBackend
    g = load_golden(name)
    ckt = parseNetlist(ac_golden_netlist(g))
    flat = abi.flatten(ckt)
    freqs, vph = np.array(g["freqs"]), cplx(g["vph"])
    ref = oracle_backend.run_ac(flat, freqs, vph)
    first = None
    for kw in (dict(), dict(force_global=True), dict(threads=64)):
        be = HipBackend(**kw)
        got = be.run_ac(flat, freqs, vph)
        assert got["status"] == 0, got["detail"]
        assert (be.info["lds_bytes"] == 0) == bool(kw.get("force_global"))
        assert cratio(got["out_v"], ref["out_v"]).max() <= 1.0 and cratio(got["out_i"], ref["out_i"]).max() <= 1.0
        if first is None:
            first = got
        assert np.array_equal(got["out_v"], first["out_v"]) and np.array_equal(got["out_i"], first["out_i"])  # geometry-independent bits
    names = ckt.nodes.rev
    for i in range(1, ckt.nodes.count()):
        assert cratio(first["out_v"][0, :, i - 1], cplx(g["V"][names[i]])).max() <= 1.0


def test_ac_public_api_default_backend_reference_snapshot():
    """simulate() with the default (HIP) backend reproduces the reference's inline snapshot of
    tests/basics/basics01.test.ts / README.md character for character."""
    from spicey_amd.simulate import simulate
    g = load_golden("ac_readme")
    out = simulate(golden_netlist(g))
    assert out["tran"] is None and sac.formatAcResult(out["ac"]) == g["formatted"]
    assert list(out["ac"]["elementCurrents"]) == g["keysI"]


def test_ac_errors():
    g = load_golden("ac_err_float")
    with pytest.raises(sac.SingularComplexMatrixError):
        sac.simulateAC(parseNetlist(golden_netlist(g)))
    tiny = parseNetlist("* tiny\nV1 1 0 ac 1\nR1 1 0 1k\nC1 1 2 1e-12\nC2 2 0 1e-12\n.ac lin 2 1 2\n.end")
    with pytest.raises(ZeroDivisionError, match="Complex divide by ~0"):
        sac.simulateAC(tiny)
    with pytest.raises(ValueError, match="R R1 must be > 0"):
        sac.simulateAC(parseNetlist(golden_netlist(load_golden("ac_err_r0"))))


def test_ac_baseline_sized_ladder_and_batch(oracle_backend):
    """1001 unknowns (workspace 95 KB of LDS) at the golden's 16 frequencies; then 8 swept instances x 16 frequencies
    in one launch against the oracle."""
    from spicey_amd.lib import HipBackend
    g = load_golden("ac_rc1000")
    ckt = parseNetlist(ac_golden_netlist(g))
    flat = abi.flatten(ckt)
    be = HipBackend()
    got = be.run_ac(flat, np.array(g["freqs"]), cplx(g["vph"]))
    assert got["status"] == 0 and be.info["n_var"] == 1001 and be.info["lds_bytes"] > 90000
    names = ckt.nodes.rev
    col = {names[i]: i - 1 for i in range(1, ckt.nodes.count())}
    for k, v in g["V"].items():
        assert cratio(got["out_v"][0, :, col[k]], cplx(v)).max() <= 1.0, k
    elem = [e.name for e in ckt.R] + [e.name for e in ckt.C] + [e.name for e in ckt.L] + [e.name for e in ckt.V]
    for k, v in g["I"].items():
        assert cratio(got["out_i"][0, :, elem.index(k)], cplx(v)).max() <= 1.0, k
    flat8, _, _, _ = synth.chain_batch("rc_ladder", 300, range(1, 9), tran=".tran 1e-6 3e-5")
    freqs = np.array(sac.logspace(1e3, 1e8, 3))
    ref = oracle_backend.run_ac(flat8, freqs, np.array([1.0 + 0.0j]))
    got = HipBackend().run_ac(flat8, freqs, np.array([1.0 + 0.0j]))
    assert got["status"] == 0 and got["out_v"].shape == (8, len(freqs), 300)
    assert cratio(got["out_v"], ref["out_v"]).max() <= 1.0 and cratio(got["out_i"], ref["out_i"]).max() <= 1.0


def test_ac_large_mesh_global_workspace(oracle_backend):
    """rcd_mesh(34x34): 1157 unknowns but ~25 000 L+U entries -> the complex workspace (16 B per entry) leaves LDS."""
    from spicey_amd.lib import HipBackend
    text = "\n".join(ln + " ac 1" if ln.startswith("V1 ") else ln for ln in synth.rcd_mesh(34, seed=11, tran=".ac dec 2 1e5 1e8").split("\n"))
    ckt = parseNetlist(text)
    flat = abi.flatten(ckt)
    freqs = np.array(sac.buildFrequencyArray(**ckt.analyses["ac"]))
    vph = sac.source_phasors(ckt)
    be = HipBackend()
    got = be.run_ac(flat, freqs, vph)
    assert got["status"] == 0 and be.info["lds_bytes"] == 0
    ref = oracle_backend.run_ac(flat, freqs, vph)
    assert cratio(got["out_v"], ref["out_v"]).max() <= 1.0 and cratio(got["out_i"], ref["out_i"]).max() <= 1.0


def test_ac_resident_sweep_on_gpu(oracle_backend):
    """Batches that outnumber the CUs take the resident sweep (persistent workgroup per instance and frequency class, task
    records and frequency-independent stamp parts in registers): same parity bar as the per-solve kernel, to which it must
    agree within rounding; the info block says which one ran."""
    from spicey_amd import synth
    from spicey_amd.lib import AcHandle
    flat, _, _, _ = synth.chain_batch("rc_ladder", 300, range(1, 9), tran=".tran 1e-6 3e-5")  # 8 instances
    freqs = np.array(sac.logspace(1e3, 1e8, 16))[:80]                                          # x 80 frequencies = 640 solves
    vph = np.array([1.0 + 0.25j])
    ref = oracle_backend.run_ac(flat, freqs, vph)
    h = AcHandle(flat)
    got = h.run(freqs, vph)
    assert got["status"] == 0 and h.info()["interpreter"] == 2 and h.info()["resident_tasks"] > 0
    tol = lambda a, b: (np.abs(a - b) / (1e-9 * np.abs(b) + 1e-12)).max()
    assert tol(got["out_v"], ref["out_v"]) <= 1.0 and tol(got["out_i"], ref["out_i"]) <= 1.0
    h2 = AcHandle(flat, no_resident=True)
    one = h2.run(freqs, vph)
    assert one["status"] == 0 and h2.info()["interpreter"] == 1
    assert tol(got["out_v"], one["out_v"]) <= 1.0 and tol(got["out_i"], one["out_i"]) <= 1.0
    # RLC (entries with inductor stamps keep the reference's per-stamp sequence) and an error inside a batch
    g = load_golden("ac_rlc")
    ckt = parseNetlist(golden_netlist(g))
    f3 = abi.flatten(ckt).replicate(6)
    fr = np.array(sac.logspace(10.0, 1e7, 20))[:110]
    r3 = AcHandle(f3)
    g3 = r3.run(fr, sac.source_phasors(ckt))
    o3 = oracle_backend.run_ac(f3, fr, sac.source_phasors(ckt))
    assert g3["status"] == 0 and r3.info()["interpreter"] == 2 and tol(g3["out_v"], o3["out_v"]) <= 1.0 and tol(g3["out_i"], o3["out_i"]) <= 1.0


def test_ac_resonance_dense_fallback_on_gpu(oracle_backend):
    """A series L - C node cancels at its resonance: the static pivot order hits ~0 there, the reference's partial
    pivoting does not.  The HIP path repeats exactly those (instance, frequency) solves with dense partial pivoting
    (`spicey_ac_dense_kernel`) and matches the reference at, next to and away from the resonance — in a batch whose other
    instances resonate elsewhere, through both sweep kernels; with the fallback switched off the sweep fails; what is
    singular for the reference stays singular."""
    import math
    from random_circuits import series_rlc_ladder
    from spicey_amd.lib import AcHandle
    flats = [abi.flatten(parseNetlist(series_rlc_ladder(12, l=1e-3 * (1 + 0.25 * k)))) for k in range(4)]
    flat = abi.stack_instances(flats)
    f0 = 1.0 / (2.0 * math.pi * math.sqrt(1e-3 * 1e-6))
    freqs = np.array([f0 * (1.0 + d) for d in (1e-2, 1e-6, 1e-9, 1e-12, 0.0, -1e-10)] + [f0 / math.sqrt(1.25), 777.0])
    vph = np.ones(flat.nV, np.complex128)
    ref = oracle_backend.run_ac(flat, freqs, vph)
    assert ref["status"] == 0
    for kw in (dict(), dict(no_resident=True), dict(force_global=True)):
        h = AcHandle(flat, **kw)
        got = h.run(freqs, vph)
        n_dense = h.info()["tail_levels"]
        h.close()
        assert got["status"] == 0, got["detail"]
        assert 4 <= n_dense <= 8                       # instance 0 next to f0 (4-5 points), instance 1 at its own resonance
        assert cratio(got["out_v"], ref["out_v"]).max() <= 1.0 and cratio(got["out_i"], ref["out_i"]).max() <= 1.0
    h = AcHandle(flat, no_dense=True)
    assert h.run(freqs, vph)["status"] == abi.ERR_COMPLEX_DIV
    h.close()
    text = "* floating tank\nV1 in 0 AC 1\nR1 in 0 1k\nL1 a 0 1\nC1 a 0 1\n.ac lin 1 1 1\n.end\n"
    flat = abi.flatten(parseNetlist(text))
    freqs = np.array([1.0 / (2.0 * math.pi)])
    vph = np.ones(flat.nV, np.complex128)
    ref = oracle_backend.run_ac(flat, freqs, vph)
    h = AcHandle(flat)
    got = h.run(freqs, vph)
    h.close()
    assert ref["status"] != 0 and got["status"] == ref["status"]
