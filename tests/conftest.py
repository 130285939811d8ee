import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def golden_netlist(g):
    from spicey_amd import synth
    if "netlist_file" in g:
        with open(os.path.join(GOLD, g["netlist_file"])) as f:
            return f.read()
    gen, kw = g["generator"]
    return getattr(synth, gen)(**kw)


def fnum(v):
    """driver.mjs encodes non-finite doubles as strings ("Infinity", "NaN")."""
    return float(v) if isinstance(v, str) else v


def farr(seq):
    return np.array([fnum(v) for v in seq], dtype=np.float64)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return ((a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b)))


SMALL_GOLDENS = ["two_probes", "transient01", "case_insensitive", "switch_vt_vh", "vswitch_pwl", "diode_switch",
                 "boost_probe", "bridge_rectifier", "bridge_bleed", "star_hub", "lc_tank", "relay_osc", "half_bridge", "units_title", "float_cap",
                 "steps_round", "ladder20", "dchain20", "mesh6", "mesh9x5",
                 # floating voltage sources (static pivots must stay +-1) and near-singular pivots the reference still solves
                 "fv_bridge", "fv_cap", "fv_hang", "fv_diode", "fv_chain", "near_sing_a", "near_sing_c", "near_sing_e"]
# netlists on which the reference throws Error("Singular matrix (real)") (solveReal.ts:28): structurally singular ones and
# pivots below EPS = 1e-15 that BOTH the partial-pivot order and this build's static order run into
# the reference's own quirk (solveReal.ts:45: row updates with |multiplier| < 1e-15 are skipped): the oracle reproduces both
# bit for bit; the sparse static-order path reproduces the second one and, by construction, not the first (test_oracle.py)
QUIRK_GOLDENS = ["skip_quirk", "skip_quirk_ref"]
SINGULAR_GOLDENS = ["err_singular", "err_vloop", "near_sing_b", "near_sing_d", "near_sing_f"]
LARGE_GOLDENS = ["rc1000_200", "dchain1000_200", "mesh20_30"]


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle.pyoracle import OracleBackend
    return OracleBackend()
