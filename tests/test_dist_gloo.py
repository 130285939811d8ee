"""N>1 path on CPU: world_size-2 gloo run of the instance-sharded transient (spicey_amd/dist.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import REPO
from spicey_amd import dist as sdist
from spicey_amd import synth


def test_shard_range_partitions():
    for n, w in ((7, 2), (256, 8), (5, 8), (512, 4), (1, 1)):
        parts = [list(sdist.shard_range(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert list(sdist.shard_range(256, 3, 8)) == list(range(96, 128))  # config 4: 32 instances per GPU


def test_two_rank_gloo_matches_single_process(tmp_path, oracle_backend):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n_total = 7
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_worker.py"), str(tmp_path), str(n_total)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    flat, dt, steps, src = synth.chain_batch("diode_chain", 24, range(1, n_total + 1), tran=".tran 1e-6 2e-5")
    ref = oracle_backend.run(flat, steps, dt, src)
    got = np.zeros_like(ref["out_v"])
    seen = []
    for r in range(2):
        ids = np.load(tmp_path / f"ids_{r}.npy")
        got[ids] = np.load(tmp_path / f"out_v_{r}.npy")
        seen += list(ids)
    assert sorted(seen) == list(range(n_total))
    err = np.abs(got - ref["out_v"]) / (1e-9 * np.abs(ref["out_v"]) + 1e-12)
    assert err.max() <= 1.0
    summ = np.load(tmp_path / "summary.npy")
    assert summ[0] == n_total * (steps + 1) and summ[1] == 2.0 and summ[2] == 2.0
    assert abs(summ[3] + summ[4] - got[:, -1, :].sum()) <= 1e-9 * abs(got[:, -1, :].sum())
