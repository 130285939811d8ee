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
    # the launcher bench.py uses for `--gpus N` without torchrun (spicey_amd/launch.py), here with 2 gloo ranks
    from spicey_amd.launch import spawn_local_ranks
    n_total = 7
    code, _ = spawn_local_ranks([sys.executable, os.path.join(REPO, "tests", "dist_worker.py"), str(tmp_path), str(n_total)], 2, timeout=300)
    assert code == 0
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
    # the root's gathered probe columns are the instance-ordered concatenation of both shards
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), got[:, :, [0, 5, 23]])
    summ = np.load(tmp_path / "summary.npy")
    assert summ[0] == n_total * (steps + 1) and summ[1] == 2.0 and summ[2] == 2.0
    assert abs(summ[3] + summ[4] - got[:, -1, :].sum()) <= 1e-9 * abs(got[:, -1, :].sum())


def test_launcher_reports_a_failing_rank_and_stops_the_others():
    from spicey_amd.launch import spawn_local_ranks
    t0 = __import__("time").monotonic()
    code, out = spawn_local_ranks([sys.executable, "-c", "import os, sys, time; r = int(os.environ['RANK']); print('rank', r, os.environ['WORLD_SIZE'], "
                                   "os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'], flush=True); sys.exit(5) if r == 2 else time.sleep(60)"], 3)
    assert code == 5 and out == "rank 0 3 0 127.0.0.1\n" and __import__("time").monotonic() - t0 < 30


def test_bench_gpus_flag_is_live():
    """`python3 bench.py --gpus 2` starts TWO ranks (here both stop at "needs a GPU": the product path has no CPU
    fallback) and exits non-zero; a WORLD_SIZE that contradicts --gpus is refused before anything runs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs a GPU") >= 1 and "[spawn_local_ranks] rank" in p.stderr
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
