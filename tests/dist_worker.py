"""Worker of tests/test_dist_gloo.py: one rank of the instance-sharded path on CPU (gloo).
Compute goes through the CPU emulator of the device program (test infrastructure); on the GPU box the
same sharding code (spicey_amd/dist.py, bench.py) drives libspicey_hip.so over RCCL."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from emul.pyemul import EmulBackend  # noqa: E402
from spicey_amd import abi, synth  # noqa: E402
from spicey_amd import dist as sdist  # noqa: E402
from spicey_amd.netlist import parseNetlist  # noqa: E402


def main():
    out_dir, n_total = sys.argv[1], int(sys.argv[2])
    sdist.init("gloo")
    r, w = sdist.rank(), sdist.world()
    tran = ".tran 1e-6 2e-5"
    src_np = None
    if r == 0:
        ckt = parseNetlist(synth.diode_chain(24, seed=1, tran=tran))
        dt, steps = abi.computeEffectiveTimeStep(1e-6, 2e-5)
        src_np = abi.source_table(ckt, dt, steps)
    src = sdist.broadcast_f64(src_np).numpy()
    mine = sdist.shard_range(n_total)
    # rank 0 holds the parsed batch (every instance); the others get the topology by broadcast and their block of the
    # per-instance values by scatter (spicey_amd/dist.py) — and must end up with what they would have built
    full = None
    if r == 0:
        full, _, _, _ = synth.chain_batch("diode_chain", 24, range(1, n_total + 1), tran=tran)
        full.C_vprev[:] = np.arange(n_total)[:, None] * 1e-3  # (state travels too)
    topo = sdist.broadcast_topology(full)
    flat = sdist.scatter_params_from_root(topo, full, n_total)
    own, dt, steps, _ = synth.chain_batch("diode_chain", 24, [i + 1 for i in mine], tran=tran)
    assert flat.n_inst == len(mine) == own.n_inst and flat.n_nodes == own.n_nodes
    for k in abi.FlatCircuit.TOPO:
        assert np.array_equal(getattr(flat, k), getattr(own, k)), k
    for k in abi.FlatCircuit.VALS:
        if k != "C_vprev":
            assert np.array_equal(getattr(flat, k), getattr(own, k)), k
    assert np.array_equal(flat.C_vprev, np.repeat(np.array(list(mine))[:, None] * 1e-3, flat.nC, axis=1))
    flat.C_vprev[:] = 0.0
    ids = sdist.assert_distinct_devices(100 + r)
    assert ids == [100 + i for i in range(w)]
    try:
        sdist.assert_distinct_devices(7)
        raise SystemExit("assert_distinct_devices accepted two ranks on one device")
    except RuntimeError:
        pass
    be = EmulBackend(2, 64)
    res = be.run(flat, steps, dt, src)
    assert res["status"] == 0
    np.save(os.path.join(out_dir, f"out_v_{r}.npy"), res["out_v"])
    np.save(os.path.join(out_dir, f"ids_{r}.npy"), np.array(list(mine)))
    # probe-filtered result gather to rank 0 (bench.py does the same over RCCL): columns 0, 5, 23 of every instance
    cols = torch.tensor([0, 5, 23])
    g = sdist.gather_rows_to_root(torch.from_numpy(res["out_v"]).index_select(2, cols), n_total)
    assert (g is None) == (r != 0)
    if r == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), g.numpy())
    chk = sdist.gather_to_all(torch.tensor([res["out_v"][:, -1, :].sum()], dtype=torch.float64))
    total = sdist.sum_over_ranks(float(be.solves))
    tmax = sdist.max_over_ranks(float(r + 1))
    sdist.barrier()
    if r == 0:
        np.save(os.path.join(out_dir, "summary.npy"), np.array([total, tmax, w] + [float(c.item()) for c in chk]))


if __name__ == "__main__":
    main()
