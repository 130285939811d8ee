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
    flat, dt, steps, _ = synth.chain_batch("diode_chain", 24, [i + 1 for i in mine], tran=tran)
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
