#!/usr/bin/env python3
"""bench.py — timestep solves/s of the transient hot path on N MI355X of one node.

Workload (BASELINE.json configs[2], the configuration the metric's target is quoted on): the
1000-node nonlinear diode chain, .tran 1e-6 1e-2 = 10 000 timesteps (10 001 points), as a batch of
`--batch` parameter-swept instances per GPU (seeds as BASELINE config 4 sweeps the RC ladder: same
topology, per-instance r_k, c_k).  One bench "step" = one spicey_run_batch of the whole batch: every
instance stamps, factors and solves at every timestep (no factor reuse: the circuit is nonlinear), and
records every node voltage and every element current like the reference's simulateTRAN does
(simulateTRAN.ts:164-219).  Inputs are resident in HBM before the timed region; results land in HBM.

value = total solves (= sum of iterations over steps over instances over ranks) / wall time, where a
solve is one pass of simulateTRAN.ts:152-160.  Multi-GPU: instances are sharded across ranks with no
data-path collective (weak scaling: --batch instances PER GPU); RCCL is used to broadcast the shared
topology / source table from rank 0 and to gather per-rank checksums.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from spicey_amd import abi, synth  # noqa: E402
from spicey_amd import dist as sdist  # noqa: E402
from spicey_amd.lib import Handle  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(solves_per_launch, workload_key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes).  Counters
    cannot be read from inside the timed run, so the value is only reported when the committed measurement
    was taken on the same workload (same solves per launch); otherwise null."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
            if int(t["solves_per_launch"]) == int(solves_per_launch) and t.get("workload_key") == workload_key:
                return float(t["traffic_bytes_per_launch"]), os.path.basename(f)
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def cpu_baseline(workload, n, seconds_target=12.0):
    """Oracle (bit-exact restatement of the reference's dense-GE algorithm), 1 thread, bounded sample."""
    from oracle.pyoracle import OracleBackend
    ob = OracleBackend()
    flat, dt, _, _ = synth.chain_batch(workload, n, [1], tran=".tran 1e-6 1e-2")
    from spicey_amd.netlist import parseNetlist
    ckt = parseNetlist(getattr(synth, workload)(n, seed=1))
    # calibrate on 50 steps, then run a sample sized for ~seconds_target
    src = abi.source_table(ckt, dt, 50)
    t0 = time.perf_counter()
    ob.run(flat, 50, dt, src, want_currents=True)
    per = (time.perf_counter() - t0) / 51
    steps = int(max(100, min(10000, seconds_target / per)))
    src = abi.source_table(ckt, dt, steps)
    t0 = time.perf_counter()
    r = ob.run(flat, steps, dt, src, want_currents=True)
    el = time.perf_counter() - t0
    solves = int(r["iters"].sum())
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": solves / el, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"1 instance x {steps + 1} timesteps of the same {n}-node {workload} netlist "
                      f"({el:.1f} s, oracle/spicey_ref.c dense GE in the reference's operation order; host: {cpu}, "
                      f"{os.cpu_count()} logical cores, 1 used)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="diode_chain", choices=["diode_chain", "rc_ladder"])
    ap.add_argument("--nodes", type=int, default=1000)
    ap.add_argument("--timesteps", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=512, help="instances per GPU")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--inst-per-wg", type=int, default=0)
    ap.add_argument("--geometry", type=int, default=0, help="0 auto, 1 latency (one workgroup per CU), 2 throughput (two per CU)")
    ap.add_argument("--no-currents", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-instance", action="store_true", help="also time ONE instance (config 2/3 as written)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    sdist.init("nccl", dev)  # RCCL over xGMI
    rank = sdist.rank()
    n_gpus = world

    B, n, tsteps = args.batch, args.nodes, args.timesteps
    dt = 1e-6
    tran = f".tran 1e-6 {tsteps * 1e-6!r}"
    # ---- rank 0 evaluates the shared source table (waveform closures are host-side) and broadcasts it
    from spicey_amd.netlist import parseNetlist
    src_np = None
    if rank == 0:
        ckt = parseNetlist(getattr(synth, args.workload)(n, seed=1, tran=tran))
        src_np = abi.source_table(ckt, dt, tsteps)
    src = sdist.broadcast_f64(src_np, device=dev)  # 8*(steps+1)*nV bytes
    # ---- instance shard of this rank: weak scaling, B instances per GPU, seeds = global instance id + 1
    mine = sdist.shard_range(B * world)
    flat, _, _, _ = synth.chain_batch(args.workload, n, [i + 1 for i in mine], tran=tran)

    def alloc_and_make(batch_flat):
        h = Handle(batch_flat, device=local_rank, threads=args.threads, inst_per_wg=args.inst_per_wg, geometry=args.geometry)
        info = h.info()
        ov = torch.empty((batch_flat.n_inst, tsteps + 1, info["n_out"]), dtype=torch.float64, device=dev)
        oi = None if args.no_currents else torch.empty((batch_flat.n_inst, tsteps + 1, info["n_cur"]), dtype=torch.float64, device=dev)
        return h, info, ov, oi

    h, info, out_v, out_i = alloc_and_make(flat)

    def one_step():
        h.run_device(tsteps, dt, src.data_ptr(), out_v.data_ptr(), out_i.data_ptr() if out_i is not None else 0)
        rc = h.sync()
        if rc != 0:
            raise RuntimeError(f"spicey run failed: {h.error()}")

    for _ in range(args.warmup):
        one_step()
    sdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    solves_rank = 0
    for _ in range(args.steps):
        one_step()
        kernel_ms.append(h.kernel_ms())  # HIP events around the launch, on the launch stream
        solves_rank += h.solves()
    torch.cuda.synchronize()
    sdist.barrier()
    el = sdist.max_over_ranks(time.perf_counter() - t0, dev)
    total_solves = sdist.sum_over_ranks(float(solves_rank), dev)
    chks = sdist.gather_to_all(out_v[:, -1, :].sum().reshape(1))

    if rank == 0:
        finite = all(bool(torch.isfinite(c).item()) for c in chks)
        k_ms = float(np.mean(kernel_ms))
        solves_per_launch = solves_rank / args.steps
        # SURVEY.md §8(d) formula.  The library evaluates it with ITS nnz(L+U) (nested dissection trades fill for
        # parallel levels: 4967 on the chain); the roofline figure uses the fill-free count of the natural order
        # (nnzA + 2, the BASELINE.md table: 216 048 / 240 024 B per solve) so that extra fill never counts as progress.
        algo_own = info["algorithmic_bytes_solve"]
        algo = algo_own - 20 * max(0, info["nnz_lu"] - (info["nnz_a"] + 2))
        achieved = algo * solves_per_launch / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(solves_per_launch, f"{args.workload}:{n}:{tsteps}:{B}:{int(not args.no_currents)}")
        rec = {
            "metric": "Newton-LU timestep solves/sec, 1000-node netlist",
            "value": total_solves / el,
            "unit": "solves/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}({n}) BASELINE configs[{2 if args.workload == 'diode_chain' else 1}]: "
                            f"{tsteps} timesteps, {B} parameter-swept instances per GPU, all node voltages"
                            + ("" if args.no_currents else " + all element currents") + " recorded",
                "nodes": n, "unknowns": info["n_var"], "timesteps": tsteps, "instances_per_gpu": B,
                "instances_total": B * n_gpus, "parallelism": f"instance-sharded x{n_gpus}, no data-path collective",
                "inst_per_workgroup": info["inst_per_wg"], "threads": info["threads"], "lds_bytes": info["lds_bytes"],
                "nnz_a": info["nnz_a"], "nnz_lu": info["nnz_lu"], "levels": info["n_levels"],
                "factor_reuse": info.get("factor_reuse", 0),  # 1 = linear circuit: "solve-only" rate, factors of step 0 reused
                "interpreter": info["interpreter"], "geometry": info["geometry"], "tail_levels": info["tail_levels"], "resident_tasks": info["resident_tasks"], "streamed_tasks": info["streamed_tasks"],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                "algorithmic_bytes_per_solve": algo, "algorithmic_bytes_own_ordering": algo_own, "solves_per_launch": solves_per_launch, "kernel_ms": k_ms,
                "kernel": "spicey_tran_kernel_v2" if info.get("interpreter") == 2 else "spicey_tran_kernel",
            },
            "results_finite": finite,
        }
        if args.single_instance:
            f1, _, _, _ = synth.chain_batch(args.workload, n, [1], tran=tran)
            h1, i1, ov1, oi1 = alloc_and_make(f1)
            best = None
            for _ in range(3):
                h1.run_device(tsteps, dt, src.data_ptr(), ov1.data_ptr(), oi1.data_ptr() if oi1 is not None else 0)
                assert h1.sync() == 0
                best = h1.kernel_ms() if best is None else min(best, h1.kernel_ms())
            rec["single_instance"] = {"solves_per_s": h1.solves() / (best * 1e-3), "us_per_timestep": best * 1e3 / (tsteps + 1),
                                      "threads": i1["threads"]}
            h1.close()
        if n_gpus == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.workload, n)
        print(json.dumps(rec), flush=True)
    h.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
