#!/usr/bin/env python3
"""bench.py — timestep solves/s of the transient hot path on N MI355X of one node.

Workload (BASELINE.json configs[2], the configuration the metric's target is quoted on): the
1000-node nonlinear diode chain, .tran 1e-6 1e-2 = 10 000 timesteps (10 001 points), as a batch of
`--batch` parameter-swept instances per GPU (seeds as BASELINE config 4 sweeps the RC ladder: same
topology, per-instance r_k, c_k).  One bench "step" = one spicey_run of the whole batch: every
instance stamps, factors and solves at every timestep (no factor reuse: the circuit is nonlinear), and
records every node voltage and every element current like the reference's simulateTRAN does
(simulateTRAN.ts:164-219).  Inputs are resident in HBM before the timed region; results land in HBM.

value = total solves (= sum of iterations over steps over instances over ranks) / wall time, where a
solve is one pass of simulateTRAN.ts:152-160.

Multi-GPU: instances are block-partitioned over ranks, one process per GPU, no
data-path collective (weak scaling: --batch instances PER GPU).  Rank 0 holds the parsed batch: RCCL broadcasts
the shared topology and the source table, scatters every rank's block of per-instance parameters before the
run, and gathers the probe-filtered results (simulateTRAN.ts:240-249 is the filter) to rank 0 after it; the
gather is timed separately (`gather_ms`), outside the timed region.  With N > 1 the ranks must sit on N
distinct devices (PCI bus ids are compared) or the run exits non-zero.
`python3 bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (fresh child
processes, before anything touches the GPU); under `python -m torch.distributed.run` it joins the job and
checks that the world size is N.

After the timed region rank 0 checks the TIMED buffers against the oracle: three instances of its shard
(first, middle, last seed), first 200 timesteps, every node voltage and element current, at
|x - ref| <= 1e-9 |ref| + 1e-12 (`parity_max_over_tol`, `parity_ok`), plus — on N > 1 — the gathered probe
columns of the globally last instance (computed on the last rank).

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PARITY_STEPS = 200
RTOL, ATOL = 1e-9, 1e-12


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
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
    ap.add_argument("--probes", type=int, default=8, help="node-voltage columns gathered to rank 0 after the run")
    ap.add_argument("--no-currents", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-instance", action="store_true", help="skip the ONE-instance timing (configs 2/3 exactly as written)")
    ap.add_argument("--single-instance", action="store_true", help="(default now; kept for old command lines)")
    return ap.parse_args(argv)


def committed_pmc(solves_per_launch, workload_key):
    """HBM bytes per launch and VALU issue fraction from the committed rocprofv3 PMC passes (profiles/rNN_pmc_*.json:
    separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be read from
    inside the timed run, so the values are only reported when the committed measurement was taken on the same
    workload (same key, same solves per launch); otherwise null."""
    import glob
    traffic = src = valu = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
            if int(t["solves_per_launch"]) == int(solves_per_launch) and t.get("workload_key") == workload_key:
                traffic, src = float(t["traffic_bytes_per_launch"]), os.path.basename(f)
                break
        except (OSError, KeyError, ValueError):
            pass
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_sq_lds.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
            if t.get("workload_key", "diode_chain:1000:10000:512:1") == workload_key:
                valu = float(t["derived"]["valu_issue_utilisation_at_4_cycles_per_instruction"])
                break
        except (OSError, KeyError, ValueError):
            pass
    return traffic, src, valu


def tol_ratio(got, ref):
    import numpy as np
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    with np.errstate(invalid="ignore"):
        r = np.abs(got - ref) / (RTOL * np.abs(ref) + ATOL)
    same_nonfinite = (~np.isfinite(ref)) & ((got == ref) | (np.isnan(got) & np.isnan(ref)))
    r = np.where(same_nonfinite, 0.0, r)  # the unclamped diode current may be Infinity in the reference too
    return float(np.nan_to_num(r, nan=np.inf).max()) if r.size else 0.0


def oracle_prefix(workload, n, seeds, tran, src_np, steps):
    """Oracle (checker) on the first `steps` timesteps of the given seeds of the bench workload."""
    from oracle.pyoracle import OracleBackend
    from spicey_amd import synth
    flat, dt, _, _ = synth.chain_batch(workload, n, seeds, tran=tran)
    return OracleBackend().run(flat, steps, dt, src_np[: steps + 1], want_currents=True)


def cpu_baseline(workload, n, seconds_target=12.0):
    """Oracle (bit-exact restatement of the reference's dense-GE algorithm), 1 thread, bounded sample."""
    from oracle.pyoracle import OracleBackend
    from spicey_amd import abi, synth
    from spicey_amd.netlist import parseNetlist
    ob = OracleBackend()
    flat, dt, _, _ = synth.chain_batch(workload, n, [1], tran=".tran 1e-6 1e-2")
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
                      f"{os.cpu_count()} logical cores, 1 used)",
            "reference_js": {"value": 28.7 if workload == "diode_chain" else 22.4, "unit": "solves/s", "cores": 1,
                             "where": "the reference's own TypeScript path (type-erased, Node 12) timed in the BUILD container "
                                      "(Xeon 2.1 GHz) while generating tests/golden/*_full.json — a build-container figure: the "
                                      "reference's source does not travel to the GPU box, so it cannot be timed there"}}


def run_rank(args):
    import numpy as np
    import torch
    from spicey_amd import abi, synth
    from spicey_amd import dist as sdist
    from spicey_amd.lib import Handle
    from spicey_amd.netlist import parseNetlist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or without torchrun: bench.py starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank with LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    sdist.init("nccl", dev)  # RCCL over xGMI
    rank = sdist.rank()
    if sdist.world() != args.gpus:
        raise SystemExit(f"communicator has {sdist.world()} ranks, --gpus says {args.gpus}")
    n_gpus = world
    if world > 1:
        import torch.distributed as tdist
        if tdist.get_world_size() != args.gpus:
            raise SystemExit(f"torch.distributed world size {tdist.get_world_size()} != --gpus {args.gpus}")
    props = torch.cuda.get_device_properties(local_rank)
    pci = (int(getattr(props, "pci_domain_id", 0)) << 16) | (int(getattr(props, "pci_bus_id", local_rank)) << 8) | int(getattr(props, "pci_device_id", 0))
    log(f"[rank {rank}/{world}] device cuda:{local_rank} = {props.name}, pci {getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}, "
        f"{props.total_memory / 2**30:.0f} GiB, {props.multi_processor_count} CUs")
    try:
        pci_ids = sdist.assert_distinct_devices(pci, dev)  # N ranks on fewer than N GPUs: exit non-zero, no JSON line
    except RuntimeError as e:
        raise SystemExit(f"bench.py --gpus {args.gpus}: {e}")

    B, n, tsteps = args.batch, args.nodes, args.timesteps
    dt = 1e-6
    tran = f".tran 1e-6 {tsteps * 1e-6!r}"
    # ---- rank 0 evaluates the shared source table (waveform closures are host-side) and broadcasts it
    src_np = None
    if rank == 0:
        ckt = parseNetlist(getattr(synth, args.workload)(n, seed=1, tran=tran))
        src_np = abi.source_table(ckt, dt, tsteps)
    src = sdist.broadcast_f64(src_np, device=dev)  # 8*(steps+1)*nV bytes
    # ---- instance shard of this rank: weak scaling, B instances per GPU, seeds = global instance id + 1
    n_total = B * world
    mine = sdist.shard_range(n_total)
    # rank 0 holds the parsed batch (all instances of all ranks); topology by broadcast, each rank's block of the
    # per-instance parameters by scatter (spicey_amd/dist.py; §8(e): ~32 KB of topology, 32 KB per instance)
    full = None
    if rank == 0:
        full, _, _, _ = synth.chain_batch(args.workload, n, range(1, n_total + 1), tran=tran)
    t_h = time.perf_counter()
    topo = sdist.broadcast_topology(full, dev)
    flat = sdist.scatter_params_from_root(topo, full, n_total, dev)
    handout_ms = (time.perf_counter() - t_h) * 1e3
    del full

    def alloc_and_make(batch_flat):
        h = Handle(batch_flat, device=local_rank, threads=args.threads, inst_per_wg=args.inst_per_wg, geometry=args.geometry)
        info = h.info()
        ov = torch.empty((batch_flat.n_inst, tsteps + 1, info["n_out"]), dtype=torch.float64, device=dev)
        oi = None if args.no_currents else torch.empty((batch_flat.n_inst, tsteps + 1, info["n_cur"]), dtype=torch.float64, device=dev)
        return h, info, ov, oi

    h, info, out_v, out_i = alloc_and_make(flat)

    def one_step():
        # every bench step is the SAME transient (from the state the netlist starts in), not a continuation of the
        # previous step's end state: device-to-device restore of 16 B per capacitor/diode, enqueued before the launch
        h.reset_state()
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

    # ---- result gather (outside the timed region, timed on its own): probe filter, then every rank's block to rank 0
    np_cols = max(1, min(args.probes, info["n_out"]))
    cols = sorted({int(round(i * (info["n_out"] - 1) / max(1, np_cols - 1))) for i in range(np_cols)})
    cols_t = torch.tensor(cols, device=dev)
    sdist.barrier()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    probe_local = out_v.index_select(2, cols_t)  # [B, steps+1, n_probe]: the .PRINT filter of simulateTRAN.ts:240-249
    gathered = sdist.gather_rows_to_root(probe_local, n_total)
    torch.cuda.synchronize()
    sdist.barrier()
    gather_ms = sdist.max_over_ranks((time.perf_counter() - tg) * 1e3, dev)

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        solves_per_launch = solves_rank / args.steps
        # SURVEY.md §8(d) formulas.  The library evaluates the full formula with ITS nnz(L+U) (nested dissection trades
        # fill for parallel levels: 4967 on the chain); the roofline figure uses the fill-free count of the natural order
        # (nnzA + 2, the BASELINE.md table: 216 048 / 240 024 B per solve) so that extra fill never counts as progress.
        # A linear circuit reuses the factors of step 0 (info.factor_reuse): the timed kernel then only solves, and is
        # priced with B_solve-only = 12 nnzLU + 24 Nvar + 16 E_C + 8 (nNodes + E_total)  (100 024 B on the ladder).
        algo_own = info["algorithmic_bytes_solve"]
        nnz_lu_nofill = info["nnz_a"] + 2
        algo_full = algo_own - 20 * max(0, info["nnz_lu"] - nnz_lu_nofill)
        e_total = flat.nR + flat.nC + flat.nL + flat.nV + flat.nS + flat.nD
        algo_solve_only = 12 * nnz_lu_nofill + 24 * info["n_var"] + 16 * flat.nC + 8 * (flat.n_nodes + e_total)
        reuse = bool(info.get("factor_reuse", 0))
        algo = algo_solve_only if reuse else algo_full
        achieved = algo * solves_per_launch / (k_ms * 1e-3) / 1e9
        wkey = f"{args.workload}:{n}:{tsteps}:{B}:{int(not args.no_currents)}"
        traffic, traffic_src, valu = committed_pmc(solves_per_launch, wkey)
        result_bytes = 8.0 * (info["n_out"] + (0 if args.no_currents else info["n_cur"])) * solves_per_launch
        # LEAD roofline figure = what the kernel really moves to and from HBM per launch / its measured duration / 8 TB/s:
        # counter traffic of the committed PMC passes of this same command where they exist (counters cannot be read from
        # inside a timed run), else the compulsory part that is known exactly — the result stream this launch wrote.  The
        # SURVEY 8(d) formula figure (bytes a STREAMING implementation would move) is kept as `frac_formula`: this kernel keeps
        # matrix and factors on chip, so that figure can exceed 1 and is no utilisation.
        moved = traffic if traffic else result_bytes
        moved_gbs = moved / (k_ms * 1e-3) / 1e9
        if moved_gbs > HBM_PEAK_GBS:
            raise SystemExit(f"bench.py: {moved_gbs:.0f} GB/s of HBM traffic is above the {HBM_PEAK_GBS:.0f} GB/s peak: the timing or the byte count is wrong")
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
                            + ("" if args.no_currents else " + all element currents") + " recorded"
                            + ("; linear circuit: factors of step 0 reused = SOLVE-ONLY rate (SURVEY.md §8(d))" if reuse else ""),
                "nodes": n, "unknowns": info["n_var"], "timesteps": tsteps, "instances_per_gpu": B,
                "instances_total": n_total, "parallelism": f"instance-sharded x{n_gpus}, no data-path collective",
                "inst_per_workgroup": info["inst_per_wg"], "threads": info["threads"], "lds_bytes": info["lds_bytes"],
                "nnz_a": info["nnz_a"], "nnz_lu": info["nnz_lu"], "levels": info["n_levels"],
                "factor_reuse": int(reuse),
                "interpreter": info["interpreter"], "geometry": info["geometry"], "tail_levels": info["tail_levels"],
                "resident_tasks": info["resident_tasks"], "streamed_tasks": info["streamed_tasks"],
            },
            "roofline": {
                "bound": "hbm", "achieved": moved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": moved_gbs / HBM_PEAK_GBS,
                "frac_means": ("HBM bytes per launch from the PMC counters (2*FETCH_SIZE + WRITE_SIZE of the committed passes of this same "
                               "command)" if traffic else "HBM bytes of the result stream this launch wrote (no committed PMC pass for this workload)")
                              + " / measured kernel time / 8 TB/s.  The kernel is NOT bandwidth-bound: matrix, factors and state stay on chip "
                              "(LDS + registers), HBM sees the result stream and a few KB of task records per solve; the limiter is VALU "
                              "issue + dependent LDS chains (`valu_issue_frac`).  `frac_formula` = SURVEY.md 8(d) ALGORITHMIC bytes (what a "
                              "streaming implementation would move) per second / 8 TB/s, kept for comparison across rounds; it can exceed 1",
                "achieved_formula": achieved, "frac_formula": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                "hbm_utilisation": (traffic / (k_ms * 1e-3) / (HBM_PEAK_GBS * 1e9)) if traffic else None,
                "hbm_utilisation_results_only": result_bytes / (k_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                "valu_issue_frac": valu,
                "algorithmic_bytes_per_solve": algo, "algorithmic_bytes_formula": "solve-only" if reuse else "full (stamp + factor + solve)",
                "algorithmic_bytes_own_ordering": algo_own, "solves_per_launch": solves_per_launch, "kernel_ms": k_ms,
                "kernel": "spicey_tran_kernel_v2" if info.get("interpreter") == 2 else "spicey_tran_kernel",
            },
            "handout": {"ms": handout_ms, "what": "topology broadcast + per-instance parameter scatter from rank 0 (RCCL; before the timed region)",
                        "devices_pci": pci_ids},
            "gather": {"ms": gather_ms, "bytes_to_root": int(n_total * (tsteps + 1) * len(cols) * 8), "probe_columns": cols,
                       "what": "probe-filtered node voltages of every instance of every rank gathered to rank 0 (RCCL gather); "
                               "not part of the timed region"},
        }
        # ---- parity of the TIMED buffers against the oracle (checker only; after the timed region)
        ps = min(PARITY_STEPS, tsteps)
        idx = sorted({0, len(mine) // 2, len(mine) - 1})
        seeds = [mine[i] + 1 for i in idx]
        ref = oracle_prefix(args.workload, n, seeds, tran, src_np, ps)
        worst = tol_ratio(out_v[idx, : ps + 1].cpu().numpy(), ref["out_v"])
        if out_i is not None:
            worst = max(worst, tol_ratio(out_i[idx, : ps + 1].cpu().numpy(), ref["out_i"]))
        checked = [f"inst {mine[i]} (seed {mine[i] + 1})" for i in idx]
        if world > 1:  # the gathered block of the LAST rank: its compute and the gather itself
            ref_last = oracle_prefix(args.workload, n, [n_total], tran, src_np, ps)
            worst = max(worst, tol_ratio(gathered[n_total - 1, : ps + 1].cpu().numpy(), ref_last["out_v"][0][:, cols]))
            ref_first = ref["out_v"][0][:, cols]
            worst = max(worst, tol_ratio(gathered[0, : ps + 1].cpu().numpy(), ref_first))
            checked.append(f"gathered probes of inst {n_total - 1} (rank {world - 1})")
        finite = bool(torch.isfinite(out_v[:, -1, :]).all().item())
        rec["parity_max_over_tol"] = worst
        rec["parity_ok"] = bool(worst <= 1.0 and finite)
        rec["parity_checked"] = f"timed buffers, first {ps + 1} timesteps, all voltages + currents: " + ", ".join(checked) + \
                                f"; tolerance {RTOL:g}*|ref| + {ATOL:g}; last step finite on every instance: {finite}"
        if not args.no_single_instance:
            f1, _, _, _ = synth.chain_batch(args.workload, n, [1], tran=tran)
            h1, i1, ov1, oi1 = alloc_and_make(f1)
            best = None
            for _ in range(3):
                h1.reset_state()
                h1.run_device(tsteps, dt, src.data_ptr(), ov1.data_ptr(), oi1.data_ptr() if oi1 is not None else 0)
                assert h1.sync() == 0
                best = h1.kernel_ms() if best is None else min(best, h1.kernel_ms())
            rec["single_instance"] = {"solves_per_s": h1.solves() / (best * 1e-3), "us_per_timestep": best * 1e3 / (tsteps + 1),
                                      "threads": i1["threads"], "geometry": i1["geometry"],
                                      "what": f"ONE {args.workload}({n}) netlist, {tsteps} timesteps: BASELINE configs as written"}
            h1.close()
        if not args.no_cpu_baseline:  # rank 0's host cores, at every N
            rec["cpu_baseline"] = cpu_baseline(args.workload, n)
        print(json.dumps(rec), flush=True)
        if not rec["parity_ok"]:
            h.close()
            raise SystemExit("bench.py: the timed buffers do not match the oracle")
    h.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # start the ranks ourselves: nothing in THIS process has touched the GPU (no torch import, no HIP call)
        from spicey_amd.launch import spawn_local_ranks
        code, out = spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.stdout.write(out)
        sys.stdout.flush()
        if code == 0 and not out.strip().startswith("{"):
            code = 1
        raise SystemExit(code)
    run_rank(args)


if __name__ == "__main__":
    main()
