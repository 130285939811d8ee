#!/usr/bin/env python3
"""gpurun_out/prof_round/ (tools/profile_round.sh, run on the GPU box) -> profiles/rNN_* (tracked)."""
import csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof_round")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
DST = os.path.join(REPO, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True), key=os.path.getmtime)
    assert f, pattern
    return f[-1]  # gpurun merges into gpurun_out/ without deleting earlier runs' files: take the newest


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


def counters(which, kernel_substr):
    """{counter: value summed over the XCDs' rows of ONE dispatch (averaged over dispatches)}, kernel name, rows kept"""
    src = one(which + "/**/*counter_collection.csv")
    rows = [r for r in csv.DictReader(open(src)) if kernel_substr in r["Kernel_Name"]]
    assert rows, (which, "no rows for", kernel_substr)
    ndisp = len({r["Dispatch_Id"] for r in rows})
    tot = {}
    for r in rows:
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / ndisp
    with open(src) as f:
        lines = f.read().splitlines()
    keep = [lines[0]] + [l for l in lines[1:] if kernel_substr in l]
    return tot, rows[0]["Kernel_Name"], keep, rows[0]


for name in ("bench_line", "bench_line_rc_ladder", "bench_line_under_rocprof"):
    rec = last_json_line(os.path.join(SRC, name + ".json"))
    json.dump(rec, open(os.path.join(DST, f"{rnd}_{name}.json"), "w"), indent=1)
bench = last_json_line(os.path.join(SRC, "bench_line.json"))

shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{rnd}_bench_kernel_stats.csv"))
shutil.copy(one("ac_stats/**/*kernel_stats.csv"), os.path.join(DST, f"{rnd}_ac_kernel_stats.csv"))
aclines = [json.loads(l) for l in open(os.path.join(SRC, "ac_probe.json")) if l.startswith("{")]
# ---- HBM traffic of the AC sweep kernel (resident sweep, 64 instances x 201 frequencies; 3 timed repetitions per run)
try:
    fa, kac, keep_fa, _ = counters("ac_fetch", "spicey_ac_kernel")
    wa, _, keep_wa, _ = counters("ac_write", "spicey_ac_kernel")
    open(os.path.join(DST, f"{rnd}_ac_pmc_counter_collection.csv"), "w").write("\n".join(keep_fa + keep_wa[1:]) + "\n")
    for rec_ac in aclines:
        if rec_ac["inst"] != 64:
            continue
        tr = 2.0 * fa["FETCH_SIZE"] * 1024.0 + wa["WRITE_SIZE"] * 1024.0  # per dispatch
        r = rec_ac["roofline"]
        r.update(frac_formula=r["frac"], achieved_formula=r["achieved"], traffic=tr, traffic_unit="bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes)",
                 traffic_bytes_per_solve=tr / rec_ac["solves"], achieved=tr / (rec_ac["kernel_ms"] * 1e-3) / 1e9, kernel=kac)
        r["frac"] = r["achieved"] / r["peak"]
        r["compulsory_bytes_per_solve"] = 16 * (1000 + 2000 + 1)  # one complex value per recorded node and element
except AssertionError as e:
    print("no AC PMC passes in this profile round:", e)
json.dump(aclines, open(os.path.join(DST, f"{rnd}_ac_probe.json"), "w"), indent=1)

# ---- HBM traffic of the bench kernel: separate FETCH_SIZE / WRITE_SIZE passes
tot = {}
kernel = None
for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    t, kernel, keep, _ = counters(which, "spicey_tran_kernel")
    tot[counter] = t[counter]
    open(os.path.join(DST, f"{rnd}_{which}_counter_collection.csv"), "w").write("\n".join(keep) + "\n")
one_launch = last_json_line(os.path.join(SRC, "pmc_fetch.json"))
spl = int(one_launch["roofline"]["solves_per_launch"])
cfg = one_launch["config"]
wkey = f"diode_chain:{cfg['nodes']}:{cfg['timesteps']}:{cfg['instances_per_gpu']}:1"
fetch_raw = tot["FETCH_SIZE"] * 1024.0  # KB -> bytes
write = tot["WRITE_SIZE"] * 1024.0
traffic = 2.0 * fetch_raw + write
rec = {
    "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-instance  (separate passes for FETCH_SIZE and WRITE_SIZE; tools/profile_round.sh)",
    "workload": cfg["workload"],
    "workload_key": wkey,
    "solves_per_launch": spl,
    "kernel": kernel,
    "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
    "fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": 2.0 * fetch_raw, "write_bytes": write,
    "traffic_bytes_per_launch": traffic, "traffic_bytes_per_solve": traffic / spl,
    "note": "gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> doubled. "
            "WRITE_SIZE is exact for streaming stores: expected result bytes = instances*(steps+1)*(nodes + currents)*8.",
}
json.dump(rec, open(os.path.join(DST, f"{rnd}_pmc_traffic.json"), "w"), indent=1)

# ---- SQ / LDS counters of the bench kernel (what actually bounds it)
sq, _, _, row0 = counters("pmc_sq", "spicey_tran_kernel")
lds, _, _, _ = counters("pmc_lds", "spicey_tran_kernel")
kms = last_json_line(os.path.join(SRC, "pmc_sq.json"))["roofline"]["kernel_ms"]
wave_cycles = sq["SQ_WAVE_CYCLES"]
derived = {
    "valu_wave_instructions_per_solve": sq["SQ_INSTS_VALU"] / spl,
    "salu_wave_instructions_per_solve": sq["SQ_INSTS_SALU"] / spl,
    "lds_wave_instructions_per_solve": sq["SQ_INSTS_LDS"] / spl,
    "vmem_read_wave_instructions_per_solve": lds["SQ_INSTS_VMEM_RD"] / spl,
    "vmem_write_wave_instructions_per_solve": lds["SQ_INSTS_VMEM_WR"] / spl,
    # VALU issue: 4 cycles per wave64 instruction on a SIMD; SQ_BUSY_CYCLES is per XCD-summed shader-engine cycles -> use kernel time
    "valu_issue_utilisation_at_4_cycles_per_instruction": sq["SQ_INSTS_VALU"] * 4.0 / (256 * 4 * (kms * 1e-3) * 2.4e9),
    "wave_cycles_waiting_fraction": sq["SQ_WAIT_ANY"] / wave_cycles,
    "wave_cycles_issuing_fraction": sq["SQ_ACTIVE_INST_ANY"] / wave_cycles,
    "wave_cycles_valu_fraction": sq["SQ_ACTIVE_INST_VALU"] / wave_cycles,
    "lds_bank_conflict_share_of_lds_cycles": lds["SQ_LDS_BANK_CONFLICT"] / max(lds["SQ_LDS_IDX_ACTIVE"], 1.0),
    "note": "utilisation at the 2.4 GHz peak clock over 256 CUs x 4 SIMDs for the kernel's duration under the profiler; SQ_* cycle counters are quad-cycles",
}
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY (pass 1); "
                      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY (pass 2) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-instance",
           "workload": cfg["workload"], "workload_key": wkey, "solves_per_launch": spl, "kernel_ms_under_profiler": kms,
           "kernel": kernel, "vgpr": row0["VGPR_Count"], "sgpr": row0["SGPR_Count"], "lds_block": row0["LDS_Block_Size"],
           "counters": {**sq, **lds}, "derived": derived}, open(os.path.join(DST, f"{rnd}_pmc_sq_lds.json"), "w"), indent=1)

# ---- BASELINE config 5
c5 = last_json_line(os.path.join(SRC, "config5_full.json"))
c5p = last_json_line(os.path.join(SRC, "config5_5000.json"))
shutil.copy(one("c5_stats/**/*kernel_stats.csv"), os.path.join(DST, f"{rnd}_config5_kernel_stats.csv"))
f5, k5, keep_f, _ = counters("c5_fetch", "spicey_tran_kernel")
w5, _, keep_w, _ = counters("c5_write", "spicey_tran_kernel")
open(os.path.join(DST, f"{rnd}_config5_pmc_counter_collection.csv"), "w").write("\n".join(keep_f + keep_w[1:]) + "\n")
solves_p = c5p["solves"]
traffic5 = 2.0 * f5["FETCH_SIZE"] * 1024.0 + w5["WRITE_SIZE"] * 1024.0
for rec5 in (c5, c5p):
    rec5["roofline"]["traffic"] = traffic5 / solves_p * rec5["solves"]
    rec5["roofline"]["traffic_unit"] = "bytes per launch (rocprofv3 PMC on the 5 001-point run: 2*FETCH_SIZE + WRITE_SIZE, scaled by solves)"
    rec5["roofline"]["traffic_bytes_per_solve"] = traffic5 / solves_p
    rec5["roofline"]["hbm_utilisation"] = traffic5 / solves_p * rec5["solves_per_s"] / 8e12
    rec5["roofline"]["kernel"] = k5
json.dump(c5, open(os.path.join(DST, f"{rnd}_config5_full.json"), "w"), indent=1)
variants = {"profiled_5001_points": c5p}
for k in ("config5_tasklists_2000", "config5_x4", "config5_x16"):
    variants[k] = last_json_line(os.path.join(SRC, k + ".json"))
json.dump(variants, open(os.path.join(DST, f"{rnd}_config5_variants.json"), "w"), indent=1)

print(json.dumps({"value": bench["value"], "kernel_ms": bench["roofline"]["kernel_ms"], "frac": bench["roofline"]["frac"],
                  "traffic_per_solve": rec["traffic_bytes_per_solve"], "single": bench.get("single_instance"), "parity": bench.get("parity_max_over_tol"),
                  "valu_issue": derived["valu_issue_utilisation_at_4_cycles_per_instruction"], "salu_per_solve": derived["salu_wave_instructions_per_solve"],
                  "config5_ms_per_step": c5["ms_per_step"], "config5_traffic_per_solve": traffic5 / solves_p}, indent=1))
for fn in (f"{rnd}_bench_kernel_stats.csv", f"{rnd}_config5_kernel_stats.csv"):
    for row in csv.DictReader(open(os.path.join(DST, fn))):
        if "spicey" in row["Name"]:
            print(fn, row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e6, "ms avg")
