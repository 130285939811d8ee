#!/usr/bin/env python3
"""gpurun_out/prof_round/ (tools/profile_round.sh, run on the GPU box) -> profiles/rNN_* (tracked)."""
import csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof_round")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
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


for name in ("bench_line", "bench_line_rc_ladder", "bench_line_under_rocprof"):
    rec = last_json_line(os.path.join(SRC, name + ".json"))
    json.dump(rec, open(os.path.join(DST, f"{rnd}_{name}.json"), "w"), indent=1)
bench = last_json_line(os.path.join(SRC, "bench_line.json"))

shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{rnd}_bench_kernel_stats.csv"))
shutil.copy(one("ac_stats/**/*kernel_stats.csv"), os.path.join(DST, f"{rnd}_ac_kernel_stats.csv"))
aclines = [json.loads(l) for l in open(os.path.join(SRC, "ac_probe.json")) if l.startswith("{")]
json.dump(aclines, open(os.path.join(DST, f"{rnd}_ac_probe.json"), "w"), indent=1)

tot = {}
kernel = None
for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    src = one(which + "/**/*counter_collection.csv")
    rows = [r for r in csv.DictReader(open(src)) if "spicey_tran_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    assert rows, (which, "no spicey kernel rows")
    kernel = rows[0]["Kernel_Name"]
    tot[counter] = sum(float(r["Counter_Value"]) for r in rows) / len({r["Dispatch_Id"] for r in rows})
    # keep the spicey rows plus the header (the full file also lists torch's fill / copy kernels)
    with open(src) as f:
        lines = f.read().splitlines()
    keep = [lines[0]] + [l for l in lines[1:] if "spicey_" in l]
    open(os.path.join(DST, f"{rnd}_{which}_counter_collection.csv"), "w").write("\n".join(keep) + "\n")
one_launch = last_json_line(os.path.join(SRC, "pmc_fetch.json"))
spl = int(one_launch["roofline"]["solves_per_launch"])
cfg = one_launch["config"]
fetch_raw = tot["FETCH_SIZE"] * 1024.0  # KB -> bytes
write = tot["WRITE_SIZE"] * 1024.0
traffic = 2.0 * fetch_raw + write
rec = {
    "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline  (separate passes for FETCH_SIZE and WRITE_SIZE; tools/profile_round.sh)",
    "workload": cfg["workload"],
    "workload_key": f"diode_chain:{cfg['nodes']}:{cfg['timesteps']}:{cfg['instances_per_gpu']}:1",
    "solves_per_launch": spl,
    "kernel": kernel,
    "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
    "fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": 2.0 * fetch_raw, "write_bytes": write,
    "traffic_bytes_per_launch": traffic, "traffic_bytes_per_solve": traffic / spl,
    "note": "gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> doubled. "
            "WRITE_SIZE is exact for streaming stores: expected result bytes = instances*(steps+1)*(nodes + currents)*8.",
}
json.dump(rec, open(os.path.join(DST, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({"value": bench["value"], "kernel_ms": bench["roofline"]["kernel_ms"], "frac": bench["roofline"]["frac"],
                  "traffic_per_solve": rec["traffic_bytes_per_solve"], "single": bench.get("single_instance")}, indent=1))
for row in csv.DictReader(open(os.path.join(DST, f"{rnd}_bench_kernel_stats.csv"))):
    if "spicey" in row["Name"]:
        print(row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e6, "ms avg")
