#!/usr/bin/env python3
"""Generate tests/golden/ac_*.json by running the REFERENCE's own AC path (type-erased, Node 12).

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference and node).  Same recipe as
make_golden.py: erase_types.py --ac writes type-erased twins into a mkdtemp scratch directory (deleted at exit, never
inside the repo), driver_ac.mjs imports the reference's parseNetlist + simulateAC + formatAcResult from there, and only
numbers (and the reference's formatted text output) are stored under tests/golden/.

It also writes tests/golden/vgraph_*.json: the reference's spiceyTranToVGraphs / eecEngineTranToVGraphs
(lib/formatting/formatToVGraph.ts, through driver_vgraph.mjs) on two of the reference's own test netlists.

Usage: python3 tools/js_oracle/make_golden_ac.py [--only NAME ...]
"""
import argparse
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
from spicey_amd import synth  # noqa: E402

NODE = ["node", "--harmony-nullish", "--harmony-optional-chaining", "--max-old-space-size=16000"]
SMALL = ["ac_readme", "ac_rlc", "ac_two_src", "ac_err_r0", "ac_err_float", "ac_none", "ac_fv"]
SYNTH = {
    "ac_ladder30": ("rc_ladder", dict(n=30, seed=4, tran=".ac dec 10 1e3 1e8")),
    "ac_mesh6": ("rcd_mesh", dict(rows=6, seed=3, tran=".ac dec 5 1e4 1e9")),
}
LARGE = {  # BASELINE-sized topology: 1001 unknowns, 16 frequencies (~1 min on the reference JS path)
    "ac_rc1000": ("rc_ladder", dict(n=1000, seed=1, tran=".ac dec 5 1e3 1e6")),
}


def ac_netlist(gen, kw):
    """Synthetic TRAN netlists re-used for AC: the PULSE source gets an `ac 1` phasor, `.tran` becomes `.ac`."""
    text = getattr(synth, gen)(**kw)
    lines = []
    for ln in text.split("\n"):
        if ln.startswith("V1 "):
            ln = ln + " ac 1"
        lines.append(ln)
    return "\n".join(lines)


def run_driver(root, netlist_text):
    with tempfile.TemporaryDirectory(prefix="spicey_gold_") as td:
        cir, out = os.path.join(td, "in.cir"), os.path.join(td, "out.json")
        with open(cir, "w") as f:
            f.write(netlist_text)
        subprocess.run(NODE + [os.path.join(HERE, "driver_ac.mjs"), root, cir, out], check=True)
        with open(out) as f:
            return json.load(f)


def sha_freq_major(series_by_key, keys, nf):
    h = hashlib.sha256()
    for k in range(nf):
        for name in keys:
            h.update(struct.pack("<2d", *series_by_key[name][k]))
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix="spicey_oracle_")
    try:
        subprocess.run([sys.executable, os.path.join(HERE, "erase_types.py"), root, "--ac"], check=True)
        want = lambda n: not args.only or n in args.only  # noqa: E731
        for name in SMALL:
            if not want(name):
                continue
            res = run_driver(root, open(os.path.join(GOLD, "netlists", name + ".cir")).read())
            res["netlist_file"] = f"netlists/{name}.cir"
            json.dump(res, open(os.path.join(GOLD, name + ".json"), "w"))
            print(name, "error=" + repr(res.get("error")), res.get("ms"), len(res.get("freqs") or []))
        for name, (gen, kw) in SYNTH.items():
            if not want(name):
                continue
            res = run_driver(root, ac_netlist(gen, kw))
            res["generator"] = [gen, kw, "V1 += ' ac 1'"]
            json.dump(res, open(os.path.join(GOLD, name + ".json"), "w"))
            print(name, res.get("error"), res.get("ms"), len(res["freqs"]))
        for name, (gen, kw) in LARGE.items():
            if not want(name):
                continue
            res = run_driver(root, ac_netlist(gen, kw))
            nf = len(res["freqs"])
            keep = [k for k in res["keysV"] if k in ("n1", "n2", "n3", "n10", "n100", "n500", "n1000")]
            keepi = [k for k in res["keysI"] if k in ("V1", "R1", "C1", "R500", "C999")]
            out = {k: res[k] for k in ("nodes", "counts", "acSpec", "freqs", "vph", "ms")}
            out.update(generator=[gen, kw, "V1 += ' ac 1'"], nkeysV=len(res["keysV"]), nkeysI=len(res["keysI"]),
                       keysV_head=res["keysV"][:5], keysI_head=res["keysI"][:5],
                       V={k: res["V"][k] for k in keep}, I={k: res["I"][k] for k in keepi},
                       sha256_V=sha_freq_major(res["V"], res["keysV"], nf), sha256_I=sha_freq_major(res["I"], res["keysI"], nf),
                       formatted_head="\n".join(res["formatted"].split("\n")[:3])[:2000])
            out["nodes"] = out["nodes"][:8] + ["..."]
            json.dump(out, open(os.path.join(GOLD, name + ".json"), "w"))
            print(name, out["ms"], "ms", nf, "freqs", out["sha256_V"])
        for name in ("two_probes", "switch_vt_vh"):
            if not want("vgraph_" + name):
                continue
            with tempfile.TemporaryDirectory(prefix="spicey_gold_") as td:
                out = os.path.join(td, "out.json")
                subprocess.run(NODE + [os.path.join(HERE, "driver_vgraph.mjs"), root, os.path.join(GOLD, "netlists", name + ".cir"), out], check=True)
                g = json.load(open(out))
            g["netlist_file"] = f"netlists/{name}.cir"
            json.dump(g, open(os.path.join(GOLD, f"vgraph_{name}.json"), "w"))
            print("vgraph_" + name, len(g["graphs"]), "graphs")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
