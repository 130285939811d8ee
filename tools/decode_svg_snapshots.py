#!/usr/bin/env python3
"""Decode the reference's committed SVG snapshots into numeric golden series.

TEST INFRASTRUCTURE; runs only where /root/reference exists.  Each
`<path class="simulation-line" d="M x y L x y …">` in
/root/reference/tests/transient/__snapshots__/*.snap.svg holds one series, one vertex per
timestep; y is printed to 6 decimals over a 456-px plot (y=520 <-> axis min, y=64 <-> axis max;
axis range from the axis-label-y tick texts).  Output: tests/golden/svg_series.json with the
spicey series (legend entries without "(ngspice)") converted back to volts.  Only numbers and
labels are stored ("The SVG snapshots are decodable golden vectors").
"""
import glob
import json
import os
import re

SNAP = "/root/reference/tests/transient/__snapshots__"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "svg_series.json")
# snapshot file -> golden netlist that produced it
NETLIST = {
    "two-probes-two-probes-graph": "two_probes",
    "transient01-rc-pulse-comparison": "transient01",
    "switch-vt-vh-switch-vt-vh-graph": "switch_vt_vh",
    "vswitch-pwl-vswitch-pwl-control": "vswitch_pwl",
    "boost-converter-probe-boost-converter-probe": "boost_probe",
}


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(SNAP, "*.snap.svg"))):
        name = os.path.basename(path)[: -len(".snap.svg")]
        svg = open(path).read()
        yl = re.findall(r'class="axis-label axis-label-y" x="[^"]*" y="([^"]*)"[^>]*>([^<]*)<', svg)
        ys = sorted((float(y), float(v)) for y, v in yl)
        (y_top, v_max), (y_bot, v_min) = ys[0], ys[-1]
        legends = re.findall(r'class="legend-label"[^>]*>([^<]*)<', svg)
        paths = re.findall(r'<path class="simulation-line" d="([^"]*)"', svg)
        assert len(legends) == len(paths), (name, len(legends), len(paths))
        series = {}
        for lab, d in zip(legends, paths):
            if "(ngspice)" in lab:
                continue
            pts = re.findall(r"[ML] (\S+) (\S+)", d)
            series[lab] = [v_min + (y_bot - float(y)) / (y_bot - y_top) * (v_max - v_min) for _, y in pts]
        out[name] = {"netlist": NETLIST[name], "axis": [v_min, v_max], "quantum": (v_max - v_min) / (y_bot - y_top) * 1e-6,
                     "series": series}
        print(name, {k: len(v) for k, v in series.items()}, "axis", v_min, v_max)
    json.dump(out, open(OUT, "w"))


if __name__ == "__main__":
    main()
