#!/usr/bin/env python3
"""Fail the build if any gfx950 kernel spills vector registers to memory (ScratchSize > 0 with VGPR spills) or keeps a
register array in scratch (a large frame) — see kernels.hip: the toolchain hazard is a spill store issued where EXEC is
empty, and a demoted register array is a performance cliff.  A small frame WITHOUT vector spills (a dead stack object the
backend never addresses: no scratch instruction in the kernel) is allowed and noted."""
import re
import sys

log = open(sys.argv[1]).read()
errs = [l for l in log.splitlines() if " error: " in l]
if errs:
    print("\n".join(errs))
    sys.exit(1)
names = re.findall(r"Function Name: (\S+)", log)
scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", log)]
vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", log)]
vspill = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", log)]
sspill = [int(x) for x in re.findall(r"SGPRs Spill: (\d+)", log)]
bad = []
for n, v, s, vs, ss in zip(names, vgprs, scratch, vspill, sspill):
    note = ""
    if s > 0 and (vs > 0 or s > 64):
        bad.append((n, s, vs))
    elif s > 0:
        note = f"  (frame of {s} B, no vector spill)"
    print(f"  {n[:70]:70s} VGPRs {v:3d} scratch {s}{note}")
if bad:
    print("register spills are not allowed:", bad)
    sys.exit(1)
