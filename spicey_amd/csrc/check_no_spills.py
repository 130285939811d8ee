#!/usr/bin/env python3
"""Fail the build if any gfx950 kernel spills registers (ScratchSize > 0) — see kernels.hip."""
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
bad = [(n, s) for n, s in zip(names, scratch) if s > 0]
for n, v, s in zip(names, vgprs, scratch):
    print(f"  {n[:70]:70s} VGPRs {v:3d} scratch {s}")
if bad:
    print("register spills are not allowed:", bad)
    sys.exit(1)
