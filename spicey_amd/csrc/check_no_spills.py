#!/usr/bin/env python3
"""Fail the build if any gfx950 kernel spills VECTOR registers — to scratch memory (ScratchSize > 0) or to accumulator
registers (`VGPRs Spill` with scratch 0: v_accvgpr_write / v_accvgpr_read pairs) — see kernels.hip: the toolchain hazard
is a spill store of a value defined in a divergent loop that is issued where EXEC is empty; v_accvgpr_write is
EXEC-masked exactly like a scratch store, so both kinds are refused.  A stack frame without any vector spill is accepted
only when the kernel's ISA holds no scratch instruction (checked by the caller passing --isa <disassembly>), otherwise
refused as well.  SGPR spills (v_writelane into a VGPR lane, EXEC-independent) are reported, not refused.

Usage: check_no_spills.py resource_usage.log [--isa kernels.s ...]"""
import re
import sys


def parse(log):
    """One dict per kernel, in log order (the -Rpass-analysis=kernel-resource-usage remark block of each)."""
    out = []
    cur = None
    for line in log.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        if cur is None:
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\d+)", line)
        if m:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def main(argv):
    log = open(argv[1]).read()
    errs = [l for l in log.splitlines() if " error: " in l]
    if errs:
        print("\n".join(errs))
        return 1
    isa = ""
    if "--isa" in argv:
        for p in argv[argv.index("--isa") + 1:]:
            isa += open(p).read()
    bad = []
    for k in parse(log):
        n, v, a = k["name"], k.get("VGPRs", 0), k.get("AGPRs", 0)
        s, vs, ss = k.get("ScratchSize", 0), k.get("VGPRs Spill", 0), k.get("SGPRs Spill", 0)
        note = ""
        if vs > 0:
            bad.append((n, f"{vs} vector registers spilled ({'scratch' if s else 'to AGPRs'})"))
        elif s > 0:
            # a frame without vector spills: only a dead stack object is acceptable (no scratch_ instruction in the kernel)
            body = re.search(re.escape(n) + r":\n(.*?)\n\s*s_endpgm", isa, re.S) if isa else None
            if body is None or re.search(r"\bscratch_(load|store)|\bbuffer_(load|store)[^\n]*\boffen\b[^\n]*\bs\[0:3\]", body.group(1)):
                bad.append((n, f"stack frame of {s} B per lane" + ("" if body else " (no ISA given to prove it dead)")))
            else:
                note = f"  (dead frame of {s} B: no scratch instruction in the ISA)"
        print(f"  {n[:78]:78s} VGPRs {v:3d} AGPRs {a:3d} SGPR spills {ss:3d} scratch {s}{note}")
    if bad:
        print("vector register spills / live stack frames are not allowed:")
        for b in bad:
            print("   ", b[0], "--", b[1])
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
