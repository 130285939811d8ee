#!/usr/bin/env python3
"""Throw-away TypeScript type-erasure for the reference's TRAN path.

TEST INFRASTRUCTURE ONLY.  This script reads the 15 files of the reference's
transient path from /root/reference at run time and writes type-erased `.mjs`
twins into a scratch directory (a mkdtemp under /tmp, removed by the caller).
Nothing it produces is committed or shipped: only the *numeric* outputs of the
reference (tests/golden/*) enter the repository.

Only type syntax is removed; every arithmetic line stays byte-identical (the
caller can verify with `--diff`).  Node 12 needs `--harmony-nullish
--harmony-optional-chaining` for `??` and `?.`.
"""
import argparse
import difflib
import os
import re
import sys

REF = "/root/reference/lib"
FILES = [
    "constants/EPS.ts",
    "constants/physics.ts",
    "math/solveReal.ts",
    "stamping/stampAdmittanceReal.ts",
    "stamping/stampCurrentReal.ts",
    "stamping/stampVoltageSourceReal.ts",
    "parsing/NodeIndex.ts",
    "parsing/parseNetlist.ts",
    "parsing/parseNumberWithUnits.ts",
    "parsing/parsePulseArgs.ts",
    "parsing/parsePwlArgs.ts",
    "parsing/pulseValue.ts",
    "parsing/pwlValue.ts",
    "analysis/simulateTRAN.ts",
    "formatting/formatTranResult.ts",
]
# AC sweep path, erased with `--ac`
AC_FILES = [
    "math/Complex.ts",
    "math/solveComplex.ts",
    "stamping/stampAdmittanceComplex.ts",
    "stamping/stampVoltageSourceComplex.ts",
    "utils/logspace.ts",
    "analysis/simulateAC.ts",
    "formatting/formatAcResult.ts",
    "formatting/formatToVGraph.ts",
]


def strip_type_blocks(src: str) -> str:
    """Remove `import type`, top-level `type X = ...` and `export type {...}`."""
    lines = src.split("\n")
    out = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if re.match(r"^import type\b", ln):
            i += 1
            continue
        if re.match(r"^(export )?type \w+", ln) or re.match(r"^export type \{", ln):
            # consume until braces balance and the next line does not continue
            depth = 0
            started = False
            while i < len(lines):
                l2 = lines[i]
                depth += l2.count("{") + l2.count("<") - l2.count("}") - l2.count(">")
                started = True
                i += 1
                nxt = lines[i] if i < len(lines) else ""
                if depth <= 0 and not re.match(r"^\s*[|&]", nxt) and not l2.rstrip().endswith(("=", "|", "&")):
                    break
            continue
        out.append(ln)
        i += 1
    return "\n".join(out)


def erase_param_types(params: str) -> str:
    """Inside a parameter list: drop `: Type` after each identifier."""
    # split on top-level commas
    parts, depth, cur = [], 0, ""
    for ch in params:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    cleaned = []
    for p in parts:
        m = re.match(r"^(\s*\w+)\??\s*:\s*[\s\S]*$", p)
        cleaned.append(m.group(1) if m else p)
    return ",".join(cleaned)


def erase(src: str, rel: str) -> str:
    s = strip_type_blocks(src)
    # class field declarations (NodeIndex)
    s = re.sub(r"^\s+private map: .*\n", "", s, flags=re.M)
    s = re.sub(r"^\s+rev: string\[\]\n", "", s, flags=re.M)

    # formatTranResult's inline object parameter type
    if rel.endswith("formatTranResult.ts"):
        s = re.sub(r"function formatTranResult\(\s*tran: \{[\s\S]*?\} \| null,\s*\)",
                   "function formatTranResult(tran)", s)

    if rel.endswith("formatToVGraph.ts"):  # multi-line typed parameter lists with generic return types
        s = re.sub(r"export function (\w+)\(\s*tranResult: [^,]+,\s*ckt: ParsedCircuit,[^\n]*\n\s*simulation_experiment_id: string,\s*\): SimulationTransientVoltageGraph\[\] \{",
                   r"export function \1(tranResult, ckt, simulation_experiment_id) {", s)
        s = re.sub(r"const graphs: SimulationTransientVoltageGraph\[\] = \[\]", "const graphs = []", s)
    if rel.endswith("formatAcResult.ts"):
        s = re.sub(r"function formatAcResult\(\s*ac: \{[\s\S]*?\} \| null,\s*\)", "function formatAcResult(ac)", s)
    if rel.endswith("simulateAC.ts"):  # object-typed return annotation
        s = re.sub(r"\): \{ A: Complex\[\]\[\]; b: Complex\[\] \} \{", ") {", s)
    if rel.endswith("Complex.ts"):  # class field declarations and typed method parameters
        s = re.sub(r"^  (re|im): number\n", "", s, flags=re.M)
        s = re.sub(r"^(  (?:static )?\w+)\(([^)]*)\) \{", lambda m: f"{m.group(1)}({erase_param_types(m.group(2))}) {{", s, flags=re.M)

    # function declarations: parameter lists and return annotations
    def fn_repl(m):
        return f"{m.group(1)}({erase_param_types(m.group(2))}) {{"
    s = re.sub(r"(function \w+)\(([^)]*)\)(?:\s*:\s*[^{]+?)?\s*\{", fn_repl, s)
    # class methods
    s = re.sub(r"^(\s+(?:getOrCreate|get|matrixIndexOfNode))\(([^)]*)\) \{",
               lambda m: f"{m.group(1)}({erase_param_types(m.group(2))}) {{", s, flags=re.M)
    # arrow functions with annotated single params
    s = re.sub(r"\((\w+): number\) =>", r"(\1) =>", s)

    # multi-line `const spec: Omit<...> & {...} =`
    s = re.sub(r"const spec: Omit<[\s\S]*?> & \{ index\?: number \} =", "const spec =", s)
    # variable annotations: `const|let x: T =` / `let m: T` (no space before colon)
    s = re.sub(r"\b(const|let) (\w+): [^=\n]+? =", r"\1 \2 =", s)
    s = re.sub(r"\blet (\w+): [^=\n]+$", r"let \1", s, flags=re.M)

    s = re.sub(r"new Map<[^>]*>\(", "new Map(", s)
    s = s.replace(" as const", "")
    s = s.replace(" as keyof typeof unitMul", "")
    s = s.replace(" as number", "")
    # postfix non-null assertion
    s = re.sub(r"(?<=[\]\)\w])!(?=[\.,\)\];\s]|$)", "", s, flags=re.M)
    # logical-or assignment
    s = re.sub(r"\((\w+\[[\w\.]+\]) \|\|= \[\]\)", r"(\1 || (\1 = []))", s)
    # import specifiers
    s = re.sub(r'from "(\.[^"]+)"', r'from "\1.mjs"', s)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--diff", action="store_true", help="print source→erased diffs for review")
    ap.add_argument("--ac", action="store_true", help="also erase the AC sweep path")
    args = ap.parse_args()
    for rel in FILES + (AC_FILES if args.ac else []):
        with open(os.path.join(REF, rel)) as f:
            src = f.read()
        out = erase(src, rel)
        dst = os.path.join(args.outdir, "lib", rel[:-3] + ".mjs")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(out)
        if args.diff:
            sys.stdout.writelines(difflib.unified_diff(
                src.splitlines(True), out.splitlines(True), rel, rel + " (erased)", n=0))


if __name__ == "__main__":
    main()
