#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: type-erase THIS repository's ts/*.ts (the bun:ffi drop-in layer) into .mjs files that the
Node 12 of the image can execute, with `bun:ffi` redirected to the N-API stand-in (tools/node_shim/bun_ffi.mjs) and the
imports of reference modules (`../lib/...`) redirected to small local stand-ins (tests/node/stubs/): the reference
does not exist on the GPU box.  Only type syntax is removed; see --diff.

Usage: erase_own_ts.py <outdir> [--diff]"""
import difflib
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FILES = ["abiLayout.ts", "spiceyHip.ts", "simulateTRAN.ts", "simulateAC.ts", "constants.ts", "types.ts", "NodeIndex.ts", "numbers.ts",
         "waveforms.ts", "parseNetlist.ts", "Complex.ts", "logspace.ts", "format.ts", "simulate.ts", "index.ts"]
STUBS = {}  # (ts/ is self-contained since round 3: nothing is imported from the reference's lib/ any more)


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    return parts


def strip_param_types(params):
    out = []
    for p in split_top(params):
        depth, cut = 0, None
        for i, ch in enumerate(p):
            if ch in "([{<":
                depth += 1
            elif ch in ")]}>":
                depth -= 1
            elif ch == ":" and depth == 0:
                cut = i
                break
        out.append(p if cut is None else p[:cut].rstrip().rstrip("?"))
    return ",".join(out)


def matching_paren(s, i):
    depth = 0
    for j in range(i, len(s)):
        if s[j] == "(":
            depth += 1
        elif s[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced")


def erase(src, stub_dir, shim_path):
    lines = src.split("\n")
    out, i = [], 0
    while i < len(lines):  # drop `import type`, `export type X = ...` / `type X = ...` blocks
        ln = lines[i]
        if re.match(r"^import type\b", ln):
            i += 1
            continue
        if re.match(r"^export type \{", ln):  # `export type { A, B } from "./x"`, possibly over several lines
            while i < len(lines):
                done = "}" in lines[i]
                i += 1
                if done:
                    break
            continue
        if re.match(r"^(export )?type \w+", ln):
            depth = 0
            while i < len(lines):
                depth += lines[i].count("{") - lines[i].count("}")
                i += 1
                if depth <= 0:
                    break
            continue
        out.append(ln)
        i += 1
    # classes: field declarations go, method signatures lose their parameter and return types
    out2, in_class = [], False
    for ln in out:
        if re.match(r"^(export )?class \w+", ln):
            in_class = True
        elif in_class and ln.startswith("}"):
            in_class = False
        elif in_class:
            if re.match(r"^  (?:private |public |readonly )*\w+[?!]?: [^=(){}]+$", ln):
                continue
            m = re.match(r"^(  (?:static )?(?!if\b|for\b|while\b|switch\b|catch\b)\w+)\((.*)\)(?:: [^{]+)? \{$", ln)
            if m:
                ln = f"{m.group(1)}({strip_param_types(m.group(2))}) {{"
        out2.append(ln)
    s = "\n".join(out2)
    s = re.sub(r",\s*type \w+(?=\s*[,}])", "", s)  # `import { a, type B }`
    # function declarations: parameter and return types
    res, pos = "", 0
    for m in re.finditer(r"\bfunction \w+\(", s):
        if m.start() < pos:
            continue
        o = m.end() - 1
        c = matching_paren(s, o)
        eol = s.index("\n", c)
        body = s.rindex("{", c, eol)
        res += s[pos:o] + "(" + strip_param_types(s[o + 1:c]) + ") "
        pos = body
    s = res + s[pos:]
    # arrow functions: typed parameter lists and return types
    def arrow(m):
        return "(" + strip_param_types(m.group(1)) + ") =>"
    s = re.sub(r"\(([^()]*)\)\s*(?::\s*[\w\[\]<>| ]+?)?\s*=>", arrow, s)
    # variable annotations
    s = re.sub(r"\b(const|let) (\w+): [^=\n]+? =", r"\1 \2 =", s)
    s = re.sub(r"\blet (\w+): [^=\n;]+$", r"let \1", s, flags=re.M)  # `let x: T` without an initialiser
    # casts
    s = re.sub(r" as unknown as \w+", "", s)
    s = re.sub(r" as (any|const|number|\w+Array)\b", "", s)
    # non-null assertions:  x!  x]!  x)!   before . , ) ] ; or end of line / space
    s = re.sub(r"(?<=[\]\)\w])!(?=[\.,\)\];\s]|$)", "", s, flags=re.M)
    # logical-or assignment:  (a[b[c]] ||= [])  ->  (a[b[c]] || (a[b[c]] = []))
    s = re.sub(r"\((\w+\[(?:[^\[\]]|\[[^\[\]]*\])+\]) \|\|= \[\]\)", r"(\1 || (\1 = []))", s)
    # module specifiers
    s = s.replace('from "bun:ffi"', f'from "{shim_path}"')
    for ref, stub in STUBS.items():
        s = s.replace(f'from "{ref}"', f'from "{stub_dir}/{stub}.mjs"')
    s = re.sub(r'from "(\./[^"]+)"', r'from "\1.mjs"', s)
    s = s.replace("import.meta.dir", 'new URL(".", import.meta.url).pathname.replace(/\\/$/, "")')
    return s


def main():
    outdir = sys.argv[1]
    os.makedirs(outdir, exist_ok=True)
    stub_dir = os.path.join(REPO, "tests", "node", "stubs")
    shim = os.path.join(REPO, "tools", "node_shim", "bun_ffi.mjs")
    for f in FILES:
        src = open(os.path.join(REPO, "ts", f)).read()
        dst = erase(src, stub_dir, shim)
        open(os.path.join(outdir, f[:-3] + ".mjs"), "w").write(dst)
        if "--diff" in sys.argv:
            sys.stdout.writelines(difflib.unified_diff(src.splitlines(True), dst.splitlines(True), f, f + " (erased)", n=0))


if __name__ == "__main__":
    main()
