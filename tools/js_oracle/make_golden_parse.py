#!/usr/bin/env python3
"""tests/golden/parser_cases.json: the reference's own parseNetlist (type-erased, Node 12) on tricky netlist snippets —
unit suffixes, titles / comments / continuations, source specifications, models, analysis cards, error messages.
TEST INFRASTRUCTURE ONLY (same recipe as make_golden.py; numbers / strings only are stored).

Usage: python3 tools/js_oracle/make_golden_parse.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")
NODE = ["node", "--harmony-nullish", "--harmony-optional-chaining"]

UNITS = ["1", "1.5", "1k", "1K", "2.2meg", "3MEG", "4m", "5M", "6u", "7n", "8p", "9f", "1g", "2G", "3t", "1T", "1e3", "1E-3", "2.5e-3k", "1kohm", "10uF",
         "100nH", "1mil", "-3.3", "+4", ".5", "5.", "1e", "abc", "1k5", "1 k", "0x10", "1_000", "1,5", "Infinity", "NaN", "1e400", "4.7µ", "10Ω"]
CASES = [f"* units\nR1 a 0 {u}\n.end" for u in UNITS]
CASES += [
    "title line without star\nR1 1 0 1k\n.end",
    "Demo of a simple AC circuit\nv1 1 0 dc 0 ac 1\nr1 1 2 30\n.end",
    "\n\n* blank lines first\nR1 1 0 1k\n\n\nC1 1 0 1u\n.end\nR2 1 0 5\n",
    "* comments\nR1 1 0 1k ; trailing\nR2 1 0 2k $ dollar\n* full line\n; semicolon line\nR3 1 0 3k\n.end",
    "* continuation\nV1 in 0 PULSE(0 5\n+ 0 1n 1n\n+ 5u 10u)\nR1 in 0 1k\n.end",
    "* case\nr1 A 0 1K\nR2 a B 2k\nc1 b GND 1u\nC2 B gnd 2u\nl1 b 0 1m\n.END",
    "* ground names\nR1 a gnd 1\nR2 b GND 1\nR3 c Gnd 1\nR4 d 0 1\nR5 e 00 1\nR6 f ground 1\n.end",
    "* duplicate names\nR1 a 0 1\nR1 a b 2\nC1 b 0 1u\nC1 a 0 2u\n.end",
    "* numeric-like node names order\nR1 10 2 1\nR2 2 b 1\nR3 b 1 1\nR4 1 0 1\nR5 007 0 1\n.end",
    "* two terminals same node\nR1 a a 1k\nC1 0 0 1u\nV1 a 0 1\n.end",
    "* missing value\nR1 a 0\n.end",
    "* missing node\nR1 a\n.end",
    "* bad value\nC1 a 0 xyz\n.end",
    "* unknown element\nQ1 c b e mod\nM1 d g s b nmos\nX1 a b sub\nR1 a 0 1\n.end",
    "* unknown directives\n.option reltol=1e-3\n.include foo.lib\n.param x=1\n.ic v(a)=1\nR1 a 0 1\n.end",
    "* V dc forms\nV1 a 0 5\nV2 b 0 dc 3.3\nV3 c 0 DC 1 AC 2\nV4 d 0 ac 1 45\nV5 e 0 AC 1m -90 dc 2\nV6 f 0\n.end",
    "* V pulse\nV1 a 0 PULSE(0 5 0 1n 1n 5u 10u)\nV2 b 0 pulse(1 -1 1u 2u 3u 4u 20u 3)\nV3 c 0 dc 1 PULSE(0 3.3 2e-6 1e-6 1e-6 8e-6 2e-5)\nV4 d 0 PULSE 0 5 0 1n 1n 5u 10u\n.end",
    "* V pulse spaced\nV1 a 0 PULSE ( 0 5 0 1n 1n 5u 10u )\nV2 b 0 PULSE(0,5,0,1n,1n,5u,10u)\n.end",
    "* V pulse short\nV1 a 0 PULSE(0 5 0 1n 1n 5u)\n.end",
    "* V pulse bad number\nV1 a 0 PULSE(0 5 0 1n xx 5u 10u)\n.end",
    "* V pulse unbalanced\nV1 a 0 PULSE(0 5 0 1n 1n 5u 10u\n.end",
    "* V pwl\nV1 a 0 PWL(0 0 1u 5 2u 5 3u 0)\nV2 b 0 pwl(0 1 1m 2)\nV3 c 0 PWL(1u 3)\n.end",
    "* V pwl odd\nV1 a 0 PWL(0 0 1u 5 2u)\n.end",
    "* V pwl bad\nV1 a 0 PWL(0 0 1u zz)\n.end",
    "* V sin unsupported\nV1 a 0 SIN(0 1 1k)\nR1 a 0 1\n.end",
    "* switch ok\n.model SW1 SW(Ron=1 Roff=1e6 Vt=2.5 Vh=0.5)\nS1 a b c 0 SW1\nR1 a 0 1\n.end",
    "* switch von voff\n.model SW2 SW(Ron=0.1 Roff=1Meg Von=3 Voff=1)\nS1 a b c d SW2\n.end",
    "* switch defaults\n.model SW3 SW()\n.model SW4 SW\nS1 a b c d SW3\nS2 a b c d SW4\n.end",
    "* switch on\n.model SW1 SW(Ron=1 Roff=1e6 Vt=1 Vh=0)\nS1 a b c d SW1 ON\nS2 a b c d SW1 off\n.end",
    "* switch unknown model\nS1 a b c d NOPE\n.end",
    "* switch model later\nS1 a b c d SWL\n.model SWL SW(Ron=2 Roff=2k Vt=1 Vh=0.1)\n.end",
    "* switch missing nodes\n.model SW1 SW(Ron=1)\nS1 a b c SW1\n.end",
    "* model case\n.MODEL sw1 sw(RON=1 roff=10k VT=1 vh=0)\nS1 a b c d SW1\n.end",
    "* diode ok\n.model DM D(Is=1e-14 N=1)\nD1 a 0 DM\nD2 0 a dm\n.end",
    "* diode defaults\n.model DX D\n.model DY D()\n.model DZ D(N=2)\nD1 a 0 DX\nD2 a 0 DY\nD3 a 0 DZ\n.end",
    "* diode unknown model\nD1 a 0 NOPE\n.end",
    "* diode extra params\n.model DM D(Is=1e-12 N=1.5 Rs=0.1 Cjo=1p BV=50)\nD1 a k DM\n.end",
    "* model other type\n.model QN NPN(BF=100)\n.model RR R(R=1)\nR1 a 0 1\n.end",
    "* model malformed\n.model\nR1 a 0 1\n.end",
    "* tran forms\nR1 a 0 1\n.tran 1u 1m\n.end",
    "* tran upper\nR1 a 0 1\n.TRAN 1US 10MS\n.end",
    "* tran three\nR1 a 0 1\n.tran 1u 1m 0.5m\n.end",
    "* tran one\nR1 a 0 1\n.tran 1u\n.end",
    "* tran zero dt\nR1 a 0 1\n.tran 0 1m\n.end",
    "* tran uic\nR1 a 0 1\n.tran 1u 1m uic\n.end",
    "* tran twice\nR1 a 0 1\n.tran 1u 1m\n.tran 2u 2m\n.end",
    "* ac forms\nR1 a 0 1\n.ac dec 10 1 1k\n.end",
    "* ac lin\nR1 a 0 1\n.AC LIN 5 10 50\n.end",
    "* ac oct\nR1 a 0 1\n.ac oct 10 1 1k\n.end",
    "* ac short\nR1 a 0 1\n.ac dec 10 1\n.end",
    "* ac float points\nR1 a 0 1\n.ac dec 2.5 1 1k\n.end",
    "* print\nR1 a 0 1\nR2 b a 1\n.print tran v(a) v(b)\n.end",
    "* print upper\nR1 A 0 1\n.PRINT TRAN V(A) V(nope) I(R1)\n.end",
    "* print ac\nR1 a 0 1\n.print ac v(a)\n.end",
    "* print spaces\nR1 a 0 1\n.print tran v( a ) v(a,0)\n.end",
    "* probe\nR1 a 0 1\n.probe v(a)\n.plot tran v(a)\n.end",
    "* inductor & cap ic\nL1 a 0 1m ic=1\nC1 a 0 1u IC=2\n.end",
    "* whitespace tabs\nR1\ta\t0\t1k\n  R2   a   b   2k  \n.end",
    "* crlf\r\nR1 a 0 1k\r\nC1 a 0 1u\r\n.end\r\n",
    "",
    "* only comment",
    ".end",
]


def main():
    sys.path.insert(0, HERE)
    root = tempfile.mkdtemp(prefix="spicey_oracle_")
    try:
        subprocess.run([sys.executable, os.path.join(HERE, "erase_types.py"), root], check=True)
        cases = os.path.join(root, "cases.json")
        out = os.path.join(root, "out.json")
        json.dump(CASES, open(cases, "w"))
        subprocess.run(NODE + [os.path.join(HERE, "driver_parse.mjs"), root, cases, out], check=True)
        g = json.load(open(out))
        json.dump(g, open(os.path.join(GOLD, "parser_cases.json"), "w"))
        nerr = sum(1 for r in g["results"] if "error" in r)
        print(len(g["cases"]), "cases,", nerr, "errors")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
