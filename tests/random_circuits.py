"""Seeded random netlists for property tests (test infrastructure): connected R/C/L/V/D/S mixes in the reference's
netlist dialect.  Every circuit has a resistive spanning tree to ground (so the DC operating matrix is regular) plus
random extra elements, including floating capacitors / inductors, several sources and switches with random control
pairs."""
import random


def random_netlist(seed: int, max_nodes: int = 10, floating_sources: bool = False) -> str:
    rnd = random.Random(seed)
    n = rnd.randint(2, max_nodes)
    nodes = [f"n{i}" for i in range(1, n + 1)]
    lines = [f"* random circuit seed {seed}"]
    has_d = rnd.random() < 0.6
    has_s = rnd.random() < 0.4
    if has_d:
        lines.append(f".model DM D(Is={rnd.choice(['1e-14', '2e-12', '5e-15'])} N={rnd.choice(['1', '1.5', '2'])})")
    if has_s:
        lines.append(f".model SW1 SW(Ron={rnd.choice(['1', '0.5', '10'])} Roff={rnd.choice(['1e6', '1e9'])} Vt={rnd.choice(['1', '2.5'])} Vh={rnd.choice(['0', '0.2', '0.5'])})")
    k = 0

    def name(p):
        nonlocal k
        k += 1
        return f"{p}{k}"

    def val(lo, hi):
        import math
        return f"{math.exp(rnd.uniform(math.log(lo), math.log(hi))):.6g}"

    # sources: 1-2, each from its own node to ground (no source loops)
    nsrc = rnd.randint(1, min(2, n))
    src_nodes = rnd.sample(nodes, nsrc)
    for sn in src_nodes:
        kind = rnd.random()
        if kind < 0.45:
            w = f"PULSE(0 {rnd.choice(['5', '3.3', '-2', '12'])} {rnd.choice(['0', '2e-6'])} 1e-6 1e-6 {rnd.choice(['5e-6', '8e-6'])} {rnd.choice(['1.2e-5', '2e-5'])})"
        elif kind < 0.75:
            w = f"PWL(0 0 4e-6 {rnd.choice(['5', '-3'])} 9e-6 {rnd.choice(['1', '0'])} 1.6e-5 {rnd.choice(['4', '6'])})"
        else:
            w = f"dc {rnd.choice(['5', '1.5', '-3'])}"
        lines.append(f"{name('V')} {sn} 0 {w}")
    # resistive spanning tree over {ground} + nodes
    order = nodes[:]
    rnd.shuffle(order)
    placed = ["0"]
    for nd in order:
        lines.append(f"{name('R')} {nd} {rnd.choice(placed)} {val(10, 1e5)}")
        placed.append(nd)

    def pair():
        a = rnd.choice(nodes)
        b = rnd.choice(["0"] + nodes)
        while b == a:
            b = rnd.choice(["0"] + nodes)
        return a, b

    for _ in range(rnd.randint(0, n)):
        a, b = pair()
        lines.append(f"{name('R')} {a} {b} {val(10, 1e5)}")
    for _ in range(rnd.randint(1, n + 1)):
        a, b = pair()
        lines.append(f"{name('C')} {a} {b} {val(1e-10, 1e-6)}")
    for _ in range(rnd.randint(0, max(1, n // 3))):
        a, b = pair()
        if a in src_nodes and b in src_nodes + ["0"]:
            continue  # an ideal inductor straight across ideal sources is singular at t = 0 in any MNA
        lines.append(f"{name('L')} {a} {b} {val(1e-6, 1e-2)}")
    if has_d:
        for _ in range(rnd.randint(1, max(1, n // 2))):
            a, b = pair()
            lines.append(f"{name('D')} {a} {b} DM")
    if has_s:
        for _ in range(rnd.randint(1, 2)):
            a, b = pair()
            c, d = pair()
            lines.append(f"{name('S')} {a} {b} {c} {d} SW1")
    if floating_sources:
        # 1-3 sources between two non-ground nodes (own generator: the base circuit of a seed stays what it was), kept
        # loop-free together with the grounded ones (a loop of ideal sources is singular in any MNA)
        r2 = random.Random(seed * 7919 + 17)
        comp = {nd: nd for nd in nodes + ["0"]}

        def find(x):
            while comp[x] != x:
                x = comp[x]
            return x

        for sn in src_nodes:
            comp[find(sn)] = find("0")
        for _ in range(r2.randint(1, 3)):
            a, b = r2.sample(nodes, 2) if n >= 2 else (nodes[0], nodes[0])
            if find(a) == find(b):
                continue
            comp[find(a)] = find(b)
            kind = r2.random()
            if kind < 0.4:
                w = f"PULSE(0 {r2.choice(['2', '-1.5', '0.7'])} {r2.choice(['0', '3e-6'])} 1e-6 1e-6 {r2.choice(['4e-6', '7e-6'])} 1.5e-5)"
            elif kind < 0.6:
                w = f"PWL(0 0 5e-6 {r2.choice(['1', '-2'])} 1.2e-5 0.5)"
            else:
                w = f"dc {r2.choice(['1', '0.3', '-2', '0'])}"
            lines.append(f"{name('V')} {a} {b} {w}")
    lines.append(f".tran 1e-6 {rnd.choice(['2e-5', '1.5e-5', '3e-5'])}")
    lines.append(".end")
    return "\n".join(lines)


def series_diode_chain(seed: int, n: int) -> str:
    """A chain whose links are diodes (either direction) or small resistors, driven hard (up to 200 V): conducting series
    diodes couple neighbouring rows far more strongly than anything ties them to ground — the weakly diagonally dominant
    case that separates elimination orders numerically (the tridiagonal top's cyclic reduction against LU)."""
    rng = random.Random(1000 + seed)
    amp = 10 ** rng.uniform(0, 2.3)
    lines = [f"* series diode chain seed {seed}", ".model DM D(Is=1e-14 N=1)", f"V1 n1 0 PULSE(0 {amp:.4g} 0 1e-6 1e-6 5e-6 2e-5)"]
    for k in range(1, n):
        u = rng.random()
        if u < 0.45:
            lines.append(f"D{k} n{k} n{k+1} DM")
        elif u < 0.55:
            lines.append(f"D{k} n{k+1} n{k} DM")
        else:
            lines.append(f"RS{k} n{k} n{k+1} {10 ** rng.uniform(0, 4):.5g}")
        lines.append(f"R{k} n{k+1} 0 {10 ** rng.uniform(1, 5):.5g}")
        if rng.random() < 0.7:
            lines.append(f"C{k} n{k+1} 0 {10 ** rng.uniform(-10, -7):.4g}")
        if rng.random() < 0.2:
            lines.append(f"DG{k} n{k+1} 0 DM")
    lines += [".tran 1e-6 2.5e-5", ".end", ""]
    return "\n".join(lines)


def series_rlc_ladder(stages: int, r: float = 10.0, l: float = 1e-3, c: float = 1e-6) -> str:
    """R - L - C stages in series: the node between L and C carries 1/(jwL) + jwC, which cancels at f0 = 1/(2 pi sqrt(LC)).
    The reference's partial pivoting takes another row there; a static diagonal pivot order divides by ~0."""
    lines = ["* series RLC ladder", "V1 in 0 AC 1"]
    prev = "in"
    for k in range(stages):
        lines += [f"R{k} {prev} a{k} {r!r}", f"L{k} a{k} b{k} {l!r}", f"C{k} b{k} 0 {c!r}"]
        prev = f"b{k}"
    lines += [".ac lin 3 1000 2000", ".end", ""]
    return "\n".join(lines)
