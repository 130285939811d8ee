// ts/simulateAC.ts — drop-in replacement of lib/analysis/simulateAC.ts with the native solver underneath.
// Same signature and result shape ({ freqs, nodeVoltages, elementCurrents } of Complex), same Error messages.
// What stays in TypeScript is what depends on the JS engine's Math (Math.pow in logspace, Math.cos / Math.sin in
// Complex.fromPolar) and the argument checks that throw before any arithmetic; every per-frequency complex solve
// (simulateAC.ts:80-126) is one native call for the whole sweep.
import { EPS } from "./constants"
import { Complex } from "./Complex"
import type { ParsedCircuit } from "./types"
import { logspace } from "./logspace"
import { runAcNative, type FlatCircuit } from "./spiceyHip"

function frequencies(ac: NonNullable<ParsedCircuit["analyses"]["ac"]>): number[] {
  if (ac.mode === "dec") return logspace(ac.f1, ac.f2, ac.N)
  const npts = Math.max(2, ac.N)
  const step = (ac.f2 - ac.f1) / (npts - 1)
  return Array.from({ length: npts }, (_, i) => ac.f1 + i * step)
}

function flattenLinear(ckt: ParsedCircuit): FlatCircuit {
  const i32 = (a: number[]) => Int32Array.from(a)
  const f64 = (a: number[]) => Float64Array.from(a)
  const none = { i: new Int32Array(0), f: new Float64Array(0) }
  return {
    nNodes: ckt.nodes.count() - 1,
    R: { n1: i32(ckt.R.map((e) => e.n1)), n2: i32(ckt.R.map((e) => e.n2)), val: f64(ckt.R.map((e) => e.R)) },
    C: { n1: i32(ckt.C.map((e) => e.n1)), n2: i32(ckt.C.map((e) => e.n2)), val: f64(ckt.C.map((e) => e.C)), vPrev: f64(ckt.C.map(() => 0)) },
    L: { n1: i32(ckt.L.map((e) => e.n1)), n2: i32(ckt.L.map((e) => e.n2)), val: f64(ckt.L.map((e) => e.L)), iPrev: f64(ckt.L.map(() => 0)) },
    V: { n1: i32(ckt.V.map((e) => e.n1)), n2: i32(ckt.V.map((e) => e.n2)) },
    // the AC analysis has no diode / switch stamps (simulateAC.ts:38-59)
    S: { n1: none.i, n2: none.i, cp: none.i, cn: none.i, ron: none.f, roff: none.f, von: none.f, voff: none.f, isOn: none.i },
    D: { np: none.i, nm: none.i, is: none.f, n: none.f, vdPrev: none.f },
  }
}

function simulateAC(ckt: ParsedCircuit) {
  if (!ckt.analyses.ac) return null
  const freqs = frequencies(ckt.analyses.ac)

  // errors the reference throws while building the system (simulateAC.ts:39, :51-53 via Complex.div)
  for (const r of ckt.R) if (r.R <= 0) throw new Error(`R ${r.name} must be > 0`)
  for (const f of freqs)
    for (const l of ckt.L) {
      const w = 2 * Math.PI * f * l.L
      if (!(Math.abs(w) < EPS) && w * w < EPS) throw new Error("Complex divide by ~0")
    }

  const nV = ckt.V.length
  const vph = new Float64Array(nV * 2)
  ckt.V.forEach((vs, k) => {
    const z = Complex.fromPolar(vs.acMag || 0, vs.acPhaseDeg || 0)
    vph[2 * k] = z.re
    vph[2 * k + 1] = z.im
  })

  const flat = flattenLinear(ckt)
  const res = runAcNative(flat, Float64Array.from(freqs), vph)

  const nNodes = flat.nNodes
  const nodeVoltages: Record<string, Complex[]> = {}
  ckt.nodes.rev.forEach((name, id) => {
    if (id !== 0) nodeVoltages[name] = []
  })
  const names = [...ckt.R.map((e) => e.name), ...ckt.C.map((e) => e.name), ...ckt.L.map((e) => e.name), ...ckt.V.map((e) => e.name)]
  const nCur = names.length
  const elementCurrents: Record<string, Complex[]> = {}
  for (let k = 0; k < freqs.length; k++) {
    for (let id = 1; id <= nNodes; id++) {
      const series = nodeVoltages[ckt.nodes.rev[id]!]
      if (series) series.push(Complex.from(res.outV[(k * nNodes + id - 1) * 2]!, res.outV[(k * nNodes + id - 1) * 2 + 1]!))
    }
    for (let j = 0; j < nCur; j++)
      (elementCurrents[names[j]!] ||= []).push(Complex.from(res.outI[(k * nCur + j) * 2]!, res.outI[(k * nCur + j) * 2 + 1]!))
  }
  return { freqs, nodeVoltages, elementCurrents }
}

export { simulateAC }
