// ts/simulateTRAN.ts — drop-in replacement of lib/analysis/simulateTRAN.ts with the native solver underneath.
// Same signature, same result shape, same in-place mutation of `ckt` state, same Error messages.
// Everything the reference does outside its time loop stays here, in the reference's own order.
import { EPS } from "./constants"
import type { ParsedCircuit } from "./types"
import { runTransientNative, type FlatCircuit } from "./spiceyHip"

/** simulateTRAN.ts:14-19 verbatim semantics: must be the same double operations in the same order. */
function computeEffectiveTimeStep(dtRequested: number, tstop: number) {
  const dtEff = dtRequested > EPS ? dtRequested : Math.max(tstop / 1000, EPS)
  const steps = Math.max(1, Math.ceil(tstop / Math.max(dtEff, EPS)))
  const dt = steps > 0 ? tstop / steps : tstop
  return { dt, steps }
}

function flatten(ckt: ParsedCircuit): FlatCircuit {
  const S = ckt.S.filter((s) => s.model)
  const D = ckt.D.filter((d) => d.model)
  const i32 = (a: number[]) => Int32Array.from(a)
  const f64 = (a: number[]) => Float64Array.from(a)
  return {
    nNodes: ckt.nodes.count() - 1,
    R: { n1: i32(ckt.R.map((e) => e.n1)), n2: i32(ckt.R.map((e) => e.n2)), val: f64(ckt.R.map((e) => e.R)) },
    C: { n1: i32(ckt.C.map((e) => e.n1)), n2: i32(ckt.C.map((e) => e.n2)), val: f64(ckt.C.map((e) => e.C)), vPrev: f64(ckt.C.map((e) => e.vPrev)) },
    L: { n1: i32(ckt.L.map((e) => e.n1)), n2: i32(ckt.L.map((e) => e.n2)), val: f64(ckt.L.map((e) => e.L)), iPrev: f64(ckt.L.map((e) => e.iPrev)) },
    V: { n1: i32(ckt.V.map((e) => e.n1)), n2: i32(ckt.V.map((e) => e.n2)) },
    S: {
      n1: i32(S.map((e) => e.n1)), n2: i32(S.map((e) => e.n2)), cp: i32(S.map((e) => e.ncPos)), cn: i32(S.map((e) => e.ncNeg)),
      ron: f64(S.map((e) => e.model!.Ron)), roff: f64(S.map((e) => e.model!.Roff)),
      von: f64(S.map((e) => e.model!.Von)), voff: f64(S.map((e) => e.model!.Voff)), isOn: i32(S.map((e) => (e.isOn ? 1 : 0))),
    },
    D: {
      np: i32(D.map((e) => e.nPlus)), nm: i32(D.map((e) => e.nMinus)), is: f64(D.map((e) => e.model!.Is)),
      n: f64(D.map((e) => e.model!.N)), vdPrev: f64(D.map((e) => e.vdPrev)),
    },
  }
}

function simulateTRAN(ckt: ParsedCircuit) {
  if (!ckt.analyses.tran) return null
  const { dt: dtRequested, tstop } = ckt.analyses.tran
  const { dt, steps } = computeEffectiveTimeStep(dtRequested, tstop)
  const nV = ckt.V.length

  // waveform closures cannot cross the FFI: pre-evaluate `vs.waveform ? vs.waveform(t) : vs.dc || 0` (:67) at t = step*dt (:147)
  const src = new Float64Array((steps + 1) * nV)
  const times: number[] = []
  let t = 0
  for (let step = 0; step <= steps; step++, t = step * dt) {
    times.push(t)
    for (let k = 0; k < nV; k++) {
      const vs = ckt.V[k]!
      src[step * nV + k] = vs.waveform ? vs.waveform(t) : vs.dc || 0
    }
  }

  const flat = flatten(ckt)
  // `.PRINT TRAN v(..)` probes are resolved to node ids here and applied by the device, which then writes, and this
  // function re-keys, only those columns (the reference records every node and filters at the end, :240-249)
  const wanted = ckt.probes.tran.map((p) => p.toUpperCase())
  const recorded: number[] = []
  for (let id = 1; id <= flat.nNodes; id++) {
    if (wanted.length === 0 || wanted.includes(ckt.nodes.rev[id]!.toUpperCase())) recorded.push(id)
  }
  if (wanted.length > 0) flat.outNodes = Int32Array.from(recorded)
  const res = runTransientNative(flat, steps, dt, src)

  // re-key: one strided column copy per recorded node / per element, then a plain number[] (the reference's result type);
  // keys enter the objects in the reference's insertion order so that JS key order (integer-like names first) matches
  const np1 = steps + 1
  const column = (buf: Float64Array, stride: number, col: number): number[] => {
    const out = new Float64Array(np1)
    for (let s = 0, k = col; s < np1; s++, k += stride) out[s] = buf[k]!
    return Array.from(out)
  }
  const nOut = recorded.length
  const nodeVoltages: Record<string, number[]> = {}
  recorded.forEach((id, c) => {
    nodeVoltages[ckt.nodes.rev[id]!] = column(res.outV, nOut, c)
  })
  const names = [
    ...ckt.R.map((e) => e.name), ...ckt.C.map((e) => e.name), ...ckt.L.map((e) => e.name), ...ckt.V.map((e) => e.name),
    ...ckt.S.filter((s) => s.model).map((e) => e.name), ...ckt.D.filter((d) => d.model).map((e) => e.name),
  ]
  const nCur = names.length
  const elementCurrents: Record<string, number[]> = {}
  const columnsOf: Map<string, number[]> = new Map()
  names.forEach((nm, j) => {
    const cols = columnsOf.get(nm)
    if (cols) cols.push(j)
    else columnsOf.set(nm, [j])
  })
  for (const [nm, cols] of columnsOf) {
    if (cols.length === 1) {
      elementCurrents[nm] = column(res.outI, nCur, cols[0]!)
    } else {
      // elements that share a name share one array in the reference, interleaved per step
      const out: number[] = []
      for (let s = 0; s < np1; s++) for (const j of cols) out.push(res.outI[s * nCur + j]!)
      elementCurrents[nm] = out
    }
  }

  // state write-back (:221-237, :122-124): a second simulateTRAN(ckt) continues from here
  ckt.C.forEach((c, i) => (c.vPrev = res.state.vPrev[i]!))
  ckt.L.forEach((l, i) => (l.iPrev = res.state.iPrev[i]!))
  ckt.D.filter((d) => d.model).forEach((d, i) => (d.vdPrev = res.state.vdPrev[i]!))
  ckt.S.filter((s) => s.model).forEach((s, i) => (s.isOn = res.state.isOn[i] !== 0))

  // (`skipRisk` is this solver's addition to the reference's three keys: > 0 says that the reference's own row-update skip,
  // solveReal.ts:45, may have made ITS numbers differ from these — ts/spiceyHip.ts, NativeTranResult.skipRisk)
  return { times, nodeVoltages, elementCurrents, skipRisk: res.skipRisk }
}

export { simulateTRAN }
