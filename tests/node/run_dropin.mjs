// tests/node/run_dropin.mjs — TEST INFRASTRUCTURE: executes the type-erased TypeScript drop-in layer (ts/simulateTRAN.ts,
// ts/simulateAC.ts, ts/spiceyHip.ts) under Node 12 against libspicey_hip.so.
//   node --harmony-nullish --harmony-optional-chaining run_dropin.mjs <erased_dir> <circuit.json> <out.json>
// circuit.json is the ParsedCircuit the Python parser mirror produced (elements, nodes.rev, analyses, probes; source
// waveforms as tables over the step grid).  The harness rebuilds the object shape parseNetlist returns (NodeIndex with
// count() / rev, waveform closures) and calls the drop-in exactly like lib/analysis/simulate.ts would.
import fs from "fs"
import path from "path"
import { pathToFileURL } from "url"

const [, , erased, cktPath, outPath] = process.argv
const main = async () => {
  const out = {}
  try {
    const { simulateTRAN } = await import(pathToFileURL(path.join(erased, "simulateTRAN.mjs")).href)
    const { simulateAC } = await import(pathToFileURL(path.join(erased, "simulateAC.mjs")).href)
    const { nativeVersion } = await import(pathToFileURL(path.join(erased, "spiceyHip.mjs")).href)
    const j = JSON.parse(fs.readFileSync(cktPath, "utf8"))
    if (j.netlist != null) {
      // the whole TypeScript package on its own: text -> ts/parseNetlist.ts -> ts/simulate.ts -> native, then the formatter
      const { simulate, formatTranResult } = await import(pathToFileURL(path.join(erased, "index.mjs")).href)
      const r = simulate(j.netlist)
      const enc = (x) => (Number.isFinite(x) ? x : String(x))
      out.full = { nodes: r.circuit.nodes.rev, times: r.tran.times, keysV: Object.keys(r.tran.nodeVoltages), keysI: Object.keys(r.tran.elementCurrents),
                   V: r.tran.nodeVoltages, I: {}, skipRisk: r.tran.skipRisk, text_head: formatTranResult(r.tran).split("\n").slice(0, 4),
                   state: { vPrev: r.circuit.C.map((c) => c.vPrev), iPrev: r.circuit.L.map((l) => l.iPrev), vdPrev: r.circuit.D.map((d) => d.vdPrev), isOn: r.circuit.S.map((s) => s.isOn) } }
      for (const k of out.full.keysI) out.full.I[k] = r.tran.elementCurrents[k].map(enc)
      fs.writeFileSync(outPath, JSON.stringify(out))
      return
    }
    const ckt = {
      nodes: { rev: j.nodes, count: () => j.nodes.length },
      R: j.R, C: j.C, L: j.L, S: j.S, D: j.D,
      V: j.V.map((v) => ({ ...v, waveform: v.table ? (t) => v.table[Math.round(t / j.dt)] : null })),
      analyses: j.analyses, probes: j.probes,
    }
    out.version = nativeVersion()
    if (j.analyses.tran) {
      const r1 = simulateTRAN(ckt)
      out.tran = { times: r1.times, keysV: Object.keys(r1.nodeVoltages), keysI: Object.keys(r1.elementCurrents), V: r1.nodeVoltages, I: r1.elementCurrents,
                   state: { vPrev: ckt.C.map((c) => c.vPrev), iPrev: ckt.L.map((l) => l.iPrev), vdPrev: ckt.D.map((d) => d.vdPrev), isOn: ckt.S.map((s) => s.isOn) } }
      const enc = (x) => (Number.isFinite(x) ? x : String(x))
      for (const k of out.tran.keysI) out.tran.I[k] = out.tran.I[k].map(enc)
      if (j.second_run) {  // a second call continues from the state written back into ckt
        const r2 = simulateTRAN(ckt)
        out.tran2 = { V: r2.nodeVoltages }
      }
    }
    if (j.analyses.ac) {
      const a = simulateAC(ckt)
      const pack = (rec) => { const o = {}; for (const k of Object.keys(rec)) o[k] = rec[k].map((z) => [z.re, z.im]); return o }
      out.ac = { freqs: a.freqs, keysV: Object.keys(a.nodeVoltages), keysI: Object.keys(a.elementCurrents), V: pack(a.nodeVoltages), I: pack(a.elementCurrents) }
    }
  } catch (e) {
    out.error = String(e && e.message ? e.message : e)
  }
  fs.writeFileSync(outPath, JSON.stringify(out))
}
main()
