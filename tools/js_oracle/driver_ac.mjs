// TEST INFRASTRUCTURE ONLY — runs the reference's own (type-erased) AC path under Node 12 and dumps numeric
// results.  Usage:
//   node --harmony-nullish --harmony-optional-chaining driver_ac.mjs <erased_root> <netlist.cir> <out.json>
// Output JSON: { nodes, counts, acSpec, freqs, vph: [[re,im] per source], keysV, keysI, V: {name: [[re,im]...]},
//   I: {...}, formatted, ms, error? }.  Doubles print in shortest round-trip form (bit-exact through json.load).
import fs from "fs"
import path from "path"
import { pathToFileURL } from "url"

const [, , root, netlistPath, outPath] = process.argv
const imp = (rel) => import(pathToFileURL(path.join(root, rel)).href)

const main = async () => {
  const { parseNetlist } = await imp("lib/parsing/parseNetlist.mjs")
  const { simulateAC } = await imp("lib/analysis/simulateAC.mjs")
  const { formatAcResult } = await imp("lib/formatting/formatAcResult.mjs")
  const { Complex } = await imp("lib/math/Complex.mjs")
  const text = fs.readFileSync(netlistPath, "utf8")
  const out = {}
  try {
    const ckt = parseNetlist(text)
    out.nodes = ckt.nodes.rev
    out.counts = { R: ckt.R.length, C: ckt.C.length, L: ckt.L.length, V: ckt.V.length, S: ckt.S.length, D: ckt.D.length }
    out.acSpec = ckt.analyses.ac
    out.vph = ckt.V.map((vs) => { const z = Complex.fromPolar(vs.acMag || 0, vs.acPhaseDeg || 0); return [z.re, z.im] })
    const t0 = Date.now()
    const res = simulateAC(ckt)
    out.ms = Date.now() - t0
    if (res) {
      const pack = (rec) => { const o = {}; for (const k of Object.keys(rec)) o[k] = rec[k].map((z) => [z.re, z.im]); return o }
      out.freqs = res.freqs
      out.keysV = Object.keys(res.nodeVoltages)
      out.keysI = Object.keys(res.elementCurrents)
      out.V = pack(res.nodeVoltages)
      out.I = pack(res.elementCurrents)
      out.formatted = formatAcResult(res)
    } else out.none = true
  } catch (e) {
    out.error = String(e && e.message ? e.message : e)
  }
  fs.writeFileSync(outPath, JSON.stringify(out))
}
main()
