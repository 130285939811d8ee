// TEST INFRASTRUCTURE ONLY — runs the reference's own (type-erased) TRAN path under Node 12
// and dumps numeric results.  Usage:
//   node --harmony-nullish --harmony-optional-chaining driver.mjs <erased_root> <netlist.cir> <out.json> [repeat]
// `erased_root` is the scratch directory produced by erase_types.py; nothing from it is committed.
// Output JSON: { nodes: rev[], probes, steps, keysV: [...], keysI: [...], times, V: {name: [...]}, I: {name: [...]},
//   state: {C_vPrev, L_iPrev, D_vdPrev, S_isOn}, counts, error?: string }
// JS prints doubles in shortest round-trip form, so Python's json.load recovers them bit-exactly.
import fs from "fs"
import path from "path"
import { pathToFileURL } from "url"

const [, , root, netlistPath, outPath, repeatArg] = process.argv
const repeat = repeatArg ? parseInt(repeatArg, 10) : 1

const main = async () => {
  const { parseNetlist } = await import(pathToFileURL(path.join(root, "lib/parsing/parseNetlist.mjs")).href)
  const { simulateTRAN } = await import(pathToFileURL(path.join(root, "lib/analysis/simulateTRAN.mjs")).href)
  const { formatTranResult } = await import(pathToFileURL(path.join(root, "lib/formatting/formatTranResult.mjs")).href)
  const text = fs.readFileSync(netlistPath, "utf8")
  const out = {}
  try {
    const ckt = parseNetlist(text)
    out.nodes = ckt.nodes.rev
    out.probes = ckt.probes.tran
    out.skipped = ckt.skipped
    out.counts = { R: ckt.R.length, C: ckt.C.length, L: ckt.L.length, V: ckt.V.length, S: ckt.S.length, D: ckt.D.length }
    out.elements = {
      R: ckt.R.map((e) => [e.name, e.n1, e.n2, e.R]),
      C: ckt.C.map((e) => [e.name, e.n1, e.n2, e.C]),
      L: ckt.L.map((e) => [e.name, e.n1, e.n2, e.L]),
      V: ckt.V.map((e) => [e.name, e.n1, e.n2, e.dc, e.waveform ? 1 : 0, e.index]),
      S: ckt.S.map((e) => [e.name, e.n1, e.n2, e.ncPos, e.ncNeg, e.model.Ron, e.model.Roff, e.model.Von, e.model.Voff]),
      D: ckt.D.map((e) => [e.name, e.nPlus, e.nMinus, e.model.Is, e.model.N]),
    }
    out.tranSpec = ckt.analyses.tran
    out.runs = []
    for (let r = 0; r < repeat; r++) {
      const t0 = Date.now()
      const res = simulateTRAN(ckt)
      const ms = Date.now() - t0
      if (!res) {
        out.runs.push(null)
        continue
      }
      const run = {
        ms,
        keysV: Object.keys(res.nodeVoltages),
        keysI: Object.keys(res.elementCurrents),
        times: res.times,
        V: res.nodeVoltages,
        // JSON cannot carry Infinity/NaN: encode non-finite currents as strings
        I: {},
        state: {
          C_vPrev: ckt.C.map((e) => e.vPrev),
          L_iPrev: ckt.L.map((e) => e.iPrev),
          D_vdPrev: ckt.D.map((e) => e.vdPrev),
          S_isOn: ckt.S.map((e) => (e.isOn ? 1 : 0)),
        },
      }
      for (const k of Object.keys(res.elementCurrents)) {
        run.I[k] = res.elementCurrents[k].map((v) => (Number.isFinite(v) ? v : String(v)))
      }
      if (r === 0) run.formatted_head = formatTranResult(res).split("\n").slice(0, 4)
      out.runs.push(run)
    }
  } catch (err) {
    out.error = String(err && err.message ? err.message : err)
  }
  fs.writeFileSync(outPath, JSON.stringify(out))
}
main()
