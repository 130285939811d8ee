// tests/node/run_parser.mjs — TEST INFRASTRUCTURE: this build's own TypeScript parser (ts/parseNetlist.ts, type-erased by
// tools/node_shim/erase_own_ts.py) under Node 12 on a list of netlist texts; dumps for every text the parsed structure
// (elements, models, analyses, probes, skipped lines, source waveforms sampled at fixed times) or the Error message, in the
// layout of tests/golden/parser_cases.json (which holds the REFERENCE's parser's answers to the same texts).
//   node --harmony-nullish --harmony-optional-chaining run_parser.mjs <erased_dir> <cases.json> <out.json>
import fs from "fs"
import path from "path"
import { pathToFileURL } from "url"

const [, , erased, casesPath, outPath] = process.argv
const main = async () => {
  const { parseNetlist } = await import(pathToFileURL(path.join(erased, "parseNetlist.mjs")).href)
  const input = JSON.parse(fs.readFileSync(casesPath, "utf8"))
  const num = (x) => (typeof x === "number" && !Number.isFinite(x) ? String(x) : x)
  const ts = input.ts
  const results = input.cases.map((text) => {
    try {
      const c = parseNetlist(text)
      return {
        nodes: c.nodes.rev,
        R: c.R.map((e) => [e.name, e.n1, e.n2, num(e.R)]),
        C: c.C.map((e) => [e.name, e.n1, e.n2, num(e.C), num(e.vPrev)]),
        L: c.L.map((e) => [e.name, e.n1, e.n2, num(e.L), num(e.iPrev)]),
        V: c.V.map((e) => [e.name, e.n1, e.n2, num(e.dc), num(e.acMag), num(e.acPhaseDeg), e.index, e.waveform ? ts.map((t) => num(e.waveform(t))) : null]),
        S: c.S.map((e) => [e.name, e.n1, e.n2, e.ncPos, e.ncNeg, e.modelName, e.isOn, e.model ? [e.model.name, num(e.model.Ron), num(e.model.Roff), num(e.model.Von), num(e.model.Voff)] : null]),
        D: c.D.map((e) => [e.name, e.nPlus, e.nMinus, e.modelName, num(e.vdPrev), e.model ? [e.model.name, num(e.model.Is), num(e.model.N)] : null]),
        analyses: JSON.parse(JSON.stringify(c.analyses)),
        probes: c.probes,
        skipped: c.skipped,
        count: c.nodes.count(), ground: c.nodes.get("0"), row: c.nodes.matrixIndexOfNode(c.nodes.count() - 1),
      }
    } catch (e) {
      return { error: String(e && e.message ? e.message : e) }
    }
  })
  fs.writeFileSync(outPath, JSON.stringify({ results }))
}
main()
