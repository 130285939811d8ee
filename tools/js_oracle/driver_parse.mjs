// TEST INFRASTRUCTURE ONLY — the reference's own parseNetlist (lib/parsing/*, type-erased) on a list of netlist
// snippets: dumps the parsed structure (or the Error message) of each.
//   node --harmony-nullish --harmony-optional-chaining driver_parse.mjs <erased_root> <cases.json> <out.json>
import fs from "fs"
import path from "path"
import { pathToFileURL } from "url"

const [, , root, casesPath, outPath] = process.argv
const main = async () => {
  const { parseNetlist } = await import(pathToFileURL(path.join(root, "lib/parsing/parseNetlist.mjs")).href)
  const cases = JSON.parse(fs.readFileSync(casesPath, "utf8"))
  const enc = (x) => (typeof x === "number" && !Number.isFinite(x) ? String(x) : x)
  const ts = [0, 1e-9, 5e-7, 1e-6, 2.5e-6, 1e-5, 3.3e-5, 1e-4, 1e-3, 0.0123, 1]
  const out = cases.map((text) => {
    try {
      const c = parseNetlist(text)
      return {
        nodes: c.nodes.rev,
        R: c.R.map((e) => [e.name, e.n1, e.n2, enc(e.R)]),
        C: c.C.map((e) => [e.name, e.n1, e.n2, enc(e.C), enc(e.vPrev)]),
        L: c.L.map((e) => [e.name, e.n1, e.n2, enc(e.L), enc(e.iPrev)]),
        V: c.V.map((e) => [e.name, e.n1, e.n2, enc(e.dc), enc(e.acMag), enc(e.acPhaseDeg), e.index, e.waveform ? ts.map((t) => enc(e.waveform(t))) : null]),
        S: c.S.map((e) => [e.name, e.n1, e.n2, e.ncPos, e.ncNeg, e.modelName, e.isOn, e.model ? [e.model.name, enc(e.model.Ron), enc(e.model.Roff), enc(e.model.Von), enc(e.model.Voff)] : null]),
        D: c.D.map((e) => [e.name, e.nPlus, e.nMinus, e.modelName, enc(e.vdPrev), e.model ? [e.model.name, enc(e.model.Is), enc(e.model.N)] : null]),
        analyses: JSON.parse(JSON.stringify(c.analyses)),
        probes: c.probes,
        skipped: c.skipped,
      }
    } catch (e) {
      return { error: String(e && e.message ? e.message : e) }
    }
  })
  fs.writeFileSync(outPath, JSON.stringify({ ts, cases, results: out }))
}
main()
