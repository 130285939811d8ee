// TEST INFRASTRUCTURE ONLY — the reference's own spiceyTranToVGraphs / eecEngineTranToVGraphs
// (lib/formatting/formatToVGraph.ts, type-erased) on a netlist, dumped as JSON.
//   node --harmony-nullish --harmony-optional-chaining driver_vgraph.mjs <erased_root> <netlist.cir> <out.json>
import fs from "fs"
import path from "path"
import { pathToFileURL } from "url"

const [, , root, netlistPath, outPath] = process.argv
const imp = (rel) => import(pathToFileURL(path.join(root, rel)).href)
const main = async () => {
  const { parseNetlist } = await imp("lib/parsing/parseNetlist.mjs")
  const { simulateTRAN } = await imp("lib/analysis/simulateTRAN.mjs")
  const { spiceyTranToVGraphs, eecEngineTranToVGraphs } = await imp("lib/formatting/formatToVGraph.mjs")
  const ckt = parseNetlist(fs.readFileSync(netlistPath, "utf8"))
  const res = simulateTRAN(ckt)
  const out = { graphs: spiceyTranToVGraphs(res, ckt, "exp_1") }
  out.eec = eecEngineTranToVGraphs({ time_s: [0, 1e-3, 2.5e-3], voltages: { out: [0, 1.5, 3.25], "2": [1, 2, 3] } }, ckt, "exp_2")
  out.empty = spiceyTranToVGraphs(null, ckt, "x").length
  fs.writeFileSync(outPath, JSON.stringify(out))
}
main()
