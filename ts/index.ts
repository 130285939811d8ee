// ts/index.ts — the package surface, name for name what the reference's lib/index.ts:1-12 exports.
export { parseNetlist } from "./parseNetlist"
export { simulate } from "./simulate"
export { simulateAC } from "./simulateAC"
export { simulateTRAN } from "./simulateTRAN"
export { formatAcResult, formatTranResult, spiceyTranToVGraphs, eecEngineTranToVGraphs } from "./format"
export type { EecEngineTranResult } from "./format"
export { Complex } from "./Complex"
