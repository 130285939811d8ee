// ts/simulate.ts — the public entry (reference: lib/analysis/simulate.ts:5-10): parse, AC sweep if the netlist has an .ac
// card, transient if it has a .tran card.  Both analyses run natively (ts/spiceyHip.ts -> libspicey_hip.so).
import { parseNetlist } from "./parseNetlist"
import { simulateAC } from "./simulateAC"
import { simulateTRAN } from "./simulateTRAN"

export function simulate(netlistText: string) {
  const circuit = parseNetlist(netlistText)
  const ac = simulateAC(circuit)
  const tran = simulateTRAN(circuit)
  return { circuit, ac, tran }
}
