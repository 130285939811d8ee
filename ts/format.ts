// ts/format.ts — result presentation (reference: lib/formatting/formatTranResult.ts, formatAcResult.ts, formatToVGraph.ts):
// the CSV-like text of a transient / AC result (every number Number.prototype.toPrecision(6)) and the circuit-json
// voltage-graph objects.  For very large transients the same text comes out of the native formatter
// (include/spicey_hip.h: spicey_format_tran, pinned on 9 701 values formatted by the JS engine); this is the plain path.
import type { Complex } from "./Complex"
import type { ParsedCircuit } from "./types"

type TranLike = { times: number[]; nodeVoltages: Record<string, number[]> } | null
type AcLike = { freqs: number[]; nodeVoltages: Record<string, Complex[]> } | null
export type EecEngineTranResult = { time_s: number[]; voltages: Record<string, number[]> }
export type TransientVoltageGraph = {
  type: "simulation_transient_voltage_graph"
  simulation_transient_voltage_graph_id: string
  simulation_experiment_id: string
  timestamps_ms: number[]
  voltage_levels: number[]
  time_per_step: number
  start_time_ms: number
  end_time_ms: number
  name: string
}

const six = (x: number) => (+x).toPrecision(6)

export function formatTranResult(tran: TranLike): string {
  if (!tran) return "No TRAN analysis.\n"
  const keys = Object.keys(tran.nodeVoltages)
  const out = [["t(s)", ...keys.map((k) => `${k}:V`)].join(", ")]
  tran.times.forEach((t, step) => {
    if (t == null) return
    const cells = [six(t)]
    for (const k of keys) {
      const v = tran.nodeVoltages[k]?.[step]
      if (v != null) cells.push(six(v))
    }
    out.push(cells.join(", "))
  })
  return out.join("\n")
}

export function formatAcResult(ac: AcLike): string {
  if (!ac) return "No AC analysis.\n"
  const keys = Object.keys(ac.nodeVoltages)
  const out = ["f(Hz), " + keys.map((k) => `${k}:|V|,∠V(deg)`).join(", ")]
  ac.freqs.forEach((f, i) => {
    if (f == null) return
    const cells = [six(f)]
    for (const k of keys) {
      const z = ac.nodeVoltages[k]?.[i]
      if (z) cells.push(`${six(z.abs())},${six(z.phaseDeg())}`)
    }
    out.push(cells.join(", "))
  })
  return out.join("\n")
}

function graphs(timesS: number[], series: Record<string, number[]>, ckt: ParsedCircuit, experiment: string, idTail: string, nameTail: string): TransientVoltageGraph[] {
  if (!ckt.analyses.tran) return []
  const { dt, tstop } = ckt.analyses.tran
  return Object.keys(series).map((node) => ({
    type: "simulation_transient_voltage_graph",
    simulation_transient_voltage_graph_id: `stvg_${experiment}_${node}${idTail}`,
    simulation_experiment_id: experiment,
    timestamps_ms: timesS.map((t) => t * 1000),
    voltage_levels: series[node]!,
    time_per_step: dt * 1000,
    start_time_ms: 0,
    end_time_ms: tstop * 1000,
    name: `V(${node})${nameTail}`,
  }))
}

export function spiceyTranToVGraphs(tran: TranLike, ckt: ParsedCircuit, simulation_experiment_id: string): TransientVoltageGraph[] {
  return tran ? graphs(tran.times, tran.nodeVoltages, ckt, simulation_experiment_id, "", "") : []
}

export function eecEngineTranToVGraphs(res: EecEngineTranResult, ckt: ParsedCircuit, simulation_experiment_id: string): TransientVoltageGraph[] {
  return graphs(res.time_s, res.voltages, ckt, simulation_experiment_id, "_eec", " (ngspice)")
}
