// ts/types.ts — the data model between parser and analyses: what parseNetlist returns and simulateTRAN / simulateAC read
// and (state fields) write.  Field names and meanings are the reference's (lib/parsing/parseNetlist.ts:12-105,
// lib/types/simulation.ts:1-10): they ARE the drop-in surface — callers and the formatters index these objects by name.
import type { NodeIndex } from "./NodeIndex"

export type PulseSpec = { v1: number; v2: number; td: number; tr: number; tf: number; ton: number; period: number; ncycles: number }
export type PwlPoint = { t: number; v: number }
export type Waveform = ((t: number) => number) | null

export type ParsedResistor = { name: string; n1: number; n2: number; R: number }
export type ParsedCapacitor = { name: string; n1: number; n2: number; C: number; vPrev: number }
export type ParsedInductor = { name: string; n1: number; n2: number; L: number; iPrev: number }
export type ParsedVoltageSource = {
  name: string; n1: number; n2: number
  dc: number; acMag: number; acPhaseDeg: number
  waveform: Waveform
  index: number
}
export type ParsedVSwitchModel = { name: string; Ron: number; Roff: number; Von: number; Voff: number }
export type ParsedDiodeModel = { name: string; Is: number; N: number }
export type ParsedSwitch = {
  name: string; n1: number; n2: number; ncPos: number; ncNeg: number
  modelName: string; model: ParsedVSwitchModel | null
  isOn: boolean
}
export type ParsedDiode = { name: string; nPlus: number; nMinus: number; modelName: string; model: ParsedDiodeModel | null; vdPrev: number }
export type ParsedACAnalysis = { mode: "dec" | "lin"; N: number; f1: number; f2: number } | null
export type ParsedTranAnalysis = { dt: number; tstop: number } | null
export type CircuitNodeIndex = NodeIndex
export type ParsedCircuit = {
  nodes: CircuitNodeIndex
  R: ParsedResistor[]; C: ParsedCapacitor[]; L: ParsedInductor[]; V: ParsedVoltageSource[]; S: ParsedSwitch[]; D: ParsedDiode[]
  analyses: { ac: ParsedACAnalysis; tran: ParsedTranAnalysis }
  probes: { tran: string[] }
  skipped: string[]
  models: { vswitch: Map<string, ParsedVSwitchModel>; diode: Map<string, ParsedDiodeModel> }
}
