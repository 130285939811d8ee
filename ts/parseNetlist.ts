// ts/parseNetlist.ts — netlist text -> ParsedCircuit: the parser of the drop-in surface, written for this build.
//
// Behaviour contract = the reference's parser (lib/parsing/parseNetlist.ts:123-481) as SURVEY.md Appendix D lists it and as
// tests/golden/parser_cases.json pins it (101 snippets run through the reference itself: structures, waveform samples,
// every Error text).  Structure is this build's own: a line scanner, one table of directive cards, one table of element
// letters.  It runs once per netlist (microseconds to milliseconds) and is not part of the accelerated path.
//
// Quirks that callers depend on (all pinned by the fixture):
//   * the FIRST line that is neither blank, a `*` comment, a directive nor starts with one of r c l v g s m i q d is the
//     title and vanishes — wherever it stands; a title that does start with such a letter is parsed as an element
//   * `//` and `;` start a comment anywhere in a line; parsing stops at `.end`; `+` continuation lines are not joined
//   * ground is only the node literally called "0"; node names are case-insensitive, first spelling wins
//   * sources: a leading bare number is the DC value; `dc v`, `ac mag [phase]`, `PULSE(7 or 8 numbers)`, `PWL(t v ...)`;
//     anything else (SIN(...), ...) is skipped word by word; the waveform wins over DC in a transient
//   * g m i q lines, diode lines that do not have exactly 4 tokens, unknown cards and `.print` of anything but `tran`
//     land in `skipped`; `.tran` reads two numbers and ignores the rest (uic ...)
//   * errors of an element line are wrapped as `Parse error on line: "<line>"\n<cause>`; errors of cards are not
import { NodeIndex } from "./NodeIndex"
import { parseNumberWithUnits } from "./numbers"
import { parsePulseArgs, parsePwlArgs, pulseValue, pwlValue } from "./waveforms"
import type { ParsedCircuit, ParsedDiodeModel, ParsedVoltageSource, ParsedVSwitchModel } from "./types"

type Card = (words: string[], ckt: ParsedCircuit, line: string) => void
type ElementLine = (words: string[], ckt: ParsedCircuit, line: string) => void

// one token: a quoted string, NAME(...) with its parenthesis, a bare parenthesis group, or a run of non-blanks
function tokenize(line: string): string[] {
  return line.match(/"[^"]*"|\w+\s*\([^)]*\)|\([^()]*\)|\S+/g) || []
}

function word(words: string[], at: number, complaint: string): string {
  const w = words[at]
  if (w == null) throw new Error(complaint)
  return w
}

// ---- .model NAME TYPE(k=v ...) | TYPE (k=v ...) | TYPE k=v ... -----------------------------------------------------------
function modelAssignments(words: string[]): { kind: string; pairs: Array<[string, number]> } {
  let kind = word(words, 2, ".model missing type")
  let inside = ""
  const open = kind.indexOf("(")
  if (open >= 0) {
    inside = kind.slice(open + 1)
    kind = kind.slice(0, open)
  }
  const rest = words.slice(3).join(" ")
  let text = inside ? `${inside} ${rest.replace(/\)$/, "")}`.trim() : rest.replace(/^\(/, "").replace(/\)$/, "")
  text = text.replace(/^\(/, "").replace(/\)$/, "").trim()
  const pairs: Array<[string, number]> = []
  if (text.length > 0)
    for (const item of text.split(/[\s,]+/)) {
      if (!item) continue
      const sides = item.split("=")
      if (!sides[0] || sides.length < 2) continue
      const value = parseNumberWithUnits(sides[1])
      if (!Number.isNaN(value)) pairs.push([sides[0].toLowerCase(), value])
    }
  return { kind, pairs }
}

const CARDS: Record<string, Card> = {
  ".ac": (words, ckt) => {
    const mode = word(words, 1, ".ac missing mode").toLowerCase()
    if (mode !== "dec" && mode !== "lin") throw new Error(".ac supports 'dec' or 'lin'")
    const N = parseInt(word(words, 2, ".ac missing point count"), 10)
    const f1 = parseNumberWithUnits(word(words, 3, ".ac missing start frequency"))
    const f2 = parseNumberWithUnits(word(words, 4, ".ac missing stop frequency"))
    ckt.analyses.ac = { mode, N, f1, f2 }
  },
  ".tran": (words, ckt) => {
    const dt = parseNumberWithUnits(word(words, 1, ".tran missing timestep"))
    const tstop = parseNumberWithUnits(word(words, 2, ".tran missing stop time"))
    ckt.analyses.tran = { dt, tstop }
  },
  ".print": (words, ckt, line) => {
    if (word(words, 1, ".print missing analysis type").toLowerCase() !== "tran") {
      ckt.skipped.push(line)
      return
    }
    for (const w of words.slice(2)) {
      const probe = /^v\(([^)]+)\)$/i.exec(w)
      if (!probe || !probe[1]) continue
      const name = probe[1]
      if (!ckt.probes.tran.some((p) => p.toUpperCase() === name.toUpperCase())) ckt.probes.tran.push(name)
    }
  },
  ".model": (words, ckt, line) => {
    const name = word(words, 1, ".model missing name")
    const { kind, pairs } = modelAssignments(words)
    const k = kind.toLowerCase()
    if (k === "vswitch" || k === "sw") {
      const m: ParsedVSwitchModel = { name, Ron: 1, Roff: 1e12, Von: 0, Voff: 0 }
      let vt: number | undefined
      let vh: number | undefined
      for (const [key, value] of pairs) {
        if (key === "ron") m.Ron = value
        else if (key === "roff") m.Roff = value
        else if (key === "von") m.Von = value
        else if (key === "voff") m.Voff = value
        else if (key === "vt") vt = value
        else if (key === "vh") vh = value
      }
      if (vt !== undefined) {
        const h = vh === undefined ? 0 : vh
        m.Von = vt + h / 2
        m.Voff = vt - h / 2
      }
      ckt.models.vswitch.set(name.toLowerCase(), m)
    } else if (k === "d") {
      const m: ParsedDiodeModel = { name, Is: 1e-14, N: 1 }
      for (const [key, value] of pairs) {
        if (key === "is") m.Is = value
        else if (key === "n") m.N = value
      }
      ckt.models.diode.set(name.toLowerCase(), m)
    } else {
      ckt.skipped.push(line)
    }
  },
}

// ---- element lines ------------------------------------------------------------------------------------------------------
function twoNodes(words: string[], ckt: ParsedCircuit, what: string): [number, number] {
  const a = ckt.nodes.getOrCreate(word(words, 1, `${what} missing node`))
  const b = ckt.nodes.getOrCreate(word(words, 2, `${what} missing node`))
  return [a, b]
}

function sourceLine(words: string[], ckt: ParsedCircuit): void {
  const [n1, n2] = twoNodes(words, ckt, "Voltage source")
  const vs: ParsedVoltageSource = { name: words[0]!, n1, n2, dc: 0, acMag: 0, acPhaseDeg: 0, waveform: null, index: -1 }
  let at = 3
  if (at < words.length && !/^[a-zA-Z]/.test(words[at]!)) vs.dc = parseNumberWithUnits(words[at++])
  while (at < words.length) {
    const key = words[at]!.toLowerCase()
    if (key === "dc") {
      vs.dc = parseNumberWithUnits(word(words, at + 1, "DC value missing"))
      at += 2
    } else if (key === "ac") {
      vs.acMag = parseNumberWithUnits(word(words, at + 1, "AC magnitude missing"))
      const phase = words[at + 2]
      if (phase != null && /^[+-]?\d/.test(phase)) {
        vs.acPhaseDeg = parseNumberWithUnits(phase)
        at += 3
      } else at += 2
    } else if (key.startsWith("pulse") || key.startsWith("pwl")) {
      const pulse = key.startsWith("pulse")
      const label = pulse ? "PULSE" : "PWL"
      const own = key.includes("(")
      const spec = own ? key : word(words, at + 1, `${label}() missing arguments`)
      if (!spec || !/\(.*\)/.test(spec)) throw new Error(`Malformed ${label}() specification`)
      if (pulse) {
        const p = parsePulseArgs(spec)
        vs.waveform = (t: number) => pulseValue(p, t)
      } else {
        const points = parsePwlArgs(spec)
        vs.waveform = (t: number) => pwlValue(points, t)
      }
      at += own ? 1 : 2
    } else at += 1
  }
  ckt.V.push(vs)
}

const ELEMENTS: Record<string, ElementLine> = {
  r: (words, ckt) => {
    const [n1, n2] = twoNodes(words, ckt, "Resistor")
    ckt.R.push({ name: words[0]!, n1, n2, R: parseNumberWithUnits(word(words, 3, "Resistor missing value")) })
  },
  c: (words, ckt) => {
    const [n1, n2] = twoNodes(words, ckt, "Capacitor")
    ckt.C.push({ name: words[0]!, n1, n2, C: parseNumberWithUnits(word(words, 3, "Capacitor missing value")), vPrev: 0 })
  },
  l: (words, ckt) => {
    const [n1, n2] = twoNodes(words, ckt, "Inductor")
    ckt.L.push({ name: words[0]!, n1, n2, L: parseNumberWithUnits(word(words, 3, "Inductor missing value")), iPrev: 0 })
  },
  v: sourceLine,
  s: (words, ckt) => {
    const [n1, n2] = twoNodes(words, ckt, "Switch")
    const ncPos = ckt.nodes.getOrCreate(word(words, 3, "Switch missing control node"))
    const ncNeg = ckt.nodes.getOrCreate(word(words, 4, "Switch missing control node"))
    const modelName = word(words, 5, "Switch missing model").toLowerCase()
    ckt.S.push({ name: words[0]!, n1, n2, ncPos, ncNeg, modelName, model: null, isOn: false })
  },
  d: (words, ckt, line) => {
    if (words.length !== 4) {
      ckt.skipped.push(line)
      return
    }
    const [nPlus, nMinus] = twoNodes(words, ckt, "Diode")
    ckt.D.push({ name: words[0]!, nPlus, nMinus, modelName: word(words, 3, "Diode missing model").toLowerCase(), model: null, vdPrev: 0 })
  },
}

export function parseNetlist(text: string): ParsedCircuit {
  const ckt: ParsedCircuit = {
    nodes: new NodeIndex(),
    R: [], C: [], L: [], V: [], S: [], D: [],
    analyses: { ac: null, tran: null },
    probes: { tran: [] },
    skipped: [],
    models: { vswitch: new Map(), diode: new Map() },
  }
  let titleSeen = false
  for (const physical of text.split(/\r?\n/)) {
    let line = physical.trim()
    if (!line || line.startsWith("*")) continue
    if (/^\s*\.end\b/i.test(line)) break
    line = line.replace(/\/\/.*$/, "").replace(/;.*$/, "")
    const words = tokenize(line)
    const head = words[0]
    if (!head) continue
    if (head.startsWith(".")) {
      const card = CARDS[head.toLowerCase()]
      if (card) card(words, ckt, line)
      else ckt.skipped.push(line)
      continue
    }
    if (!titleSeen && !/^[rclvgsmiqd]\w*$/i.test(head)) {
      titleSeen = true
      continue
    }
    const handler = ELEMENTS[head[0]!.toLowerCase()]
    if (!handler) {
      ckt.skipped.push(line)
      continue
    }
    try {
      handler(words, ckt, line)
    } catch (err) {
      const cause = err instanceof Error ? err.message : String(err)
      throw new Error(`Parse error on line: "${line}"\n${cause}`)
    }
  }
  // sources take the unknowns behind the nodes; switches and diodes are bound to their models (which may follow them)
  const nodeUnknowns = ckt.nodes.count() - 1
  ckt.V.forEach((vs, k) => (vs.index = nodeUnknowns + k))
  for (const sw of ckt.S) {
    const m = ckt.models.vswitch.get(sw.modelName)
    if (!m) throw new Error(`Unknown .model ${sw.modelName} referenced by switch ${sw.name}`)
    sw.model = m
    sw.isOn = false
  }
  for (const d of ckt.D) {
    const m = ckt.models.diode.get(d.modelName)
    if (!m) throw new Error(`Unknown .model ${d.modelName} referenced by diode ${d.name}`)
    d.model = m
  }
  return ckt
}

export type {
  ParsedCircuit, ParsedACAnalysis, ParsedTranAnalysis, ParsedResistor, ParsedCapacitor, ParsedInductor, ParsedDiode,
  ParsedDiodeModel, ParsedVoltageSource, ParsedVSwitchModel, ParsedSwitch, CircuitNodeIndex,
} from "./types"
