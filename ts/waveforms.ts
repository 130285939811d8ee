// ts/waveforms.ts — PULSE / PWL source specifications and their values at time t (reference behaviour:
// lib/parsing/parsePulseArgs.ts, parsePwlArgs.ts, pulseValue.ts:4-22, pwlValue.ts:3-16; SURVEY.md Appendix A5).
// The values are evaluated on the host at t = step * dt and handed to the native solver as a table: closures cannot cross
// the FFI.  The arithmetic below is order-exact (every expression rounds like the reference's).
import { EPS } from "./constants"
import { parseNumberWithUnits } from "./numbers"
import type { PulseSpec, PwlPoint } from "./types"

// "PULSE ( a b, c )" -> ["a", "b", "c"]: keyword and the outer parentheses dropped, blanks and commas both separate
function argumentList(token: string, keyword: RegExp): string[] {
  const inner = token.trim().replace(keyword, "(").replace(/^\(/, "").replace(/\)$/, "").trim()
  return inner.split(/[\s,]+/).filter((piece) => piece.length > 0)
}

export function parsePulseArgs(token: string): PulseSpec {
  const words = argumentList(token, /^pulse\s*\(/i)
  if (words.length < 7) throw new Error("PULSE(...) requires 7 or 8 args")
  const x = words.map((w) => parseNumberWithUnits(w))
  if (x.some((value) => Number.isNaN(value))) throw new Error("Invalid PULSE() numeric value")
  return { v1: x[0]!, v2: x[1]!, td: x[2]!, tr: x[3]!, tf: x[4]!, ton: x[5]!, period: x[6]!, ncycles: words.length > 7 ? x[7]! : Infinity }
}

export function parsePwlArgs(token: string): PwlPoint[] {
  const words = argumentList(token, /^pwl\s*\(/i)
  if (words.length === 0 || words.length % 2 !== 0) throw new Error("PWL(...) requires an even number of time/value pairs")
  const points: PwlPoint[] = []
  for (let i = 0; i < words.length; i += 2) {
    const t = parseNumberWithUnits(words[i])
    const v = parseNumberWithUnits(words[i + 1])
    if (Number.isNaN(t) || Number.isNaN(v)) throw new Error("Invalid PWL() numeric value")
    points.push({ t, v })
  }
  return points
}

export function pulseValue(p: PulseSpec, t: number): number {
  if (t < p.td) return p.v1
  const sinceDelay = t - p.td
  const cycle = Math.floor(sinceDelay / p.period)
  if (cycle >= p.ncycles) return p.v1
  const tc = sinceDelay - cycle * p.period
  if (tc < p.tr) return p.v1 + (p.v2 - p.v1) * (tc / Math.max(p.tr, EPS))
  const plateauEnd = p.tr + p.ton
  if (tc < plateauEnd) return p.v2
  if (tc < p.tr + p.ton + p.tf) return p.v2 + (p.v1 - p.v2) * ((tc - plateauEnd) / Math.max(p.tf, EPS))
  return p.v1
}

export function pwlValue(points: PwlPoint[], t: number): number {
  if (points.length === 0) return 0
  if (t <= points[0]!.t) return points[0]!.v
  for (let i = 1; i < points.length; i++) {
    const from = points[i - 1]!
    const to = points[i]!
    if (t <= to.t) return from.v + (to.v - from.v) * ((t - from.t) / Math.max(to.t - from.t, EPS))
  }
  return points[points.length - 1]!.v
}
