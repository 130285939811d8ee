// ts/numbers.ts — SPICE numbers with unit suffixes (reference behaviour: lib/parsing/parseNumberWithUnits.ts:1-30, pinned
// by tests/golden/parser_cases.json).  Quirks kept on purpose: one trailing unit word ohm | v | a | s | h | f is stripped
// BEFORE the multiplier is looked up, so "1f" and "100f" lose their femto; "meg" is the only multi-letter multiplier;
// anything unknown ("1mil", "1k5") keeps the bare mantissa; non-numbers are NaN.
const PLAIN = /^[+-]?\d*\.?\d+(?:[eE][+-]?\d+)?$/
const WITH_LETTERS = /^([+-]?\d*\.?\d+(?:[eE][+-]?\d+)?)([a-zA-Z]+)$/
const MULTIPLIER: Record<string, number> = { t: 1e12, g: 1e9, meg: 1e6, k: 1e3, m: 1e-3, u: 1e-6, n: 1e-9, p: 1e-12, f: 1e-15 }

export function parseNumberWithUnits(raw: unknown): number {
  if (raw == null) return NaN
  const text = String(raw).trim()
  if (text === "") return NaN
  if (PLAIN.test(text)) return parseFloat(text)
  const parts = WITH_LETTERS.exec(text)
  if (!parts) return parseFloat(text)
  const mantissa = parseFloat(parts[1]!)
  const letters = parts[2]!.toLowerCase().replace(/(ohm|v|a|s|h|f)$/, "")
  const scale = letters === "meg" || letters.length === 1 ? MULTIPLIER[letters] : undefined
  return scale === undefined ? mantissa : mantissa * scale
}
