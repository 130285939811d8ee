// stand-in for the reference's lib/utils/logspace.ts (decade grid, overshooting last point kept, stop frequency
// appended when the grid falls short)
export function logspace(f1, f2, pointsPerDecade) {
  if (f1 <= 0 || f2 <= 0) throw new Error(".ac frequencies must be > 0")
  if (f2 < f1) { const t = f1; f1 = f2; f2 = t }
  const n = Math.max(1, Math.ceil(Math.log10(f2 / f1) * pointsPerDecade))
  const out = []
  for (let i = 0; i <= n; i++) out.push(f1 * Math.pow(10, i / pointsPerDecade))
  if (out[out.length - 1] < f2 * (1 - 1e-15)) out.push(f2)
  return out
}
