// ts/logspace.ts — the frequency grid of `.ac dec N f1 f2` (reference behaviour: lib/utils/logspace.ts:3-15, pinned by the
// 201-line snapshot of tests/basics/basics01.test.ts): N points per decade from the lower to the upper frequency, the last
// grid point may overshoot; the stop frequency is appended when the grid ends below it.  Math.pow is engine-defined, which
// is why the grid is built on the host and handed to the native sweep as a list.
import { EPS } from "./constants"

export function logspace(f1: number, f2: number, pointsPerDecade: number): number[] {
  if (f1 <= 0 || f2 <= 0) throw new Error(".ac frequencies must be > 0")
  const lo = Math.min(f1, f2)
  const hi = f2 < f1 ? f1 : f2
  const steps = Math.max(1, Math.ceil(Math.log10(hi / lo) * pointsPerDecade))
  const grid: number[] = []
  for (let i = 0; i <= steps; i++) grid.push(lo * Math.pow(10, i / pointsPerDecade))
  if (grid[grid.length - 1]! < hi * (1 - EPS)) grid.push(hi)
  return grid
}
