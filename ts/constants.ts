// ts/constants.ts — the two constants of the hot path (reference: lib/constants/EPS.ts:1, lib/constants/physics.ts:1).
// The native side carries the same values (spicey_amd/csrc/program.h: SPICEY_EPS, SPICEY_VT300).
export const EPS = 1e-15
export const VT_300K = 0.02585
