// ts/spiceyHip.ts — bun:ffi binding of libspicey_hip.so (include/spicey_hip.h).
//
// This is the reference-side binding a spicey maintainer adds: it replaces the BODY of
// lib/analysis/simulateTRAN.ts:146-238 (the `for step … for iter …` nest) by one blocking FFI call and keeps
// everything around it (parseNetlist, computeEffectiveTimeStep :14-19, waveform closures :67, result keys,
// probe filtering :240-249, state write-back :221-237) in TypeScript.  No numerics live here.
//
// NOT executed in the build container (no Bun there); the same C-ABI is exercised by the Python mirror
// (spicey_amd/lib.py) in the test-suite, and the struct offsets below are generated + checked
// (tools/gen_ts_layout.py, tests/test_ts_layout.py).
import { dlopen, FFIType, ptr, CString, type Pointer } from "bun:ffi"
import { SpiceyDescLayout, SpiceyOptionsLayout } from "./abiLayout"

export const SPICEY_ABI_VERSION = 2
export const SPICEY_OK = 0
export const SPICEY_ERR_SINGULAR = 1
export const SPICEY_ERR_BAD_DESC = 2
export const SPICEY_ERR_HIP = 3
export const SPICEY_ERR_NO_DEVICE = 4
export const SPICEY_ERR_COMPLEX_DIV = 5

const libPath = process.env.SPICEY_HIP_LIB ?? `${import.meta.dir}/../spicey_amd/libspicey_hip.so`

const { symbols: C } = dlopen(libPath, {
  spicey_create: { args: [FFIType.ptr, FFIType.ptr, FFIType.ptr], returns: FFIType.i32 },
  spicey_run: {
    args: [FFIType.ptr, FFIType.i64, FFIType.f64, FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr],
    returns: FFIType.i32,
  },
  spicey_get_state: { args: [FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr], returns: FFIType.i32 },
  // several devices behind one handle: instances block-partitioned over `devices`, results gathered into one buffer
  spicey_create_multi: { args: [FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.i32, FFIType.ptr], returns: FFIType.i32 },
  spicey_run_multi: {
    args: [FFIType.ptr, FFIType.i64, FFIType.f64, FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr],
    returns: FFIType.i32,
  },
  spicey_get_state_multi: { args: [FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr], returns: FFIType.i32 },
  spicey_multi_last_error: { args: [FFIType.ptr], returns: FFIType.ptr },
  spicey_destroy_multi: { args: [FFIType.ptr], returns: FFIType.void },
  spicey_last_error: { args: [FFIType.ptr], returns: FFIType.ptr },
  spicey_last_solve_count: { args: [FFIType.ptr], returns: FFIType.i64 },
  // diagnostics (SpiceyOptions.diagnostics bit 0): solves in which the reference's `|f| < EPS` row-update skip may bite
  spicey_last_skip_risk: { args: [FFIType.ptr, FFIType.ptr], returns: FFIType.i64 },
  spicey_destroy: { args: [FFIType.ptr], returns: FFIType.void },
  spicey_version: { args: [], returns: FFIType.ptr },
  spicey_ac_create: { args: [FFIType.ptr, FFIType.ptr, FFIType.ptr], returns: FFIType.i32 },
  spicey_ac_run: { args: [FFIType.ptr, FFIType.i64, FFIType.ptr, FFIType.ptr, FFIType.ptr, FFIType.ptr], returns: FFIType.i32 },
  spicey_ac_last_error: { args: [FFIType.ptr], returns: FFIType.ptr },
  spicey_ac_destroy: { args: [FFIType.ptr], returns: FFIType.void },
})

/** SoA view of one ParsedCircuit (parseNetlist.ts:85-105); node ids are the reference's (0 = ground). */
export type FlatCircuit = {
  nNodes: number
  R: { n1: Int32Array; n2: Int32Array; val: Float64Array }
  C: { n1: Int32Array; n2: Int32Array; val: Float64Array; vPrev: Float64Array }
  L: { n1: Int32Array; n2: Int32Array; val: Float64Array; iPrev: Float64Array }
  V: { n1: Int32Array; n2: Int32Array }
  S: {
    n1: Int32Array; n2: Int32Array; cp: Int32Array; cn: Int32Array
    ron: Float64Array; roff: Float64Array; von: Float64Array; voff: Float64Array; isOn: Int32Array
  }
  D: { np: Int32Array; nm: Int32Array; is: Float64Array; n: Float64Array; vdPrev: Float64Array }
  /** node ids (1-based) to record, in this order; absent / empty = every node in id order (SpiceyDesc.out_nodes) */
  outNodes?: Int32Array
}

// bun:ffi cannot take a pointer to an empty TypedArray; zero-length arrays are passed as NULL
const P = (a: ArrayBufferView): bigint => (a.byteLength === 0 ? 0n : BigInt(ptr(a as any) as unknown as number))

function packDesc(f: FlatCircuit): { buf: ArrayBuffer; keep: ArrayBufferView[] } {
  const L = SpiceyDescLayout.fields
  const buf = new ArrayBuffer(SpiceyDescLayout.size)
  const dv = new DataView(buf)
  const i32 = (k: keyof typeof L, v: number) => dv.setInt32(L[k].offset, v, true)
  const p64 = (k: keyof typeof L, a: ArrayBufferView) => dv.setBigUint64(L[k].offset, P(a), true)
  i32("abi_version", SPICEY_ABI_VERSION)
  i32("n_nodes", f.nNodes)
  i32("n_inst", 1)
  i32("nR", f.R.n1.length); i32("nC", f.C.n1.length); i32("nL", f.L.n1.length)
  i32("nV", f.V.n1.length); i32("nS", f.S.n1.length); i32("nD", f.D.np.length)
  p64("R_n1", f.R.n1); p64("R_n2", f.R.n2); p64("R_val", f.R.val)
  p64("C_n1", f.C.n1); p64("C_n2", f.C.n2); p64("C_val", f.C.val); p64("C_vprev", f.C.vPrev)
  p64("L_n1", f.L.n1); p64("L_n2", f.L.n2); p64("L_val", f.L.val); p64("L_iprev", f.L.iPrev)
  p64("V_n1", f.V.n1); p64("V_n2", f.V.n2)
  p64("S_n1", f.S.n1); p64("S_n2", f.S.n2); p64("S_cp", f.S.cp); p64("S_cn", f.S.cn)
  p64("S_ron", f.S.ron); p64("S_roff", f.S.roff); p64("S_von", f.S.von); p64("S_voff", f.S.voff); p64("S_ison", f.S.isOn)
  p64("D_np", f.D.np); p64("D_nm", f.D.nm); p64("D_is", f.D.is); p64("D_n", f.D.n); p64("D_vdprev", f.D.vdPrev)
  // .PRINT probes are applied on the device: only the probed columns are written and copied back (the reference
  // computes every node and filters afterwards, simulateTRAN.ts:240-249)
  const outNodes = f.outNodes && f.outNodes.length ? f.outNodes : new Int32Array(0)
  i32("n_out", outNodes.length)
  p64("out_nodes", outNodes)
  const keep = [f.R.n1, f.R.n2, f.R.val, f.C.n1, f.C.n2, f.C.val, f.C.vPrev, f.L.n1, f.L.n2, f.L.val, f.L.iPrev, f.V.n1, f.V.n2,
    f.S.n1, f.S.n2, f.S.cp, f.S.cn, f.S.ron, f.S.roff, f.S.von, f.S.voff, f.S.isOn, f.D.np, f.D.nm, f.D.is, f.D.n, f.D.vdPrev, outNodes]
  return { buf, keep }
}

function lastError(h: Pointer | null): string {
  const p = C.spicey_last_error(h)
  return p ? new CString(p).toString() : ""
}

export type NativeTranResult = {
  outV: Float64Array // [steps+1][nOut], nOut = outNodes.length or nNodes
  outI: Float64Array // [steps+1][nCur], order R, C, L, V, S, D
  iters: Int32Array // [steps+1]
  state: { vPrev: Float64Array; iPrev: Float64Array; vdPrev: Float64Array; isOn: Int32Array }
  /** (solve, column) pairs in which the stamped matrix had a nonzero entry below 1e-15 x its column's largest: where the
   *  reference's `if (Math.abs(f) < EPS) continue` (solveReal.ts:45) drops a row update this solver performs; 0 = none */
  skipRisk: number
}

/** One transient run on the GPU.  Throws Error("Singular matrix (real)") like solveReal.ts:28. */
export function runTransientNative(f: FlatCircuit, steps: number, dt: number, srcTable: Float64Array): NativeTranResult {
  const { buf, keep } = packDesc(f)
  const opt = new ArrayBuffer(SpiceyOptionsLayout.size) // zeros: device 0, auto geometry
  new DataView(opt).setInt32(SpiceyOptionsLayout.fields.want_currents.offset, 1, true)
  new DataView(opt).setInt32(SpiceyOptionsLayout.fields.diagnostics.offset, 1, true)
  const hOut = new BigUint64Array(1)
  let rc = C.spicey_create(ptr(buf), ptr(opt), ptr(hOut))
  void keep
  if (rc !== SPICEY_OK) throw new Error(`spicey_create failed (${rc}): ${lastError(null)}`)
  const h = Number(hOut[0]) as unknown as Pointer
  try {
    const nCur = f.R.n1.length + f.C.n1.length + f.L.n1.length + f.V.n1.length + f.S.n1.length + f.D.np.length
    const nOut = f.outNodes && f.outNodes.length ? f.outNodes.length : f.nNodes
    const outV = new Float64Array((steps + 1) * Math.max(nOut, 1))
    const outI = new Float64Array((steps + 1) * Math.max(nCur, 1))
    const iters = new Int32Array(steps + 1)
    rc = C.spicey_run(h, BigInt(steps), dt, srcTable.length ? ptr(srcTable) : null, ptr(outV), ptr(outI), ptr(iters))
    if (rc === SPICEY_ERR_SINGULAR) throw new Error("Singular matrix (real)")
    if (rc !== SPICEY_OK) throw new Error(`spicey_run failed (${rc}): ${lastError(h)}`)
    const state = {
      vPrev: new Float64Array(f.C.n1.length), iPrev: new Float64Array(f.L.n1.length),
      vdPrev: new Float64Array(f.D.np.length), isOn: new Int32Array(f.S.n1.length),
    }
    rc = C.spicey_get_state(h, state.vPrev.length ? ptr(state.vPrev) : null, state.iPrev.length ? ptr(state.iPrev) : null,
      state.vdPrev.length ? ptr(state.vdPrev) : null, state.isOn.length ? ptr(state.isOn) : null)
    if (rc !== SPICEY_OK) throw new Error(`spicey_get_state failed (${rc}): ${lastError(h)}`)
    const skipRisk = Number(C.spicey_last_skip_risk(h, null))
    return { outV, outI, iters, state, skipRisk }
  } finally {
    C.spicey_destroy(h)
  }
}

export type NativeAcResult = {
  outV: Float64Array // [nFreq][nNodes][2] (re, im)
  outI: Float64Array // [nFreq][nR+nC+nL+nV][2], order R, C, L, V
}

/** One AC sweep on the GPU: every frequency is an independent complex solve (simulateAC.ts:80-126), one launch.
 *  vph = [nV][2] source phasors.  Throws the reference's Error messages (solveComplex.ts:28, Complex.ts:42). */
export function runAcNative(f: FlatCircuit, freqs: Float64Array, vph: Float64Array): NativeAcResult {
  const { buf, keep } = packDesc(f)
  const opt = new ArrayBuffer(SpiceyOptionsLayout.size)
  const hOut = new BigUint64Array(1)
  let rc = C.spicey_ac_create(ptr(buf), ptr(opt), ptr(hOut))
  void keep
  if (rc !== SPICEY_OK) {
    const p = C.spicey_ac_last_error(null)
    throw new Error(`spicey_ac_create failed (${rc}): ${p ? new CString(p).toString() : ""}`)
  }
  const h = Number(hOut[0]) as unknown as Pointer
  try {
    const nCur = f.R.n1.length + f.C.n1.length + f.L.n1.length + f.V.n1.length
    const outV = new Float64Array(freqs.length * Math.max(f.nNodes, 1) * 2)
    const outI = new Float64Array(freqs.length * Math.max(nCur, 1) * 2)
    rc = C.spicey_ac_run(h, BigInt(freqs.length), freqs.length ? ptr(freqs) : null, vph.length ? ptr(vph) : null, ptr(outV), ptr(outI))
    if (rc === SPICEY_ERR_SINGULAR) throw new Error("Singular matrix (complex)")
    if (rc === SPICEY_ERR_COMPLEX_DIV) throw new Error("Complex divide by ~0")
    if (rc !== SPICEY_OK) {
      const p = C.spicey_ac_last_error(h)
      throw new Error(`spicey_ac_run failed (${rc}): ${p ? new CString(p).toString() : ""}`)
    }
    return { outV, outI }
  } finally {
    C.spicey_ac_destroy(h)
  }
}

export function nativeVersion(): string {
  const p = C.spicey_version()
  return p ? new CString(p).toString() : ""
}
