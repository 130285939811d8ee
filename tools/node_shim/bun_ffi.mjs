// tools/node_shim/bun_ffi.mjs — TEST INFRASTRUCTURE: the subset of Bun's `bun:ffi` module that ts/spiceyHip.ts uses,
// implemented on the N-API addon bunffi.node, so that the TypeScript drop-in layer runs under Node 12.
// Semantics kept from Bun: ptr(view) returns a Number, i32 results are Numbers, i64 results BigInts, pointer results
// Numbers (0 = null), `null` is accepted for pointer arguments.
import { createRequire } from "module"
import { fileURLToPath } from "url"
import path from "path"

const require = createRequire(import.meta.url)
const addon = require(path.join(path.dirname(fileURLToPath(import.meta.url)), "bunffi.node"))

export const FFIType = { ptr: "ptr", i32: "i32", i64: "i64", f64: "f64", void: "void" }

export function ptr(view) {
  return Number(addon.addressOf(view)) + (ArrayBuffer.isView(view) ? 0 : 0)
}

export class CString {
  constructor(p) { this.value = addon.cstring(BigInt(p)) }
  toString() { return this.value }
}

export function dlopen(libPath, decls) {
  const handle = addon.dlopen(libPath)
  const symbols = {}
  for (const name of Object.keys(decls)) {
    const { args, returns } = decls[name]
    const fn = addon.dlsym(handle, name)
    symbols[name] = (...actual) => {
      if (actual.length !== args.length) throw new TypeError(`${name}: expected ${args.length} arguments, got ${actual.length}`)
      const ints = [], dbls = []
      args.forEach((t, i) => {
        const v = actual[i]
        if (t === FFIType.f64) dbls.push(Number(v))
        else ints.push(v == null ? 0n : typeof v === "bigint" ? v : BigInt(Math.trunc(Number(v))))
      })
      const r = addon.call(fn, ints, dbls, returns === FFIType.ptr ? "ptr" : returns)
      if (returns === FFIType.ptr) return Number(r)
      return r
    }
  }
  return { symbols }
}
