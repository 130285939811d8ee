// TEST INFRASTRUCTURE ONLY — Number.prototype.toPrecision(6) of the JS engine (Node 12 / V8 7.8) on tricky doubles
// -> tests/golden/toprecision6.json.  No reference code involved: this pins the ENGINE semantics that
// lib/formatting/formatTranResult.ts:15,19 and formatAcResult.ts:17,21 rely on (ties go to the larger digit string,
// exponential notation for e < -6 or e >= 6).   Usage: node tools/js_oracle/gen_toprecision6.js tests/golden/toprecision6.json
const vals = []
const push = (x) => { vals.push(x); vals.push(-x) }
;[0, 1, 0.5, 2.5, 100000.5, 100001.5, 999999.5, 999999.4999999999, 1000005, 1234565, 1234575, 12345650, 0.000001, 0.0000012345675, 1e-7, 9.999995e-7, 9.999994999e-7,
  1e21, 1e20, 123456789012345680000, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 0.1, 0.2, 0.3, 1/3, 2/3, 1e5, 1e6, 999999, 1000000, 99999.95, 99999.949999,
  9.999995, 9.9999949999, 0.9999995, 1.0000005, 1.000005, 1.00000500000001, 4.35, 0.000123456789, 123456.5, 1234567.5, 0.015625, 0.0078125, 1.5e-10, 2.5e-7, 3.5e-7].forEach(push)
vals.push(Infinity, -Infinity, NaN)
let s = 12345
const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296 }
for (let i = 0; i < 3000; i++) { const e = Math.floor(rnd() * 60) - 30; push((rnd() * 9 + 1) * Math.pow(10, e)) }
for (let i = 0; i < 600; i++) { const n = Math.floor(rnd() * 9e6) + 1e5; push(n + 0.5); push(n * 10 + 5); push((n * 10 + 5) / 1e9) }   // ties and near-ties
const enc = (x) => (Number.isFinite(x) ? x : String(x))
require("fs").writeFileSync(process.argv[2], JSON.stringify({ engine: process.version, values: vals.map(enc), strings: vals.map((x) => x.toPrecision(6)) }))
