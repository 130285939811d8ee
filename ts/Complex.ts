// ts/Complex.ts — the complex value type of the AC results (public export of the package, reference: lib/math/Complex.ts).
// The sweep itself runs natively (include/spicey_hip.h: spicey_ac_run); this class carries its results and keeps the
// arithmetic a caller may do on them, with the reference's guards: dividing or inverting by a value whose squared
// magnitude is below 1e-15 throws.
import { EPS } from "./constants"

export class Complex {
  re: number
  im: number

  constructor(re = 0, im = 0) {
    this.re = re
    this.im = im
  }

  static from(re: number, im = 0): Complex {
    return new Complex(re, im)
  }

  /** magnitude and phase in DEGREES (the `ac mag phase` of a source line) */
  static fromPolar(mag: number, deg = 0): Complex {
    const phase = (deg * Math.PI) / 180
    return new Complex(mag * Math.cos(phase), mag * Math.sin(phase))
  }

  clone(): Complex {
    return new Complex(this.re, this.im)
  }

  add(b: Complex): Complex {
    return new Complex(this.re + b.re, this.im + b.im)
  }

  sub(b: Complex): Complex {
    return new Complex(this.re - b.re, this.im - b.im)
  }

  mul(b: Complex): Complex {
    return new Complex(this.re * b.re - this.im * b.im, this.re * b.im + this.im * b.re)
  }

  div(b: Complex): Complex {
    const norm = b.re * b.re + b.im * b.im
    if (norm < EPS) throw new Error("Complex divide by ~0")
    return new Complex((this.re * b.re + this.im * b.im) / norm, (this.im * b.re - this.re * b.im) / norm)
  }

  inv(): Complex {
    const norm = this.re * this.re + this.im * this.im
    if (norm < EPS) throw new Error("Complex invert by ~0")
    return new Complex(this.re / norm, -this.im / norm)
  }

  abs(): number {
    return Math.hypot(this.re, this.im)
  }

  phaseDeg(): number {
    return (Math.atan2(this.im, this.re) * 180) / Math.PI  // (this order of operations: the text of formatAcResult depends on the last bit)
  }
}
