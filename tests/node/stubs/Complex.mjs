// stand-in for the reference's lib/math/Complex.ts: the two members ts/simulateAC.ts uses
export class Complex {
  constructor(re = 0, im = 0) { this.re = re; this.im = im }
  static from(re, im = 0) { return new Complex(re, im) }
  static fromPolar(mag, deg = 0) {
    const ph = (deg * Math.PI) / 180
    return new Complex(mag * Math.cos(ph), mag * Math.sin(ph))
  }
}
