/*
 * oracle/spicey_ref_ac.c — CPU restatement of the reference's AC sweep.
 *
 * TEST INFRASTRUCTURE.  Only tests/ may build, load or call this file; the product (libspicey_hip.so,
 * spicey_amd/) never does.
 *
 * Parity status: PINNED.  Checked bit-for-bit (tests/test_oracle_ac.py) against tests/golden/ac_*.json, outputs of
 * the reference's own TypeScript AC path executed in the build container (tools/js_oracle/make_golden_ac.py:
 * type-erasure + Node 12); that run in turn reproduces the reference's own inline snapshot of
 * tests/basics/basics01.test.ts (201 lines of formatAcResult output) character for character.
 *
 * Operation order follows the reference so that IEEE-754 results are identical (-O2 -ffp-contract=off):
 *   buildLinearSystemForAC   /root/reference/lib/analysis/simulateAC.ts:25-62   (R, C, L, V in this order)
 *   stampAdmittanceComplex   /root/reference/lib/stamping/stampAdmittanceComplex.ts:4-30
 *   stampVoltageSourceComplex /root/reference/lib/stamping/stampVoltageSourceComplex.ts:5-35
 *   solveComplex             /root/reference/lib/math/solveComplex.ts:4-73 (dense GE, partial pivoting on |z|,
 *                            |f| < EPS skip, update j = k..n, sequential back-substitution)
 *   Complex add/sub/mul/div/abs  /root/reference/lib/math/Complex.ts:25-58
 *   result recording         simulateAC.ts:84-126 (node voltages; currents R, C, L, V)
 * The frequency list (simulateAC.ts:9-23, utils/logspace.ts) and the source phasors (Complex.fromPolar,
 * Complex.ts:16-19) are HOST work in the drop-in (Math.pow / Math.cos / Math.sin are engine-defined); they are inputs
 * here.  The one engine-defined operation inside is Math.hypot (Complex.abs), used only in comparisons (pivot choice,
 * |f| < EPS): ref_hypot() restates V8's algorithm (scale by the maximum, Kahan-compensated sum of squares, sqrt).
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/spicey_hip.h"

#define EPS 1e-15

typedef struct { double re, im; } cx;

static inline cx cx_add(cx a, cx b) { cx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cx cx_sub(cx a, cx b) { cx r = {a.re - b.re, a.im - b.im}; return r; }
static inline cx cx_mul(cx a, cx b) { cx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
/* Complex.div: returns 0 and sets *bad when d < EPS ("Complex divide by ~0") */
static inline cx cx_div(cx a, cx b, int *bad) {
  const double d = b.re * b.re + b.im * b.im;
  cx r = {0.0, 0.0};
  if (d < EPS) { *bad = 1; return r; }
  r.re = (a.re * b.re + a.im * b.im) / d;
  r.im = (a.im * b.re - a.re * b.im) / d;
  return r;
}
/* V8 Math.hypot for two arguments */
static double ref_hypot(double x, double y) {
  const double ax = fabs(x), ay = fabs(y);
  double mx = 0.0;
  if (isnan(x) || isnan(y)) return (isinf(x) || isinf(y)) ? INFINITY : NAN;
  if (ax > mx) mx = ax;
  if (ay > mx) mx = ay;
  if (mx == INFINITY) return INFINITY;
  if (mx == 0.0) return 0.0;
  double sum = 0.0, comp = 0.0;
  const double v[2] = {ax, ay};
  for (int i = 0; i < 2; i++) {
    const double n = v[i] / mx;
    const double summand = n * n - comp;
    const double prelim = sum + summand;
    comp = (prelim - sum) - summand;
    sum = prelim;
  }
  return sqrt(sum) * mx;
}
static inline double cx_abs(cx a) { return ref_hypot(a.re, a.im); }

static void stamp_adm(cx **A, int n1, int n2, cx Y) {
  const int i1 = n1 - 1, i2 = n2 - 1; /* NodeIndex.matrixIndexOfNode: ground -> -1 */
  if (i1 >= 0) A[i1][i1] = cx_add(A[i1][i1], Y);
  if (i2 >= 0) A[i2][i2] = cx_add(A[i2][i2], Y);
  if (i1 >= 0 && i2 >= 0) {
    A[i1][i2] = cx_sub(A[i1][i2], Y);
    A[i2][i1] = cx_sub(A[i2][i1], Y);
  }
}

/* rows[i] has n+1 entries (augmented).  0 ok, 1 "Singular matrix (complex)", 5 "Complex divide by ~0" */
static int solve_complex(cx **rows, int n, cx *x) {
  for (int k = 0; k < n; k++) {
    int imax = k;
    double vmax = cx_abs(rows[k][k]);
    for (int i = k + 1; i < n; i++) {
      const double v = cx_abs(rows[i][k]);
      if (v > vmax) { vmax = v; imax = i; }
    }
    if (vmax < EPS) return 1;
    if (imax != k) { cx *t = rows[k]; rows[k] = rows[imax]; rows[imax] = t; }
    cx *pr = rows[k];
    const cx pivot = pr[k];
    for (int i = k + 1; i < n; i++) {
      cx *row = rows[i];
      int bad = 0;
      const cx f = cx_div(row[k], pivot, &bad);
      if (bad) return 5;
      if (cx_abs(f) < EPS) continue;
      for (int j = k; j <= n; j++) row[j] = cx_sub(row[j], cx_mul(f, pr[j]));
    }
  }
  for (int i = n - 1; i >= 0; i--) {
    cx *row = rows[i];
    cx s = row[n];
    for (int j = i + 1; j < n; j++) s = cx_sub(s, cx_mul(row[j], x[j]));
    int bad = 0;
    x[i] = cx_div(s, row[i], &bad);
    if (bad) return 5;
  }
  return 0;
}

/*
 * One AC sweep of instance `inst` of descriptor d (diodes and switches are ignored, like simulateAC.ts does).
 *   freqs   [n_freq]
 *   vph     [nV][2] source phasors (re, im)
 *   out_v   [n_freq][n_nodes][2]
 *   out_i   [n_freq][nR+nC+nL+nV][2] or NULL
 * Returns 0, 1 (singular), 5 (complex divide by ~0), 6 (resistor <= 0: "R <name> must be > 0", index in *err_index).
 */
int32_t spicey_ref_ac(const SpiceyDesc *d, int32_t inst, int64_t n_freq, const double *freqs, const double *vph,
                      double *out_v, double *out_i, int32_t *err_index) {
  const int nN = d->n_nodes, nV = d->nV, n = nN + nV;
  const int nR = d->nR, nC = d->nC, nL = d->nL;
  const double *Rv = d->R_val + (size_t)inst * nR, *Cv = d->C_val + (size_t)inst * nC, *Lv = d->L_val + (size_t)inst * nL;
  const double twoPi = 2 * 3.141592653589793; /* 2 * Math.PI */
  const int nCur = nR + nC + nL + nV;
  cx *store = (cx *)malloc(sizeof(cx) * (size_t)n * (n + 1));
  cx **rows = (cx **)malloc(sizeof(cx *) * (size_t)(n > 0 ? n : 1));
  cx *x = (cx *)malloc(sizeof(cx) * (size_t)(n > 0 ? n : 1));
  int rc = 0;
  const cx zero = {0.0, 0.0}, one = {1.0, 0.0};
  for (int64_t fi = 0; fi < n_freq && rc == 0; fi++) {
    const double f = freqs[fi];
    for (int i = 0; i < n; i++) {
      rows[i] = store + (size_t)i * (n + 1);
      for (int j = 0; j <= n; j++) rows[i][j] = zero;
    }
    for (int i = 0; i < nR; i++) {
      if (Rv[i] <= 0) { rc = 6; if (err_index) *err_index = i; break; }
      const cx Y = {1 / Rv[i], 0.0};
      stamp_adm(rows, d->R_n1[i], d->R_n2[i], Y);
    }
    if (rc) break;
    for (int i = 0; i < nC; i++) {
      const cx Y = {0.0, twoPi * f * Cv[i]};
      stamp_adm(rows, d->C_n1[i], d->C_n2[i], Y);
    }
    for (int i = 0; i < nL; i++) {
      const cx denom = {0.0, twoPi * f * Lv[i]};
      cx Y = zero;
      if (!(cx_abs(denom) < EPS)) {
        int bad = 0;
        Y = cx_div(one, denom, &bad);
        if (bad) { rc = 5; break; }
      }
      stamp_adm(rows, d->L_n1[i], d->L_n2[i], Y);
    }
    if (rc) break;
    for (int k = 0; k < nV; k++) {
      const int i1 = d->V_n1[k] - 1, i2 = d->V_n2[k] - 1, j = nN + k;
      if (i1 >= 0) rows[i1][j] = cx_add(rows[i1][j], one);
      if (i2 >= 0) rows[i2][j] = cx_sub(rows[i2][j], one);
      if (i1 >= 0) rows[j][i1] = cx_add(rows[j][i1], one);
      if (i2 >= 0) rows[j][i2] = cx_sub(rows[j][i2], one);
      const cx v = {vph[2 * k], vph[2 * k + 1]};
      rows[j][n] = cx_add(rows[j][n], v);
    }
    rc = solve_complex(rows, n, x);
    if (rc) break;
    double *ov = out_v + (size_t)fi * nN * 2;
    for (int i = 0; i < nN; i++) { ov[2 * i] = x[i].re; ov[2 * i + 1] = x[i].im; }
    if (out_i) {
      double *oi = out_i + (size_t)fi * nCur * 2;
      int c = 0;
#define NODEV(nd) ((nd) == 0 ? zero : x[(nd) - 1])
      for (int i = 0; i < nR; i++, c++) {
        const cx Y = {1 / Rv[i], 0.0};
        const cx cur = cx_mul(Y, cx_sub(NODEV(d->R_n1[i]), NODEV(d->R_n2[i])));
        oi[2 * c] = cur.re; oi[2 * c + 1] = cur.im;
      }
      for (int i = 0; i < nC; i++, c++) {
        const cx Y = {0.0, twoPi * f * Cv[i]};
        const cx cur = cx_mul(Y, cx_sub(NODEV(d->C_n1[i]), NODEV(d->C_n2[i])));
        oi[2 * c] = cur.re; oi[2 * c + 1] = cur.im;
      }
      for (int i = 0; i < nL; i++, c++) {
        const cx denom = {0.0, twoPi * f * Lv[i]};
        cx Y = zero;
        if (!(cx_abs(denom) < EPS)) { int bad = 0; Y = cx_div(one, denom, &bad); }
        const cx cur = cx_mul(Y, cx_sub(NODEV(d->L_n1[i]), NODEV(d->L_n2[i])));
        oi[2 * c] = cur.re; oi[2 * c + 1] = cur.im;
      }
      for (int k = 0; k < nV; k++, c++) { oi[2 * c] = x[nN + k].re; oi[2 * c + 1] = x[nN + k].im; }
#undef NODEV
    }
  }
  free(store); free(rows); free(x);
  return rc;
}
