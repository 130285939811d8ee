/*
 * oracle/spicey_ref.c — CPU restatement of the reference's transient hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this file; the product (libspicey_hip.so, spicey_amd/) never does.
 *
 * Parity status: PINNED.  This restatement is checked bit-for-bit (tests/test_oracle.py) against
 * tests/golden/<name>.json, which are outputs of the reference's own TypeScript TRAN path executed in
 * the build container (tools/js_oracle/make_golden.py: type-erasure + Node 12), and against the
 * reference's five committed SVG snapshots decoded in tests/golden/svg_series.json.
 *
 * It keeps the reference's operation order so that IEEE-754 results are identical
 * (compile with -O2 -ffp-contract=off):
 *   time loop / iteration loop      /root/reference/lib/analysis/simulateTRAN.ts:146-162
 *   stampAllElementsAtTime          simulateTRAN.ts:25-102
 *   stampAdmittanceReal             /root/reference/lib/stamping/stampAdmittanceReal.ts:3-29
 *   stampCurrentReal                /root/reference/lib/stamping/stampCurrentReal.ts:3-14
 *   stampVoltageSourceReal          /root/reference/lib/stamping/stampVoltageSourceReal.ts:4-32
 *   solveReal (dense GE, partial pivoting, |f|<EPS skip)   /root/reference/lib/math/solveReal.ts:3-73
 *   updateSwitchStatesFromSolution  simulateTRAN.ts:108-128
 *   recording + state update        simulateTRAN.ts:164-237
 *   EPS = 1e-15, VT_300K = 0.02585  /root/reference/lib/constants/EPS.ts:1, physics.ts:1
 * The only implementation-defined operation is Math.exp (simulateTRAN.ts:93,216).  The goldens
 * were produced under Node 12 / V8 7.8, whose Math.exp is base::ieee754::exp, a port of Sun's
 * fdlibm __ieee754_exp (third-party dependency of the JS engine, absent from /root/reference).
 * ref_exp() below restates that published algorithm (argument reduction by ln2 hi/lo, degree-5
 * Remez polynomial in r*r, scaling by 2^k); with it every golden is reproduced bit-for-bit,
 * whereas glibc's exp differs in the last bit on a few per cent of the inputs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/spicey_hip.h"

#define EPS 1e-15
#define VT_300K 0.02585

/* fdlibm e_exp.c (Sun Microsystems, 1993/2004) as used by V8 base::ieee754::exp. */
static double ref_exp(double x) {
  static const double one = 1.0, halF[2] = {0.5, -0.5}, huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302,
                      o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                      ln2HI[2] = {6.93147180369123816490e-01, -6.93147180369123816490e-01},
                      ln2LO[2] = {1.90821492927058770002e-10, -1.90821492927058770002e-10},
                      invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                      P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08, E = 2.718281828459045;
  double y, hi = 0.0, lo = 0.0, c, t, twopk;
  int32_t k = 0, xsb;
  uint64_t bits;
  uint32_t hx, lx;
  memcpy(&bits, &x, 8);
  hx = (uint32_t)(bits >> 32);
  lx = (uint32_t)bits;
  xsb = (hx >> 31) & 1;
  hx &= 0x7fffffff;
  if (hx >= 0x40862E42) { /* |x| >= 709.78... */
    if (hx >= 0x7ff00000) {
      if (((hx & 0xfffff) | lx) != 0) return x + x; /* NaN */
      return (xsb == 0) ? x : 0.0;                    /* exp(+-inf) = {inf, 0} */
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42) {   /* |x| > 0.5 ln2 */
    if (hx < 0x3FF0A2B2) { /* and |x| < 1.5 ln2 */
      if (x == 1.0) return E;
      hi = x - ln2HI[xsb];
      lo = ln2LO[xsb];
      k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + halF[xsb]);
      t = k;
      hi = x - t * ln2HI[0];
      lo = t * ln2LO[0];
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) { /* |x| < 2**-28 */
    if (huge + x > one) return one + x;
  } else {
    k = 0;
  }
  t = x * x;
  if (k >= -1021) {
    bits = (uint64_t)(uint32_t)(0x3ff00000 + (int32_t)((uint32_t)k << 20)) << 32;
  } else {
    bits = (uint64_t)(uint32_t)(0x3ff00000 + (int32_t)((uint32_t)(k + 1000) << 20)) << 32;
  }
  memcpy(&twopk, &bits, 8);
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return one - ((x * c) / (c - 2.0) - x);
  y = one - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    if (k == 1024) return y * 2.0 * 8.98846567431158e+307; /* 0x1p1023 */
    return y * twopk;
  }
  return y * twopk * twom1000;
}

static inline double dmax(double a, double b) {
  /* Math.max: NaN if either is NaN */
  if (a != a || b != b) return NAN;
  return a > b ? a : b;
}

/* stampAdmittanceReal.ts:3-29 (row index = node id - 1, ground dropped) */
static inline void stamp_adm(double **A, int n1, int n2, double Y) {
  int i1 = n1 - 1, i2 = n2 - 1;
  if (i1 >= 0) A[i1][i1] = A[i1][i1] + Y;
  if (i2 >= 0) A[i2][i2] = A[i2][i2] + Y;
  if (i1 >= 0 && i2 >= 0) {
    A[i1][i2] = A[i1][i2] - Y;
    A[i2][i1] = A[i2][i1] - Y;
  }
}

/* stampCurrentReal.ts:3-14 */
static inline void stamp_cur(double *b, int np, int nm, double cur) {
  int ip = np - 1, im = nm - 1;
  if (ip >= 0) b[ip] = b[ip] - cur;
  if (im >= 0) b[im] = b[im] + cur;
}

/* Test knobs of the checker (never part of the reference's behaviour unless stated):
 *   skip_off    1 = run solveReal WITHOUT `if (Math.abs(f) < EPS) continue` (solveReal.ts:45): the same algorithm with every
 *               row update performed.  Used to show that the product's only semantic difference to the reference is that line.
 *   skipped     how many NONZERO multipliers the last run skipped (exact zeros are structural and change nothing)
 *   skip_solves how many solves of the last run skipped at least one
 *   lin_err     if set, [steps+1] per step max over the diodes of |vd(x) - vd_lin|, vd_lin = the junction voltage the step's
 *               last solve was stamped with (simulateTRAN.ts:81-85) */
static int g_skip_off = 0;
static int64_t g_skipped = 0, g_skip_solves = 0;
static double *g_lin_err = 0;
void spicey_ref_set_knobs(int32_t skip_off, double *lin_err) { g_skip_off = skip_off; g_lin_err = lin_err; }
void spicey_ref_get_skips(int64_t *skipped, int64_t *skip_solves) { if (skipped) *skipped = g_skipped; if (skip_solves) *skip_solves = g_skip_solves; }

/* solveReal.ts:3-73; rows[] are pointers into an (n x (n+1)) slab, augmented column = b.
 * Returns 0, or 1 for "Singular matrix (real)". */
static int solve_real(double **rows, int n, double *x) {
  int64_t skipped_here = 0;
  for (int k = 0; k < n; k++) {
    int imax = k;
    double vmax = fabs(rows[k][k]);
    for (int i = k + 1; i < n; i++) {
      double v = fabs(rows[i][k]);
      if (v > vmax) {
        vmax = v;
        imax = i;
      }
    }
    if (vmax < EPS) return 1;
    if (imax != k) {
      double *tmp = rows[k];
      rows[k] = rows[imax];
      rows[imax] = tmp;
    }
    const double *prow = rows[k];
    const double pivot = prow[k];
    for (int i = k + 1; i < n; i++) {
      double *row = rows[i];
      double f = row[k] / pivot;
      if (fabs(f) < EPS) {
        if (f != 0.0) skipped_here++;
        if (!g_skip_off || f == 0.0) continue;
      }
      for (int j = k; j <= n; j++) row[j] = row[j] - f * prow[j];
    }
  }
  g_skipped += skipped_here;
  g_skip_solves += skipped_here > 0;
  for (int i = n - 1; i >= 0; i--) {
    const double *row = rows[i];
    double s = row[n];
    for (int j = i + 1; j < n; j++) s -= row[j] * x[j];
    x[i] = s / row[i];
  }
  return 0;
}

/*
 * One instance, one transient run.  State arrays (may be NULL = start from the descriptor and
 * discard) are read as the state entering the run and overwritten with the state leaving it.
 * Returns SPICEY_OK, SPICEY_ERR_SINGULAR (err_step/err_iter filled) or SPICEY_ERR_BAD_DESC.
 */
int32_t spicey_ref_run(const SpiceyDesc *d, int32_t inst, int64_t steps, double dt,
                       const double *src_table, double *out_v, double *out_i, int32_t *iters,
                       double *C_vprev, double *L_iprev, double *D_vdprev, int32_t *S_ison,
                       int64_t *err_step, int32_t *err_iter) {
  if (!d || d->abi_version != SPICEY_ABI_VERSION || inst < 0 || inst >= d->n_inst) return SPICEY_ERR_BAD_DESC;
  const int nN = d->n_nodes, nR = d->nR, nC = d->nC, nL = d->nL, nV = d->nV, nS = d->nS, nD = d->nD;
  const int n = nN + nV;
  const int n_out = (d->n_out > 0 && d->out_nodes) ? d->n_out : nN;
  const int n_cur = nR + nC + nL + nV + nS + nD;
  const double *Rv = d->R_val ? d->R_val + (size_t)inst * nR : 0;
  const double *Cv = d->C_val ? d->C_val + (size_t)inst * nC : 0;
  const double *Lv = d->L_val ? d->L_val + (size_t)inst * nL : 0;
  const double *Ron = d->S_ron ? d->S_ron + (size_t)inst * nS : 0;
  const double *Roff = d->S_roff ? d->S_roff + (size_t)inst * nS : 0;
  const double *Von = d->S_von ? d->S_von + (size_t)inst * nS : 0;
  const double *Voff = d->S_voff ? d->S_voff + (size_t)inst * nS : 0;
  const double *Dis = d->D_is ? d->D_is + (size_t)inst * nD : 0;
  const double *Dn = d->D_n ? d->D_n + (size_t)inst * nD : 0;

  double *vprev = (double *)calloc(nC + 1, sizeof(double));
  double *iprev = (double *)calloc(nL + 1, sizeof(double));
  double *vdprev = (double *)calloc(nD + 1, sizeof(double));
  int32_t *ison = (int32_t *)calloc(nS + 1, sizeof(int32_t));
  for (int i = 0; i < nC; i++) vprev[i] = C_vprev ? C_vprev[i] : (d->C_vprev ? d->C_vprev[(size_t)inst * nC + i] : 0.0);
  for (int i = 0; i < nL; i++) iprev[i] = L_iprev ? L_iprev[i] : (d->L_iprev ? d->L_iprev[(size_t)inst * nL + i] : 0.0);
  for (int i = 0; i < nD; i++) vdprev[i] = D_vdprev ? D_vdprev[i] : (d->D_vdprev ? d->D_vdprev[(size_t)inst * nD + i] : 0.0);
  for (int i = 0; i < nS; i++) ison[i] = S_ison ? S_ison[i] : (d->S_ison ? d->S_ison[(size_t)inst * nS + i] : 0);

  double *slab = (double *)malloc((size_t)(n > 0 ? n : 1) * (n + 1) * sizeof(double));
  double **rows = (double **)malloc((size_t)(n > 0 ? n : 1) * sizeof(double *));
  double *b = (double *)malloc((size_t)(n + 1) * sizeof(double));
  double *x = (double *)malloc((size_t)(n + 1) * sizeof(double));
  double *vdlin = (double *)malloc((size_t)(nD + 1) * sizeof(double));
  int32_t rc = SPICEY_OK;
  g_skipped = 0; g_skip_solves = 0;

#define VOLT(node) ((node) == 0 ? 0.0 : x[(node)-1])
  for (int64_t step = 0; step <= steps && rc == SPICEY_OK; step++) {
    const double *src = src_table + (size_t)step * nV;
    for (int i = 0; i < n; i++) x[i] = 0.0; /* :149 */
    int iter = 0;
    for (; iter < 20; iter++) {
      /* :152-153 fresh zero A, b */
      memset(slab, 0, (size_t)n * (n + 1) * sizeof(double));
      for (int i = 0; i < n; i++) {
        rows[i] = slab + (size_t)i * (n + 1);
        b[i] = 0.0;
      }
      /* stampAllElementsAtTime :25-102, order R, C, L, S, V, D */
      for (int i = 0; i < nR; i++) stamp_adm(rows, d->R_n1[i], d->R_n2[i], 1 / Rv[i]);
      for (int i = 0; i < nC; i++) {
        double Gc = Cv[i] / dmax(dt, EPS);
        stamp_adm(rows, d->C_n1[i], d->C_n2[i], Gc);
        double Ieq = -Gc * vprev[i];
        stamp_cur(b, d->C_n1[i], d->C_n2[i], Ieq);
      }
      for (int i = 0; i < nL; i++) {
        double Gl = dmax(dt, EPS) / Lv[i];
        stamp_adm(rows, d->L_n1[i], d->L_n2[i], Gl);
        stamp_cur(b, d->L_n1[i], d->L_n2[i], iprev[i]);
      }
      for (int i = 0; i < nS; i++) {
        double Rvalue = ison[i] ? Ron[i] : Roff[i];
        double Rcl = dmax(fabs(Rvalue), EPS);
        stamp_adm(rows, d->S_n1[i], d->S_n2[i], 1 / Rcl);
      }
      for (int k = 0; k < nV; k++) { /* stampVoltageSourceReal.ts:4-32 */
        int i1 = d->V_n1[k] - 1, i2 = d->V_n2[k] - 1, j = nN + k;
        if (i1 >= 0) rows[i1][j] = rows[i1][j] + 1;
        if (i2 >= 0) rows[i2][j] = rows[i2][j] - 1;
        if (i1 >= 0) rows[j][i1] = rows[j][i1] + 1;
        if (i2 >= 0) rows[j][i2] = rows[j][i2] - 1;
        b[j] = b[j] + src[k];
      }
      for (int i = 0; i < nD; i++) { /* :72-101 */
        int np = d->D_np[i], nm = d->D_nm[i];
        double vd_iter = VOLT(np) - VOLT(nm);
        double vd = iter == 0 ? vdprev[i] : vd_iter;
        vdlin[i] = vd;
        double vt = Dn[i] * VT_300K;
        double vl = vd;
        if (vd > 0.8) vl = 0.8;
        if (vd < -1.0) vl = -1.0;
        double e = ref_exp(vl / vt);
        double id = Dis[i] * (e - 1);
        double gd = dmax((Dis[i] / vt) * e, 1e-12);
        double ieq = id - gd * vl;
        stamp_adm(rows, np, nm, gd);
        stamp_cur(b, np, nm, ieq);
      }
      /* solveReal: augmented column */
      for (int i = 0; i < n; i++) rows[i][n] = b[i];
      if (solve_real(rows, n, x)) {
        rc = SPICEY_ERR_SINGULAR;
        if (err_step) *err_step = step;
        if (err_iter) *err_iter = iter;
        break;
      }
      /* updateSwitchStatesFromSolution :108-128 */
      int switched = 0;
      for (int i = 0; i < nS; i++) {
        double vp = VOLT(d->S_cp[i]), vn = VOLT(d->S_cn[i]);
        double vctrl = vp - vn;
        int next = ison[i];
        if (ison[i]) {
          if (vctrl < Voff[i]) next = 0;
        } else if (vctrl > Von[i]) {
          next = 1;
        }
        if (next != ison[i]) {
          ison[i] = next;
          switched = 1;
        }
      }
      if (!switched) break;
      if (iter == 19) break;
    }
    if (rc != SPICEY_OK) break;
    if (iters) iters[step] = iter + 1;
    if (g_lin_err) {
      double m = 0.0;
      for (int i = 0; i < nD; i++) {
        double e = fabs((VOLT(d->D_np[i]) - VOLT(d->D_nm[i])) - vdlin[i]);
        if (e > m) m = e;
      }
      g_lin_err[step] = m;
    }

    /* record node voltages :164-171 */
    if (out_v) {
      double *ov = out_v + (size_t)step * n_out;
      if (d->n_out > 0 && d->out_nodes)
        for (int i = 0; i < n_out; i++) ov[i] = VOLT(d->out_nodes[i]);
      else
        for (int i = 0; i < nN; i++) ov[i] = x[i];
    }
    /* element currents :173-219 */
    if (out_i) {
      double *oi = out_i + (size_t)step * n_cur;
      int o = 0;
      for (int i = 0; i < nR; i++) oi[o++] = (VOLT(d->R_n1[i]) - VOLT(d->R_n2[i])) / Rv[i];
      for (int i = 0; i < nC; i++)
        oi[o++] = (Cv[i] * (VOLT(d->C_n1[i]) - VOLT(d->C_n2[i]) - vprev[i])) / dmax(dt, EPS);
      for (int i = 0; i < nL; i++) {
        double Gl = dmax(dt, EPS) / Lv[i];
        oi[o++] = Gl * (VOLT(d->L_n1[i]) - VOLT(d->L_n2[i])) + iprev[i];
      }
      for (int k = 0; k < nV; k++) oi[o++] = x[nN + k];
      for (int i = 0; i < nS; i++) {
        double Rvalue = ison[i] ? Ron[i] : Roff[i];
        double Rcl = dmax(fabs(Rvalue), EPS);
        oi[o++] = (VOLT(d->S_n1[i]) - VOLT(d->S_n2[i])) / Rcl;
      }
      for (int i = 0; i < nD; i++) {
        double vd = VOLT(d->D_np[i]) - VOLT(d->D_nm[i]);
        double vt = Dn[i] * VT_300K;
        oi[o++] = Dis[i] * (ref_exp(vd / vt) - 1);
      }
    }
    /* state update :221-237 */
    for (int i = 0; i < nC; i++) vprev[i] = VOLT(d->C_n1[i]) - VOLT(d->C_n2[i]);
    for (int i = 0; i < nL; i++) {
      double Gl = dmax(dt, EPS) / Lv[i];
      iprev[i] = Gl * (VOLT(d->L_n1[i]) - VOLT(d->L_n2[i])) + iprev[i];
    }
    for (int i = 0; i < nD; i++) vdprev[i] = VOLT(d->D_np[i]) - VOLT(d->D_nm[i]);
  }
#undef VOLT

  if (C_vprev) memcpy(C_vprev, vprev, sizeof(double) * nC);
  if (L_iprev) memcpy(L_iprev, iprev, sizeof(double) * nL);
  if (D_vdprev) memcpy(D_vdprev, vdprev, sizeof(double) * nD);
  if (S_ison) memcpy(S_ison, ison, sizeof(int32_t) * nS);
  free(vprev); free(iprev); free(vdprev); free(ison);
  free(slab); free(rows); free(b); free(x); free(vdlin);
  return rc;
}

/* computeEffectiveTimeStep, simulateTRAN.ts:14-19 (same double operations in the same order). */
void spicey_ref_timestep(double dt_requested, double tstop, double *dt_out, int64_t *steps_out) {
  double dtEff = dt_requested > EPS ? dt_requested : dmax(tstop / 1000, EPS);
  double s = ceil(tstop / dmax(dtEff, EPS));
  double steps = dmax(1, s);
  *dt_out = steps > 0 ? tstop / steps : tstop;
  *steps_out = (int64_t)steps;
}
