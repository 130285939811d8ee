// ac_exec.h — AC sweep: one complex MNA solve per (instance, frequency).
//
// Replaces the per-frequency body of simulateAC (/root/reference/lib/analysis/simulateAC.ts:80-126):
//   buildLinearSystemForAC (:25-62)  -> S: entries gathered from the static stamp lists with complex admittances
//   solveComplex (lib/math/solveComplex.ts:4-73) -> U_l / K_l: the same level-scheduled sparse LU as the transient
//                                       kernel (symbolic.cpp), in complex arithmetic, fixed pivot order
//   recording (:84-126)              -> Z: node voltages, currents of R, C, L, V
// Every (instance, frequency) pair is independent: one workgroup each, workspace in LDS (16 bytes per entry) or,
// for large circuits, in global memory.  The phase code is shared with the test suite's CPU emulator (same
// SPICEY_HD / Exec.phase() arrangement as tran_exec.h); diodes and switches do not take part (the reference's AC
// analysis ignores them), so the program is built from the descriptor with nS = nD = 0.
#pragma once
#include "tran_exec.h"

#define SPICEY_ERR_COMPLEX_DIV_CODE 5

struct SpiceyAcRun {
  const double *R_inv, *C_val, *L_val;  // [n_inst][n<kind>]; R_inv = 1 / R, formed once per handle on the host (the same IEEE
                                        // quotient the reference forms at every frequency, simulateAC.ts:39-41)
  const double *freqs;                  // [n_freq]
  const double *vph;                    // [n_inst][nV][2] source phasors
  double *out_v;                        // [n_inst][n_freq][nOut][2]
  double *out_i;                        // [n_inst][n_freq][nR+nC+nL+nV][2] or null
  double *gW;                           // [n_workgroups][nW][2] global workspace, or null when the workspace is in LDS
  int32_t *status;                      // [n_inst * n_freq] 0 ok, 1 singular, 5 complex divide by ~0
  int64_t n_freq;
  int64_t slot_base;                    // first (instance, frequency) slot of this launch (large sweeps run in chunks)
  int32_t n_inst;
};

struct SpiceyCx { double re, im; };

SPICEY_HD SpiceyCx cx_mul(SpiceyCx a, SpiceyCx b) { return SpiceyCx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
SPICEY_HD SpiceyCx cx_sub(SpiceyCx a, SpiceyCx b) { return SpiceyCx{a.re - b.re, a.im - b.im}; }
SPICEY_HD SpiceyCx cx_add(SpiceyCx a, SpiceyCx b) { return SpiceyCx{a.re + b.re, a.im + b.im}; }

template <class Exec>
struct AcPhases {
  const SpiceyProg &P;
  const SpiceyAcRun &R;
  SpiceyCx *W;      // [nW]
  int32_t *flags;   // [0] error code of this solve
  int T;
  size_t inst;
  double w;         // 2 pi f   (simulateAC.ts:33,43: twoPi * f * C = (twoPi * f) * C)

  // 1/z for pivots, with the reference's guards: |z| < EPS -> "Singular matrix (complex)" (solveComplex.ts:28),
  // |z|^2 < EPS -> "Complex divide by ~0" (Complex.ts:40-42, reached through entry.div(pivot))
  SPICEY_HD SpiceyCx pivot_inv(SpiceyCx z) const {
    const double d = z.re * z.re + z.im * z.im;
    if (d < SPICEY_EPS) {
      const int code = d < SPICEY_EPS * SPICEY_EPS ? 1 : SPICEY_ERR_COMPLEX_DIV_CODE;
      if (flags[0] == 0 || code == 1) flags[0] = code;  // benign race: any writer leaves a non-zero code
      return SpiceyCx{0.0, 0.0};
    }
    return SpiceyCx{z.re / d, -z.im / d};
  }
  // admittance of static-stamp slot idx: [1/R | j w C | 1/(j w L) | 1] (simulateAC.ts:38-55)
  SPICEY_HD SpiceyCx admittance(uint32_t idx) const {
    if (idx < (uint32_t)P.nR) return SpiceyCx{R.R_inv[inst * P.nR + idx], 0.0};
    idx -= P.nR;
    if (idx < (uint32_t)P.nC) return SpiceyCx{0.0, w * R.C_val[inst * P.nC + idx]};
    idx -= P.nC;
    if (idx < (uint32_t)P.nL) {
      const double wl = w * R.L_val[inst * P.nL + idx];
      if (fabs(wl) < SPICEY_EPS) return SpiceyCx{0.0, 0.0};
      const double d = wl * wl;
      if (d < SPICEY_EPS) { flags[0] = SPICEY_ERR_COMPLEX_DIV_CODE; return SpiceyCx{0.0, 0.0}; }
      return SpiceyCx{0.0 / d, (0.0 - wl) / d};  // Complex.from(1,0).div(Complex.from(0, wl))
    }
    return SpiceyCx{1.0, 0.0};
  }
  SPICEY_HD SpiceyCx volt(int32_t xi) const { return xi < 0 ? SpiceyCx{0.0, 0.0} : W[xi]; }

  // raw: the entries as stamped (the dense fallback wants A itself, not the reciprocals of leaf diagonals)
  SPICEY_HD void s_stamp(int tid, bool raw = false) const {
    SPICEY_NOUNROLL
    for (int e = tid; e < P.nLU; e += T) {
      SpiceyCx v{0.0, 0.0};
      for (uint32_t j = P.stat_ptr[e]; j < P.stat_ptr[e + 1]; j++) {
        const uint32_t ix = P.stat_idx[j];
        const SpiceyCx g = admittance(SPICEY_IDX(ix));
        v = (ix & SPICEY_NEG) ? cx_sub(v, g) : cx_add(v, g);
      }
      if (!raw && (P.ent_flag[e] & 1)) v = pivot_inv(v);  // leaf diagonal: final as stamped
      W[e] = v;
    }
    const uint32_t oV = (uint32_t)(P.nC + P.nL);
    SPICEY_NOUNROLL
    for (int r = tid; r < P.n; r += T) {  // b[j] += Vph of source rows (stampVoltageSourceComplex.ts:34)
      SpiceyCx acc{0.0, 0.0};
      for (uint32_t j = P.rhs_ptr[r]; j < P.rhs_ptr[r + 1]; j++) {
        const uint32_t ix = P.rhs_idx[j], u = SPICEY_IDX(ix);
        if (u < oV || u >= oV + (uint32_t)P.nV) continue;  // capacitor / inductor companions are transient-only
        const double *ph = R.vph + (inst * P.nV + (u - oV)) * 2;
        const SpiceyCx v{ph[0], ph[1]};
        acc = (ix & SPICEY_NEG) ? cx_sub(acc, v) : cx_add(acc, v);
      }
      W[P.nLU + r] = acc;
    }
  }
  SPICEY_HD void u_level(int tid, int l) const {
    const int nw = T >> 6, wv = tid >> 6, lane = tid & 63;
    for (uint32_t s = P.lvl_slice[l] + wv; s < P.lvl_slice[l + 1]; s += nw) {
      const uint32_t t = s * 64 + lane;
      const uint32_t tgt = P.upd_tgt[t];
      if (tgt == SPICEY_TGT_PAD) continue;
      const uint32_t cnt = P.upd_cnt[t];
      const uint32_t off = P.upd_slice[s].off + lane;
      const uint32_t ti = SPICEY_IDX(tgt);
      SpiceyCx acc = W[ti];
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t li = P.upd_pairs[off + (j * 3 + 0) * 64];
        const uint32_t di = P.upd_pairs[off + (j * 3 + 1) * 64];
        const uint32_t ui = P.upd_pairs[off + (j * 3 + 2) * 64];
        acc = cx_sub(acc, cx_mul(cx_mul(W[li], W[di]), W[ui]));
      }
      if (tgt & SPICEY_TGT_RECIP) acc = pivot_inv(acc);
      W[ti] = acc;
    }
  }
  SPICEY_HD void k_level(int tid, int l) const {
    const int nw = T >> 6, wv = tid >> 6, lane = tid & 63;
    for (uint32_t s = P.bk_lvl_slice[l] + wv; s < P.bk_lvl_slice[l + 1]; s += nw) {
      const uint32_t t = s * 64 + lane;
      const uint32_t yi = P.bk_x[t];
      if (yi == SPICEY_TGT_PAD) continue;
      const uint32_t cnt = P.bk_cnt[t];
      const uint32_t off = P.bk_slice[s].off + lane;
      SpiceyCx acc = W[yi];
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t ki = P.bk_pairs[off + (j * 3 + 0) * 64];
        const uint32_t di = P.bk_pairs[off + (j * 3 + 1) * 64];
        const uint32_t ui = P.bk_pairs[off + (j * 3 + 2) * 64];
        acc = cx_sub(acc, cx_mul(cx_mul(W[ki], W[di]), W[ui]));
      }
      W[yi] = acc;
    }
  }
  SPICEY_HD void k_scale(int tid) const {
    SPICEY_NOUNROLL
    for (int i = tid; i < P.n; i += T) W[P.nLU + i] = cx_mul(W[P.nLU + i], W[P.bk_d[i]]);
  }
  SPICEY_HD void z_record(int tid, int64_t fi) const {
    const size_t slot = inst * (size_t)R.n_freq + (size_t)fi;
    double *ov = R.out_v + slot * (size_t)P.nOut * 2;
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nOut; i += T) {
      const SpiceyCx v = volt(P.out_x[i]);
      ov[2 * i] = v.re; ov[2 * i + 1] = v.im;
    }
    if (!R.out_i) return;
    const int nCur = P.nR + P.nC + P.nL + P.nV;
    double *oi = R.out_i + slot * (size_t)nCur * 2;
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nR; i += T) {
      const SpiceyCx cur = cx_mul(admittance((uint32_t)i), cx_sub(volt(P.R_a[i]), volt(P.R_b[i])));
      oi[2 * i] = cur.re; oi[2 * i + 1] = cur.im;
    }
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nC; i += T) {
      const SpiceyCx cur = cx_mul(admittance((uint32_t)(P.nR + i)), cx_sub(volt(P.C_a[i]), volt(P.C_b[i])));
      oi[2 * (P.nR + i)] = cur.re; oi[2 * (P.nR + i) + 1] = cur.im;
    }
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nL; i += T) {
      const SpiceyCx cur = cx_mul(admittance((uint32_t)(P.nR + P.nC + i)), cx_sub(volt(P.L_a[i]), volt(P.L_b[i])));
      oi[2 * (P.nR + P.nC + i)] = cur.re; oi[2 * (P.nR + P.nC + i) + 1] = cur.im;
    }
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nV; i += T) {
      const SpiceyCx cur = W[P.V_x[i]];
      oi[2 * (P.nR + P.nC + P.nL + i)] = cur.re; oi[2 * (P.nR + P.nC + P.nL + i) + 1] = cur.im;
    }
  }
};

// ---- dense fallback with partial pivoting ---------------------------------------------------------------------------------
// The static pivot order is safe for G + jwC; with inductors a diagonal can cancel (1/(jwL) + jwC = 0 at a resonance) where
// the reference's partial pivoting simply takes another row (solveComplex.ts:16-36).  A solve that trips the pivot guards is
// therefore repeated here the reference's way: dense A | b in global memory in the reference's own numbering (rows = nodes
// then branches), per column k the row of largest |A[i][k]| (first one among equals), `vmax < EPS` -> "Singular matrix
// (complex)", a pivot with |p|^2 < EPS -> "Complex divide by ~0" (Complex.div), rows whose multiplier has |f| < EPS skipped,
// back substitution.  One workgroup per solve; O(n^2) memory, O(n * nnz)-ish work on circuit matrices (most multipliers are
// zero).  Exec additionally supplies atomic_inc(int *) (an LDS counter).
// Scratch: sd[T + 2 n + 2] doubles, si[T + n + 4] ints (LDS on the GPU).
// |z| without overflow / underflow of the squares (Math.hypot's contract; used in comparisons only)
SPICEY_HD double cx_abs(SpiceyCx z) {
  const double a = fabs(z.re), b = fabs(z.im);
  const double m = a > b ? a : b, q = a > b ? b : a;
  if (m == 0.0) return 0.0;
  const double r = q / m;
  return m * sqrt(1.0 + r * r);
}
template <class Exec>
SPICEY_HD void spicey_ac_dense_solve(Exec &ex, const SpiceyProg &P, const SpiceyAcRun &R, SpiceyCx *Ws, SpiceyCx *A, double *sd, int32_t *si,
                                     int32_t *flags, int64_t slot) {
  const size_t inst = (size_t)(slot / R.n_freq);
  const int64_t fi = slot % R.n_freq;
  const double two_pi = 2 * 3.141592653589793;
  const int T = ex.threads(), n = P.n;
  const size_t ld = (size_t)n + 1;
  AcPhases<Exec> ph{P, R, Ws, flags, T, inst, two_pi * R.freqs[fi]};
  double *red_v = sd;                           // [T] per-thread column maxima
  SpiceyCx *act_f = (SpiceyCx *)(sd + T);       // [n] multipliers of the rows to update
  int32_t *red_i = si, *act_i = si + T, *scal = si + T + n;  // scal: [0] pivot row, [1] active rows
  ex.phase(SPICEY_PH_PRO, [&](int tid) { if (tid == 0) { flags[0] = 0; scal[0] = 0; scal[1] = 0; } });
  ex.phase(SPICEY_PH_B, [&](int tid) {
    ph.s_stamp(tid, true);
    for (size_t i = (size_t)tid; i < (size_t)n * ld; i += (size_t)T) A[i] = SpiceyCx{0.0, 0.0};
  });
  ex.phase(SPICEY_PH_B, [&](int tid) {
    for (int e = tid; e < P.nLU; e += T) A[(size_t)P.ent_ro[e] * ld + (size_t)P.ent_co[e]] = Ws[e];
    for (int r = tid; r < n; r += T) A[(size_t)P.pos_row[r] * ld + (size_t)n] = Ws[P.nLU + r];
  });
  int code = flags[0];  // (an inductor admittance the reference would refuse as well)
  for (int k = 0; k < n && code == 0; k++) {
    ex.phase(SPICEY_PH_U0, [&](int tid) {
      double bv = -1.0;
      int bi = -1;
      for (int i = k + tid; i < n; i += T) {
        const SpiceyCx z = A[(size_t)i * ld + (size_t)k];
        const double v = cx_abs(z);
        if (v > bv) { bv = v; bi = i; }
      }
      red_v[tid] = bv; red_i[tid] = bi;
    });
    ex.phase(SPICEY_PH_U0, [&](int tid) {
      if (tid != 0) return;
      double bv = -1.0;
      int bi = -1;
      const int lim = n - k < T ? n - k : T;
      for (int t = 0; t < lim; t++)
        if (red_v[t] > bv || (red_v[t] == bv && red_i[t] < bi)) { bv = red_v[t]; bi = red_i[t]; }
      scal[0] = bi; scal[1] = 0;
      if (bv < SPICEY_EPS) flags[0] = 1;
    });
    code = flags[0];
    if (code) break;
    const int imax = scal[0];
    if (imax != k)
      ex.phase(SPICEY_PH_U0, [&](int tid) {
        for (int j = tid; j <= n; j += T) {
          // (field by field: a struct temporary here becomes a 16-byte memcpy through a stack slot, i.e. a scratch frame)
          SpiceyCx *pk = A + (size_t)k * ld + (size_t)j, *pi = A + (size_t)imax * ld + (size_t)j;
          const double kr = pk->re, ki = pk->im, ir = pi->re, ii = pi->im;
          pk->re = ir; pk->im = ii;
          pi->re = kr; pi->im = ki;
        }
      });
    ex.phase(SPICEY_PH_U0, [&](int tid) {
      const SpiceyCx pv = A[(size_t)k * ld + (size_t)k];
      const double d = pv.re * pv.re + pv.im * pv.im;
      if (d < SPICEY_EPS) {  // entry.div(pivot) throws for the first row below; the last pivot is divided by on the way back
        if (tid == 0) flags[0] = SPICEY_ERR_COMPLEX_DIV_CODE;
        return;
      }
      for (int i = k + 1 + tid; i < n; i += T) {
        const SpiceyCx en = A[(size_t)i * ld + (size_t)k];
        const SpiceyCx f{(en.re * pv.re + en.im * pv.im) / d, (en.im * pv.re - en.re * pv.im) / d};
        if (cx_abs(f) < SPICEY_EPS) continue;
        const int a = ex.atomic_inc(&scal[1]);
        act_i[a] = i; act_f[a] = f;
      }
    });
    code = flags[0];
    if (code) break;
    const int na = scal[1];
    if (na > 0)
      ex.phase(SPICEY_PH_U0, [&](int tid) {
        const int nw = T >> 6, wv = tid >> 6, lane = tid & 63;
        for (int a = wv; a < na; a += nw) {
          const int i = act_i[a];
          const SpiceyCx f = act_f[a];
          for (int j = k + lane; j <= n; j += 64) {
            const SpiceyCx src = A[(size_t)k * ld + (size_t)j];
            A[(size_t)i * ld + (size_t)j] = cx_sub(A[(size_t)i * ld + (size_t)j], cx_mul(f, src));
          }
        }
      });
  }
  for (int j = n - 1; j >= 0 && code == 0; j--) {
    ex.phase(SPICEY_PH_K0, [&](int tid) {
      if (tid != 0) return;
      const SpiceyCx pv = A[(size_t)j * ld + (size_t)j], s = A[(size_t)j * ld + (size_t)n];
      const double d = pv.re * pv.re + pv.im * pv.im;
      if (d < SPICEY_EPS) { flags[0] = SPICEY_ERR_COMPLEX_DIV_CODE; return; }
      A[(size_t)j * ld + (size_t)n] = SpiceyCx{(s.re * pv.re + s.im * pv.im) / d, (s.im * pv.re - s.re * pv.im) / d};
    });
    code = flags[0];
    if (code) break;
    if (j > 0)
      ex.phase(SPICEY_PH_K0, [&](int tid) {
        const SpiceyCx xj = A[(size_t)j * ld + (size_t)n];
        for (int i = tid; i < j; i += T) {
          const SpiceyCx cf = A[(size_t)i * ld + (size_t)j];
          if (cf.re == 0.0 && cf.im == 0.0) continue;
          A[(size_t)i * ld + (size_t)n] = cx_sub(A[(size_t)i * ld + (size_t)n], cx_mul(cf, xj));
        }
      });
  }
  const int final_code = code;
  ex.phase(SPICEY_PH_K0, [&](int tid) {
    if (final_code == 0)
      for (int kk = tid; kk < n; kk += T) Ws[P.nLU + kk] = A[(size_t)P.pos_col[kk] * ld + (size_t)n];
  });
  ex.phase(SPICEY_PH_Z, [&](int tid) {
    if (final_code == 0) ph.z_record(tid, fi);
    if (tid == 0) R.status[slot] = final_code;
  });
}

// One (instance, frequency) solve by one workgroup.  All control flow is workgroup-uniform.
template <class Exec>
SPICEY_HD void spicey_ac_solve(Exec &ex, const SpiceyProg &P, const SpiceyAcRun &R, SpiceyCx *W, int32_t *flags, int64_t slot) {
  const size_t inst = (size_t)(slot / R.n_freq);
  const int64_t fi = slot % R.n_freq;
  const double two_pi = 2 * 3.141592653589793;
  AcPhases<Exec> ph{P, R, W, flags, ex.threads(), inst, two_pi * R.freqs[fi]};
  ex.phase(SPICEY_PH_PRO, [&](int tid) { if (tid == 0) flags[0] = 0; });
  ex.phase(SPICEY_PH_B, [&](int tid) { ph.s_stamp(tid); });
  for (int l = 0; l < P.nLevels; l++) {
    if (P.lvl_slice[l] == P.lvl_slice[l + 1]) continue;
    ex.phase(SPICEY_PH_U0, [&](int tid) { ph.u_level(tid, l); });
  }
  for (int l = P.nLevels - 1; l >= 0; l--) {
    if (P.bk_lvl_slice[l] == P.bk_lvl_slice[l + 1]) continue;
    ex.phase(SPICEY_PH_K0, [&](int tid) { ph.k_level(tid, l); });
  }
  ex.phase(SPICEY_PH_K0, [&](int tid) { ph.k_scale(tid); });
  const int code = flags[0];
  ex.phase(SPICEY_PH_Z, [&](int tid) {
    if (code == 0) ph.z_record(tid, fi);
    if (tid == 0) R.status[slot] = code;
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// Resident sweep: ONE persistent workgroup runs many frequencies of one instance.  What does not depend on the frequency
// is fetched once and kept in registers — the step the transient kernel took from v1 to v2:
//   * the factor / backward task records (16-byte records with 16-bit indices, the layout of spicey_build_resident: every
//     (wave, slot) chunk belongs to one phase; phases that do not fit the RMAX slots stay streamed from P.rec16);
//   * of every entry e = tid + j T the frequency-independent parts: gr = sum of +-1/R and +-1 (source incidences),
//     gc = sum of +-C, so that the stamp of a frequency is (gr, w gc) — the reference adds the admittances one by one
//     (simulateAC.ts:38-55): the imaginary part differs by the rounding of w (C1 + C2) against w C1 + w C2, ~1e-16 relative.
//     Entries with an inductor stamp keep the exact per-stamp path (guards of :47-55 included).
// The backward substitution uses the ROW-oriented records (x_k = (y_k - sum U_kb x_b) / u_kk), so no scaling phase.
template <int RMAX, int NSE>
struct AcResRegs {
  uint32_t w0[RMAX], w1[RMAX], w2[RMAX], w3[RMAX];
  uint32_t phv[RMAX / 4];  // phase of every slot, one byte each (0xFF = unused); wave-uniform
  double gr[NSE], gc[NSE];
  uint32_t ef;           // 3 bits per entry slot: bit 0 leaf diagonal (invert as stamped), bit 1 has an inductor stamp (generic path), bit 2 = the slot holds an entry
  uint32_t rowsrc;       // bit j: row tid + j T has source contributions (its right-hand side is not zero)
};

template <class Exec, int RMAX, int NSE>
struct AcResident {
  typedef AcResRegs<RMAX, NSE> Regs;
  const SpiceyProg &P;
  const SpiceyResident &Q;
  const SpiceyAcRun &R;
  SpiceyCx *W;
  int32_t *flags;
  int T;
  size_t inst;

  SPICEY_HD void load(int tid, Regs &rr, const AcPhases<Exec> &gen) const {
    for (int s = 0; s < RMAX; s++) {
      const bool have = s < Q.rmax;
      const uint32_t *src = Q.res + ((size_t)(have ? s : 0) * T + tid) * 4;
      rr.w0[s] = have ? src[0] : 0u; rr.w1[s] = have ? src[1] : 0u; rr.w2[s] = have ? src[2] : 0u; rr.w3[s] = have ? src[3] : 0u;
    }
    for (int s4 = 0; s4 < RMAX / 4; s4++) {
      uint32_t pk = 0;
      for (int b = 0; b < 4; b++) {
        const int s = s4 * 4 + b;
        const int ph = s < Q.rmax ? Q.res_phase[(size_t)(tid >> 6) * Q.rmax + s] : -1;
        pk |= (uint32_t)(ph < 0 ? 0xff : (ph & 0xff)) << (8 * b);
      }
      rr.phv[s4] = pk;
    }
    rr.ef = 0u;
    for (int j = 0; j < NSE; j++) {
      const int e = tid + j * T;
      double gr = 0.0, gc = 0.0;
      uint32_t f = 0u;
      if (e < P.nLU) {
        f = 4u | (uint32_t)(P.ent_flag[e] & 1);
        for (uint32_t k = P.stat_ptr[e]; k < P.stat_ptr[e + 1]; k++) {
          const uint32_t ix = P.stat_idx[k], idx = SPICEY_IDX(ix);
          if (idx >= (uint32_t)(P.nR + P.nC) && idx < (uint32_t)(P.nR + P.nC + P.nL)) { f |= 2u; continue; }
          const SpiceyCx g = idx < (uint32_t)P.nR || idx >= (uint32_t)(P.nR + P.nC + P.nL) ? gen.admittance(idx)
                                                                                           : SpiceyCx{0.0, R.C_val[inst * P.nC + (idx - P.nR)]};
          if (ix & SPICEY_NEG) { gr -= g.re; gc -= g.im; } else { gr += g.re; gc += g.im; }
        }
      }
      rr.gr[j] = gr; rr.gc[j] = gc; rr.ef |= f << (3 * j);
    }
    rr.rowsrc = 0u;
    const uint32_t oV = (uint32_t)(P.nC + P.nL);
    for (int j = 0; j < 32; j++) {
      const int r = tid + j * T;
      if (r >= P.n) break;
      for (uint32_t k = P.rhs_ptr[r]; k < P.rhs_ptr[r + 1]; k++) {
        const uint32_t u = SPICEY_IDX(P.rhs_idx[k]);
        if (u >= oV && u < oV + (uint32_t)P.nV) rr.rowsrc |= 1u << j;
      }
    }
  }

  SPICEY_HD void stamp(int tid, const Regs &rr, const AcPhases<Exec> &gen) const {
    const double w = gen.w;
    SPICEY_UNROLL
    for (int j = 0; j < NSE; j++) {
      const uint32_t f = (rr.ef >> (3 * j)) & 7u;
      if (!(f & 4u)) continue;
      const int e = tid + j * T;
      SpiceyCx v{rr.gr[j], w * rr.gc[j]};
      if (f & 2u) {  // an inductor among the stamps: the reference's own sequence for this entry
        v = SpiceyCx{0.0, 0.0};
        for (uint32_t k = P.stat_ptr[e]; k < P.stat_ptr[e + 1]; k++) {
          const uint32_t ix = P.stat_idx[k];
          const SpiceyCx g = gen.admittance(SPICEY_IDX(ix));
          v = (ix & SPICEY_NEG) ? cx_sub(v, g) : cx_add(v, g);
        }
      }
      if (f & 1u) v = gen.pivot_inv(v);
      W[e] = v;
    }
    SPICEY_NOUNROLL
    for (int e = tid + NSE * T; e < P.nLU; e += T) {  // entries beyond the resident capacity
      SpiceyCx v{0.0, 0.0};
      for (uint32_t k = P.stat_ptr[e]; k < P.stat_ptr[e + 1]; k++) {
        const uint32_t ix = P.stat_idx[k];
        const SpiceyCx g = gen.admittance(SPICEY_IDX(ix));
        v = (ix & SPICEY_NEG) ? cx_sub(v, g) : cx_add(v, g);
      }
      if (P.ent_flag[e] & 1) v = gen.pivot_inv(v);
      W[e] = v;
    }
    const uint32_t oV = (uint32_t)(P.nC + P.nL);
    for (int j = 0, r = tid; r < P.n; j++, r += T) {
      SpiceyCx acc{0.0, 0.0};
      if (j >= 32 || ((rr.rowsrc >> j) & 1u)) {
        for (uint32_t k = P.rhs_ptr[r]; k < P.rhs_ptr[r + 1]; k++) {
          const uint32_t ix = P.rhs_idx[k], u = SPICEY_IDX(ix);
          if (u < oV || u >= oV + (uint32_t)P.nV) continue;
          const double *phs = R.vph + (inst * P.nV + (u - oV)) * 2;
          const SpiceyCx v{phs[0], phs[1]};
          acc = (ix & SPICEY_NEG) ? cx_sub(acc, v) : cx_add(acc, v);
        }
      }
      W[P.nLU + r] = acc;
    }
  }

  // one 16-byte record in complex arithmetic (formats: program.h "compact 16-bit task records")
  template <bool ktask>
  SPICEY_HD void exec(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const AcPhases<Exec> &gen) const {
    const uint32_t meta = w0 >> 16;
    if (!(meta & (SPICEY_R16_VALID << 8))) return;
    const uint32_t tgt = w0 & 0xffffu, cnt = meta & 0xffu;
    SpiceyCx acc = W[tgt];
    if (ktask) {
      const uint32_t d = w1 & 0xffffu;
      if (cnt <= 2) {
        if (cnt >= 1) acc = cx_sub(acc, cx_mul(W[w1 >> 16], W[w2 & 0xffffu]));
        if (cnt == 2) acc = cx_sub(acc, cx_mul(W[w2 >> 16], W[w3 & 0xffffu]));
      } else {
        const uint16_t *o = P.ovf16 + w3;
        for (uint32_t j = 0; j < cnt; j++) acc = cx_sub(acc, cx_mul(W[o[2 * j]], W[o[2 * j + 1]]));
      }
      W[tgt] = cx_mul(acc, W[d]);
    } else {
      if (cnt <= 2) {
        if (cnt >= 1) acc = cx_sub(acc, cx_mul(cx_mul(W[w1 & 0xffffu], W[w1 >> 16]), W[w2 & 0xffffu]));
        if (cnt == 2) acc = cx_sub(acc, cx_mul(cx_mul(W[w2 >> 16], W[w3 & 0xffffu]), W[w3 >> 16]));
      } else {
        const uint16_t *o = P.ovf16 + w3;
        for (uint32_t j = 0; j < cnt; j++) acc = cx_sub(acc, cx_mul(cx_mul(W[o[3 * j]], W[o[3 * j + 1]]), W[o[3 * j + 2]]));
      }
      if (meta & (SPICEY_R16_RECIP << 8)) acc = gen.pivot_inv(acc);
      W[tgt] = acc;
    }
  }
  template <bool ktask>
  SPICEY_HD void phase_tasks(int tid, const Regs &rr, int p, const AcPhases<Exec> &gen) const {
    SPICEY_UNROLL
    for (int s = 0; s < RMAX; s++)
      if (SPICEY_UNIFORM((int)((rr.phv[s >> 2] >> ((s & 3) * 8)) & 0xffu)) == p) {
        exec<ktask>(rr.w0[s], rr.w1[s], rr.w2[s], rr.w3[s], gen);
        SPICEY_SCHED_FENCE;  // one task at a time: interleaving the slots' operand fetches only costs registers
      }
    const uint32_t sc = Q.st_cnt[p];
    if (sc) {
      const uint32_t *base = P.rec16 + (size_t)Q.st_first[p] * 4;
      SPICEY_NOUNROLL
      for (uint32_t j = (uint32_t)tid; j < sc; j += (uint32_t)T) {
        const uint32_t *r = base + (size_t)j * 4;
        exec<ktask>(r[0], r[1], r[2], r[3], gen);
      }
    }
  }
};

// All frequencies fi = f0, f0 + fstride, ... of instance `inst` by one workgroup.  Control flow is workgroup-uniform.
template <int RMAX, int NSE, class Exec>
SPICEY_HD void spicey_ac_sweep_resident(Exec &ex, const SpiceyProg &P, const SpiceyResident &Q, const SpiceyAcRun &R, SpiceyCx *W, int32_t *flags,
                                        size_t inst, int64_t f0, int64_t fstride) {
  typedef AcResRegs<RMAX, NSE> Regs;
  const double two_pi = 2 * 3.141592653589793;
  const int T = ex.threads();
  AcResident<Exec, RMAX, NSE> rs{P, Q, R, W, flags, T, inst};
  {
    AcPhases<Exec> gen{P, R, W, flags, T, inst, 0.0};
    ex.phase(SPICEY_PH_PRO, [&](int tid) {
      if (tid == 0) flags[0] = 0;
      rs.load(tid, ex.template regs<Regs>(tid), gen);
    });
  }
  const int nL = P.nLevels;
  for (int64_t fi = f0; fi < R.n_freq; fi += fstride) {
    AcPhases<Exec> gen{P, R, W, flags, T, inst, two_pi * R.freqs[fi]};
    ex.phase(SPICEY_PH_B, [&](int tid) { rs.stamp(tid, ex.template regs<Regs>(tid), gen); });
    for (int p = 0; p < 2 * nL; p++) {
      if (P.ph_cnt[p] == 0) continue;
      if (p < nL) ex.phase(SPICEY_PH_U0, [&](int tid) { rs.template phase_tasks<false>(tid, ex.template regs<Regs>(tid), p, gen); });
      else ex.phase(SPICEY_PH_K0, [&](int tid) { rs.template phase_tasks<true>(tid, ex.template regs<Regs>(tid), p, gen); });
    }
    const int code = flags[0];
    ex.phase(SPICEY_PH_Z, [&](int tid) {
      if (code == 0) gen.z_record(tid, fi);
      if (tid == 0) {
        R.status[inst * (size_t)R.n_freq + (size_t)fi] = code;
        flags[0] = 0;  // for the next frequency (nothing else touches the flag in this phase)
      }
    });
  }
}
