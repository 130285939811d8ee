// tran_exec.h — the per-workgroup transient program interpreter (device code, host-compilable).
//
// One workgroup owns K instances (same topology, interleaved [index][K] in LDS) and runs the whole
// `for step … for iter …` nest of /root/reference/lib/analysis/simulateTRAN.ts:146-238 for them
// inside ONE kernel launch.  Every function below is the body of one barrier-separated PHASE;
// inside a phase the threads are independent (gather form: each thread owns what it writes, reads
// only data finalised in earlier phases), so the same code can be
//   * the HIP kernel (kernels.hip): `phase(f)` = f(threadIdx.x); __syncthreads();
//   * the CPU test emulator (tests/emul): `phase(f)` = for tid in 0..T-1: f(tid)
// The emulator is test infrastructure for the symbolic phase and this interpreter; the product
// path only ever runs the HIP kernel.
//
// Phases per time step (nLev = elimination-tree height):
//   B    dynamic stamps (switch / diode conductances) + right-hand side          simulateTRAN.ts:25-102
//   U_l  l = 0..nLev-2: Schur updates of level l with the forward elimination fused in as an extra
//        column; diagonals that become final are stored as reciprocals (singularity check = solveReal.ts:28)
//   K_l  l = nLev-1..0: backward substitution                                    solveReal.ts:56-72
//   S    switch hysteresis + iteration control (only if the circuit has switches) simulateTRAN.ts:108-128,151-162
//   Z    recording, state update, and the NEXT step's element evaluation          simulateTRAN.ts:164-237
#pragma once
#include <math.h>
#include <stdint.h>

#include "program.h"

#if defined(__HIPCC__)
#define SPICEY_HD __host__ __device__ __forceinline__
#else
#define SPICEY_HD inline
#endif
#ifndef SPICEY_EXP
#define SPICEY_EXP 0  // timing experiments only (tools/exp_build.sh): bit0 no result stores, bit1 no diode section, bit2 no capacitor section, bit3 no parameter loads, bit4 no voltage / resistor section, bit5 no remainder loops
#endif
#ifndef SPICEY_MARK_TID
#define SPICEY_MARK_TID 0
#endif
#if (SPICEY_EXP & 64) && defined(__HIP_DEVICE_COMPILE__)
#define SPICEY_MARK(c, n) do { if ((c).zprof && threadIdx.x == (SPICEY_MARK_TID)) { unsigned long long t_ = clock64(); if ((n) < 15) (c).zprof[n] += t_ - (c).zprof[15]; (c).zprof[15] = t_; } } while (0)
#else
#define SPICEY_MARK(c, n) do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SPICEY_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)  // value is wave-uniform by construction
// Keeps a register-resident packed word packed: without this hipcc hoists the field decode (8+ VGPRs and
// a mask pair per record) out of the time loop and spills.
#define SPICEY_OPAQUE(x) asm volatile("" : "+v"(x))
// Same for wave-uniform values (instance index): per-instance base pointers derived from it are then formed
// inside the phase that needs them instead of living in (spilled) SGPRs across the whole time loop.
#define SPICEY_OPAQUE_S(x) asm volatile("" : "+s"(x))
// wave vote: true if the condition holds in any active lane (a scalar branch: whole waves skip work nobody needs)
#define SPICEY_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
// result streams are written once and never read by the kernel: non-temporal stores keep them from evicting the
// L2-resident program / parameter lines
#if SPICEY_EXP & 1
#define SPICEY_STREAM_STORE(ptr, val) do { if ((val) == 1.2345e-300) __builtin_nontemporal_store((val), (ptr)); } while (0)
#else
#define SPICEY_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif
// diagnostics (SpiceyRun::skip_risk / lin_err): 64-bit integer atomics on global memory; the maximum over a wave by
// cross-lane shuffles (every lane of the wave must arrive: call it outside divergent branches); one lane per wave reports
#define SPICEY_ATOMIC_ADD_U64(p, v) atomicAdd((unsigned long long *)(p), (unsigned long long)(v))
#define SPICEY_ATOMIC_MAX_U64(p, v) atomicMax((unsigned long long *)(p), (unsigned long long)(v))
static __device__ __forceinline__ double spicey_wave_max(double x) {
  for (int off = 32; off > 0; off >>= 1) {
    const double y = __shfl_xor(x, off);
    x = (y > x) ? y : x;
  }
  return x;
}
#define SPICEY_WAVE_MAX(x) spicey_wave_max(x)
#define SPICEY_WAVE_LEADER(tid) (((tid) & 63) == 0)
// cross-lane moves that do not go through the LDS crossbar (used on dependent chains of the dense fronts):
// the value of ONE lane to all (wave-uniform: two v_readlane into scalars) ...
static __device__ __forceinline__ double spicey_readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  asm volatile("" : "+v"(lo), "+v"(hi));  // back into vector registers at once: the kernels that use this have no scalar registers to spare
  return __hiloint2double(hi, lo);
}
// ... and the value of lane q of every quad (4 consecutive lanes) to the quad (DPP quad_perm: a VALU move)
template <int Q>
static __device__ __forceinline__ double spicey_quad_bcast_q(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), Q * 0x55, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), Q * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double spicey_quad_bcast_f64(double v, int q) {
  switch (q & 3) {
    case 0: return spicey_quad_bcast_q<0>(v);
    case 1: return spicey_quad_bcast_q<1>(v);
    case 2: return spicey_quad_bcast_q<2>(v);
    default: return spicey_quad_bcast_q<3>(v);
  }
}
#define SPICEY_NOUNROLL _Pragma("unroll 1")  // thread-strided loops run 1-2 trips: unrolling only costs VGPRs
#define SPICEY_UNROLL _Pragma("unroll")      // small fixed-trip loops over a register array: without it the array is indexed through s_set_gpr_idx
#define SPICEY_SCHED_FENCE __builtin_amdgcn_sched_barrier(0)  // keep the K instances' code from being interleaved
#else
#define SPICEY_NOUNROLL
#define SPICEY_UNROLL
#define SPICEY_SCHED_FENCE
#define SPICEY_UNIFORM(x) (x)
#define SPICEY_OPAQUE(x) (void)(x)
#define SPICEY_OPAQUE_S(x) (void)(x)
#define SPICEY_WAVE_ANY(c) true
#define SPICEY_STREAM_STORE(ptr, val) (*(ptr) = (val))
#define SPICEY_ATOMIC_ADD_U64(p, v) (*(p) += (unsigned long long)(v))
#define SPICEY_ATOMIC_MAX_U64(p, v) do { if ((unsigned long long)(v) > *(p)) *(p) = (unsigned long long)(v); } while (0)
#define SPICEY_WAVE_MAX(x) (x)  // (the emulator runs one thread at a time: every thread reports for itself)
#define SPICEY_WAVE_LEADER(tid) true
#endif

// phase tags (profiling slots, SpiceyRun::prof)
#define SPICEY_PH_PRO 0
#define SPICEY_PH_B 1
#define SPICEY_PH_S 2
#define SPICEY_PH_A 3
#define SPICEY_PH_Z 4
#define SPICEY_PH_U0 8
#define SPICEY_PH_K0 40
#define SPICEY_PH_SLOTS 72

template <int K>
struct WgCtx {
  double *W;     // [nW][K]   L+U entries, then rhs / x'
  double *G;     // hybrid workspace (SpiceyProg::hybrid): leaf-owned entries in global memory, [nLU][K] by entry id; else null
  double *u;     // [nU][K]   vPrev | iPrev | V(t) | diode ieq
  double *gd;    // [nGdyn][K] switch conductances | diode gd
  int32_t *ison; // [nS][K]
  int32_t *flags;  // [0] switched, [1] singular code, [2] singular inst
  uint32_t *tail;  // [tail_n][64][4] task records of the tail phases (v2), or null
#if SPICEY_EXP & 64
  unsigned long long *zprof;  // experiment builds: 16 profiling slots for marks inside B / Z ([15] = last timestamp)
#endif
  int32_t inst[K];
  int32_t valid[K];
};

// 1/x for pivots: hardware reciprocal seed + two Newton steps (<= 1 ulp; the result feeds a 1e-9 parity
// budget, and the reference's own quotient order differs anyway).  The IEEE-exact quotient hipcc emits
// for `1.0 / x` is ~3x longer and sits on the critical path of every factor level.
SPICEY_HD double spicey_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

SPICEY_HD double spicey_max_nan(double a, double b) {  // Math.max semantics
  return (a > b || a != a) ? a : b;
}

// Diode companion model, simulateTRAN.ts:87-98, and (when `want_i`) the recorded current of :214-217,
// which uses the UNCLAMPED junction voltage.  One exp serves both whenever vd lies inside the clamp
// window [-1, 0.8].  The per-diode constants 1/(N VT) and Is/(N VT) are formed once per evaluation
// from Is, N (two divisions); callers on the hot path pass them precomputed.
SPICEY_HD void spicey_diode_k(double vd, double is, double inv_vt, double is_vt, bool want_i, double &gd, double &ieq, double &irec) {
  double vl = vd;
  if (vd > 0.8) vl = 0.8;
  if (vd < -1.0) vl = -1.0;
  const double e = exp(vl * inv_vt);
  const double id = is * (e - 1.0);
  gd = spicey_max_nan(is_vt * e, 1e-12);
  ieq = id - gd * vl;
  irec = id;
  if (want_i && vl != vd) irec = is * (exp(vd * inv_vt) - 1.0);
}
SPICEY_HD void spicey_diode(double vd, double is, double nn, double &gd, double &ieq) {
  const double vt = nn * SPICEY_VT300;
  double irec;
  spicey_diode_k(vd, is, 1.0 / vt, is / vt, false, gd, ieq, irec);
}

SPICEY_HD double spicey_switch_g(int on, double ron, double roff) {  // simulateTRAN.ts:59-61
  const double r = on ? ron : roff;
  return 1.0 / spicey_max_nan(fabs(r), SPICEY_EPS);
}

// ---- diagnostics -------------------------------------------------------------------------------------------------------
// Right after phase B the workspace holds the stamped matrix A (leaf diagonals as reciprocals).  The reference eliminates
// with partial pivoting, so its multiplier for row i at column k is a_ik / max_j |a_jk| (of the matrix as updated so far)
// and `if (Math.abs(f) < EPS) continue` (solveReal.ts:45) SKIPS the row update when that is below 1e-15 — a nonzero
// coupling silently dropped, which a static sparse order does not reproduce (deviations).  This pass counts the
// columns of the STAMPED matrix in which some nonzero entry is below 1e-15 x the column's largest: the first-order
// indicator of that situation (exact for the first pivot; fills and updated entries are not looked at).  One thread per
// column, read-only, no influence on the solve.  `weight` = solves the count stands for (a linear circuit's matrix is
// looked at once, at step 0, for all its steps).
template <int K, bool HYB = false>
SPICEY_HD void spicey_skip_risk(const SpiceyProg &P, const SpiceyRun &R, const WgCtx<K> &c, int tid, int T, unsigned long long weight) {
  SPICEY_NOUNROLL
  for (int col = tid; col < P.n; col += T) {
    const uint32_t j0 = P.col_ptr[col], j1 = P.col_ptr[col + 1];
    for (int k = 0; k < K; k++) {
      if (!c.valid[k]) continue;
      double mx = 0.0, mn = 1.0e308;
      bool any = false;
      for (uint32_t j = j0; j < j1; j++) {
        const uint32_t e = P.col_ent[j];
        const uint32_t id = SPICEY_IDX(e);
        double v;
        if (HYB) {  // hybrid workspace: leaf-owned entries in the global array, the others at their LDS index
          const uint32_t g0 = (uint32_t)P.hyb_g0, nr = (uint32_t)P.nRestore, g2 = (uint32_t)P.hyb_g2;
          if (id < g0 || (id >= nr && id < nr + g2)) v = fabs(c.G[(size_t)id * K + k]);
          else v = fabs(c.W[(size_t)(id - g0 - (id >= nr ? g2 : 0u)) * K + k]);
        } else {
          v = fabs(c.W[(size_t)id * K + k]);
        }
        if (e & SPICEY_TGT_RECIP) v = 1.0 / v;
        if (v != 0.0) { any = true; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
      }
      if (any && mn / mx < SPICEY_EPS) SPICEY_ATOMIC_ADD_U64(R.skip_risk + c.inst[k], weight);  // (a quotient, like the reference's f)
    }
  }
}
// the one-shot linearisation error of a step (SpiceyRun::lin_err): every wave reports the largest |vd(x) - vd_lin| of its diodes
SPICEY_HD void spicey_lin_err_report(const SpiceyRun &R, size_t inst, int64_t step, int tid, double lerr) {
  const double m = SPICEY_WAVE_MAX(lerr);
  if (SPICEY_WAVE_LEADER(tid) && m > 0.0) {
    unsigned long long bits;
    __builtin_memcpy(&bits, &m, 8);
    SPICEY_ATOMIC_MAX_U64(R.lin_err + inst * (size_t)(R.steps + 1) + (size_t)step, bits);
  }
}

template <int K>
struct TranPhases {
  const SpiceyProg &P;
  const SpiceyRun &R;
  WgCtx<K> &c;
  int T;  // threads
  // the diagnostics of SpiceyOptions.diagnostics are compiled into the kernels with K <= 2 only (the 4-instance kernels have
  // no registers to spare: with them the build reports a stack frame); the host keeps K <= 2 when the option is set
  static constexpr bool DIAG = K <= 2;

  SPICEY_HD double volt(int32_t xi, int k) const { return xi < 0 ? 0.0 : c.W[(size_t)xi * K + k]; }

  // ---- prologue -----------------------------------------------------------------------------
  SPICEY_HD void p0_gstat(int tid) const {
    const double dtc = spicey_max_nan(R.dt, SPICEY_EPS);
    for (int k = 0; k < K; k++) {
      if (!c.valid[k]) continue;
      const size_t in = (size_t)c.inst[k];
      double *g = R.gstat + in * P.nGstat;
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nGstat; i += T) {
        double v;
        if (i < P.nR) v = 1.0 / R.R_val[in * P.nR + i];
        else if (i < P.nR + P.nC) v = R.C_val[in * P.nC + (i - P.nR)] / dtc;
        else if (i < P.nR + P.nC + P.nL) v = dtc / R.L_val[in * P.nL + (i - P.nR - P.nC)];
        else v = 1.0;
        g[i] = v;
      }
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nD; i += T) {
        const double vt = R.D_n[in * P.nD + i] * SPICEY_VT300;
        R.dpar[(in * P.nD + i) * 2 + 0] = 1.0 / vt;
        R.dpar[(in * P.nD + i) * 2 + 1] = R.D_is[in * P.nD + i] / vt;
      }
    }
  }
  SPICEY_HD void p1_static(int tid) const {
    for (int k = 0; k < K; k++) {
      if (!c.valid[k]) continue;
      const size_t in = (size_t)c.inst[k];
      const double *g = R.gstat + in * P.nGstat;
      double *sv = R.statv + in * P.nLU;
      SPICEY_NOUNROLL
      for (int e = tid; e < P.nLU; e += T) {
        double v = 0.0;
        for (uint32_t j = P.stat_ptr[e]; j < P.stat_ptr[e + 1]; j++) {
          const uint32_t ix = P.stat_idx[j];
          const double gv = g[SPICEY_IDX(ix)];
          v = (ix & SPICEY_NEG) ? v - gv : v + gv;
        }
        if (P.ent_flag[e] == 1) {  // static leaf diagonal: pre-invert once per run
          if (fabs(v) < SPICEY_EPS) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
          v = 1.0 / v;
        }
        sv[e] = v;
      }
      double *rc = R.rcoef + in * P.nRhsIdx;
      SPICEY_NOUNROLL
      for (int j = tid; j < P.nRhsIdx; j += T) rc[j] = g[P.rhs_cof[j]];
    }
  }
  // evaluate elements from the state entering the run (step 0, iter 0)
  SPICEY_HD void a0_initial(int tid) const {
    const int oL = P.nC, oV = P.nC + P.nL, oD = P.nC + P.nL + P.nV;
    for (int k = 0; k < K; k++) {
      const size_t in = (size_t)c.inst[k];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nC; i += T) c.u[(size_t)i * K + k] = R.C_vprev[in * P.nC + i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nL; i += T) c.u[(size_t)(oL + i) * K + k] = R.L_iprev[in * P.nL + i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nV; i += T) c.u[(size_t)(oV + i) * K + k] = R.src[i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nS; i += T) {
        const int on = R.S_ison[in * P.nS + i];
        c.ison[(size_t)i * K + k] = on;
        c.gd[(size_t)i * K + k] = spicey_switch_g(on, R.S_ron[in * P.nS + i], R.S_roff[in * P.nS + i]);
      }
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nD; i += T) {
        double g, q;
        spicey_diode(R.D_vdprev[in * P.nD + i], R.D_is[in * P.nD + i], R.D_n[in * P.nD + i], g, q);
        c.gd[(size_t)(P.nS + i) * K + k] = g;
        c.u[(size_t)(oD + i) * K + k] = q;
        if (DIAG && R.lin_vd && c.valid[k]) R.lin_vd[in * P.nD + i] = R.D_vdprev[in * P.nD + i];
      }
    }
    static_copy(tid, true);
    if (tid == 0) c.flags[0] = 0;
  }
  SPICEY_HD void static_copy(int tid, bool all = false) const {
    const int ne = all ? P.nLU : P.nRestore;  // entries >= nRestore are never written after the first copy
    for (int k = 0; k < K; k++) {
      const double *sv = R.statv + (size_t)c.inst[k] * P.nLU;
      SPICEY_NOUNROLL
      for (int e = tid; e < ne; e += T) c.W[(size_t)e * K + k] = sv[e];
    }
  }

  // ---- B: dynamic stamps + right-hand side ----------------------------------------------------
  SPICEY_HD void b_stamp(int tid) const {
    if (tid == 0) c.flags[0] = 0;
    SPICEY_NOUNROLL
    for (int t = tid; t < P.nDynEnt; t += T) {
      const uint32_t et = P.dyn_ent[t];
      const uint32_t e = SPICEY_IDX(et);
      const uint32_t j0 = P.dyn_ptr[t], j1 = P.dyn_ptr[t + 1];
      for (int k = 0; k < K; k++) {
        double v = R.statv[(size_t)c.inst[k] * P.nLU + e];
        for (uint32_t j = j0; j < j1; j++) {
          const uint32_t ix = P.dyn_idx[j];
          const double gv = c.gd[(size_t)SPICEY_IDX(ix) * K + k];
          v = (ix & SPICEY_NEG) ? v - gv : v + gv;
        }
        if (et & SPICEY_TGT_RECIP) {
          if (fabs(v) < SPICEY_EPS && c.valid[k]) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
          v = spicey_rcp(v);
        }
        c.W[(size_t)e * K + k] = v;
      }
    }
    SPICEY_NOUNROLL
    for (int r = tid; r < P.n; r += T) {
      const uint32_t j0 = P.rhs_ptr[r], j1 = P.rhs_ptr[r + 1];
      for (int k = 0; k < K; k++) {
        const double *rc = R.rcoef + (size_t)c.inst[k] * P.nRhsIdx;
        double acc = 0.0;
        for (uint32_t j = j0; j < j1; j++) {
          const uint32_t ix = P.rhs_idx[j];
          const double t = rc[j] * c.u[(size_t)SPICEY_IDX(ix) * K + k];
          acc = (ix & SPICEY_NEG) ? acc - t : acc + t;
        }
        c.W[(size_t)(P.nLU + r) * K + k] = acc;
      }
    }
  }

  // ---- U_l: Schur updates of one elimination-tree level ----------------------------------------
  SPICEY_HD void u_level(int tid, int l, bool reuse = false) const {
    const int nw = T >> 6, w = tid >> 6, lane = tid & 63;
    for (uint32_t s = P.lvl_slice[l] + w; s < P.lvl_slice[l + 1]; s += nw) u_slice(s, lane, reuse);
  }
  // the slices of level l that belong to the bins g, g + G, ... (subtree-local levels below the front cut, program.h),
  // dealt to this workgroup's waves in one round-robin over all of them
  SPICEY_HD void u_bins(int tid, int l, int g, int G, bool reuse) const {
    const uint32_t nw = (uint32_t)(T >> 6), w = (uint32_t)(tid >> 6);
    const int lane = tid & 63;
    const uint32_t *bs = P.bin_upd + (size_t)l * (size_t)(P.nBins + 1);
    uint32_t i = 0;
    for (int b = g; b < P.nBins; b += G) {
      const uint32_t s0 = bs[b], s1 = bs[b + 1];
      for (uint32_t s = s0 + (w + nw - i % nw) % nw; s < s1; s += nw) u_slice(s, lane, reuse);
      i += s1 - s0;
    }
  }
  SPICEY_HD void u_slice(uint32_t s, int lane, bool reuse) const {
    {
      const uint32_t t = s * 64 + lane;
      const uint32_t tgt = P.upd_tgt[t];
      if (tgt == SPICEY_TGT_PAD) return;
      const uint32_t cnt = P.upd_cnt[t];
      const uint32_t off = P.upd_slice[s].off + lane;
      const uint32_t ti = SPICEY_IDX(tgt);
      if (reuse && ti < (uint32_t)P.nLU) return;  // reused factorisation: right-hand-side column only
      double acc[K];
      for (int k = 0; k < K; k++) acc[k] = c.W[(size_t)ti * K + k];
      uint32_t j = 0;
      // long product lists (dense fronts of large circuits): 4 products' indices and operands are in flight at once —
      // one dependent L2 round trip per 4 products instead of per product; the summation order is unchanged
      for (; j + 4 <= cnt; j += 4) {
        uint32_t li[4], di[4], ui[4];
        for (int q = 0; q < 4; q++) {
          li[q] = P.upd_pairs[off + ((j + q) * 3 + 0) * 64];
          di[q] = P.upd_pairs[off + ((j + q) * 3 + 1) * 64];
          ui[q] = P.upd_pairs[off + ((j + q) * 3 + 2) * 64];
        }
        double lv[4][K], dv[4][K], uv[4][K];
        for (int q = 0; q < 4; q++)
          for (int k = 0; k < K; k++) {
            lv[q][k] = c.W[(size_t)li[q] * K + k]; dv[q][k] = c.W[(size_t)di[q] * K + k]; uv[q][k] = c.W[(size_t)ui[q] * K + k];
          }
        for (int q = 0; q < 4; q++)
          for (int k = 0; k < K; k++) acc[k] = fma(-(lv[q][k] * dv[q][k]), uv[q][k], acc[k]);
      }
      for (; j < cnt; j++) {
        const uint32_t li = P.upd_pairs[off + (j * 3 + 0) * 64];
        const uint32_t di = P.upd_pairs[off + (j * 3 + 1) * 64];
        const uint32_t ui = P.upd_pairs[off + (j * 3 + 2) * 64];
        for (int k = 0; k < K; k++)
          acc[k] = fma(-(c.W[(size_t)li * K + k] * c.W[(size_t)di * K + k]), c.W[(size_t)ui * K + k], acc[k]);
      }
      if (tgt & SPICEY_TGT_RECIP) {
        for (int k = 0; k < K; k++) {
          if (fabs(acc[k]) < SPICEY_EPS && c.valid[k]) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
          acc[k] = spicey_rcp(acc[k]);
        }
      }
      for (int k = 0; k < K; k++) c.W[(size_t)ti * K + k] = acc[k];
    }
  }

  // ---- K_l: backward substitution, column-oriented: the pivots of level l update the rows below them --------
  SPICEY_HD void k_level(int tid, int l) const {
    const int nw = T >> 6, w = tid >> 6, lane = tid & 63;
    for (uint32_t s = P.bk_lvl_slice[l] + w; s < P.bk_lvl_slice[l + 1]; s += nw) k_slice(s, lane);
  }
  SPICEY_HD void k_bins(int tid, int l, int g, int G) const {  // see u_bins
    const uint32_t nw = (uint32_t)(T >> 6), w = (uint32_t)(tid >> 6);
    const int lane = tid & 63;
    const uint32_t *bs = P.bin_bk + (size_t)l * (size_t)(P.nBins + 1);
    uint32_t i = 0;
    for (int b = g; b < P.nBins; b += G) {
      const uint32_t s0 = bs[b], s1 = bs[b + 1];
      for (uint32_t s = s0 + (w + nw - i % nw) % nw; s < s1; s += nw) k_slice(s, lane);
      i += s1 - s0;
    }
  }
  SPICEY_HD void k_slice(uint32_t s, int lane) const {
    {
      const uint32_t t = s * 64 + lane;
      const uint32_t yi = P.bk_x[t];
      if (yi == SPICEY_TGT_PAD) return;
      const uint32_t cnt = P.bk_cnt[t];
      const uint32_t off = P.bk_slice[s].off + lane;
      double acc[K];
      for (int k = 0; k < K; k++) acc[k] = c.W[(size_t)yi * K + k];
      uint32_t j = 0;
      // (as in u_slice: 4 products' indices, then their operands, in flight together; the order of the sum is unchanged)
      if constexpr (K <= 2)  // (the 4-instance kernels have no registers to spare)
      for (; j + 4 <= cnt; j += 4) {
        uint32_t ki[4], di[4], ui[4];
        for (int q = 0; q < 4; q++) {
          ki[q] = P.bk_pairs[off + ((j + q) * 3 + 0) * 64];
          di[q] = P.bk_pairs[off + ((j + q) * 3 + 1) * 64];
          ui[q] = P.bk_pairs[off + ((j + q) * 3 + 2) * 64];
        }
        double kv[4][K], dv[4][K], uv[4][K];
        for (int q = 0; q < 4; q++)
          for (int k = 0; k < K; k++) {
            kv[q][k] = c.W[(size_t)ki[q] * K + k]; dv[q][k] = c.W[(size_t)di[q] * K + k]; uv[q][k] = c.W[(size_t)ui[q] * K + k];
          }
        for (int q = 0; q < 4; q++)
          for (int k = 0; k < K; k++) acc[k] = fma(-(kv[q][k] * dv[q][k]), uv[q][k], acc[k]);
      }
      for (; j < cnt; j++) {
        const uint32_t ki = P.bk_pairs[off + (j * 3 + 0) * 64];
        const uint32_t di = P.bk_pairs[off + (j * 3 + 1) * 64];
        const uint32_t ui = P.bk_pairs[off + (j * 3 + 2) * 64];
        for (int k = 0; k < K; k++)
          acc[k] = fma(-(c.W[(size_t)ki * K + k] * c.W[(size_t)di * K + k]), c.W[(size_t)ui * K + k], acc[k]);
      }
      for (int k = 0; k < K; k++) c.W[(size_t)yi * K + k] = acc[k];
    }
  }
  // x[i] = y[i] * dinv[i] for every unknown (after the last level)
  SPICEY_HD void k_scale(int tid) const {
    SPICEY_NOUNROLL
    for (int i = tid; i < P.n; i += T) {
      const uint32_t di = P.bk_d[i];
      for (int k = 0; k < K; k++) c.W[(size_t)(P.nLU + i) * K + k] *= c.W[(size_t)di * K + k];
    }
  }

  // ---- S: switch hysteresis (updateSwitchStatesFromSolution, simulateTRAN.ts:108-128) -----------
  SPICEY_HD void s_switches(int tid) const {
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nS; i += T)
      for (int k = 0; k < K; k++) {
        const size_t in = (size_t)c.inst[k];
        const double vctrl = volt(P.S_cp[i], k) - volt(P.S_cn[i], k);
        const int on = c.ison[(size_t)i * K + k];
        int next = on;
        if (on) {
          if (vctrl < R.S_voff[in * P.nS + i]) next = 0;
        } else if (vctrl > R.S_von[in * P.nS + i]) {
          next = 1;
        }
        if (next != on) {
          c.ison[(size_t)i * K + k] = next;
          c.flags[0] = 1;
        }
      }
  }
  // ---- A': re-linearise for iteration >= 1 (diodes from x, simulateTRAN.ts:81-85) ----------------
  SPICEY_HD void a_reiterate(int tid) const {
    const int oD = P.nC + P.nL + P.nV;
    for (int k = 0; k < K; k++) {
      const size_t in = (size_t)c.inst[k];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nS; i += T)
        c.gd[(size_t)i * K + k] = spicey_switch_g(c.ison[(size_t)i * K + k], R.S_ron[in * P.nS + i], R.S_roff[in * P.nS + i]);
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nD; i += T) {
        double g, q;
        const double vd = volt(P.D_a[i], k) - volt(P.D_b[i], k);
        spicey_diode(vd, R.D_is[in * P.nD + i], R.D_n[in * P.nD + i], g, q);
        c.gd[(size_t)(P.nS + i) * K + k] = g;
        c.u[(size_t)(oD + i) * K + k] = q;
        if (DIAG && R.lin_vd && c.valid[k]) R.lin_vd[in * P.nD + i] = vd;
      }
    }
    static_copy(tid);
  }

  // ---- Z: record, update state, evaluate the next step's companions ----------------------------
  SPICEY_HD void z_record(int tid, int64_t step, bool keep_factors = false) const {
    const bool last = step == R.steps;
    const int oL = P.nC, oV = P.nC + P.nL, oD = P.nC + P.nL + P.nV;
    const int cR = 0, cC = P.nR, cL = P.nR + P.nC, cV = cL + P.nL, cS = cV + P.nV, cD = cS + P.nS;
    for (int k = 0; k < K; k++) {
      if (!c.valid[k]) continue;
      const size_t in = (size_t)c.inst[k];
      double *ov = R.out_v + (in * (size_t)(R.steps + 1) + (size_t)step) * P.nOut;
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nOut; i += T) ov[i] = volt(P.out_x[i], k);
      const bool cur = R.out_i != nullptr;
      double *oi = cur ? R.out_i + (in * (size_t)(R.steps + 1) + (size_t)step) * P.nCur : nullptr;
      const double *g = R.gstat + in * P.nGstat;
      if (cur)
        SPICEY_NOUNROLL
        for (int i = tid; i < P.nR; i += T) oi[cR + i] = (volt(P.R_a[i], k) - volt(P.R_b[i], k)) * g[i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nC; i += T) {
        const double dv = volt(P.C_a[i], k) - volt(P.C_b[i], k);
        if (cur) oi[cC + i] = g[P.nR + i] * (dv - c.u[(size_t)i * K + k]);
        c.u[(size_t)i * K + k] = dv;
        if (last) R.C_vprev[in * P.nC + i] = dv;
      }
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nL; i += T) {
        const double dv = volt(P.L_a[i], k) - volt(P.L_b[i], k);
        const double il = g[P.nR + P.nC + i] * dv + c.u[(size_t)(oL + i) * K + k];
        if (cur) oi[cL + i] = il;
        c.u[(size_t)(oL + i) * K + k] = il;
        if (last) R.L_iprev[in * P.nL + i] = il;
      }
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nV; i += T) {
        if (cur) oi[cV + i] = c.W[(size_t)P.V_x[i] * K + k];
        if (!last) c.u[(size_t)(oV + i) * K + k] = R.src[(size_t)(step + 1) * P.nV + i];
      }
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nS; i += T) {
        const int on = c.ison[(size_t)i * K + k];
        const double gs = spicey_switch_g(on, R.S_ron[in * P.nS + i], R.S_roff[in * P.nS + i]);
        if (cur) oi[cS + i] = (volt(P.S_a[i], k) - volt(P.S_b[i], k)) * gs;
        c.gd[(size_t)i * K + k] = gs;
        if (last) R.S_ison[in * P.nS + i] = on;
      }
      double lerr = 0.0;
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nD; i += T) {
        const double vd = volt(P.D_a[i], k) - volt(P.D_b[i], k);
        const double is = R.D_is[in * P.nD + i];
        const double *dp = R.dpar + (in * P.nD + i) * 2;  // {1/(N VT), Is/(N VT)} from the prologue
        double gg, q, irec;
        spicey_diode_k(vd, is, dp[0], dp[1], cur, gg, q, irec);
        if (cur) oi[cD + i] = irec;  // unclamped, simulateTRAN.ts:214-217
        c.gd[(size_t)(P.nS + i) * K + k] = gg;
        c.u[(size_t)(oD + i) * K + k] = q;
        if (last) R.D_vdprev[in * P.nD + i] = vd;
        if (DIAG && R.lin_vd) {  // diagnostics: how far the junction moved from where this solve had it linearised
          const double e = fabs(vd - R.lin_vd[in * P.nD + i]);
          lerr = e > lerr ? e : lerr;
          R.lin_vd[in * P.nD + i] = vd;
        }
      }
      if (DIAG && R.lin_err) spicey_lin_err_report(R, in, step, tid, lerr);
    }
    if (!keep_factors) static_copy(tid);  // a linear circuit keeps the factors of step 0 in W
  }
};

// ---------------------------------------------------------------------------------------------
// v2: register-resident program.  The factor / backward task lists are step-invariant, so every
// thread keeps its share as RMAX 16-byte records in VGPRs for the whole transient (the register file,
// 512 KB per CU, is the largest low-latency store of the chip); only phases that do not fit are
// streamed from L2.  Each (wave, slot) chunk belongs to one phase, so dispatch is wave-uniform.
// Register arrays that are indexed with a wave-uniform RUNTIME index (the slot cursor): as native vector
// types hipcc addresses them through the VGPR index register (s_set_gpr_idx), O(1), instead of a compare
// chain over all slots or a scratch round trip.
#if defined(__clang__)
template <int N> struct U32Vec { typedef uint32_t type __attribute__((ext_vector_type(N))); };
#else
template <int N> struct U32Arr { uint32_t v[N]; uint32_t &operator[](int i) { return v[i]; } const uint32_t &operator[](int i) const { return v[i]; } };
template <int N> struct U32Vec { typedef U32Arr<N> type; };  // (host build: a plain array; gcc's vector types want a power of two)
#endif

template <int K, int RMAX, int NSV, int NEL>
struct ResRegs {
  // factor / backward task records, one 16-byte record per slot, word-major; the slots of a wave are sorted
  // by phase, `phv` holds the phase id of every slot (one byte each, 0xFF = unused), `cursor` the next slot
  typename U32Vec<RMAX>::type w0, w1, w2, w3;
  typename U32Vec<(RMAX + 3) / 4>::type phv;
  int32_t cursor;
  // entries with dynamic stamps are numbered first: only the first NDD slots can hold one and need a descriptor
  static constexpr int NDD = NSV == 6 ? 2 : NSV / 2;
  double sv[NSV][K];    // static part of the entries this thread re-stamps (e = tid + j T)
  uint32_t dd[NDD];     // dynamic-stamp descriptors of the first NDD of them
  uint32_t rhs[NEL][2]; // right-hand-side descriptors of rows tid + j T
  uint32_t eR[NEL], eC[NEL], eD[NEL], ox[NEL];  // packed terminals of elements tid + j T; W index of output tid + j T (ox[0] >> 16: source tid's branch current)
  double vprev[NEL][K]; // vPrev of capacitors tid + j T (simulateTRAN.ts:221-225), exact
  // Z's element parameters {1/R, C/dt, Is, 1/(N VT), Is/(N VT)} of items tid + j T and the next source value:
  // fetched at the end of the last backward phase so that the L2 round trip (~1900 cycles measured) overlaps that
  // phase's barrier; live only from there to Z (K == 1 geometries)
  double pf[NEL][5];
};

// One task.  For the common inline case (<= 2 products) ALL operands are fetched up front — unused index fields
// are 0, a valid address — and the unused products are masked by selects: one LDS round trip per task instead of
// one per product (the dependent ds_read -> wait -> fma chains dominated the small phases).
// OPG (hybrid workspace, SpiceyProg::hybrid): the phase eliminates / back-substitutes the LEAVES of the elimination tree —
// the pivot's own entries (L, reciprocal diagonal, U) are read from the global array c.G by entry id, every target and
// every right-hand-side / solution operand from LDS as always (`xoff` = first LDS index of the right-hand side: the third
// operand of a right-hand-side task is y_k, not an entry).
template <int K, bool KTASK, bool OPG = false>
SPICEY_HD void spicey_exec_rec16(const WgCtx<K> &c, const uint16_t *ovf, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                 uint32_t keep_from = 0u, uint32_t xoff = 0u) {
  const uint32_t meta = w0 >> 16;
  if (!(meta & (SPICEY_R16_VALID << 8))) return;
  const uint32_t tgt = w0 & 0xffffu, cnt = meta & 0xffu;
  // a reused factorisation (linear circuit, step > 0) runs only the right-hand-side column of the factor tasks:
  // keep_from = first right-hand-side index then, 0 otherwise
  if (!KTASK && tgt < keep_from) return;
  const double *E = OPG ? c.G : c.W;  // where the pivot's own entries are
  double acc[K];
  if (KTASK) {
    const uint32_t d = w1 & 0xffffu;
    if (cnt <= 2) {
      const uint32_t u0 = w1 >> 16, x0 = w2 & 0xffffu, u1 = w2 >> 16, x1 = w3 & 0xffffu;
      const bool two = SPICEY_WAVE_ANY(cnt == 2);  // tasks are sorted by count: most waves are uniform
      double a0[K], b0[K], a1[K], b1[K], dv[K];
      for (int k = 0; k < K; k++) {
        acc[k] = c.W[(size_t)tgt * K + k];
        a0[k] = E[(size_t)u0 * K + k]; b0[k] = c.W[(size_t)x0 * K + k];
        dv[k] = E[(size_t)d * K + k];
      }
      if (two)
        for (int k = 0; k < K; k++) { a1[k] = E[(size_t)u1 * K + k]; b1[k] = c.W[(size_t)x1 * K + k]; }
      for (int k = 0; k < K; k++) {  // explicit fma: the same rounding in every interpreter and geometry
        const double s0 = fma(-a0[k], b0[k], acc[k]);
        acc[k] = cnt >= 1 ? s0 : acc[k];
        if (two) {
          const double s1 = fma(-a1[k], b1[k], acc[k]);
          acc[k] = cnt == 2 ? s1 : acc[k];
        }
        acc[k] *= dv[k];
      }
    } else {
      for (int k = 0; k < K; k++) acc[k] = c.W[(size_t)tgt * K + k];
      const uint16_t *o = ovf + w3;
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t u = o[2 * j], x = o[2 * j + 1];
        for (int k = 0; k < K; k++) acc[k] = fma(-E[(size_t)u * K + k], c.W[(size_t)x * K + k], acc[k]);
      }
      for (int k = 0; k < K; k++) acc[k] *= E[(size_t)d * K + k];
    }
    for (int k = 0; k < K; k++) c.W[(size_t)tgt * K + k] = acc[k];
  } else {
    // (hybrid: the third operand is an entry of the pivot's U row — global — for a matrix target, y_k — LDS — for a
    // right-hand-side target)
    const bool third_lds = !OPG || tgt >= xoff;
    if (cnt <= 2) {
      const uint32_t l0 = w1 & 0xffffu, d0 = w1 >> 16, u0 = w2 & 0xffffu, l1 = w2 >> 16, d1 = w3 & 0xffffu, u1 = w3 >> 16;
      const bool two = SPICEY_WAVE_ANY(cnt == 2);
      double p0[K], q0[K], r0[K], p1[K], q1[K], r1[K];
      for (int k = 0; k < K; k++) {
        acc[k] = c.W[(size_t)tgt * K + k];
        p0[k] = E[(size_t)l0 * K + k]; q0[k] = E[(size_t)d0 * K + k];
        r0[k] = (!OPG || third_lds) ? c.W[(size_t)u0 * K + k] : c.G[(size_t)u0 * K + k];
      }
      if (two)
        for (int k = 0; k < K; k++) {
          p1[k] = E[(size_t)l1 * K + k]; q1[k] = E[(size_t)d1 * K + k];
          r1[k] = (!OPG || third_lds) ? c.W[(size_t)u1 * K + k] : c.G[(size_t)u1 * K + k];
        }
      for (int k = 0; k < K; k++) {
        const double s0 = fma(-(p0[k] * q0[k]), r0[k], acc[k]);
        acc[k] = cnt >= 1 ? s0 : acc[k];
        if (two) {
          const double s1 = fma(-(p1[k] * q1[k]), r1[k], acc[k]);
          acc[k] = cnt == 2 ? s1 : acc[k];
        }
      }
    } else {
      for (int k = 0; k < K; k++) acc[k] = c.W[(size_t)tgt * K + k];
      const uint16_t *o = ovf + w3;
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t l = o[3 * j], d = o[3 * j + 1], u = o[3 * j + 2];
        for (int k = 0; k < K; k++) {
          const double uv = (!OPG || third_lds) ? c.W[(size_t)u * K + k] : c.G[(size_t)u * K + k];
          acc[k] = fma(-(E[(size_t)l * K + k] * E[(size_t)d * K + k]), uv, acc[k]);
        }
      }
    }
    if (meta & (SPICEY_R16_RECIP << 8)) {
      for (int k = 0; k < K; k++) {
        if (fabs(acc[k]) < SPICEY_EPS && c.valid[k]) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
        acc[k] = spicey_rcp(acc[k]);
      }
    }
    for (int k = 0; k < K; k++) c.W[(size_t)tgt * K + k] = acc[k];
  }
}

// One ROW record of a factor phase (program.h: fus16): the targets a_ii, y_i and the (at most two) fills of row i from its
// (at most two) pivots of this level, sharing the multipliers -(L_ik d_k).  The products and their order are those of the
// generic tasks it stands for.  rhs_only: a reused factorisation updates y_i alone.
template <int K, bool OPG = false>
SPICEY_HD void spicey_exec_row16(const WgCtx<K> &c, const uint32_t *w, bool rhs_only) {
  const double *E = OPG ? c.G : c.W;  // (hybrid workspace: the pivots' own entries L_ik, d_k, U_ki, U_k,o come from the global array)
  const uint32_t meta = w[0] >> 16;
  if (!(meta & (SPICEY_R16_VALID << 8))) return;
  const uint32_t iaa = w[0] & 0xffffu, iy = w[1] & 0xffffu;
  const uint32_t l0 = w[1] >> 16, d0 = w[2] & 0xffffu, u0 = w[2] >> 16, y0 = w[3] & 0xffffu, f0 = w[3] >> 16, t0 = w[4] & 0xffffu;
  const uint32_t l1 = w[4] >> 16, d1 = w[5] & 0xffffu, u1 = w[5] >> 16, y1 = w[6] & 0xffffu, f1 = w[6] >> 16, t1 = w[7] & 0xffffu;
  const bool two = (meta & 3u) == 2u, o0 = (meta >> 4) & 1u, o1 = (meta >> 5) & 1u;
  for (int k = 0; k < K; k++) {
    // every operand in one LDS round trip (an unused second pivot / fill: index 0, a valid address; results masked)
    double aii = c.W[(size_t)iaa * K + k], yi = c.W[(size_t)iy * K + k];
    const double vl0 = E[(size_t)l0 * K + k], vd0 = E[(size_t)d0 * K + k], vy0 = c.W[(size_t)y0 * K + k], vu0 = E[(size_t)u0 * K + k];
    const double vl1 = E[(size_t)l1 * K + k], vd1 = E[(size_t)d1 * K + k], vy1 = c.W[(size_t)y1 * K + k], vu1 = E[(size_t)u1 * K + k];
    const double vf0 = E[(size_t)f0 * K + k], vt0 = c.W[(size_t)t0 * K + k], vf1 = E[(size_t)f1 * K + k], vt1 = c.W[(size_t)t1 * K + k];
    const double m0 = -(vl0 * vd0), m1 = -(vl1 * vd1);
    yi = fma(m0, vy0, yi);
    aii = fma(m0, vu0, aii);
    const double y2 = fma(m1, vy1, yi), a2 = fma(m1, vu1, aii);
    yi = two ? y2 : yi;
    aii = two ? a2 : aii;
    c.W[(size_t)iy * K + k] = yi;
    if (!rhs_only) {
      if (o0) c.W[(size_t)t0 * K + k] = fma(m0, vf0, vt0);
      if (two && o1) c.W[(size_t)t1 * K + k] = fma(m1, vf1, vt1);
      if (meta & (SPICEY_R16_RECIP << 8)) {
        if (fabs(aii) < SPICEY_EPS && c.valid[k]) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
        aii = spicey_rcp(aii);
      }
      c.W[(size_t)iaa * K + k] = aii;
    }
  }
}

// Two row records of the leaves' factor phase under the hybrid workspace (OPG): the global operands of BOTH are fetched first,
// then each record runs exactly as spicey_exec_row16 would (same products, same order: the rows of one level are independent).
template <int K>
SPICEY_HD void spicey_exec_row16_x2(const WgCtx<K> &c, const uint32_t *wa, const uint32_t *wb, bool rhs_only) {
  static_assert(K == 1, "hybrid workspace: one instance per workgroup");
  const uint32_t *w2[2] = {wa, wb};
  double gl[2][2], gdg[2][2], gu[2][2], gf[2][2];
  SPICEY_UNROLL
  for (int r = 0; r < 2; r++) {
    const uint32_t *w = w2[r];
    const uint32_t l0 = w[1] >> 16, d0 = w[2] & 0xffffu, u0 = w[2] >> 16, f0 = w[3] >> 16;
    const uint32_t l1 = w[4] >> 16, d1 = w[5] & 0xffffu, u1 = w[5] >> 16, f1 = w[6] >> 16;
    gl[r][0] = c.G[l0]; gdg[r][0] = c.G[d0]; gu[r][0] = c.G[u0]; gf[r][0] = c.G[f0];
    gl[r][1] = c.G[l1]; gdg[r][1] = c.G[d1]; gu[r][1] = c.G[u1]; gf[r][1] = c.G[f1];
  }
  SPICEY_UNROLL
  for (int r = 0; r < 2; r++) {
    const uint32_t *w = w2[r];
    const uint32_t meta = w[0] >> 16;
    if (!(meta & (SPICEY_R16_VALID << 8))) continue;
    const uint32_t iaa = w[0] & 0xffffu, iy = w[1] & 0xffffu;
    const uint32_t y0 = w[3] & 0xffffu, t0 = w[4] & 0xffffu, y1 = w[6] & 0xffffu, t1 = w[7] & 0xffffu;
    const bool two = (meta & 3u) == 2u, o0 = (meta >> 4) & 1u, o1 = (meta >> 5) & 1u;
    double aii = c.W[iaa], yi = c.W[iy];
    const double vy0 = c.W[y0], vy1 = c.W[y1], vt0 = c.W[t0], vt1 = c.W[t1];
    const double m0 = -(gl[r][0] * gdg[r][0]), m1 = -(gl[r][1] * gdg[r][1]);
    yi = fma(m0, vy0, yi);
    aii = fma(m0, gu[r][0], aii);
    const double y2 = fma(m1, vy1, yi), a2 = fma(m1, gu[r][1], aii);
    yi = two ? y2 : yi;
    aii = two ? a2 : aii;
    c.W[iy] = yi;
    if (!rhs_only) {
      if (o0) c.W[t0] = fma(m0, gf[r][0], vt0);
      if (two && o1) c.W[t1] = fma(m1, gf[r][1], vt1);
      if (meta & (SPICEY_R16_RECIP << 8)) {
        if (fabs(aii) < SPICEY_EPS && c.valid[0]) { c.flags[1] = 1; c.flags[2] = c.inst[0]; }
        aii = spicey_rcp(aii);
      }
      c.W[iaa] = aii;
    }
  }
}

template <int K, int RMAX, int NSV, int NEL, bool KTASK, bool OPG = false>
SPICEY_HD void spicey_uk_phase(const SpiceyProg &P, const SpiceyResident &Q, const WgCtx<K> &c, ResRegs<K, RMAX, NSV, NEL> &rr, int tid,
                               int T, int p, bool streamed, bool reuse = false) {
  const uint32_t xoff = (uint32_t)P.xoff;  // first LDS index of the right-hand side (= nLU without the hybrid layout)
  const uint32_t keep_from = (!KTASK && reuse) ? xoff : 0u;
  if (RMAX <= 8) {
    // few slots: a static compare chain (scalar compares on the wave-uniform phase bytes).  Measured faster than
    // both indexed register access and a binary decision tree on a slot cursor (11.8 vs 16.0 / 15.2 us per step).
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < RMAX; s++) {
      const int sp = SPICEY_UNIFORM((int)((rr.phv[s >> 2] >> ((s & 3) * 8)) & 0xffu));
      if (sp == p) {
        uint32_t w0 = rr.w0[s], w1 = rr.w1[s], w2 = rr.w2[s], w3 = rr.w3[s];
        SPICEY_OPAQUE(w0); SPICEY_OPAQUE(w1); SPICEY_OPAQUE(w2); SPICEY_OPAQUE(w3);
        if (!KTASK && s + 1 < RMAX && SPICEY_UNIFORM((int)((rr.phv[(s + 1) >> 2] >> (((s + 1) & 3) * 8)) & 0xffu)) == 0xFE) {  // a chunk of row records: this slot + its continuation
          uint32_t w[8] = {w0, w1, w2, w3, rr.w0[s + 1 < RMAX ? s + 1 : s], rr.w1[s + 1 < RMAX ? s + 1 : s], rr.w2[s + 1 < RMAX ? s + 1 : s], rr.w3[s + 1 < RMAX ? s + 1 : s]};
          SPICEY_OPAQUE(w[4]); SPICEY_OPAQUE(w[5]); SPICEY_OPAQUE(w[6]); SPICEY_OPAQUE(w[7]);
          spicey_exec_row16<K, OPG>(c, w, reuse);
        } else {
          spicey_exec_rec16<K, KTASK, OPG>(c, P.ovf16, w0, w1, w2, w3, keep_from, xoff);
        }
      }
    }
  } else {
    // resident chunks of this wave that belong to phase p: consecutive slots starting at the cursor;
    // the slot index is wave-uniform, the records are fetched through the VGPR index register
    int q = rr.cursor;
    while (q < RMAX) {
      const uint32_t pw = rr.phv[q >> 2];
      const int sp = SPICEY_UNIFORM((int)((pw >> ((q & 3) * 8)) & 0xffu));
      if (sp != p) break;
      uint32_t w0 = rr.w0[q], w1 = rr.w1[q], w2 = rr.w2[q], w3 = rr.w3[q];
      SPICEY_OPAQUE(w0); SPICEY_OPAQUE(w1); SPICEY_OPAQUE(w2); SPICEY_OPAQUE(w3);
      if (!KTASK && q + 1 < RMAX && SPICEY_UNIFORM((int)((rr.phv[(q + 1) >> 2] >> (((q + 1) & 3) * 8)) & 0xffu)) == 0xFE) {  // a chunk of row records: this slot + its continuation
        uint32_t w[8] = {w0, w1, w2, w3, rr.w0[q + 1], rr.w1[q + 1], rr.w2[q + 1], rr.w3[q + 1]};
        SPICEY_OPAQUE(w[4]); SPICEY_OPAQUE(w[5]); SPICEY_OPAQUE(w[6]); SPICEY_OPAQUE(w[7]);
        spicey_exec_row16<K, OPG>(c, w, reuse);
        q += 2;
      } else {
        spicey_exec_rec16<K, KTASK, OPG>(c, P.ovf16, w0, w1, w2, w3, keep_from, xoff);
        q++;
      }
    }
    rr.cursor = q;
  }
  if (!streamed) return;
  // one 32-byte descriptor says where the phase's records are (SpiceyResident::st_desc)
  const uint32_t *dsc = Q.st_desc + (size_t)p * 8;
  const uint32_t d_rows = dsc[0], d_first = dsc[1], d_cnt = dsc[2], d_rhs = dsc[3], d_rfirst = dsc[4], d_rcnt = dsc[5], d_rrhs = dsc[6];
  uint32_t sc = (!KTASK && reuse) ? d_rhs : d_cnt;  // right-hand-side tasks lead every factor phase
  const uint32_t *base = P.rec16 + (size_t)d_first * 4;
  if (!KTASK && sc && d_rows) {
    // the phase's row-record encoding: its 32-byte row records (one per thread on the chains this is for), then the few
    // generic records of rows that do not fit the pattern
    const uint32_t npair = d_cnt;
    const uint32_t *pb = P.fus16 + (size_t)d_first * 4;
    if constexpr (OPG && NEL >= 2) {
      // (hybrid workspace: the leaves' own entries come from L2 — two row records at a time, both fetched before either is
      // executed, so that the operand loads of the second are in flight under the first; the 1024-thread build — NEL = 1 —
      // has half the records per thread and no registers for a second one)
      SPICEY_NOUNROLL
      for (uint32_t j = (uint32_t)tid; j < npair; j += 2u * (uint32_t)T) {
        const uint32_t j2 = j + (uint32_t)T;
        const bool two = j2 < npair;
        uint32_t wa[8], wb[8];
        for (int i = 0; i < 8; i++) { wa[i] = pb[(size_t)j * 8 + i]; wb[i] = pb[(size_t)(two ? j2 : j) * 8 + i]; }
        if (!two) wb[0] = 0u;  // (no VALID flag: nothing runs)
        spicey_exec_row16_x2<K>(c, wa, wb, reuse);
      }
    } else
    SPICEY_NOUNROLL
    for (uint32_t j = (uint32_t)tid; j < npair; j += (uint32_t)T) {
      uint32_t w[8];
      for (int i = 0; i < 8; i++) w[i] = pb[(size_t)j * 8 + i];
      spicey_exec_row16<K, OPG>(c, w, reuse);
    }
    base = P.fus16 + (size_t)d_rfirst * 4;
    sc = reuse ? d_rrhs : d_rcnt;
  }
  if (sc) {
    // streamed phase (did not fit the resident slots): double-buffered — the next record's L2 fetch is in flight
    // while the current task executes.  (Fetching 4 records up front was measured slower: +16 live VGPRs pushed
    // the 1024-thread kernel to its 128-register cap.)
    uint32_t j = (uint32_t)tid;
    if (j < sc) {
      const uint32_t *r = base + (size_t)j * 4;
      uint32_t c0 = r[0], c1 = r[1], c2 = r[2], c3 = r[3];
      for (;;) {
        const uint32_t jn = j + (uint32_t)T;
        const bool more = jn < sc;
        const uint32_t *rn = base + (size_t)(more ? jn : j) * 4;
        const uint32_t n0 = rn[0], n1 = rn[1], n2 = rn[2], n3 = rn[3];
        spicey_exec_rec16<K, KTASK, OPG>(c, P.ovf16, c0, c1, c2, c3, keep_from, xoff);
        if (!more) break;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        j = jn;
      }
    }
  }
}

// v2 versions of the B and Z phases: everything step-invariant that a thread needs (static entry values,
// stamp / right-hand-side descriptors, element terminals, vPrev) sits in its registers; items beyond the
// resident capacity (entries >= NSV*T, rows / elements >= T) take the streamed remainder loops.
// Difference to v1: u[c] holds the capacitor companion CURRENT gc*vPrev (so the right-hand side is a
// pure +-1 gather, stampCurrentReal.ts:12-13) and the exact vPrev lives in a register.
// HYB: the hybrid workspace layout (program.h, SpiceyProg::hybrid) — entry ids below hyb_g0 and in [nRestore, nRestore + hyb_g2)
// are leaf-owned and live in the global array c.G, all others in LDS at id - hyb_g0 (- hyb_g2 above nRestore); the
// right-hand side starts at LDS index P.xoff; c.u / c.gd point to global memory.
template <int K, int RMAX, int NSV, int NEL, bool HYB = false>
struct TranPhases2 {
  const SpiceyProg &P;
  const SpiceyRun &R;
  WgCtx<K> &c;
  int T;
  // where entry `e` (an id below nRestore: what phase B re-stamps) is stored
  SPICEY_HD void put_entry(uint32_t e, int k, double v) const {
    if (HYB) {
      if (e < (uint32_t)P.hyb_g0) c.G[(size_t)e * K + k] = v;
      else c.W[(size_t)(e - (uint32_t)P.hyb_g0) * K + k] = v;
    } else {
      c.W[(size_t)e * K + k] = v;
    }
  }
  // Which of the beyond-resident-capacity loops of B / Z have any work (wave-uniform, fixed for the run).  On the
  // circuits the resident geometry is sized for they are all empty, yet each one costs a bound fetch, address
  // arithmetic and a branch: ~1200 cycles per step in Z alone before they were put behind one test.
  uint32_t brem, zrem;
  typedef ResRegs<K, RMAX, NSV, NEL> Regs;
  // the diagnostics of SpiceyOptions.diagnostics are compiled into every geometry but the two-workgroups-per-CU one (NSV = 6:
  // 128 VGPRs and nothing to spare — with them that kernel spills, which the build refuses); the host keeps a handle with
  // the option out of that geometry
  static constexpr bool DIAG = NSV != 6 && !HYB;  // (nor into the hybrid-workspace build, for the same reason)
  // hybrid builds: items of a beyond-resident loop whose loads are in flight together (the 1024-thread build has 128 registers)
  static constexpr int BW = NEL >= 2 ? 4 : 2;   // (phase Z)
  static constexpr int BWB = BW;                 // (phase B; four at a time in the 1024-thread build compiled — 126 registers — and was 5 % slower per step)
  SPICEY_HD void set_remainders() {
    brem = (P.nRestore > NSV * T ? 1u : 0u) | (P.nDynX > 0 ? 2u : 0u) | (P.n > NEL * T ? 4u : 0u) | (P.nRowX > 0 ? 8u : 0u) |
           (P.nDynEnt > Regs::NDD * T ? 16u : 0u);
    zrem = (P.nOut > NEL * T ? 1u : 0u) | (P.nR > NEL * T ? 2u : 0u) | (P.nC > NEL * T ? 4u : 0u) | (P.nL > 0 ? 8u : 0u) |
           (P.nV > T ? 16u : 0u) | (P.nS > 0 ? 32u : 0u) | (P.nD > NEL * T ? 64u : 0u);
    brem = (uint32_t)SPICEY_UNIFORM((int)brem);
    zrem = (uint32_t)SPICEY_UNIFORM((int)zrem);
  }

  // branch-free: ground (0xFFFF) reads slot 0, a valid address, and is masked afterwards, so that the reads of
  // several elements can be issued back to back instead of one exec-masked block each
  SPICEY_HD double volt16(uint32_t xi, int k) const {
    const double v = c.W[(size_t)(xi == 0xFFFFu ? 0u : xi) * K + k];
    return xi == 0xFFFFu ? 0.0 : v;
  }
  SPICEY_HD double dv16(uint32_t ab, int k) const { return volt16(ab & 0xFFFFu, k) - volt16(ab >> 16, k); }

  SPICEY_HD void load_resident(int tid, const SpiceyResident &Q, Regs &rr) const {
    for (int s = 0; s < RMAX; s++) {
      const bool have = s < Q.rmax;
      const uint32_t *src = Q.res + ((size_t)(have ? s : 0) * T + tid) * 4;
      rr.w0[s] = have ? src[0] : 0u; rr.w1[s] = have ? src[1] : 0u; rr.w2[s] = have ? src[2] : 0u; rr.w3[s] = have ? src[3] : 0u;
    }
    for (int s4 = 0; s4 < (RMAX + 3) / 4; s4++) {
      uint32_t pk = 0;
      for (int b = 0; b < 4; b++) {
        const int s = s4 * 4 + b;
        const int ph = s < Q.rmax && s < RMAX ? Q.res_phase[(size_t)(tid >> 6) * Q.rmax + s] : -1;
        pk |= (uint32_t)(ph < 0 ? 0xff : (ph & 0xff)) << (8 * b);
      }
      rr.phv[s4] = SPICEY_UNIFORM((int)pk);
    }
    rr.cursor = 0;
    for (int j = 0; j < NEL; j++) {
      const int i = tid + j * T;
      rr.rhs[j][0] = i < P.n ? P.row_desc[(size_t)i * 2] : 0u;
      rr.rhs[j][1] = i < P.n ? P.row_desc[(size_t)i * 2 + 1] : 0xFFFFFFFFu;  // 0xFFFFFFFF = not a resident row
      rr.eR[j] = i < P.nR ? P.R_ab[i] : 0xFFFFFFFFu;
      rr.eC[j] = i < P.nC ? P.C_ab[i] : 0xFFFFFFFFu;
      rr.eD[j] = i < P.nD ? P.D_ab[i] : 0xFFFFFFFFu;
      rr.ox[j] = i < P.nOut ? (P.out_x[i] < 0 ? 0xFFFFu : (uint32_t)P.out_x[i]) : 0xFFFFu;
    }
    if (tid < P.nV) rr.ox[0] |= (uint32_t)P.V_x[tid] << 16;  // upper half of ox[0]: W index of the branch current of source tid
  }
  // after p1_static: static entry values into registers; elements from the state entering the run
  SPICEY_HD void a0_initial(int tid, Regs &rr) const {
    const int oL = P.nC, oV = P.nC + P.nL, oD = P.nC + P.nL + P.nV;
    for (int j = 0; j < NSV; j++) {
      const int e = tid + j * T;
      if (j < Regs::NDD) rr.dd[j] = e < P.nRestore ? P.ent_dd[e] : 0x80000000u;  // bit 31 = "not mine to stamp"
      for (int k = 0; k < K; k++) rr.sv[j][k] = e < P.nRestore ? R.statv[(size_t)c.inst[k] * P.nLU + e] : 0.0;
    }
    for (int k = 0; k < K; k++) {  // entries that no phase ever writes: stamped once per run
      const double *sv = R.statv + (size_t)c.inst[k] * P.nLU;
      SPICEY_NOUNROLL
      for (int e = P.nRestore + tid; e < P.nLU; e += T) {
        if (HYB) {
          if (e < P.nRestore + P.hyb_g2) c.G[(size_t)e * K + k] = sv[e];  // leaf-owned: read from the global array by phase U_0 / K_0
          else c.W[(size_t)(e - P.hyb_g0 - P.hyb_g2) * K + k] = sv[e];
        } else {
          c.W[(size_t)e * K + k] = sv[e];
        }
      }
    }
    for (int k = 0; k < K; k++) {
      const size_t in = (size_t)c.inst[k];
      const double *g = R.gstat + in * P.nGstat;
      for (int j = 0; j < NEL; j++) rr.vprev[j][k] = tid + j * T < P.nC ? R.C_vprev[in * P.nC + tid + j * T] : 0.0;
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nC; i += T) c.u[(size_t)i * K + k] = g[P.nR + i] * R.C_vprev[in * P.nC + i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nL; i += T) c.u[(size_t)(oL + i) * K + k] = R.L_iprev[in * P.nL + i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nV; i += T) c.u[(size_t)(oV + i) * K + k] = R.src[i];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nS; i += T) {
        const int on = R.S_ison[in * P.nS + i];
        c.ison[(size_t)i * K + k] = on;
        c.gd[(size_t)i * K + k] = spicey_switch_g(on, R.S_ron[in * P.nS + i], R.S_roff[in * P.nS + i]);
      }
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nD; i += T) {
        const double *dp = R.dpar + (in * P.nD + i) * 2;
        double g2, q, irec;
        spicey_diode_k(R.D_vdprev[in * P.nD + i], R.D_is[in * P.nD + i], dp[0], dp[1], false, g2, q, irec);
        c.gd[(size_t)(P.nS + i) * K + k] = g2;
        c.u[(size_t)(oD + i) * K + k] = q;
        if (DIAG && R.lin_vd && c.valid[k]) R.lin_vd[in * P.nD + i] = R.D_vdprev[in * P.nD + i];
      }
    }
    if (tid == 0) c.flags[0] = 0;
  }

  SPICEY_HD void stamp_entry(uint32_t e, uint32_t dd, const double *sv) const {  // sv[K]
    double v[K];
    for (int k = 0; k < K; k++) v[k] = sv[k];
    const uint32_t f0 = dd & 0x7fffu, f1 = (dd >> 15) & 0x7fffu;
    if (f0) {
      const uint32_t ix = (f0 & 0x3fffu) - 1;
      for (int k = 0; k < K; k++) { const double g = c.gd[(size_t)ix * K + k]; v[k] = (f0 & 0x4000u) ? v[k] - g : v[k] + g; }
    }
    if (f1) {
      const uint32_t ix = (f1 & 0x3fffu) - 1;
      for (int k = 0; k < K; k++) { const double g = c.gd[(size_t)ix * K + k]; v[k] = (f1 & 0x4000u) ? v[k] - g : v[k] + g; }
    }
    if (dd & (1u << 30))
      for (int k = 0; k < K; k++) {
        if (fabs(v[k]) < SPICEY_EPS && c.valid[k]) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
        v[k] = spicey_rcp(v[k]);
      }
    for (int k = 0; k < K; k++) put_entry(e, k, v[k]);
  }
  SPICEY_HD void rhs_row(uint32_t r, uint32_t d0, uint32_t d1) const {
    double acc[K];
    for (int k = 0; k < K; k++) acc[k] = 0.0;
    const uint32_t f[4] = {d0 & 0xffffu, d0 >> 16, d1 & 0xffffu, d1 >> 16};
    for (int i = 0; i < 4; i++)
      if (f[i]) {
        const uint32_t ix = (f[i] & 0x7fffu) - 1;
        for (int k = 0; k < K; k++) { const double t = c.u[(size_t)ix * K + k]; acc[k] = (f[i] & 0x8000u) ? acc[k] - t : acc[k] + t; }
      }
    for (int k = 0; k < K; k++) c.W[(size_t)(P.xoff + r) * K + k] = acc[k];
  }

  // ---- B: matrix = static + dynamic stamps; right-hand side -----------------------------------------
  SPICEY_HD void b_stamp(int tid, Regs &rr, bool reuse = false) const {
    SPICEY_MARK(c, 15);
    if (tid == 0) c.flags[0] = 0;
    rr.cursor = 0;  // a new solve walks the resident slots from the start
    if (!reuse) stamp_matrix(tid, rr);  // a linear circuit keeps the factors of step 0 in W
    rhs_rows(tid, rr);
  }
  // ---- batched forms of the beyond-resident-capacity loops (HYB builds, K = 1) -------------------------------------------
  // Hybrid workspace: circuits of several thousand unknowns on 512 threads — most entries lie beyond the resident slots, and
  // one at a time each of them costs two or three DEPENDENT round trips to L2 (descriptor, static value, conductances).
  // Four at a time, every load of a stage issued before the first is used (the registers are there: this build is not at
  // the 128-register cap).  Same arithmetic per entry as stamp_entry.
  SPICEY_HD void stamp_rest_batched(int tid) const {
    const double *sv0 = R.statv + (size_t)c.inst[0] * P.nLU;
    // dynamic entries beyond the descriptor slots: [NDD T, nDynEnt)
    if (brem & 16u)
    SPICEY_NOUNROLL
    for (int e0 = tid + Regs::NDD * T; e0 < P.nDynEnt; e0 += BWB * T) {
      uint32_t dd[BWB];
      double v[BWB], ga[BWB], gb[BWB];
      SPICEY_UNROLL
      for (int b = 0; b < BWB; b++) {
        const int e = e0 + b * T;
        const bool have = e < P.nDynEnt;
        dd[b] = have ? P.ent_dd[e] : 0x80000000u;
        v[b] = sv0[have ? e : e0];
      }
      SPICEY_UNROLL
      for (int b = 0; b < BWB; b++) {
        const uint32_t f0 = dd[b] & 0x7fffu, f1 = (dd[b] >> 15) & 0x7fffu;
        const bool on = !(dd[b] >> 31);
        ga[b] = c.gd[(on && f0) ? (f0 & 0x3fffu) - 1 : 0u];
        gb[b] = c.gd[(on && f1) ? (f1 & 0x3fffu) - 1 : 0u];
      }
      SPICEY_UNROLL
      for (int b = 0; b < BWB; b++) {
        if (dd[b] >> 31) continue;
        const uint32_t f0 = dd[b] & 0x7fffu, f1 = (dd[b] >> 15) & 0x7fffu;
        double x = v[b];
        if (f0) x = (f0 & 0x4000u) ? x - ga[b] : x + ga[b];
        if (f1) x = (f1 & 0x4000u) ? x - gb[b] : x + gb[b];
        if (dd[b] & (1u << 30)) {
          if (fabs(x) < SPICEY_EPS && c.valid[0]) { c.flags[1] = 1; c.flags[2] = c.inst[0]; }
          x = spicey_rcp(x);
        }
        put_entry((uint32_t)(e0 + b * T), 0, x);
      }
    }
    // static update targets beyond the resident slots: plain copies of their static value, [max(NSV T, nDynEnt), nRestore)
    if (brem & 1u) {
      int e0 = tid + NSV * T;
      if (e0 < P.nDynEnt) e0 += ((P.nDynEnt - e0 + T - 1) / T) * T;
      SPICEY_NOUNROLL
      for (; e0 < P.nRestore; e0 += BWB * T) {
        double v[BWB];
        SPICEY_UNROLL
        for (int b = 0; b < BWB; b++) v[b] = sv0[e0 + b * T < P.nRestore ? e0 + b * T : e0];
        SPICEY_UNROLL
        for (int b = 0; b < BWB; b++)
          if (e0 + b * T < P.nRestore) put_entry((uint32_t)(e0 + b * T), 0, v[b]);
      }
    }
    if (brem & 2u)
    SPICEY_NOUNROLL
    for (int t = tid; t < P.nDynX; t += T) {  // entries with > 2 dynamic stamps (rare: kept one at a time)
      const uint32_t et = P.dynx_ent[t], e = SPICEY_IDX(et);
      double x = sv0[e];
      for (uint32_t j = P.dynx_ptr[t]; j < P.dynx_ptr[t + 1]; j++) {
        const uint32_t ix = P.dynx_idx[j];
        const double g = c.gd[SPICEY_IDX(ix)];
        x = (ix & SPICEY_NEG) ? x - g : x + g;
      }
      if (et & SPICEY_TGT_RECIP) {
        if (fabs(x) < SPICEY_EPS && c.valid[0]) { c.flags[1] = 1; c.flags[2] = c.inst[0]; }
        x = spicey_rcp(x);
      }
      put_entry(e, 0, x);
    }
  }
  // right-hand-side rows beyond the resident ones, four at a time: descriptors, then all their (up to 16) contributions
  SPICEY_HD void rhs_rest_batched(int tid) const {
    SPICEY_NOUNROLL
    for (int r0 = tid + NEL * T; r0 < P.n; r0 += BWB * T) {
      uint32_t d[BWB][2];
      double t[BWB][4];
      SPICEY_UNROLL
      for (int b = 0; b < BWB; b++) {
        const int r = r0 + b * T < P.n ? r0 + b * T : r0;
        d[b][0] = P.row_desc[(size_t)r * 2]; d[b][1] = P.row_desc[(size_t)r * 2 + 1];
        if (r0 + b * T >= P.n) d[b][1] = 0xFFFFFFFFu;
      }
      SPICEY_UNROLL
      for (int b = 0; b < BWB; b++) {
        const bool on = d[b][1] != 0xFFFFFFFFu;
        const uint32_t f[4] = {d[b][0] & 0xffffu, d[b][0] >> 16, d[b][1] & 0xffffu, d[b][1] >> 16};
        SPICEY_UNROLL
        for (int i = 0; i < 4; i++) t[b][i] = c.u[(on && f[i]) ? (f[i] & 0x7fffu) - 1 : 0u];
      }
      SPICEY_UNROLL
      for (int b = 0; b < BWB; b++) {
        if (d[b][1] == 0xFFFFFFFFu) continue;
        const uint32_t f[4] = {d[b][0] & 0xffffu, d[b][0] >> 16, d[b][1] & 0xffffu, d[b][1] >> 16};
        double acc = 0.0;
        SPICEY_UNROLL
        for (int i = 0; i < 4; i++)
          if (f[i]) acc = (f[i] & 0x8000u) ? acc - t[b][i] : acc + t[b][i];
        c.W[(size_t)(P.xoff + r0 + b * T)] = acc;
      }
    }
  }

  SPICEY_HD void stamp_matrix(int tid, Regs &rr) const {
    if (HYB) {
      // hybrid workspace: the conductances live in global memory — those of all descriptor slots are fetched together (one
      // L2 round trip) before the first entry is formed; then the plain restores; then the batched rest
      double ga[Regs::NDD], gb[Regs::NDD];
      SPICEY_UNROLL
      for (int j = 0; j < Regs::NDD; j++) {
        uint32_t dd = rr.dd[j];
        SPICEY_OPAQUE(dd);
        const uint32_t f0 = dd & 0x7fffu, f1 = (dd >> 15) & 0x7fffu;
        const bool on = !(dd >> 31);
        ga[j] = c.gd[(on && f0) ? (f0 & 0x3fffu) - 1 : 0u];
        gb[j] = c.gd[(on && f1) ? (f1 & 0x3fffu) - 1 : 0u];
      }
      SPICEY_UNROLL
      for (int j = 0; j < Regs::NDD; j++) {
        uint32_t dd = rr.dd[j];
        SPICEY_OPAQUE(dd);
        if (dd >> 31) continue;
        const uint32_t f0 = dd & 0x7fffu, f1 = (dd >> 15) & 0x7fffu;
        double x = rr.sv[j][0];
        if (f0) x = (f0 & 0x4000u) ? x - ga[j] : x + ga[j];
        if (f1) x = (f1 & 0x4000u) ? x - gb[j] : x + gb[j];
        if (dd & (1u << 30)) {
          if (fabs(x) < SPICEY_EPS && c.valid[0]) { c.flags[1] = 1; c.flags[2] = c.inst[0]; }
          x = spicey_rcp(x);
        }
        put_entry((uint32_t)(tid + j * T), 0, x);
      }
      for (int j = Regs::NDD; j < NSV; j++) {
        const int e = tid + j * T;
        if (e >= P.nDynEnt && e < P.nRestore) put_entry((uint32_t)e, 0, rr.sv[j][0]);
      }
      stamp_rest_batched(tid);
      return;
    }
    for (int j = 0; j < Regs::NDD; j++) {
      const uint32_t e = (uint32_t)(tid + j * T);
      uint32_t dd = rr.dd[j];
      SPICEY_OPAQUE(dd);
      if (SPICEY_WAVE_ANY((dd & 0x7fffffffu) != 0u)) {  // dynamic entries are numbered first: only the first slot(s) take this path
        if (!(dd >> 31)) stamp_entry(e, dd, rr.sv[j]);
      } else if (!(dd >> 31)) {
        for (int k = 0; k < K; k++) put_entry(e, k, rr.sv[j][k]);
      }
    }
    for (int j = Regs::NDD; j < NSV; j++) {  // plain restores (a dynamic entry this far up is left to the loop below)
      const int e = tid + j * T;
      if (e >= P.nDynEnt && e < P.nRestore)
        for (int k = 0; k < K; k++) put_entry((uint32_t)e, k, rr.sv[j][k]);
    }
    SPICEY_MARK(c, 8);
    if (brem & 16u)
    SPICEY_NOUNROLL
    for (int e = tid + Regs::NDD * T; e < P.nDynEnt; e += T) {  // dynamic entries beyond the descriptor slots
      const uint32_t dd = P.ent_dd[e];
      if (dd >> 31) continue;
      double sv[K];
      for (int k = 0; k < K; k++) sv[k] = R.statv[(size_t)c.inst[k] * P.nLU + e];
      stamp_entry((uint32_t)e, dd, sv);
    }
    if (brem & 1u)
    SPICEY_NOUNROLL
    for (int e = tid + NSV * T; e < P.nRestore; e += T) {  // entries beyond the resident capacity
      if (e < P.nDynEnt) continue;  // done above
      const uint32_t dd = P.ent_dd[e];
      if (dd >> 31) continue;
      double sv[K];
      for (int k = 0; k < K; k++) sv[k] = R.statv[(size_t)c.inst[k] * P.nLU + e];
      stamp_entry((uint32_t)e, dd, sv);
    }
    if (brem & 2u)
    SPICEY_NOUNROLL
    for (int t = tid; t < P.nDynX; t += T) {  // entries with > 2 dynamic stamps
      const uint32_t et = P.dynx_ent[t], e = SPICEY_IDX(et);
      for (int k = 0; k < K; k++) {
        double v = R.statv[(size_t)c.inst[k] * P.nLU + e];
        for (uint32_t j = P.dynx_ptr[t]; j < P.dynx_ptr[t + 1]; j++) {
          const uint32_t ix = P.dynx_idx[j];
          const double g = c.gd[(size_t)SPICEY_IDX(ix) * K + k];
          v = (ix & SPICEY_NEG) ? v - g : v + g;
        }
        if (et & SPICEY_TGT_RECIP) {
          if (fabs(v) < SPICEY_EPS && c.valid[k]) { c.flags[1] = 1; c.flags[2] = c.inst[k]; }
          v = spicey_rcp(v);
        }
        put_entry(e, k, v);
      }
    }
  }
  SPICEY_HD void rhs_rows(int tid, Regs &rr) const {
    SPICEY_MARK(c, 9);
    if (HYB) {  // (the contributions of all resident rows in one round trip to the global element vector)
      double t[NEL][4];
      SPICEY_UNROLL
      for (int j = 0; j < NEL; j++) {
        uint32_t d0 = rr.rhs[j][0], d1 = rr.rhs[j][1];
        SPICEY_OPAQUE(d0); SPICEY_OPAQUE(d1);
        const bool on = d1 != 0xFFFFFFFFu;
        const uint32_t f[4] = {d0 & 0xffffu, d0 >> 16, d1 & 0xffffu, d1 >> 16};
        SPICEY_UNROLL
        for (int i = 0; i < 4; i++) t[j][i] = c.u[(on && f[i]) ? (f[i] & 0x7fffu) - 1 : 0u];
      }
      SPICEY_UNROLL
      for (int j = 0; j < NEL; j++) {
        uint32_t d0 = rr.rhs[j][0], d1 = rr.rhs[j][1];
        SPICEY_OPAQUE(d0); SPICEY_OPAQUE(d1);
        if (d1 == 0xFFFFFFFFu) continue;
        const uint32_t f[4] = {d0 & 0xffffu, d0 >> 16, d1 & 0xffffu, d1 >> 16};
        double acc = 0.0;
        SPICEY_UNROLL
        for (int i = 0; i < 4; i++)
          if (f[i]) acc = (f[i] & 0x8000u) ? acc - t[j][i] : acc + t[j][i];
        c.W[(size_t)(P.xoff + tid + j * T)] = acc;
      }
    } else
    for (int j = 0; j < NEL; j++) {
      uint32_t d0 = rr.rhs[j][0], d1 = rr.rhs[j][1];
      SPICEY_OPAQUE(d0); SPICEY_OPAQUE(d1);
      if (d1 != 0xFFFFFFFFu) rhs_row((uint32_t)(tid + j * T), d0, d1);
    }
    SPICEY_MARK(c, 10);
    if (HYB) {
      if (brem & 4u) rhs_rest_batched(tid);
    } else if (brem & 4u)
    SPICEY_NOUNROLL
    for (int r = tid + NEL * T; r < P.n; r += T) {
      const uint32_t d0 = P.row_desc[(size_t)r * 2], d1 = P.row_desc[(size_t)r * 2 + 1];
      if (d1 != 0xFFFFFFFFu) rhs_row((uint32_t)r, d0, d1);
    }
    if (brem & 8u)
    SPICEY_NOUNROLL
    for (int t = tid; t < P.nRowX; t += T) {  // rows with > 4 contributions: +-1 gather from the CSR lists
      const uint32_t r = P.rowx[t];
      for (int k = 0; k < K; k++) {
        double acc = 0.0;
        for (uint32_t j = P.rhs_ptr[r]; j < P.rhs_ptr[r + 1]; j++) {
          const uint32_t ix = P.rhs_idx[j];
          const double t2 = c.u[(size_t)SPICEY_IDX(ix) * K + k];
          acc = (ix & SPICEY_NEG) ? acc - t2 : acc + t2;
        }
        c.W[(size_t)(P.xoff + r) * K + k] = acc;
      }
    }
  }

  SPICEY_HD void a_reiterate(int tid) const {  // iteration >= 1: diodes from x, switches from their new state
    const int oD = P.nC + P.nL + P.nV;
    for (int k = 0; k < K; k++) {
      const size_t in = (size_t)c.inst[k];
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nS; i += T)
        c.gd[(size_t)i * K + k] = spicey_switch_g(c.ison[(size_t)i * K + k], R.S_ron[in * P.nS + i], R.S_roff[in * P.nS + i]);
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nD; i += T) {
        const double *dp = R.dpar + (in * P.nD + i) * 2;
        double g2, q, irec;
        const double vd = dv16(P.D_ab[i], k);
        spicey_diode_k(vd, R.D_is[in * P.nD + i], dp[0], dp[1], false, g2, q, irec);
        c.gd[(size_t)(P.nS + i) * K + k] = g2;
        c.u[(size_t)(oD + i) * K + k] = q;
        if (DIAG && R.lin_vd && c.valid[k]) R.lin_vd[in * P.nD + i] = vd;
      }
    }
  }

  // ---- Z: record, update state, evaluate the next step's companions --------------------------------
  SPICEY_HD void z_cap(int i, double dv, int k, size_t in, double gc, double *oi, int cC, double &vprev, bool last) const {
    if (oi) SPICEY_STREAM_STORE(&oi[cC + i], gc * (dv - vprev));
    vprev = dv;
    c.u[(size_t)i * K + k] = gc * dv;
    if (last) R.C_vprev[in * P.nC + i] = dv;
  }
  SPICEY_HD void z_dio(int i, double vd, int k, size_t in, double is, double dp0, double dp1, double *oi, int cD, int oD, bool last) const {
    double gg, q, irec;
    spicey_diode_k(vd, is, dp0, dp1, oi != nullptr, gg, q, irec);
    if (oi) SPICEY_STREAM_STORE(&oi[cD + i], irec);
    c.gd[(size_t)(P.nS + i) * K + k] = gg;
    c.u[(size_t)(oD + i) * K + k] = q;
    if (last) R.D_vdprev[in * P.nD + i] = vd;
  }
  // diagnostics pass of Z (SpiceyRun::lin_vd set; its own loop so that the production path carries no extra state):
  // |vd(x) - vd_lin| of this thread's diodes, and the new linearisation point
  SPICEY_HD double z_lin_err(int tid, int k, size_t in) const {
    double lerr = 0.0;
    SPICEY_NOUNROLL
    for (int i = tid; i < P.nD; i += T) {
      const double vd = dv16(P.D_ab[i], k);
      const double e = fabs(vd - R.lin_vd[in * P.nD + i]);
      lerr = e > lerr ? e : lerr;
      R.lin_vd[in * P.nD + i] = vd;
    }
    return lerr;
  }
  SPICEY_HD void z_prefetch(int tid, int64_t step, int k, Regs &rr) const {
    const size_t in = (size_t)(K == 1 ? c.inst[0] : (k == 0 ? c.inst[0] : c.inst[K - 1]));
    const double *g = R.gstat + in * P.nGstat;
    for (int j = 0; j < NEL; j++) {
      const int i = (SPICEY_EXP & 8) ? 0x7fffffff : tid + j * T;
      rr.pf[j][0] = i < P.nR ? g[i] : 0.0;
      rr.pf[j][1] = i < P.nC ? g[P.nR + i] : 0.0;
      const bool hd = i < P.nD;
      const double *dp = R.dpar + (in * P.nD + (hd ? i : 0)) * 2;
      rr.pf[j][2] = hd ? R.D_is[in * P.nD + i] : 0.0;
      rr.pf[j][3] = hd ? dp[0] : 0.0;
      rr.pf[j][4] = hd ? dp[1] : 0.0;
    }
  }
  // next step's source values: issued before the tasks of the last backward phase, parked in LDS after them
  SPICEY_HD double z_src_fetch(int tid, int64_t step) const {
    return (tid < P.nV && step != R.steps) ? R.src[(size_t)(step + 1) * P.nV + tid] : 0.0;
  }
  SPICEY_HD void z_src_park(int tid, double v) const {
    if (tid < P.nV) c.u[(size_t)(P.nC + P.nL + P.nV + P.nD + tid) * K] = v;
  }
  SPICEY_HD void z_prefetch_none(Regs &rr) const {
    for (int j = 0; j < NEL; j++)
      for (int q = 0; q < 5; q++) rr.pf[j][q] = 0.0;
  }
  SPICEY_HD void z_record(int tid, int64_t step, Regs &rr, bool prefetched) const {
    const bool last = step == R.steps;
    const int oL = P.nC, oV = P.nC + P.nL, oD = P.nC + P.nL + P.nV;
    const int cR = 0, cC = P.nR, cL = P.nR + P.nC, cV = cL + P.nL, cS = cV + P.nV, cD = cS + P.nS;
    // The instance loop is kept ROLLED here (one copy of the exp / store code, one instance's working set):
    // unrolled and interleaved it needs ~2x the VGPRs and the register-resident program spills.
    SPICEY_NOUNROLL
    for (int k = 0; k < K; k++) {
      const int vk = K == 1 ? c.valid[0] : (k == 0 ? c.valid[0] : c.valid[K - 1]);
      if (!vk) continue;
      const size_t in = (size_t)(K == 1 ? c.inst[0] : (k == 0 ? c.inst[0] : c.inst[K - 1]));
      double *ov = R.out_v + (in * (size_t)(R.steps + 1) + (size_t)step) * P.nOut;
      double *oi = R.out_i ? R.out_i + (in * (size_t)(R.steps + 1) + (size_t)step) * P.nCur : nullptr;
      const double *g = R.gstat + in * P.nGstat;
      if (DIAG && R.lin_vd) {  // diagnostics (wave-uniform): the step's one-shot linearisation error
        const double lerr = z_lin_err(tid, k, in);
        if (R.lin_err) spicey_lin_err_report(R, in, step, tid, lerr);
      }
      SPICEY_MARK(c, 15);
      // Element parameters of the resident items come from L2: all their loads are issued together (one round
      // trip per step instead of one per element section), normally already during the last backward phase.
      if (!(K == 1 && prefetched)) z_prefetch(tid, step, k, rr);
      double pR[NEL], pC[NEL], pIs[NEL], pD0[NEL], pD1[NEL];
      for (int j = 0; j < NEL; j++) { pR[j] = rr.pf[j][0]; pC[j] = rr.pf[j][1]; pIs[j] = rr.pf[j][2]; pD0[j] = rr.pf[j][3]; pD1[j] = rr.pf[j][4]; }
      double srcn = (K == 1 && prefetched) ? 0.0 : z_src_fetch(tid, step);
      // ... and all of them are WAITED for here, before the first result store is issued: gfx9 has one counter
      // (vmcnt) for loads and stores, which may complete out of order, so once a store is in flight a wait for any
      // load becomes vmcnt(0) = "until every result store has been acknowledged" (~1 us each time).
      for (int j = 0; j < NEL; j++) { SPICEY_OPAQUE(pR[j]); SPICEY_OPAQUE(pC[j]); SPICEY_OPAQUE(pIs[j]); SPICEY_OPAQUE(pD0[j]); SPICEY_OPAQUE(pD1[j]); }
      SPICEY_OPAQUE(srcn);
      SPICEY_MARK(c, 0);
      if (tid < P.nV) {  // source tid: branch current out, next step's value in (read by the next B only)
        uint32_t vx = rr.ox[0];
        SPICEY_OPAQUE(vx);
        if (oi) oi[cV + tid] = c.W[(size_t)(vx >> 16) * K + k];
        if (!last) c.u[(size_t)(oV + tid) * K + k] = (K == 1 && prefetched) ? c.u[(size_t)(oD + P.nD + tid) * K + k] : srcn;
      }
      SPICEY_SCHED_FENCE;
      // resident items (element / row / output tid + j T).  All their terminal voltages are read first, back to
      // back (one LDS round trip), then class by class so that the parameter registers die early.
      {
        double vo[NEL], dR[NEL];
        for (int j = 0; j < NEL; j++) {
          uint32_t ox = rr.ox[j], eR = rr.eR[j];
          SPICEY_OPAQUE(ox); SPICEY_OPAQUE(eR);
          vo[j] = volt16(ox & 0xFFFFu, k);
          dR[j] = dv16(eR, k);
        }
        SPICEY_SCHED_FENCE;
        for (int j = 0; j < NEL; j++) {
          const int i = tid + j * T;
          if (i < P.nOut && !(SPICEY_EXP & 16)) SPICEY_STREAM_STORE(&ov[i], vo[j]);
          if (oi && i < P.nR && !(SPICEY_EXP & 16)) SPICEY_STREAM_STORE(&oi[cR + i], dR[j] * pR[j]);
        }
        SPICEY_SCHED_FENCE;
      }
      double dC[NEL], dD[NEL];
      for (int j = 0; j < NEL; j++) {
        uint32_t eC = rr.eC[j], eD = rr.eD[j];
        SPICEY_OPAQUE(eC); SPICEY_OPAQUE(eD);
        dC[j] = dv16(eC, k);
        dD[j] = dv16(eD, k);
      }
      SPICEY_SCHED_FENCE;
      SPICEY_MARK(c, 1);
      for (int j = 0; j < NEL; j++) {
        const int i = tid + j * T;
        if (i < P.nC && !(SPICEY_EXP & 4)) {
          double vp = K == 1 ? rr.vprev[j][0] : (k == 0 ? rr.vprev[j][0] : rr.vprev[j][K - 1]);
          z_cap(i, dC[j], k, in, pC[j], oi, cC, vp, last);
          if (K == 1 || k == 0) rr.vprev[j][0] = vp;
          else rr.vprev[j][K - 1] = vp;
        }
      }
      SPICEY_SCHED_FENCE;
      SPICEY_MARK(c, 2);
      for (int j = 0; j < NEL; j++) {
        const int i = tid + j * T;
        if (i < P.nD && !(SPICEY_EXP & 2)) z_dio(i, dD[j], k, in, pIs[j], pD0[j], pD1[j], oi, cD, oD, last);
        SPICEY_SCHED_FENCE;
      }
      SPICEY_MARK(c, 3);
      if ((SPICEY_EXP & 32) || !zrem) continue;
      if (HYB) {
        // (hybrid workspace: four items of every kind at a time — indices and parameters of all four in flight before the
        // first terminal voltage is read; the same arithmetic per item as the loops below)
        if (zrem & 1u)
        SPICEY_NOUNROLL
        for (int i0 = tid + NEL * T; i0 < P.nOut; i0 += BW * T) {
          int32_t xi[BW];
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) xi[b] = P.out_x[i0 + b * T < P.nOut ? i0 + b * T : i0];
          double v[BW];
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) v[b] = xi[b] < 0 ? 0.0 : c.W[(size_t)xi[b]];
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++)
            if (i0 + b * T < P.nOut) SPICEY_STREAM_STORE(&ov[i0 + b * T], v[b]);
        }
        if (oi && (zrem & 2u))
        SPICEY_NOUNROLL
        for (int i0 = tid + NEL * T; i0 < P.nR; i0 += BW * T) {
          uint32_t ab[BW];
          double gg[BW], dv[BW];
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) { const int i = i0 + b * T < P.nR ? i0 + b * T : i0; ab[b] = P.R_ab[i]; gg[b] = g[i]; }
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) dv[b] = dv16(ab[b], k);
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++)
            if (i0 + b * T < P.nR) SPICEY_STREAM_STORE(&oi[cR + i0 + b * T], dv[b] * gg[b]);
        }
        if (zrem & 4u)
        SPICEY_NOUNROLL
        for (int i0 = tid + NEL * T; i0 < P.nC; i0 += BW * T) {  // beyond the resident capacity: vPrev lives in the state array
          uint32_t ab[BW];
          double gc[BW], vp[BW], dv[BW];
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) {
            const int i = i0 + b * T < P.nC ? i0 + b * T : i0;
            ab[b] = P.C_ab[i]; gc[b] = g[P.nR + i]; vp[b] = R.C_vprev[in * P.nC + i];
          }
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) dv[b] = dv16(ab[b], k);
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) {
            const int i = i0 + b * T;
            if (i >= P.nC) continue;
            z_cap(i, dv[b], k, in, gc[b], oi, cC, vp[b], false);
            R.C_vprev[in * P.nC + i] = vp[b];
          }
        }
        if (zrem & 64u)
        SPICEY_NOUNROLL
        for (int i0 = tid + NEL * T; i0 < P.nD; i0 += BW * T) {
          uint32_t ab[BW];
          double is4[BW], d0[BW], d1[BW], vd[BW];
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) {
            const int i = i0 + b * T < P.nD ? i0 + b * T : i0;
            const double *dp = R.dpar + (in * P.nD + i) * 2;
            ab[b] = P.D_ab[i]; is4[b] = R.D_is[in * P.nD + i]; d0[b] = dp[0]; d1[b] = dp[1];
          }
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++) vd[b] = dv16(ab[b], k);
          SPICEY_UNROLL
          for (int b = 0; b < BW; b++)
            if (i0 + b * T < P.nD) z_dio(i0 + b * T, vd[b], k, in, is4[b], d0[b], d1[b], oi, cD, oD, last);
        }
      }
      if (!HYB && (zrem & 1u))
      SPICEY_NOUNROLL
      for (int i = tid + NEL * T; i < P.nOut; i += T) ov[i] = P.out_x[i] < 0 ? 0.0 : c.W[(size_t)P.out_x[i] * K + k];
      if (!HYB && oi && (zrem & 2u)) {
        SPICEY_NOUNROLL
        for (int i = tid + NEL * T; i < P.nR; i += T) oi[cR + i] = dv16(P.R_ab[i], k) * g[i];
      }
      if (!HYB && (zrem & 4u))
      SPICEY_NOUNROLL
      for (int i = tid + NEL * T; i < P.nC; i += T) {  // beyond the resident capacity: vPrev lives in the state array
        double vp = R.C_vprev[in * P.nC + i];
        z_cap(i, dv16(P.C_ab[i], k), k, in, g[P.nR + i], oi, cC, vp, false);
        R.C_vprev[in * P.nC + i] = vp;
      }
      if (zrem & 8u)
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nL; i += T) {
        const double dv = dv16(P.L_ab[i], k);
        const double il = g[P.nR + P.nC + i] * dv + c.u[(size_t)(oL + i) * K + k];
        if (oi) oi[cL + i] = il;
        c.u[(size_t)(oL + i) * K + k] = il;
        if (last) R.L_iprev[in * P.nL + i] = il;
      }
      if (zrem & 16u)
      SPICEY_NOUNROLL
      for (int i = tid + T; i < P.nV; i += T) {
        if (oi) oi[cV + i] = c.W[(size_t)P.V_x[i] * K + k];
        if (!last) c.u[(size_t)(oV + i) * K + k] = R.src[(size_t)(step + 1) * P.nV + i];
      }
      if (zrem & 32u)
      SPICEY_NOUNROLL
      for (int i = tid; i < P.nS; i += T) {
        const int on = c.ison[(size_t)i * K + k];
        const double gs = spicey_switch_g(on, R.S_ron[in * P.nS + i], R.S_roff[in * P.nS + i]);
        const double va = P.S_a[i] < 0 ? 0.0 : c.W[(size_t)P.S_a[i] * K + k], vb = P.S_b[i] < 0 ? 0.0 : c.W[(size_t)P.S_b[i] * K + k];
        if (oi) oi[cS + i] = (va - vb) * gs;
        c.gd[(size_t)i * K + k] = gs;
        if (last) R.S_ison[in * P.nS + i] = on;
      }
      if (!HYB && (zrem & 64u))
      SPICEY_NOUNROLL
      for (int i = tid + NEL * T; i < P.nD; i += T) {
        const double *dp = R.dpar + (in * P.nD + i) * 2;
        z_dio(i, dv16(P.D_ab[i], k), k, in, R.D_is[in * P.nD + i], dp[0], dp[1], oi, cD, oD, last);
      }
      SPICEY_SCHED_FENCE;
    }
  }
};

// ---- tridiagonal top by parallel cyclic reduction (program.h: pcr_n, pcr_tab) -------------------------------------------
// One wave, lane i = row i of the tridiagonal Schur complement (path order), two SoA buffers {a, b, c, d}[64] in LDS used
// alternately.  Stage 0 gathers the rows from W through the index table; stage st = 1 .. S (stride 1, 2, 4, ...): row i
// eliminates its couplings to the rows i -+ stride with those rows' equations; after S = ceil(log2 n) stages every row
// stands alone and the last stage writes x_i = d_i / b_i straight into the solution slot.
// Replaces 2 x (S + 1) LDS-serial levels of the task lists; no U entries are formed for these pivots (nothing below needs
// them: the backward records of lower rows read x only).  All loads of a stage are unconditional (clamped addresses, values
// masked afterwards) so that they are issued together: one LDS round trip per stage.
template <int K>
SPICEY_HD void spicey_pcr_row(const WgCtx<K> &c, const uint16_t *tab, int n, int r, double &a, double &b, double &cc, double &d) {
  const bool on = r >= 0 && r < n;
  const int rr = on ? r : 0;
  const uint32_t ia = tab[rr * 4], ib = tab[rr * 4 + 1], ic = tab[rr * 4 + 2], id = tab[rr * 4 + 3];
  const double va = c.W[(size_t)(ia == 0xFFFFu ? ib : ia) * K], vb = c.W[(size_t)ib * K], vc = c.W[(size_t)(ic == 0xFFFFu ? ib : ic) * K],
               vd = c.W[(size_t)id * K];
  a = (on && ia != 0xFFFFu) ? va : 0.0;
  b = on ? vb : 1.0;  // rows past the end: identity
  cc = (on && ic != 0xFFFFu) ? vc : 0.0;
  d = on ? vd : 0.0;
}
// A row without a neighbour at the stage's stride has a zero coupling on that side (a_i = 0 for i < stride, c_i = 0 for
// i + stride >= n: by induction over the stages; rows past the end are identity rows), so the missing neighbour is not
// masked: its index is clamped into the buffer and whatever finite row is read there is multiplied by that zero.  (Masking
// cost 16 selects of ~70 instructions per stage, on a wave that issues one instruction per ~4.5 cycles.)
template <int K>
SPICEY_HD void spicey_pcr_stage(const WgCtx<K> &c, double *buf, const uint16_t *tab, int n, int S, int lane, int st, double *own) {
  // LDS row = {a, 1/b, c, d}: a row forms the reciprocal of its own pivot once, its two neighbours multiply with it;
  // the row's own {a, b, c, d} stay in registers (`own`) from stage to stage
  double *wr = buf + ((st & 1) ? 256 : 0);
  bool sing;
  if (st == 0) {  // gather the rows from W (stage 0 writes buffer 0)
    double a, b, cc, d;
    spicey_pcr_row<K>(c, tab, n, lane, a, b, cc, d);
    own[0] = a; own[1] = b; own[2] = cc; own[3] = d;
    sing = fabs(b) < SPICEY_EPS;
    wr[lane] = a; wr[64 + lane] = spicey_rcp(b); wr[128 + lane] = cc; wr[192 + lane] = d;
  } else {
    const double *rd = buf + (((st - 1) & 1) ? 256 : 0);
    const int h = 1 << (st - 1), im = lane - h, ip = lane + h;
    const int jm = im < 0 ? 0 : im, jp = ip > 63 ? 63 : ip;
    const double am = rd[jm], rm = rd[64 + jm], cm = rd[128 + jm], dm = rd[192 + jm];
    const double ap = rd[jp], rp = rd[64 + jp], cp = rd[128 + jp], dp = rd[192 + jp];
    const double al = -own[0] * rm;  // (a = 0 where there is no such neighbour)
    const double ga = -own[2] * rp;
    const double na = al * am, nc = ga * cp;
    const double nb = fma(ga, ap, fma(al, cm, own[1]));
    const double nd = fma(ga, dp, fma(al, dm, own[3]));
    sing = fabs(nb) < SPICEY_EPS;
    const double nr = spicey_rcp(nb);
    if (st < S) {
      own[0] = na; own[1] = nb; own[2] = nc; own[3] = nd;
      wr[lane] = na; wr[64 + lane] = nr; wr[128 + lane] = nc; wr[192 + lane] = nd;
    } else if (lane < n) {  // the rows are decoupled: x = d / b straight into the solution slot
      c.W[(size_t)tab[lane * 4 + 3] * K] = nd * nr;
    }
  }
  if (sing && lane < n && c.valid[0]) { c.flags[1] = 1; c.flags[2] = c.inst[0]; }
}
#if defined(__HIP_DEVICE_COMPILE__)
// All stages in one call for the GPU (the same arithmetic as spicey_pcr_stage, stage after stage): the stage loop is
// unrolled — strides, buffer halves and the last-stage test are constants —, the pivots are judged once at the end by
// their running minimum, and between two stages stands only a compiler fence (the LDS operations of one wave execute in
// order).  ~40 instructions per stage instead of ~70.
template <int K>
__device__ __forceinline__ void spicey_pcr_all(const WgCtx<K> &c, double *buf, const uint16_t *tab, int n, int S, int lane) {
  double a, b, cc, d;
  spicey_pcr_row<K>(c, tab, n, lane, a, b, cc, d);
  double pmin = fabs(b);  // (rows past the end: b = 1)
  buf[lane] = a; buf[64 + lane] = spicey_rcp(b); buf[128 + lane] = cc; buf[192 + lane] = d;
#pragma unroll
  for (int st = 1; st <= 6; st++) {
    if (st > S) break;  // (wave-uniform)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double *rd = buf + (((st - 1) & 1) ? 256 : 0);
    double *wr = buf + ((st & 1) ? 256 : 0);
    const int h = 1 << (st - 1);
    const int jm = max(lane - h, 0), jp = min(lane + h, 63);
    const double am = rd[jm], rm = rd[64 + jm], cm = rd[128 + jm], dm = rd[192 + jm];
    const double ap = rd[jp], rp = rd[64 + jp], cp = rd[128 + jp], dp = rd[192 + jp];
    const double al = -a * rm, ga = -cc * rp;
    const double na = al * am, nc = ga * cp;
    const double nb = fma(ga, ap, fma(al, cm, b));
    const double nd = fma(ga, dp, fma(al, dm, d));
    pmin = fmin(pmin, fabs(nb));
    const double nr = spicey_rcp(nb);
    if (st < S) {
      a = na; b = nb; cc = nc; d = nd;
      wr[lane] = na; wr[64 + lane] = nr; wr[128 + lane] = nc; wr[192 + lane] = nd;
    } else if (lane < n) {
      c.W[(size_t)tab[lane * 4 + 3] * K] = nd * nr;
    }
  }
  if (pmin < SPICEY_EPS && lane < n && c.valid[0]) { c.flags[1] = 1; c.flags[2] = c.inst[0]; }
}
#endif

// The three argument structs hold ~110 pointers: kept in SGPRs across the time loop they overflow the 102 scalar registers
// of a wave and the compiler parks them in VGPR lanes (round 1: 274 spilled SGPRs, 1 209 v_readlane in the kernel — 13 % of
// its instructions).  Every phase therefore takes the structs through `ex.fresh()`: on the GPU they live in global memory and
// `fresh` makes their address opaque for this phase, so the fields a phase needs are fetched by scalar loads inside it
// (scalar cache) and are dead at its barrier; only a handful of loop-control scalars stay live around the loop.
template <int K, int RMAX, int NSV, int NEL, bool HYB = false, class Exec>
SPICEY_HD void spicey_tran_run_v2(Exec &ex, const SpiceyProg &P, const SpiceyResident &Q, const SpiceyRun &R, WgCtx<K> &c, int wg) {
  const int T = ex.threads();
  typedef TranPhases2<K, RMAX, NSV, NEL, HYB> Ph2;
  uint32_t brem, zrem;
  {
    Ph2 p2{P, R, c, T, 0u, 0u};
    p2.set_remainders();
    brem = p2.brem; zrem = p2.zrem;
  }
  typedef ResRegs<K, RMAX, NSV, NEL> Regs;
  ex.phase(SPICEY_PH_PRO, [&](int tid) {
    const SpiceyProg Pf = ex.fresh(P); const SpiceyResident Qf = ex.fresh(Q); const SpiceyRun Rf = ex.fresh(R);
    TranPhases<K> ph{Pf, Rf, c, T};
    Ph2 p2{Pf, Rf, c, T, brem, zrem};
    if (tid == 0) { c.flags[0] = 0; c.flags[1] = 0; c.flags[2] = -1; }
    ph.p0_gstat(tid);
    p2.load_resident(tid, Qf, ex.template regs<Regs>(tid));
    if (K == 1 && Pf.pcr_n > 0) {  // tridiagonal top: its index table sits behind the two 2 KB row buffers
      uint16_t *tab = (uint16_t *)(c.tail + 1024);
      for (int i = tid; i < Pf.pcr_n * 4; i += T) tab[i] = Pf.pcr_tab[i];
    }
    for (int i = tid; i < Qf.tail_n * 64; i += T) {  // tail records -> LDS (16 bytes each; no task = all zero)
      const int p = Qf.tail_first + (i >> 6), lane = i & 63;
      const bool have = (uint32_t)lane < Pf.ph_cnt[p];
      const uint32_t *src = Pf.rec16 + ((size_t)Pf.ph_first[p] + (have ? lane : 0)) * 4;
      for (int w = 0; w < 4; w++) c.tail[(size_t)i * 4 + w] = have ? src[w] : 0u;
    }
  });
  ex.phase(SPICEY_PH_PRO, [&](int tid) { const SpiceyProg Pf = ex.fresh(P); const SpiceyRun Rf = ex.fresh(R); TranPhases<K> ph{Pf, Rf, c, T}; ph.p1_static(tid); });
  ex.phase(SPICEY_PH_PRO, [&](int tid) { const SpiceyProg Pf = ex.fresh(P); const SpiceyRun Rf = ex.fresh(R); Ph2 p2{Pf, Rf, c, T, brem, zrem}; p2.a0_initial(tid, ex.template regs<Regs>(tid)); });
  unsigned long long solves = 0;
  int32_t code = 0;
  int64_t err_step = 0;
  int32_t err_iter = 0;
  if (c.flags[1]) { code = 1; }
  // loop control: a handful of scalars
  const int nL = P.nLevels;
  const int nS = P.nS;
  const int64_t steps = R.steps;
  const int dbg_empty = R.debug_empty_phases;
  // which phases have work: kept in a scalar mask so that the phase loop issues no loads
  unsigned long long active = 0, smask = 0;
  for (int p = 0; p < 2 * nL && p < 64; p++) {
    if (SPICEY_UNIFORM((int)P.ph_cnt[p]) != 0) active |= 1ull << p;
    if (SPICEY_UNIFORM((int)Q.st_cnt[p]) != 0) smask |= 1ull << p;
  }
  const int tail_n = Q.tail_n, tail_first = Q.tail_first;
  // (the tridiagonal top's two loop-control values ride in ONE scalar across the time loop and are unpacked inside it: every
  // further live scalar there costs a lane of a spill VGPR, and the 128-register build has none to give)
  int top_pack;
  {
    const int n0 = K == 1 ? P.pcr_n : 0;
    int S0 = 0;
    while ((1 << S0) < n0) S0++;
    top_pack = n0 | (S0 << 8);
  }
  const int pcr_n = top_pack & 0xff;
  // with a tridiagonal top the factor phases end at its level and the backward phases resume below it
  const int u_end = pcr_n > 0 ? P.pcr_level : (tail_n > 0 ? tail_first : nL);
  const int k_begin = pcr_n > 0 ? 2 * nL - P.pcr_level : (tail_n > 0 ? tail_first + tail_n : nL);
  top_pack |= (K == 1 && k_begin < 2 * nL) ? 1 << 16 : 0;  // bit 16 = z_pre: Z's parameter fetch rides on the last backward phase
  // No diodes and no switches: the matrix of every step is the matrix of step 0 (dt is fixed within a run), so its
  // factors stay in W and later steps run the right-hand-side column only.  Same operands, same order: the results
  // are bit-identical to refactoring (SURVEY.md §8(d) "solve-only" rate; the reference itself never reuses).
  top_pack |= (P.nD == 0 && nS == 0 && P.nDynEnt == 0 && !R.no_reuse) ? 1 << 17 : 0;  // bit 17 = linear
  top_pack |= (Ph2::DIAG && R.skip_risk != nullptr) ? 1 << 18 : 0;  // bit 18 = diagnostics: look at the stamped matrix after B (spicey_skip_risk)
  top_pack |= (K == 1 && pcr_n > 0 && Q.k_merge == k_begin && k_begin < 2 * nL - 1) ? 1 << 19 : 0;  // bit 19 = the first backward phase runs in the top's wave
  top_pack = SPICEY_UNIFORM(top_pack);
  for (int64_t step = 0; step <= steps && code == 0; step++) {
    int iter = 0;
    for (;;) {
      int tp = top_pack;
      SPICEY_OPAQUE_S(tp);
      const int pcr_n = tp & 0xff, pcr_S = (tp >> 8) & 0xff;
      const bool linear = (tp >> 17) & 1;
      ex.phase(SPICEY_PH_B, [&](int tid) {
        const SpiceyProg Pf = ex.fresh(P);
        const SpiceyRun Rf = ex.fresh(R);
        Ph2 p2{Pf, Rf, c, T, brem, zrem};
        // the next step's source values ride on B (a long phase with few live registers): fetched first, parked in
        // LDS last; Z moves them into place
        double sn = K == 1 ? p2.z_src_fetch(tid, step) : 0.0;
        SPICEY_SCHED_FENCE;
        p2.b_stamp(tid, ex.template regs<Regs>(tid), linear && step > 0);
        SPICEY_SCHED_FENCE;
        if (K == 1) p2.z_src_park(tid, sn);
      });
      if (Ph2::DIAG && ((tp >> 18) & 1) && !(linear && step > 0))
        ex.phase(SPICEY_PH_S, [&](int tid) {
          const SpiceyProg Pf = ex.fresh(P);
          const SpiceyRun Rf = ex.fresh(R);
          spicey_skip_risk<K, HYB>(Pf, Rf, c, tid, T, linear ? (unsigned long long)(steps + 1) : 1ull);
        });
      for (int d = 0; d < dbg_empty; d++) ex.phase(SPICEY_PH_S, [&](int) {});  // diagnostics: cost of a bare phase
      // factor levels [0, u_end) | tail [u_end, k_begin) by one wave | backward levels [k_begin, 2 nL)
      for (int p = 0; p < u_end; p++) {
        if (p < 64 ? !((active >> p) & 1) : P.ph_cnt[p] == 0) continue;
        if (HYB && p == 0) {
          // hybrid workspace: phase 0 eliminates the leaves, whose own entries are read from the global array (one L2 round
          // trip for the whole level; every target is in LDS)
          ex.phase(SPICEY_PH_U0, [&](int tid) {
            const SpiceyProg Pf = ex.fresh(P);
            const SpiceyResident Qf = ex.fresh(Q);
            spicey_uk_phase<K, RMAX, NSV, NEL, false, HYB>(Pf, Qf, c, ex.template regs<Regs>(tid), tid, T, 0, ((smask >> 0) & 1) != 0, linear && step > 0);
          });
          continue;
        }
        ex.phase(SPICEY_PH_U0 + (p < 30 ? p : 30), [&](int tid) {
          const SpiceyProg Pf = ex.fresh(P);
          const SpiceyResident Qf = ex.fresh(Q);
          spicey_uk_phase<K, RMAX, NSV, NEL, false>(Pf, Qf, c, ex.template regs<Regs>(tid), tid, T, p, p < 64 ? ((smask >> p) & 1) != 0 : true, linear && step > 0);
        });
      }
      const int kmerge = (tp >> 19) & 1;
      if (pcr_n > 0) {
        // (kmerge: wave 0 goes on with the first backward phase below the top — its records are resident in this wave's
        // slots, its rows need unknowns of the top only, and the LDS operations of one wave execute in order)
        auto merged_k = [&](int lane) {
          const SpiceyProg Pf = ex.fresh(P);
          const SpiceyResident Qf = ex.fresh(Q);
          spicey_uk_phase<K, RMAX, NSV, NEL, true>(Pf, Qf, c, ex.template regs<Regs>(lane), lane, T, k_begin, false);
        };
#if defined(__HIP_DEVICE_COMPILE__)
        if (pcr_S >= 1 && pcr_S <= 6) {
          ex.wave_lockstep_keep(64, 1, [&](int lane, int, double *) {
            spicey_pcr_all<K>(c, (double *)c.tail, (const uint16_t *)(c.tail + 1024), pcr_n, pcr_S, lane);
            if (kmerge) {
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
              merged_k(lane);
            }
          });
        } else
#endif
        ex.wave_lockstep_keep(64, pcr_S + 1 + kmerge, [&](int lane, int st, double *own) {
          if (st <= pcr_S) spicey_pcr_stage<K>(c, (double *)c.tail, (const uint16_t *)(c.tail + 1024), pcr_n, pcr_S, lane, st, own);
          else merged_k(lane);
        });
      } else if (k_begin > u_end) {
        // the record of level l + 1 is fetched (LDS) while level l executes: one round trip less on the serial chain
        ex.tail_phase(SPICEY_PH_U0 + 31, k_begin - u_end,
                      [&](int tid, int lvl, uint32_t *r) {
                        const uint32_t *q = c.tail + ((size_t)lvl * 64 + tid) * 4;
                        r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
                      },
                      [&](int, int lvl, const uint32_t *r) {
                        const SpiceyProg Pf = ex.fresh(P);
                        if (u_end + lvl < nL) spicey_exec_rec16<K, false>(c, Pf.ovf16, r[0], r[1], r[2], r[3], (linear && step > 0) ? (uint32_t)Pf.xoff : 0u);
                        else spicey_exec_rec16<K, true>(c, Pf.ovf16, r[0], r[1], r[2], r[3]);
                      });
      }
      for (int p = k_begin + kmerge; p < 2 * nL - 1; p++) {
        const int l = 2 * nL - 1 - p;
        ex.phase(SPICEY_PH_K0 + (l < 31 ? l : 31), [&](int tid) {
          const SpiceyProg Pf = ex.fresh(P);
          const SpiceyResident Qf = ex.fresh(Q);
          spicey_uk_phase<K, RMAX, NSV, NEL, true>(Pf, Qf, c, ex.template regs<Regs>(tid), tid, T, p, p < 64 ? ((smask >> p) & 1) != 0 : true);
        });
      }
      // the last backward phase (level 0) is peeled: it also issues Z's parameter fetch.  (Every path through the
      // iteration defines the prefetch registers, so they are not live around the time loop.)
      if (k_begin < 2 * nL) {
        const int p = 2 * nL - 1;
        ex.phase(SPICEY_PH_K0, [&](int tid) {
          const SpiceyProg Pf = ex.fresh(P);
          const SpiceyResident Qf = ex.fresh(Q);
          spicey_uk_phase<K, RMAX, NSV, NEL, true, HYB>(Pf, Qf, c, ex.template regs<Regs>(tid), tid, T, p, p < 64 ? ((smask >> p) & 1) != 0 : true);  // (level 0: the leaves)
          SPICEY_SCHED_FENCE;  // after the tasks, not among them: their registers are free by now
          if (K == 1) { const SpiceyRun Rf = ex.fresh(R); Ph2 p2{Pf, Rf, c, T, brem, zrem}; p2.z_prefetch(tid, step, 0, ex.template regs<Regs>(tid)); }
        });
      } else if (K == 1) {
        Ph2 p2{P, R, c, T, brem, zrem};
        p2.z_prefetch_none(ex.template regs<Regs>(0));
      }
      if (c.flags[1]) { code = 1; err_step = step; err_iter = iter; break; }
      if (nS == 0) break;
      ex.phase(SPICEY_PH_S, [&](int tid) { const SpiceyProg Pf = ex.fresh(P); const SpiceyRun Rf = ex.fresh(R); TranPhases<K> ph{Pf, Rf, c, T}; ph.s_switches(tid); });
      const int switched = c.flags[0];
      if (!switched || iter == SPICEY_MAX_ITER - 1) break;
      iter++;
      ex.phase(SPICEY_PH_A, [&](int tid) { const SpiceyProg Pf = ex.fresh(P); const SpiceyRun Rf = ex.fresh(R); Ph2 p2{Pf, Rf, c, T, brem, zrem}; p2.a_reiterate(tid); });
    }
    if (code) break;
    {
      int nvalid = 0;
      for (int k = 0; k < K; k++) nvalid += c.valid[k];
      solves += (unsigned long long)(iter + 1) * (unsigned long long)nvalid;
    }
    ex.phase(SPICEY_PH_Z, [&](int tid) {
      const SpiceyRun Rf = ex.fresh(R);
      const SpiceyProg Pf = ex.fresh(P);
      Ph2 p2{Pf, Rf, c, T, brem, zrem};
      if (tid == 0 && Rf.iters)
        for (int k = 0; k < K; k++)
          if (c.valid[k]) Rf.iters[(size_t)c.inst[k] * (size_t)(steps + 1) + (size_t)step] = iter + 1;
      p2.z_record(tid, step, ex.template regs<Regs>(tid), ((top_pack >> 16) & 1) != 0);
    });
  }
  ex.phase(SPICEY_PH_PRO, [&](int tid) {
    if (tid == 0) {
      const SpiceyRun Rf = ex.fresh(R);
      Rf.status[wg * 4 + 0] = code;
      Rf.status[wg * 4 + 1] = c.flags[2];
      Rf.status[wg * 4 + 2] = (int32_t)err_step;
      Rf.status[wg * 4 + 3] = err_iter;
      Rf.solves[wg] = solves;
    }
  });
}

#include "fronts_exec.h"

// The whole run of one workgroup.  Exec supplies `phase(f)` (run f(tid) for every thread, then
// barrier) and `threads()`.  All control flow is workgroup-uniform: flags are read after barriers.
// FRONTS: compile the dense-front sweeps in (their triangular solves keep 16 doubles per thread in registers: kernels
// built with them are launched with <= 512 threads; the others keep their 1024-thread register budget untouched).
template <int K, bool FRONTS = false, class Exec>
SPICEY_HD void spicey_tran_run(Exec &ex, const SpiceyProg &P, const SpiceyRun &R, WgCtx<K> &c, int wg) {
  TranPhases<K> ph{P, R, c, ex.threads()};
  TranPhases<K> phl{P, R, c, ex.local_threads()};  // the same phases over ONE workgroup's threads (group mode)
  ex.phase(SPICEY_PH_PRO, [&](int tid) {
    if (tid == 0) { c.flags[0] = 0; c.flags[1] = 0; c.flags[2] = -1; }
    ph.p0_gstat(tid);
  });
  ex.phase(SPICEY_PH_PRO, [&](int tid) { ph.p1_static(tid); });
  ex.phase(SPICEY_PH_PRO, [&](int tid) { ph.a0_initial(tid); });
  unsigned long long solves = 0;
  int32_t code = 0;
  int64_t err_step = 0;
  int32_t err_iter = 0;
  if (c.flags[1]) { code = 1; }
  const bool linear = P.nD == 0 && P.nS == 0 && P.nDynEnt == 0 && !R.no_reuse;  // see spicey_tran_run_v2
  // dense fronts above the cut (K = 1 only; the host enables them for nonlinear circuits, so `linear` is false then)
  const bool use_fronts = FRONTS && K == 1 && P.nFronts > 0 && R.front_ws != nullptr;
  FrontsRun<Exec> fr{ex, P, R, c.W, c.flags, c.inst[0], c.valid[0], use_fronts ? R.front_ws + (size_t)wg * (size_t)P.front_ws : nullptr,
                     use_fronts ? R.front_flags + (size_t)wg * 2 * (size_t)P.nFronts : nullptr, ex.local_threads(),
                     use_fronts && R.front_ticks ? R.front_ticks + (size_t)wg * 4 * (size_t)P.nFronts : nullptr};
  unsigned int fepoch = 0;
  for (int64_t step = 0; step <= R.steps && code == 0; step++) {
    if (ex.failed()) { code = 3; err_step = step; break; }  // a cross-workgroup barrier timed out (group mode only)
    int iter = 0;
    for (;;) {
      ex.phase(SPICEY_PH_B, [&](int tid) { ph.b_stamp(tid); });
      if (TranPhases<K>::DIAG && R.skip_risk && !(linear && step > 0))  // diagnostics: the stamped matrix, before the factor levels touch it
        ex.phase(SPICEY_PH_S, [&](int tid) { spicey_skip_risk<K>(P, R, c, tid, ex.threads(), linear ? (unsigned long long)(R.steps + 1) : 1ull); });
      ex.mark(SPICEY_PH_B);
      {
        // Group mode: runs of narrow factor levels (<= 1024 tasks, one per thread: the last pivots of the top separator) also go to
        // workgroup 0 alone; a group barrier separates such a run from the next level that everybody works on.
        bool local_run = false;
        int l_first = 0;
        if constexpr (FRONTS) if (use_fronts && P.nBins > 0) {
          // subtree-local levels below the cut (program.h): every workgroup walks its bins through all those levels with
          // its own barriers; one group barrier, then the targets above the cut take their products in one phase
          ex.for_each_wg([&](int g, int G) {
            for (int l = 0; l < P.front_cut; l++) ex.wg_phase([&](int tid) { phl.u_bins(tid, l, g, G, linear && step > 0); });
          });
          ex.mark(SPICEY_PH_U0 + 18);
          ex.sync();
          ex.mark(SPICEY_PH_U0 + 19);
          l_first = P.front_cut;
        }
        // (above a front cut the lists are empty: nothing to walk; with bins, one phase is left)
        const int l_end = use_fronts ? P.front_cut + (P.nBins > 0 ? 1 : 0) : P.nLevels;
        for (int l = l_first; l < l_end; l++) {
          const uint32_t nsl = P.lvl_slice[l + 1] - P.lvl_slice[l];
          if (nsl == 0) continue;
          if (ex.serial_chain() && nsl <= 16) {
            ex.local_phase([&](int tid) { phl.u_level(tid, l, linear && step > 0); });
            local_run = true;
          } else {
            if (local_run) ex.sync();
            local_run = false;
            if (l_first > 0 && l == l_first) ex.phase_marked(SPICEY_PH_U0 + 22, [&](int tid) { ph.u_level(tid, l, linear && step > 0); });
            else ex.phase(SPICEY_PH_U0 + (l < 31 ? l : 31), [&](int tid) { ph.u_level(tid, l, linear && step > 0); });
          }
        }
        // (a trailing local run flows straight into the backward chain below, which workgroup 0 runs as well)
        if (local_run && (!ex.serial_chain() || use_fronts)) ex.sync();
      }
      ex.mark(SPICEY_PH_U0);
      if constexpr (FRONTS) if (use_fronts) {
        // upper tree: every workgroup sweeps its share of the fronts up, then down (flags between workgroups, no group
        // barrier inside); one group barrier afterwards publishes the upper unknowns to the levels below the cut
        fepoch++;
        const unsigned long long t_sweep = fr.forward(fepoch);
        ex.mark(SPICEY_PH_U0 + 1);
        fr.backward(fepoch, t_sweep);
        ex.mark(SPICEY_PH_U0 + 2);
        ex.local_phase([&](int tid) { if (tid == 0) c.W[(size_t)P.one_slot * K] = 1.0; });
        ex.sync();
        ex.mark(SPICEY_PH_U0 + 3);
      }
      if (ex.serial_chain()) {
        // Group mode: the backward levels carry little work (mesh 100^2: 172 k products over 297 levels) but each
        // would cost a cross-workgroup barrier (~4.7 us): ONE workgroup of the group walks them with its own
        // workgroup barriers, the others wait at the single group barrier behind the chain.
        // (with dense fronts the levels that are left are the WIDE ones at the bottom of the tree — thousands of rows each,
        // the interface phase included: those go to all workgroups, one group barrier each)
        bool local_run = false;
        int l_last = 0;
        if (use_fronts && P.nBins > 0) l_last = P.front_cut + 1;  // (the interface and the levels below it follow, bin by bin)
        for (int l = use_fronts ? P.front_cut : P.nLevels - 1; l >= l_last; l--) {  // (backward level `front_cut`: the interface)
          const uint32_t nsl = P.bk_lvl_slice[l + 1] - P.bk_lvl_slice[l];
          if (nsl == 0) continue;
          if (use_fronts && nsl > 16) {
            if (local_run) ex.sync();
            local_run = false;
            ex.phase(SPICEY_PH_K0 + 31, [&](int tid) { ph.k_level(tid, l); });
            continue;
          }
          ex.local_phase([&](int tid) { phl.k_level(tid, l); });
          local_run = true;
        }
        if constexpr (FRONTS) if (l_last > 0) {
          if (local_run) ex.sync();
          local_run = true;  // (one group barrier behind the bins)
          ex.mark(SPICEY_PH_U0 + 20);
          ex.for_each_wg([&](int g, int G) {
            for (int l = P.front_cut; l >= 0; l--) ex.wg_phase([&](int tid) { phl.k_bins(tid, l, g, G); });
          });
          ex.mark(SPICEY_PH_U0 + 21);
        }
        if (local_run || !use_fronts) ex.sync();
      } else {
        int l_last = 0;
        if (use_fronts && P.nBins > 0) l_last = P.front_cut + 1;
        for (int l = use_fronts ? P.front_cut : P.nLevels - 1; l >= l_last; l--) {
          if (P.bk_lvl_slice[l] == P.bk_lvl_slice[l + 1]) continue;
          ex.phase(SPICEY_PH_K0 + (l < 31 ? l : 31), [&](int tid) { ph.k_level(tid, l); });
        }
        if constexpr (FRONTS) if (l_last > 0)
          ex.for_each_wg([&](int g, int G) {
            for (int l = P.front_cut; l >= 0; l--) ex.wg_phase([&](int tid) { phl.k_bins(tid, l, g, G); });
          });
      }
      ex.phase(SPICEY_PH_K0, [&](int tid) { ph.k_scale(tid); });
      ex.mark(SPICEY_PH_K0);
      // a cross-workgroup barrier that timed out inside this iteration leaves a partially computed workspace: nothing of
      // it may be recorded or reported as a success (group mode only; the flag is sticky and uniform across the group)
      if (ex.failed()) { code = 3; err_step = step; err_iter = iter; break; }
      if (c.flags[1]) { code = 1; err_step = step; err_iter = iter; break; }
      if (P.nS == 0) break;
      ex.phase(SPICEY_PH_S, [&](int tid) { ph.s_switches(tid); });
      const int switched = c.flags[0];
      if (!switched || iter == SPICEY_MAX_ITER - 1) break;
      iter++;
      ex.phase(SPICEY_PH_A, [&](int tid) { ph.a_reiterate(tid); });  // b_stamp (next) resets flags[0] after this barrier
    }
    if (code) break;
    {
      int nvalid = 0;
      for (int k = 0; k < K; k++) nvalid += c.valid[k];
      solves += (unsigned long long)(iter + 1) * (unsigned long long)nvalid;
    }
    ex.phase(SPICEY_PH_Z, [&](int tid) {
      if (tid == 0 && R.iters)
        for (int k = 0; k < K; k++)
          if (c.valid[k]) R.iters[(size_t)c.inst[k] * (size_t)(R.steps + 1) + (size_t)step] = iter + 1;
      ph.z_record(tid, step, linear);
    });
    ex.mark(SPICEY_PH_Z);
    if (ex.failed()) { code = 3; err_step = step; break; }  // also covers the last step and runs with steps = 0
  }
  ex.phase(SPICEY_PH_PRO, [&](int tid) {
    if (tid == 0) {
      R.status[wg * 4 + 0] = code;
      R.status[wg * 4 + 1] = c.flags[2];
      R.status[wg * 4 + 2] = (int32_t)err_step;
      R.status[wg * 4 + 3] = err_iter;
      R.solves[wg] = solves;
    }
  });
}
