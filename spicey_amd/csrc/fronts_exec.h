// fronts_exec.h — dense multifrontal factorisation of the UPPER elimination tree (device code, host-compilable).
//
// Large instances (rcd_mesh(100): 10 001 unknowns, 297 elimination-tree levels) spend their step in the long
// single-pivot chains of the top separators: one barrier-separated level per pivot.  What the reference does there
// (solveReal.ts:38-54: every row below the pivot swept across the full width) IS a dense rank-1 update, so the pivots
// of level >= P.front_cut are taken out of the level-scheduled task lists and factored front by front:
//
//   front = supernode (p consecutive pivots with nested row structure = one separator of the nested dissection) plus
//           its q boundary unknowns: a dense block, local index i < p pivot k0 + i, [p, Pp) identity padding, Pp + j
//           boundary element j, column Mp the right-hand side (Pp, Mp multiples of 16).
//   forward (postorder):  zero -> take the front's own entries + right-hand side over from W (stamped by phase B,
//           updated in place by the task lists of the levels below the cut) -> add the children's contribution blocks
//           (extend-add through fr_rel) -> blocked right-looking LU, panels of 16 pivots: the 16 x 16 diagonal block by
//           ONE wave (registers + cross-lane shuffles), row / column triangular solves one thread each, trailing update
//           C -= L U in 16 x 16 tiles on v_mfma_f64_16x16x4 -> the trailing (q x q) block + rhs is the contribution
//           block for the parent.
//           A front of up to 128 rows lives in LDS for all of this (a CU moves only ~30-60 GB/s to and from L2, and
//           the right-looking update touches the whole trailing block once per panel); only its U rows (for the
//           backward solve) and its contribution block go to the front workspace in global memory.  Larger fronts stay
//           in the workspace and stage one panel at a time through LDS.
//   backward (reverse postorder): x_B from the ancestors, t = y_P - U_PB x_B, blocked back-substitution, x_P -> W.
//
// One workgroup per front; the fronts of a group's G workgroups follow the proportional-mapping schedule of
// spicey_build_front_schedule.  Fronts of different workgroups hand over through per-front flags (agent-scope
// release / acquire, the same form as the group barrier); a workgroup's own consecutive fronts need a workgroup
// barrier only.  Every sum has a fixed order that does not depend on G: results are bit-identical for every G.
#pragma once

#include "fronts_exec_consts.h"

template <class Exec>
struct FrontsRun {
  Exec &ex;
  const SpiceyProg &P;
  const SpiceyRun &R;
  double *W;        // this group's workspace (K = 1)
  int32_t *flags;   // WgCtx flags ([1] singular, [2] instance)
  int inst, valid;
  double *FW;       // this group's front workspace
  unsigned int *fl; // this group's done flags [2 nFronts]
  int T;            // threads of ONE workgroup
  unsigned long long *ticks;  // this group's per-front event times (profiling), or null

  // profiling: event `e` of front f happened now (ticks since this workgroup entered the forward sweep)
  SPICEY_HD void stamp(uint32_t f, int e, unsigned long long t0) const {
    if (ticks) ex.add_ticks(ticks + (size_t)f * 4 + e, t0);
  }

  SPICEY_HD bool foreign(uint32_t f) const { return R.fs_owner[f] != (uint32_t)ex.wg(); }
  // an LDS-resident front: the block (Mp rows of Mp + 17) plus the 16 x 16 block of L and the reciprocal pivots
  SPICEY_HD bool fits_lds(const SpiceyFront &F) const {
    return (size_t)F.Mp * (size_t)(F.Mp + SPICEY_FRONT_LDS_PAD) + 512 <= (size_t)R.front_lds_doubles;
  }

  // ---- assembly into A (row stride lda; LDS or global) ----------------------------------------------------
  // own part: zero, entries + right-hand side taken over from W (final once the levels below the cut are done: this
  // part runs BEFORE the front waits for children of other workgroups, inside what would be idle time)
  SPICEY_HD void assemble_own(const SpiceyFront &F, double *A, int lda) const {
    const int nel = F.Mp * lda;
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int i = t; i < nel; i += T) A[i] = 0.0;
    });
    ex.wg_phase([&](int t) {
      // gathers in batches of four: the index loads, then the four dependent W loads, are in flight together (one
      // dependent round trip to L2 per batch instead of per entry)
      const uint32_t *as = P.fr_asm + (size_t)F.asm0 * 2;
      SPICEY_NOUNROLL
      for (uint32_t i0 = (uint32_t)t; i0 < F.asm_n; i0 += 4u * (uint32_t)T) {
        uint32_t id[4], rc[4];
        double v[4];
        SPICEY_UNROLL
        for (int b = 0; b < 4; b++) {
          const uint32_t i = i0 + (uint32_t)b * (uint32_t)T;
          const bool have = i < F.asm_n;
          id[b] = have ? as[2 * i] : as[2 * i0];
          rc[b] = have ? as[2 * i + 1] : 0xffffffffu;
        }
        SPICEY_UNROLL
        for (int b = 0; b < 4; b++) v[b] = W[id[b]];
        SPICEY_UNROLL
        for (int b = 0; b < 4; b++)
          if (rc[b] != 0xffffffffu) A[(size_t)(rc[b] >> 16) * lda + (rc[b] & 0xffffu)] = v[b];
      }
      for (int r = F.p + t; r < F.Pp; r += T) A[(size_t)r * lda + r] = 1.0;  // identity padding of the pivot block
    });
    ex.mark(SPICEY_PH_U0 + 17);
  }
  // children's contribution blocks, in a fixed order; a child of another workgroup is waited for right before its turn
  // `irel`: LDS scratch for the child's row / column map (q <= 192 words) — the parent's own LDS front ends before it
  SPICEY_HD void assemble_children(const SpiceyFront &F, double *A, int lda, unsigned int epoch, uint32_t *irel) const {
    for (uint32_t ci = 0; ci < F.child_n; ci++) {  // extend-add
      const uint32_t cid = P.fr_child[F.child0 + ci];
      if (foreign(cid)) {
        ex.front_wait(fl + cid, epoch);
        ex.mark(SPICEY_PH_U0 + 4);
      }
      const SpiceyFront C = P.fr[cid];
      const double *Ac = FW + C.off;
      const uint32_t *rel = P.fr_rel + C.rel0;
      // the child's map goes to LDS first (one round trip for all of it): the block loop then has ONE dependent round trip
      // per turn — the child's values and the parent's (both addresses known) — instead of map -> parent entry -> sum
      ex.wg_phase([&](int t) {
        for (int i = t; i < C.q; i += T) irel[i] = rel[i];
      });
      ex.wg_phase([&](int t) {
        // a wave takes eight rows of the contribution block at a time: their loads are issued together
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int i0 = w * 8; i0 < C.q; i0 += nw * 8) {
          for (int j = lane; j <= C.q; j += 64) {
            const int sc = j < C.q ? C.Pp + j : C.Mp, dc = j < C.q ? (int)irel[j] : F.Mp;
            double v[8], o[8];
            double *dst[8];
            SPICEY_UNROLL
            for (int b = 0; b < 8; b++) {
              const int i = i0 + b < C.q ? i0 + b : i0;
              v[b] = Ac[(size_t)(C.Pp + i) * C.ld + sc];
              dst[b] = A + (size_t)irel[i] * lda + dc;
              o[b] = *dst[b];
            }
            SPICEY_UNROLL
            for (int b = 0; b < 8; b++)
              if (i0 + b < C.q) *dst[b] = o[b] + v[b];
          }
        }
      });
    }
  }

#if defined(__HIP_DEVICE_COMPILE__)
  // the body of diag_block for the 64 lanes of ONE wave (also called from inside a trailing-update phase: look-ahead)
  __device__ __forceinline__ void diag_block_wave(int t, double *Up, int su, double *Ld, double *Dinv) const {
    const int i = t >> 2, jq = t & 3;
    double a[4];
#ifdef SPICEY_DIAG_TIMING
    const long long c0 = clock64();
#endif
#pragma unroll
    for (int m = 0; m < 4; m++) a[m] = Up[(size_t)i * su + jq + 4 * m];
#ifdef SPICEY_DIAG_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long c1 = clock64();
#endif
    double dkeep = 0.0;  // lane k < 16 keeps 1 / pivot k
    double pmin = 1.0;   // smallest |pivot| so far (the identity padding behind the real rows has pivots of exactly 1)
    double *Lrow = Ld + i * SPICEY_FB;
#pragma unroll
    for (int k = 0; k < SPICEY_FB; k++) {
      // the pivot row first; the register that holds the NEXT pivot leads, so that its update is the first to finish
      const int m0 = k + 1 < SPICEY_FB ? (k + 1) >> 2 : 3;
      double u[4];
#pragma unroll
      for (int mm = 0; mm < 4; mm++) {
        const int m = (m0 + mm) & 3;
        u[m] = 4 * m + 3 > k ? __shfl(a[m], (k << 2) | jq) : 0.0;  // (columns <= k: nothing to update)
      }
      SPICEY_SCHED_FENCE;
      const double piv = spicey_readlane_f64(a[k >> 2], (k << 2) | (k & 3));
      const double d = spicey_rcp(piv);
      pmin = fmin(pmin, fabs(piv));
      dkeep = t == k ? d : dkeep;
      const double l = spicey_quad_bcast_f64(a[k >> 2], k & 3) * d;
      if (i > k) {  // (the store keeps this a branch: the updates inside run under its exec mask)
        Lrow[k] = l;  // (the four lanes of the row's quad write the same value)
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
          const int m = (m0 + mm) & 3;
          if (4 * m > k) a[m] = fma(-l, u[m], a[m]);
          else if (4 * m + 3 > k) a[m] = jq > (k & 3) ? fma(-l, u[m], a[m]) : a[m];  // the register that holds column k itself
        }
      }
      SPICEY_SCHED_FENCE;
    }
#ifdef SPICEY_DIAG_TIMING
    const long long c2 = clock64();
#endif
    if (t < SPICEY_FB) Dinv[t] = dkeep;
    if (t == 0 && pmin < SPICEY_EPS && valid) { flags[1] = 1; flags[2] = inst; }  // solveReal.ts:28
#pragma unroll
    for (int m = 0; m < 4; m++) Up[(size_t)i * su + jq + 4 * m] = a[m];
#ifdef SPICEY_DIAG_TIMING
    if (t == 0 && ex.prof) { ex.prof[60] += (unsigned long long)(c1 - c0); ex.prof[61] += (unsigned long long)(c2 - c1); ex.prof[62] += (unsigned long long)(clock64() - c2); ex.prof[63] += 1; }
#endif
  }
#endif
  // ---- 16 x 16 diagonal block of a panel: Up rows (stride su) in place, multipliers -> Ld, reciprocal pivots -> Dinv ----
  SPICEY_HD void diag_block(double *Up, int su, double *Ld, double *Dinv, int npiv_real) const {
#if defined(__HIP_DEVICE_COMPILE__)
    // wave 0, lane = (row i = lane / 4, column residue jq = lane % 4): the lane's four entries stay in registers for all
    // 16 elimination steps.  ONE wave issues an instruction every ~4.5 cycles, whatever it is (measured: 60 instructions per
    // step = 270 cycles with every cross-lane hop removed) — so the step is written for instruction count first: rows and
    // columns that a step leaves alone are masked by ONE exec-mask block per step (not a select pair per value), the
    // reciprocal pivots and the singularity verdict are kept in registers and leave once, at the end.  Second, the hops on
    // the chain pivot -> reciprocal -> multiplier -> update -> next pivot take the shortest path the hardware has: the
    // pivot is wave-uniform (v_readlane), the multiplier's source is in the lane's own quad (DPP quad_perm: a VALU move),
    // and only the pivot row goes through the LDS crossbar (ds_bpermute), issued ahead of the reciprocal.
    ex.wg_phase([&](int t) {
      if (t < 64) diag_block_wave(t, Up, su, Ld, Dinv);
    });
#else
    // the same arithmetic with the block in memory: one wave in lockstep, step k eliminates column k
    ex.wave_lockstep(64, SPICEY_FB, [&](int lane, int k) {
      const int i = lane >> 2, jq = lane & 3;
      const double piv = Up[(size_t)k * su + k];
      const double d = spicey_rcp(piv);
      if (lane == 0) {
        if (fabs(piv) < SPICEY_EPS && k < npiv_real && valid) { flags[1] = 1; flags[2] = inst; }
        Dinv[k] = d;
      }
      if (i > k) {
        const double l = Up[(size_t)i * su + k] * d;
        if (jq == 0) Ld[i * SPICEY_FB + k] = l;
        for (int j = jq; j < SPICEY_FB; j += 4)
          if (j > k) Up[(size_t)i * su + j] = fma(-l, Up[(size_t)k * su + j], Up[(size_t)i * su + j]);
      }
    });
#endif
  }

  // ---- triangular solves of a panel: one thread per row of the L panel, one per column of the U panel (rhs included) ----
  // Both kinds run the same right-looking recurrence on 16 values: v_k (x 1 / u_kk for an L row), then v_j -= v_k op(k, j)
  // for j > k — per entry the same operations in the same order as a left-looking dot product, with a dependent chain of 16
  // steps instead of 120.  The operands (a row of the U block / a column of the multipliers: the same LDS words for every
  // thread) are walked in 24 chunks of at most 8 — columns 1..7 of rows 0..7, then columns 8..15 of rows 0..7, then the
  // lower right triangle — and chunk n + 1 is fetched while chunk n is used: left to itself the compiler waits for every LDS
  // read right in front of the one or two operations that use it (60 exposed round trips per row; 16 with whole rows fetched
  // step by step), and two whole rows in flight cost 60 registers where this costs 32.
  template <bool LROW, class OP>
  SPICEY_HD void trsm16(double *v, const double *Dinv, OP op) const {
    double buf[2][8], dk[2];
    // chunk n: n < 8: (k = n, j in [n + 1, 8));  n < 16: (k = n - 8, j in [8, 16));  else (k = n - 8, j in [n - 7, 16))
    auto fetch = [&](int n) {
      const int k = n < 8 ? n : n - 8, j0 = n < 8 ? n + 1 : (n < 16 ? 8 : n - 7), j1 = n < 8 ? 8 : 16;
      SPICEY_UNROLL
      for (int j = j0; j < j1; j++) buf[n & 1][j & 7] = op(k, j);
      if (LROW && (n < 8 || n >= 16)) dk[n & 1] = Dinv[k];
    };
    fetch(0);
    SPICEY_UNROLL
    for (int n = 0; n < 24; n++) {
      if (n + 1 < 24) fetch(n + 1);
      SPICEY_SCHED_FENCE;
      const int k = n < 8 ? n : n - 8, j0 = n < 8 ? n + 1 : (n < 16 ? 8 : n - 7), j1 = n < 8 ? 8 : 16;
      if (LROW && (n < 8 || n >= 16)) v[k] *= dk[n & 1];
      SPICEY_UNROLL
      for (int j = j0; j < j1; j++) v[j] = fma(-v[k], buf[n & 1][j & 7], v[j]);
      SPICEY_SCHED_FENCE;
    }
  }
  SPICEY_HD void panel_trsm(double *Up, int su, double *Lp, int lpld, const double *Ld, const double *Dinv, int nL, int nU, int t) const {
    // (the U columns start on a wave boundary behind the L rows: a wave that held both kinds ran both branches one after
    // the other and was the phase's critical path)
    const int nL64 = (nL + 63) & ~63;
    SPICEY_NOUNROLL
    for (int it = t; it < nL64 + nU; it += T) {
      double v[SPICEY_FB];
      if (it >= nL && it < nL64) continue;
      if (it < nL) {
        double *row = Lp + (size_t)it * lpld;
        SPICEY_UNROLL
        for (int k = 0; k < SPICEY_FB; k++) v[k] = row[k];
        trsm16<true>(v, Dinv, [&](int k, int j) { return Up[(size_t)k * su + j]; });
        SPICEY_UNROLL
        for (int k = 0; k < SPICEY_FB; k++) row[k] = v[k];
      } else {
        const int c = SPICEY_FB + (it - nL64);
        SPICEY_UNROLL
        for (int k = 0; k < SPICEY_FB; k++) v[k] = Up[(size_t)k * su + c];
        trsm16<false>(v, Dinv, [&](int k, int j) { return Ld[j * SPICEY_FB + k]; });
        SPICEY_UNROLL
        for (int k = 1; k < SPICEY_FB; k++) Up[(size_t)k * su + c] = v[k];
      }
    }
  }

  // ---- trailing update C[i][j] -= sum_k Lp[i][k] Up[k][j], i < nrow (multiple of 16), j < ncol ------------------
  // Device: 16 x 16 tiles, four v_mfma_f64_16x16x4 each (A = -L tile, B = U tile; operand maps: lane l holds A[l & 15][l >> 4]
  // and B[l >> 4][l & 15], result register r holds C[(l >> 4) + 4 r][l & 15]); the columns up to the next multiple of 16
  // exist behind C and the U panel (zero padding, left unchanged).  Host: the plain sum, k ascending.
  // LOOK (look-ahead): wave 0 takes tile (0, 0) only — the next panel's diagonal block, which it then factors inside the
  // same phase — and the other waves share the rest.
  template <int NT = 2, bool LOOK = false>
  SPICEY_HD void trailing(double *C, int ldc, const double *Lp, int lpld, const double *Up, int su, int nrow, int ncol, int t) const {
    const int nw = T >> 6, w = t >> 6, lane = t & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    // Written for instruction count (a wave issues one instruction per ~4.5 cycles): the tile walk of a wave is kept in
    // scalar registers and advances without a division, the lane-dependent parts of the three operand addresses are formed
    // once per call, and all 12 operand loads of a tile (NT tiles per turn) are issued before its first MFMA — one exposed
    // LDS / L2 round trip per turn instead of one per MFMA (the first version: ~75 instructions and 4 round trips per tile).
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int ws = SPICEY_UNIFORM(w);
    const int tr = nrow >> 4, tc = (ncol + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    double *cl = C + (size_t)lk * ldc + li;          // + (16 ti + 4 r) ldc + 16 tj
    const double *al = Lp + (size_t)li * lpld + lk;  // + 16 ti lpld + 4 kk
    const double *bl = Up + (size_t)lk * su + li;    // + 4 kk su + 16 tj
    const int stride = LOOK ? (ws == 0 ? tr * tc : nw - 1) : nw;  // (LOOK: wave 0 stops after tile 0, waves 1.. walk tiles 1.. in steps of nw - 1)
    const int dti = stride / tc, dtj = stride - dti * tc;  // the step from one tile of this wave to its next (row-major walk)
    int ti = ws / tc, tj = ws - ti * tc;
    while (ti < tr) {
      double *c0[NT];
      d4 acc[NT], av[NT], bv[NT];
      bool have[NT];
      SPICEY_UNROLL
      for (int b = 0; b < NT; b++) {
        have[b] = ti < tr;  // (wave-uniform; an absent tile repeats the turn's first one and is not stored)
        const int ui = have[b] ? ti : 0, uj = have[b] ? tj : 0;
        c0[b] = cl + ((size_t)ui * 16 * ldc + (size_t)uj * 16);
        const double *la = al + (size_t)ui * 16 * lpld;
        const double *ub = bl + (size_t)uj * 16;
        acc[b][0] = c0[b][0]; acc[b][1] = c0[b][(size_t)4 * ldc]; acc[b][2] = c0[b][(size_t)8 * ldc]; acc[b][3] = c0[b][(size_t)12 * ldc];
        av[b][0] = la[0]; av[b][1] = la[4]; av[b][2] = la[8]; av[b][3] = la[12];
        bv[b][0] = ub[0]; bv[b][1] = ub[(size_t)4 * su]; bv[b][2] = ub[(size_t)8 * su]; bv[b][3] = ub[(size_t)12 * su];
        tj += dtj; ti += dti;
        if (tj >= tc) { tj -= tc; ti++; }
      }
      SPICEY_SCHED_FENCE;
      SPICEY_UNROLL
      for (int b = 0; b < NT; b++) {
        SPICEY_UNROLL
        for (int kk = 0; kk < 4; kk++) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[b][kk], bv[b][kk], acc[b], 0, 0, 0);
      }
      SPICEY_UNROLL
      for (int b = 0; b < NT; b++)
        if (have[b]) { c0[b][0] = acc[b][0]; c0[b][(size_t)4 * ldc] = acc[b][1]; c0[b][(size_t)8 * ldc] = acc[b][2]; c0[b][(size_t)12 * ldc] = acc[b][3]; }
    }
#else
    const int nchunk = (ncol + 63) >> 6;
    for (int pr = w; pr < nrow * nchunk; pr += nw) {
      const int i = pr / nchunk, j = (pr - i * nchunk) * 64 + lane;
      if (j >= ncol) continue;
      const double *l = Lp + (size_t)i * lpld;
      double acc = C[(size_t)i * ldc + j];
      for (int k = 0; k < SPICEY_FB; k++) acc = fma(-l[k], Up[(size_t)k * su + j], acc);
      C[(size_t)i * ldc + j] = acc;
    }
#endif
  }

  // ---- LEFT-looking update of a block of a staged front: C[i][j] -= sum over the panels pj < npan, k < 16 of
  //      L_pj[grow0 + i][k] * U[16 pj + k][gcol0 + j]   (i < nrow: a multiple of 16; j < ncol)
  // L_pj = the multipliers of panel pj, kept in LDS for the whole front (Lall + loff(pj), rows counted from pivot 16 pj + 16,
  // stride SPICEY_LPLD); U = the finished U rows in the front's block of the workspace (row stride ld).  C may be in LDS (the
  // panel being formed) or in the workspace (the contribution block).  Same products in the same order per entry as the
  // right-looking sweep (panel after panel, k ascending, the same MFMA tiling): bit-identical to it.
  SPICEY_HD static int loff(int pj, int Mp) { return SPICEY_LPLD * pj * (Mp - SPICEY_FB - (SPICEY_FB / 2) * (pj - 1)); }
  template <int CH = 4>
  SPICEY_HD void trailing_left(double *C, int ldc, int nrow, int ncol, int npan, const double *Lall, int Mp, int grow0, const double *A, int ld, int gcol0,
                               int t) const {
    const int nw = T >> 6, w = t >> 6, lane = t & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    // (instruction economy as in trailing(): scalar tile walk, lane parts of the addresses formed once.)  The U rows come
    // from the workspace (L2, ~1 us per round trip) and the MFMAs of a tile are nothing beside that, so what counts is round
    // trips: one tile per turn, the operands of CH panels fetched together — ceil(npan / CH) round trips per tile (one per
    // panel, as first written, was most of a staged front's time).
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int ws = SPICEY_UNIFORM(w);
    const int tr = nrow >> 4, tc = (ncol + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    double *cl = C + (size_t)lk * ldc + li;
    const double *al = Lall + (size_t)(grow0 + li - SPICEY_FB) * SPICEY_LPLD + lk;  // + 16 ti LPLD + [loff(pj) - 16 pj LPLD] + 4 kk
    const double *bl = A + (size_t)lk * ld + gcol0 + li;                            // + 16 pj ld + 16 tj + 4 kk ld
    const int dti = nw / tc, dtj = nw - dti * tc;
    int ti = ws / tc, tj = ws - ti * tc;
    while (ti < tr) {
      double *c0 = cl + ((size_t)ti * 16 * ldc + (size_t)tj * 16);
      const double *la = al + (size_t)ti * 16 * SPICEY_LPLD;
      const double *ub = bl + (size_t)tj * 16;
      d4 acc;
      acc[0] = c0[0]; acc[1] = c0[(size_t)4 * ldc]; acc[2] = c0[(size_t)8 * ldc]; acc[3] = c0[(size_t)12 * ldc];
      int lo = 0;  // loff(pj, Mp) - 16 pj LPLD, advanced by LPLD (Mp - 32 - 16 pj) per panel
      for (int p0 = 0; p0 < npan; p0 += CH) {
        d4 av[CH], bv[CH];
        SPICEY_UNROLL
        for (int c = 0; c < CH; c++) {
          const bool on = p0 + c < npan;  // (wave-uniform; an absent panel repeats the last one and is not multiplied)
          const double *x = la + lo;
          av[c][0] = x[0]; av[c][1] = x[4]; av[c][2] = x[8]; av[c][3] = x[12];
          bv[c][0] = ub[0]; bv[c][1] = ub[(size_t)4 * ld]; bv[c][2] = ub[(size_t)8 * ld]; bv[c][3] = ub[(size_t)12 * ld];
          if (p0 + c + 1 < npan) { lo += SPICEY_LPLD * (Mp - 2 * SPICEY_FB - SPICEY_FB * (p0 + c)); ub += (size_t)SPICEY_FB * ld; }
          (void)on;
        }
        SPICEY_SCHED_FENCE;
        SPICEY_UNROLL
        for (int c = 0; c < CH; c++) {
          if (p0 + c < npan) {
            SPICEY_UNROLL
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[c][kk], bv[c][kk], acc, 0, 0, 0);
          }
        }
      }
      c0[0] = acc[0]; c0[(size_t)4 * ldc] = acc[1]; c0[(size_t)8 * ldc] = acc[2]; c0[(size_t)12 * ldc] = acc[3];
      tj += dtj; ti += dti;
      if (tj >= tc) { tj -= tc; ti++; }
    }
#else
    const int nchunk = (ncol + 63) >> 6;
    for (int pr = w; pr < nrow * nchunk; pr += nw) {
      const int i = pr / nchunk, j = (pr - i * nchunk) * 64 + lane;
      if (j >= ncol) continue;
      double acc = C[(size_t)i * ldc + j];
      for (int pj = 0; pj < npan; pj++) {
        const double *l = Lall + loff(pj, Mp) + (size_t)(grow0 + i - SPICEY_FB * pj - SPICEY_FB) * SPICEY_LPLD;
        for (int k = 0; k < SPICEY_FB; k++) acc = fma(-l[k], A[(size_t)(SPICEY_FB * pj + k) * ld + gcol0 + j], acc);
      }
      C[(size_t)i * ldc + j] = acc;
    }
#endif
  }
  // the same update with a wave per tile COLUMN (for fronts with more panels than one fetch of the tile walk holds)
  template <int CH = 5, int RT = 4>
  SPICEY_HD void trailing_left_cols(double *C, int ldc, int nrow, int ncol, int npan, const double *Lall, int Mp, int grow0, const double *A, int ld, int gcol0,
                               int t) const {
    const int nw = T >> 6, w = t >> 6, lane = t & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    // Instruction economy as in trailing(): wave-uniform walk in scalar registers, lane parts of the addresses formed once.
    // The U rows come from the workspace (L2, ~1 us per round trip) and the MFMAs of a tile are nothing beside that, so the
    // walk is built around round trips: a wave owns a COLUMN of tiles — the U operands of that column (CH panels at a time;
    // every staged front of the 100 x 100 mesh has at most 5) are fetched once and serve all its row tiles.  With fewer tile
    // columns than waves the rows of a column
    // are dealt out to nw / tc waves.  (Tile by tile, one fetch per panel: 7 x 5 round trips per wave on the contribution
    // block of the (71, 100) front.)
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int ws = SPICEY_UNIFORM(w);
    const int tr = nrow >> 4, tc = (ncol + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    const int share = tc < nw ? nw / tc : 1;           // waves per tile column
    const int col0 = tc < nw ? ws % tc : ws, cstep = tc < nw ? tc : nw;
    const int row0 = tc < nw ? ws / tc : 0;            // (waves beyond share * tc: row0 >= share, they get no rows)
    if (row0 >= share) return;
    double *cl = C + (size_t)lk * ldc + li;
    const double *al = Lall + (size_t)(grow0 + li - SPICEY_FB) * SPICEY_LPLD + lk;  // + 16 ti LPLD + [loff(pj) - 16 pj LPLD] + 4 kk
    const double *bl = A + (size_t)lk * ld + gcol0 + li;                            // + 16 pj ld + 16 tj + 4 kk ld
    for (int tj = col0; tj < tc; tj += cstep) {
      int lo0 = 0;  // loff(p0, Mp) - 16 p0 LPLD
      for (int p0 = 0; p0 < npan; p0 += CH) {
        d4 bv[CH];
        SPICEY_UNROLL
        for (int c = 0; c < CH; c++) {
          const double *y = bl + (size_t)tj * 16 + (size_t)(p0 + c < npan ? p0 + c : p0) * SPICEY_FB * ld;  // (an absent panel repeats the chunk's first; not multiplied)
          bv[c][0] = y[0]; bv[c][1] = y[(size_t)4 * ld]; bv[c][2] = y[(size_t)8 * ld]; bv[c][3] = y[(size_t)12 * ld];
        }
        // the C tiles of this wave's rows of the column, RT at a time: ALL their loads go out together with the U operands
        // above — one round trip to the workspace per RT tiles (with the next tile fetched under the current one it was one
        // per tile; RT = 8 spills)
        for (int tg = row0; tg < tr; tg += RT * share) {
          d4 acc[RT];
          SPICEY_UNROLL
          for (int r = 0; r < RT; r++) {
            const int ti = tg + r * share < tr ? tg + r * share : tg;  // (wave-uniform; an absent tile repeats the group's first and is not stored)
            const double *cp = cl + ((size_t)ti * 16 * ldc + (size_t)tj * 16);
            acc[r][0] = cp[0]; acc[r][1] = cp[(size_t)4 * ldc]; acc[r][2] = cp[(size_t)8 * ldc]; acc[r][3] = cp[(size_t)12 * ldc];
          }
          SPICEY_UNROLL
          for (int r = 0; r < RT; r++) {
            const int ti = tg + r * share;
            if (ti < tr) {
              const double *la = al + (size_t)ti * 16 * SPICEY_LPLD;
              int lo = lo0;
              SPICEY_UNROLL
              for (int c = 0; c < CH; c++) {
                if (p0 + c < npan) {
                  const double *x = la + lo;
                  d4 av;
                  av[0] = x[0]; av[1] = x[4]; av[2] = x[8]; av[3] = x[12];
                  SPICEY_UNROLL
                  for (int kk = 0; kk < 4; kk++) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[kk], bv[c][kk], acc[r], 0, 0, 0);
                  lo += SPICEY_LPLD * (Mp - 2 * SPICEY_FB - SPICEY_FB * (p0 + c));
                }
              }
              double *c0 = cl + ((size_t)ti * 16 * ldc + (size_t)tj * 16);
              c0[0] = acc[r][0]; c0[(size_t)4 * ldc] = acc[r][1]; c0[(size_t)8 * ldc] = acc[r][2]; c0[(size_t)12 * ldc] = acc[r][3];
            }
          }
        }
        for (int c = 0; c < CH && p0 + c < npan; c++) lo0 += SPICEY_LPLD * (Mp - 2 * SPICEY_FB - SPICEY_FB * (p0 + c));
      }
    }
#else
    const int nchunk = (ncol + 63) >> 6;
    for (int pr = w; pr < nrow * nchunk; pr += nw) {
      const int i = pr / nchunk, j = (pr - i * nchunk) * 64 + lane;
      if (j >= ncol) continue;
      double acc = C[(size_t)i * ldc + j];
      for (int pj = 0; pj < npan; pj++) {
        const double *l = Lall + loff(pj, Mp) + (size_t)(grow0 + i - SPICEY_FB * pj - SPICEY_FB) * SPICEY_LPLD;
        for (int k = 0; k < SPICEY_FB; k++) acc = fma(-l[k], A[(size_t)(SPICEY_FB * pj + k) * ld + gcol0 + j], acc);
      }
      C[(size_t)i * ldc + j] = acc;
    }
#endif
  }

  // ---- blocked partial LU of the first Pp pivots, front resident in LDS (row stride lda = Mp + 17) ----------------
  SPICEY_HD void factor_lds(const SpiceyFront &F, double *A, int lda, double *scr) const {
    double *Ld = scr, *Dinv = scr + SPICEY_FB * SPICEY_FB;
    for (int j0 = 0; j0 < F.Pp; j0 += SPICEY_FB) {
      const int wU = F.Mp + 1 - j0, nL = F.Mp - j0 - SPICEY_FB;
      double *Up = A + (size_t)j0 * lda + j0, *Lp = A + (size_t)(j0 + SPICEY_FB) * lda + j0;
      ex.mark(SPICEY_PH_U0 + 13);
#if defined(__HIP_DEVICE_COMPILE__)
      if (j0 == 0 || T < 128)  // (later panels: factored by wave 0 inside the previous panel's trailing update)
#endif
      diag_block(Up, lda, Ld, Dinv, F.p - j0);
      ex.mark(SPICEY_PH_U0 + 14);
      ex.wg_phase([&](int t) { panel_trsm(Up, lda, Lp, lda, Ld, Dinv, nL, wU - SPICEY_FB, t); });
      ex.mark(SPICEY_PH_U0 + 15);
#if defined(__HIP_DEVICE_COMPILE__)
      // look-ahead: wave 0 updates the next panel's diagonal block first and factors it while the other waves finish the
      // trailing update (the panel's two serial parts — that block and the triangular solves — no longer add up with it)
      const bool more = j0 + SPICEY_FB < F.Pp;
      if (more && T >= 128) {
        ex.wg_phase([&](int t) {
          if (t < SPICEY_FB) Up[(size_t)t * lda + t] = Dinv[t];
          trailing<2, true>(Lp + SPICEY_FB, lda, Lp, lda, Up + SPICEY_FB, lda, nL, wU - SPICEY_FB, t);
          if (t < 64) diag_block_wave(t, Lp + SPICEY_FB, lda, Ld, Dinv);  // (Lp + 16 = the (j0 + 16, j0 + 16) corner: the next Up)
        });
        ex.mark(SPICEY_PH_U0 + 16);
        continue;
      }
#endif
      ex.wg_phase([&](int t) {
        if (t < SPICEY_FB) Up[(size_t)t * lda + t] = Dinv[t];  // reciprocal pivots on the diagonal: what the backward solve reads
        trailing(Lp + SPICEY_FB, lda, Lp, lda, Up + SPICEY_FB, lda, nL, wU - SPICEY_FB, t);
      });
      ex.mark(SPICEY_PH_U0 + 16);
    }
  }
  // U rows (backward solve) and contribution block (parent's assembly) of an LDS-resident front -> front workspace
  SPICEY_HD void store_lds_front(const SpiceyFront &F, const double *A, int lda) const {
    double *G = FW + F.off;
    ex.wg_phase([&](int t) {
      const int nw = T >> 6, w = t >> 6, lane = t & 63;
      for (int i = w; i < F.p; i += nw)
        for (int c = lane; c <= F.Mp; c += 64) G[(size_t)i * F.ld + c] = A[(size_t)i * lda + c];
      for (int i = w; i < F.q; i += nw) {
        const double *src = A + (size_t)(F.Pp + i) * lda;
        double *dst = G + (size_t)(F.Pp + i) * F.ld;
        for (int c = lane; c <= F.q; c += 64) {
          const int cc = c < F.q ? F.Pp + c : F.Mp;
          dst[cc] = src[cc];
        }
      }
    });
  }

  // ---- the same for a front that stays in the workspace: one panel at a time staged through LDS --------------------
  // LEFT-looking form of the same factorisation, for staged fronts whose multipliers fit LDS (all of the 100 x 100 mesh's):
  // the right-looking sweep below reads and writes the whole trailing block of the workspace once per panel (0.5 MB per
  // panel of a 192-row front, at the 60 GB/s one CU gets from L2: 8 us of a 15.6 us panel); here a panel is staged into
  // LDS, takes the updates of ALL earlier panels there (their multipliers stay in LDS, their U rows are read back from the
  // workspace: 26 KB per earlier panel), is factored, and only its 16 U rows go back; the contribution block takes the
  // products of all panels in ONE pass at the end.  Traffic per front: ~0.9 MB instead of ~3 MB.
  SPICEY_HD int left_lds_need(const SpiceyFront &F) const {
    return loff(F.Pp / SPICEY_FB, F.Mp) + SPICEY_FB * F.ld + SPICEY_FB * SPICEY_FB + SPICEY_FB;
  }
  SPICEY_HD void factor_global_left(const SpiceyFront &F) const {
    double *A = FW + F.off;
    double *lds = ex.lds();
    const int Mp = F.Mp, ld = F.ld;
    double *Lall = lds;                                   // multipliers of every panel, panel pj at loff(pj, Mp)
    double *Up = Lall + loff(F.Pp / SPICEY_FB, Mp);       // the 16 pivot rows of the panel being formed, stride su
    double *Ld = Up + (size_t)SPICEY_FB * ld, *Dinv = Ld + SPICEY_FB * SPICEY_FB;
    for (int j0 = 0; j0 < F.Pp; j0 += SPICEY_FB) {
      const int pj = j0 / SPICEY_FB;
      const int su = ld - j0, wU = Mp + 1 - j0, nL = Mp - j0 - SPICEY_FB;
      double *Lp = Lall + loff(pj, Mp);
      ex.wg_phase([&](int t) {  // stage the panel: its 16 rows (from the diagonal block to the right-hand side), its column block below
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        // (all loads of a thread first, then its stores: one L2 round trip for the rows instead of one per 64 columns)
        for (int k = w; k < SPICEY_FB; k += 2 * nw) {
          const bool two = k + nw < SPICEY_FB;
          const double *s0 = A + (size_t)(j0 + k) * ld + j0, *s1 = A + (size_t)(j0 + (two ? k + nw : k)) * ld + j0;
          double *d0 = Up + (size_t)k * su, *d1 = Up + (size_t)(k + nw) * su;
          for (int cb = 0; cb < su; cb += 256) {
            double v0[4], v1[4];
            SPICEY_UNROLL
            for (int b = 0; b < 4; b++) {
              const int c = cb + 64 * b + lane;
              v0[b] = s0[c < su ? c : 0]; v1[b] = s1[c < su ? c : 0];
            }
            SPICEY_UNROLL
            for (int b = 0; b < 4; b++) {
              const int c = cb + 64 * b + lane;
              if (c < su) { d0[c] = v0[b]; if (two) d1[c] = v1[b]; }
            }
          }
        }
        // the column block below: 16 lanes per row, four rows of a thread in flight together (one L2 round trip per four)
        const double *src = A + (size_t)(j0 + SPICEY_FB + (t >> 4)) * ld + j0 + (t & 15);
        double *dst = Lp + (size_t)(t >> 4) * SPICEY_LPLD + (t & 15);
        const size_t sstep = (size_t)(T >> 4) * ld, dstep = (size_t)(T >> 4) * SPICEY_LPLD;
        SPICEY_NOUNROLL
        for (int r0 = t >> 4; r0 < nL; r0 += 4 * (T >> 4)) {
          double v[4];
          SPICEY_UNROLL
          for (int b = 0; b < 4; b++) v[b] = src[(r0 + b * (T >> 4) < nL ? b : 0) * sstep];
          SPICEY_UNROLL
          for (int b = 0; b < 4; b++)
            if (r0 + b * (T >> 4) < nL) dst[b * dstep] = v[b];
          src += 4 * sstep; dst += 4 * dstep;
        }
      });
      ex.mark(SPICEY_PH_U0 + 24);
      if (pj > 0)
        ex.wg_phase([&](int t) {  // the updates of all earlier panels, in panel order
          trailing_left(Lp, SPICEY_LPLD, nL, SPICEY_FB, pj, Lall, Mp, j0 + SPICEY_FB, A, ld, j0, t);
          trailing_left(Up, su, SPICEY_FB, wU, pj, Lall, Mp, j0, A, ld, j0, t);
        });
      ex.mark(SPICEY_PH_U0 + 27);
      diag_block(Up, su, Ld, Dinv, F.p - j0);
      ex.mark(SPICEY_PH_U0 + 25);
      ex.wg_phase([&](int t) { panel_trsm(Up, su, Lp, SPICEY_LPLD, Ld, Dinv, nL, wU - SPICEY_FB, t); });
      ex.mark(SPICEY_PH_U0 + 26);
      ex.wg_phase([&](int t) {  // the finished U rows (reciprocal pivots on the diagonal) -> workspace
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int k = w; k < SPICEY_FB; k += nw) {
          double *dst = A + (size_t)(j0 + k) * ld + j0;
          for (int c = lane; c < wU; c += 64) dst[c] = c == k ? Dinv[k] : Up[(size_t)k * su + c];
        }
      });
      ex.mark(SPICEY_PH_U0 + 28);
    }
    // contribution block (rows and columns of the boundary, right-hand side included): every panel's products in one pass
    if (Mp > F.Pp)
      ex.wg_phase([&](int t) {
        trailing_left_cols(A + (size_t)F.Pp * ld + F.Pp, ld, Mp - F.Pp, Mp + 1 - F.Pp, F.Pp / SPICEY_FB, Lall, Mp, F.Pp, A, ld, F.Pp, t);
      });
    ex.mark(SPICEY_PH_U0 + 29);
  }
  SPICEY_HD void factor_global(const SpiceyFront &F) const {
    if (!R.front_right_looking && left_lds_need(F) <= SPICEY_FRONT_LDS_DOUBLES) { factor_global_left(F); return; }
    double *A = FW + F.off;
    double *lds = ex.lds();
    for (int j0 = 0; j0 < F.Pp; j0 += SPICEY_FB) {
      const int su = F.ld - j0;            // row stride of the U panel in LDS (the whole padded row: zero columns behind the rhs)
      const int wU = F.Mp + 1 - j0;        // its used width: columns j0 .. Mp (right-hand side) inclusive
      const int nL = F.Mp - j0 - SPICEY_FB;  // rows below the diagonal block
      double *Up = lds, *Ld = Up + (size_t)SPICEY_FB * su, *Dinv = Ld + SPICEY_FB * SPICEY_FB, *Lp = Dinv + SPICEY_FB;
      ex.wg_phase([&](int t) {
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int k = w; k < SPICEY_FB; k += nw) {
          const double *src = A + (size_t)(j0 + k) * F.ld + j0;
          for (int c = lane; c < su; c += 64) Up[(size_t)k * su + c] = src[c];
        }
        SPICEY_NOUNROLL
        for (int i = t; i < nL * SPICEY_FB; i += T) {
          const int r = i >> 4, k = i & 15;
          Lp[(size_t)r * SPICEY_LPLD + k] = A[(size_t)(j0 + SPICEY_FB + r) * F.ld + j0 + k];
        }
      });
      ex.mark(SPICEY_PH_U0 + 24);
      diag_block(Up, su, Ld, Dinv, F.p - j0);
      ex.mark(SPICEY_PH_U0 + 25);
      ex.wg_phase([&](int t) { panel_trsm(Up, su, Lp, SPICEY_LPLD, Ld, Dinv, nL, wU - SPICEY_FB, t); });
      ex.mark(SPICEY_PH_U0 + 26);
      ex.wg_phase([&](int t) {
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int k = w; k < SPICEY_FB; k += nw) {
          double *dst = A + (size_t)(j0 + k) * F.ld + j0;
          for (int c = lane; c < wU; c += 64) dst[c] = c == k ? Dinv[k] : Up[(size_t)k * su + c];
        }
        trailing<SPICEY_TRAIL_TILES_STAGED>(A + (size_t)(j0 + SPICEY_FB) * F.ld + j0 + SPICEY_FB, F.ld, Lp, SPICEY_LPLD, Up + SPICEY_FB, su, nL, wU - SPICEY_FB, t);
      });
      ex.mark(SPICEY_PH_U0 + 27);
    }
  }

  // ---- backward substitution of one front -----------------------------------------------------------------
  // Own data first (right-hand side after the forward sweep, the U rows): this part does not need the ancestors' unknowns and
  // runs BEFORE the front waits for its parent's workgroup; then the boundary product straight from W.
  // The front's U rows (pivot block and boundary block, p x Mp) are copied into LDS in that first part when they fit (every
  // front of the 100 x 100 mesh: at most 127 x 128): the block loop below then has no workspace (L2) round trip on its
  // dependent chain — 8 blocks x (16 lock-step LDS round trips + a strided global read of 16 columns) were 25 us on the root.
  // solve16: the 16 unknowns of one diagonal block.  Device: lane = row, its right-hand side and its row of the block in
  // registers, the solved unknown travels by v_readlane (35 cycles per step against 190 through LDS); host: the same
  // operations with the block in memory.
  SPICEY_HD void solve16(const double *Ub, int lub, int b0, int p, double *tt, double *xs) const {
#if defined(__HIP_DEVICE_COMPILE__)
    ex.wg_phase([&](int t) {
      if (t >= SPICEY_FB) return;
      const int r = b0 + t;
      double db[SPICEY_FB];
      SPICEY_UNROLL
      for (int k = 0; k < SPICEY_FB; k++) db[k] = r < p ? Ub[(size_t)t * lub + k] : (t == k ? 1.0 : 0.0);  // (rows [p, Pp): identity padding)
      double tv = tt[r], xv = 0.0;
      SPICEY_UNROLL
      for (int k = SPICEY_FB - 1; k >= 0; k--) {
        const double x = spicey_readlane_f64(tv * db[k], k);
        if (t == k) xv = x;
        if (t < k) tv = fma(-db[k], x, tv);
      }
      xs[r] = xv;
    });
#else
    // one wave, 16 lanes in lockstep: step s solves x of row k = 15 - s and removes it from the rows above it
    ex.wave_lockstep(SPICEY_FB, SPICEY_FB, [&](int lane, int s2) {
      const int k = SPICEY_FB - 1 - s2;
      if (lane > k) return;
      const double dkk = b0 + k < p ? Ub[(size_t)k * lub + k] : 1.0;
      const double x = tt[b0 + k] * dkk;
      if (lane == k) xs[b0 + k] = x;
      else tt[b0 + lane] = fma(-(b0 + lane < p ? Ub[(size_t)lane * lub + k] : 0.0), x, tt[b0 + lane]);
    });
#endif
  }
  // `in_lds`: the front is still in LDS from the forward sweep (stays_in_lds: q = 0) — its rows are read where they are, the
  // scratch of this solve goes behind it (where the factorisation kept the multipliers of a diagonal block)
  // (a template, not a run-time choice: a select between an LDS and a global pointer sends this hipcc into "Illegal
  // instruction detected: Operand has incorrect register class")
  template <bool in_lds = false>
  SPICEY_HD void solve(const SpiceyFront &F, bool wait_parent, unsigned int epoch, uint32_t f, unsigned long long t0) const {
    const int lda_l = F.Mp + SPICEY_FRONT_LDS_PAD;
    const double *A;
    double *lds;
    if constexpr (in_lds) { A = ex.lds(); lds = ex.lds() + (size_t)F.Mp * lda_l; }
    else { A = FW + F.off; lds = ex.lds(); }
    const int ldA = in_lds ? lda_l : F.ld;
    // xs: the front's unknowns [Pp], behind them the boundary unknowns xb [Mp - Pp] once the parent's are there
    double *xs = lds, *tt = xs + F.Mp, *part = tt + F.Pp;
    uint32_t *ibnd = (uint32_t *)(part + (size_t)F.Pp * 4);  // the boundary's unknown ids [q]: fetched before the wait
    double *Ul = part + (size_t)F.Pp * 4 + (F.Mp >> 1) + 1, *xb = xs + F.Pp;
    const uint32_t *bnd = P.fr_bnd + F.bnd0;
    const int lul = F.Mp + 1;  // (odd: a thread-per-row walk is bank-conflict free)
    const bool res = !in_lds && (size_t)(Ul - lds) + (size_t)F.p * (size_t)lul <= (size_t)R.front_lds_doubles;
    const double *U;  // where the block loop reads the U rows
    if constexpr (in_lds) U = A;
    else U = res ? Ul : A;
    const int lu = res ? lul : ldA;
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int i = t; i < F.Pp; i += T) tt[i] = i < F.p ? A[(size_t)i * ldA + F.Mp] : 0.0;
      SPICEY_NOUNROLL
      for (int j = t; j < F.q; j += T) ibnd[j] = bnd[j];
      if (res) {
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int i = w; i < F.p; i += nw) {
          const double *src = A + (size_t)i * F.ld;
          double *dst = Ul + (size_t)i * lul;
          for (int c0 = (i & ~15) + lane; c0 < F.Mp; c0 += 64) dst[c0] = src[c0];  // (from the row's diagonal block on)
        }
      }
    });
    if (wait_parent) ex.front_wait(fl + P.nFronts + F.parent, epoch);
    stamp(f, 2, t0);
    ex.mark(SPICEY_PH_U0 + 10);
    if (F.q > 0) {
      // the boundary unknowns into LDS first — ONE round trip to L2 for all of them (ids already here) — so that the product
      // below runs from LDS alone (index -> unknown chains, four at a time per thread, were 2 x q / 16 dependent round
      // trips: 8 us on the (71, 100) front)
      ex.wg_phase([&](int t) {
        const double *xW = W + (size_t)P.nLU;
        SPICEY_NOUNROLL
        for (int j = t; j < F.q; j += T) xb[j] = xW[ibnd[j]];
      });
      ex.wg_phase([&](int t) {  // t = y_P - U_PB x_B: four partial sums per row, combined in a fixed order
        SPICEY_NOUNROLL
        for (int it = t; it < F.p * 4; it += T) {
          const int i = it >> 2, sg = it & 3;
          const double *row = U + (size_t)i * lu + F.Pp;
          double s = 0.0;
          int j = sg;
          for (; j + 12 < F.q; j += 16) {  // four matrix loads in flight; the sum keeps its order
            const double r0 = row[j], r1 = row[j + 4], r2 = row[j + 8], r3 = row[j + 12];
            const double x0 = xb[j], x1 = xb[j + 4], x2 = xb[j + 8], x3 = xb[j + 12];
            s = fma(r0, x0, s); s = fma(r1, x1, s); s = fma(r2, x2, s); s = fma(r3, x3, s);
          }
          for (; j < F.q; j += 4) s = fma(row[j], xb[j], s);
          part[it] = s;
        }
      });
      ex.wg_phase([&](int t) {
        SPICEY_NOUNROLL
        for (int i = t; i < F.p; i += T) tt[i] -= (part[4 * i] + part[4 * i + 1]) + (part[4 * i + 2] + part[4 * i + 3]);
      });
    }
    for (int b0 = F.Pp - SPICEY_FB; b0 >= 0; b0 -= SPICEY_FB) {
      solve16(U + (size_t)b0 * lu + b0, lu, b0, F.p, tt, xs);
      if (b0 == 0) break;
      ex.wg_phase([&](int t) {  // rows above the block lose its 16 solved unknowns
        SPICEY_NOUNROLL
        for (int i = t; i < b0 && i < F.p; i += T) {
          const double *row = U + (size_t)i * lu + b0;
          double rk[SPICEY_FB];
          SPICEY_UNROLL
          for (int k = 0; k < SPICEY_FB; k++) rk[k] = row[k];
          double s = tt[i];
          SPICEY_UNROLL
          for (int k = 0; k < SPICEY_FB; k++) s = fma(-rk[k], xs[b0 + k], s);
          tt[i] = s;
        }
      });
    }
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int i = t; i < F.p; i += T) W[(size_t)P.nLU + F.k0 + i] = xs[i];
    });
  }

  // A root front (no parent, no boundary) that lives in LDS and is the LAST front of its workgroup's list stays there between
  // the sweeps: the backward sweep starts with it, from the same LDS block — no store of its U rows, no reload.
  SPICEY_HD bool stays_in_lds(const SpiceyFront &F, uint32_t s, int w) const {
    return Exec::keep_root && F.parent < 0 && F.q == 0 && s + 1 == R.fs_first[w + 1] && fits_lds(F);
  }

  // ---- the two sweeps over this workgroup's share of the front tree -------------------------------------------
  SPICEY_HD unsigned long long forward(unsigned int epoch) const {
    const int w = ex.wg();
    const unsigned long long t0 = ticks ? ex.ticks_now() : 0ull;
    // A front that is staged through the workspace takes its own entries over from W straight into the workspace — that
    // does not need the LDS, so it is done first thing in the sweep (what it reads is final since the interface phase):
    // by the time such a front's children are there, only their blocks are left to add (the 15 k gathered entries of the
    // top fronts of a 100 x 100 mesh are 24 us each, which used to sit between the last child and the first panel)
    for (uint32_t s = R.fs_first[w]; s < R.fs_first[w + 1]; s++) {
      const SpiceyFront F = P.fr[R.fs_list[s]];
      if (!fits_lds(F)) assemble_own(F, FW + F.off, F.ld);
    }
    for (uint32_t s = R.fs_first[w]; s < R.fs_first[w + 1]; s++) {
      const uint32_t f = R.fs_list[s];
      const SpiceyFront F = P.fr[f];
      if (fits_lds(F)) {
        double *A = ex.lds();
        const int lda = F.Mp + SPICEY_FRONT_LDS_PAD;
        assemble_own(F, A, lda);
        assemble_children(F, A, lda, epoch, (uint32_t *)(A + (size_t)F.Mp * lda + 272));
        stamp(f, 0, t0);
        ex.mark(SPICEY_PH_U0 + 5);
        factor_lds(F, A, lda, A + (size_t)F.Mp * lda);
        ex.mark(SPICEY_PH_U0 + 6);
        if (!stays_in_lds(F, s, w)) store_lds_front(F, A, lda);
        ex.mark(SPICEY_PH_U0 + 7);
      } else {
        assemble_children(F, FW + F.off, F.ld, epoch, (uint32_t *)ex.lds());
        stamp(f, 0, t0);
        ex.mark(SPICEY_PH_U0 + 8);
        factor_global(F);
        ex.mark(SPICEY_PH_U0 + 9);
      }
      if (F.parent >= 0 && foreign((uint32_t)F.parent)) ex.front_post(fl + f, epoch);
      stamp(f, 1, t0);
      ex.mark(SPICEY_PH_U0 + 12);
    }
    return t0;
  }
  SPICEY_HD void backward(unsigned int epoch, unsigned long long t0) const {
    const int w = ex.wg();
    for (uint32_t s = R.fs_first[w + 1]; s > R.fs_first[w]; s--) {
      const uint32_t f = R.fs_list[s - 1];
      const SpiceyFront F = P.fr[f];
      bool kept = false;
      if constexpr (Exec::keep_root) {
        kept = stays_in_lds(F, s - 1, w);
        if (kept) solve<true>(F, false, epoch, f, t0);
      }
      if (!kept) solve<false>(F, F.parent >= 0 && foreign((uint32_t)F.parent), epoch, f, t0);
      ex.mark(SPICEY_PH_U0 + 11);
      bool any = false;
      for (uint32_t ci = 0; ci < F.child_n; ci++) any = any || foreign(P.fr_child[F.child0 + ci]);
      if (any) ex.front_post(fl + P.nFronts + f, epoch);
      stamp(f, 3, t0);
    }
  }
};
