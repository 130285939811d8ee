// fronts_exec.h — dense multifrontal factorisation of the UPPER elimination tree (device code, host-compilable).
//
// Large instances (rcd_mesh(100): 10 001 unknowns, 297 elimination-tree levels) spend their step in the long
// single-pivot chains of the top separators: one barrier-separated level per pivot.  What the reference does there
// (solveReal.ts:38-54: every row below the pivot swept across the full width) IS a dense rank-1 update, so the pivots
// of level >= P.front_cut are taken out of the level-scheduled task lists and factored front by front:
//
//   front = supernode (p consecutive pivots with nested row structure = one separator of the nested dissection) plus
//           its q boundary unknowns, a dense (Mp x ld) block of the per-instance front workspace.
//   forward (postorder):  zero -> take the front's own entries + right-hand side over from W (stamped by phase B,
//           updated in place by the task lists of the levels below the cut) -> add the children's contribution blocks
//           (extend-add through fr_rel) -> blocked right-looking LU, panels of 16 pivots: panel rows/columns staged
//           in LDS, 16x16 diagonal block by one wave in lockstep, row / column triangular solves one thread each,
//           trailing update C -= L U (v_mfma_f64_16x16x4 on the device) -> the trailing (q x q) block + rhs is the
//           contribution block for the parent.
//   backward (reverse postorder): x_B from the ancestors, t = y_P - U_PB x_B, blocked back-substitution, x_P -> W.
//
// One workgroup per front; the fronts of a group's G workgroups follow the proportional-mapping schedule of
// spicey_build_front_schedule.  Fronts of different workgroups hand over through per-front flags (agent-scope
// release / acquire, the same form as the group barrier); a workgroup's own consecutive fronts need a workgroup
// barrier only.  Every sum has a fixed order that does not depend on G: results are bit-identical for every G.
#pragma once

#define SPICEY_FB 16     // panel width = MFMA tile edge
#define SPICEY_LPLD 17   // row stride (doubles) of the L panel in LDS: odd, so that a thread-per-row walk is bank-conflict free

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

  SPICEY_HD bool foreign(uint32_t f) const { return R.fs_owner[f] != (uint32_t)ex.wg(); }

  // ---- assembly ---------------------------------------------------------------------------------------------
  SPICEY_HD void assemble(const SpiceyFront &F) const {
    double *A = FW + F.off;
    const int nel = F.Mp * F.ld;
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int i = t; i < nel; i += T) A[i] = 0.0;
    });
    ex.wg_phase([&](int t) {
      const uint32_t *as = P.fr_asm + (size_t)F.asm0 * 2;
      SPICEY_NOUNROLL
      for (uint32_t i = (uint32_t)t; i < F.asm_n; i += (uint32_t)T) A[as[2 * i + 1]] = W[as[2 * i]];
      for (int r = F.p + t; r < F.Pp; r += T) A[(size_t)r * F.ld + r] = 1.0;  // identity padding of the pivot block
    });
    for (uint32_t ci = 0; ci < F.child_n; ci++) {  // extend-add, children in a fixed order
      const SpiceyFront C = P.fr[P.fr_child[F.child0 + ci]];
      const double *Ac = FW + C.off;
      const uint32_t *rel = P.fr_rel + C.rel0;
      ex.wg_phase([&](int t) {
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int i = w; i < C.q; i += nw) {
          const double *src = Ac + (size_t)(C.Pp + i) * C.ld;
          double *dst = A + (size_t)rel[i] * F.ld;
          for (int j = lane; j <= C.q; j += 64) {
            const double v = src[j < C.q ? C.Pp + j : C.Mp];
            double *d = dst + (j < C.q ? (int)rel[j] : F.Mp);
            *d += v;
          }
        }
      });
    }
  }

  // ---- blocked partial LU of the first Pp pivots ------------------------------------------------------------
  SPICEY_HD void factor(const SpiceyFront &F) const {
    double *A = FW + F.off;
    double *lds = ex.lds();
    for (int j0 = 0; j0 < F.Pp; j0 += SPICEY_FB) {
      const int su = F.ld - j0;            // row stride of the U panel in LDS
      const int wU = F.Mp + 1 - j0;        // its used width: columns j0 .. Mp (right-hand side) inclusive
      const int nL = F.Mp - j0 - SPICEY_FB;  // rows below the diagonal block
      double *Up = lds, *Ld = Up + (size_t)SPICEY_FB * su, *Dinv = Ld + SPICEY_FB * SPICEY_FB, *Lp = Dinv + SPICEY_FB;
      // a. panel -> LDS
      ex.wg_phase([&](int t) {
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int k = w; k < SPICEY_FB; k += nw) {
          const double *src = A + (size_t)(j0 + k) * F.ld + j0;
          for (int c = lane; c < wU; c += 64) Up[(size_t)k * su + c] = src[c];
        }
        SPICEY_NOUNROLL
        for (int i = t; i < nL * SPICEY_FB; i += T) {
          const int r = i >> 4, k = i & 15;
          Lp[(size_t)r * SPICEY_LPLD + k] = A[(size_t)(j0 + SPICEY_FB + r) * F.ld + j0 + k];
        }
      });
      // b. 16 x 16 diagonal block: one wave in lockstep, lane = (row i, column residue jq); step k eliminates column k
      ex.wave_lockstep(64, SPICEY_FB, [&](int lane, int k) {
        const int i = lane >> 2, jq = lane & 3;
        const double piv = Up[(size_t)k * su + k];
        const double d = spicey_rcp(piv);
        if (lane == 0) {
          if (fabs(piv) < SPICEY_EPS && j0 + k < F.p && valid) { flags[1] = 1; flags[2] = inst; }  // solveReal.ts:28
          Dinv[k] = d;
        }
        if (i > k) {
          const double l = Up[(size_t)i * su + k] * d;
          if (jq == 0) Ld[i * SPICEY_FB + k] = l;
          for (int j = jq; j < SPICEY_FB; j += 4)
            if (j > k) Up[(size_t)i * su + j] = fma(-l, Up[(size_t)k * su + j], Up[(size_t)i * su + j]);
        }
      });
      // c. triangular solves: one thread per row of the L panel, one per column of the U panel (rhs included)
      ex.wg_phase([&](int t) {
        const int nU = wU - SPICEY_FB;
        SPICEY_NOUNROLL
        for (int it = t; it < nL + nU; it += T) {
          double v[SPICEY_FB];
          if (it < nL) {
            double *row = Lp + (size_t)it * SPICEY_LPLD;
            for (int k = 0; k < SPICEY_FB; k++) v[k] = row[k];
            for (int k = 0; k < SPICEY_FB; k++) {
              double s = v[k];
              for (int q2 = 0; q2 < k; q2++) s = fma(-v[q2], Up[(size_t)q2 * su + k], s);
              v[k] = s * Dinv[k];
            }
            for (int k = 0; k < SPICEY_FB; k++) row[k] = v[k];
          } else {
            const int c = SPICEY_FB + (it - nL);
            for (int k = 0; k < SPICEY_FB; k++) v[k] = Up[(size_t)k * su + c];
            for (int k = 1; k < SPICEY_FB; k++) {
              double s = v[k];
              for (int q2 = 0; q2 < k; q2++) s = fma(-Ld[k * SPICEY_FB + q2], v[q2], s);
              v[k] = s;
            }
            for (int k = 1; k < SPICEY_FB; k++) Up[(size_t)k * su + c] = v[k];
          }
        }
      });
      // d. U rows back to the front (reciprocal pivots on the diagonal: what the backward solve reads), trailing update
      ex.wg_phase([&](int t) {
        const int nw = T >> 6, w = t >> 6, lane = t & 63;
        for (int k = w; k < SPICEY_FB; k += nw) {
          double *dst = A + (size_t)(j0 + k) * F.ld + j0;
          for (int c = lane; c < wU; c += 64) dst[c] = c == k ? Dinv[k] : Up[(size_t)k * su + c];
        }
        trailing(A + (size_t)(j0 + SPICEY_FB) * F.ld + j0 + SPICEY_FB, F.ld, Lp, Up + SPICEY_FB, su, nL, wU - SPICEY_FB, t);
      });
    }
  }

  // C[i][j] -= sum_k Lp[i][k] Up[k][j], i < nrow, j < ncol (k ascending: the same order on host and device VALU path)
  SPICEY_HD void trailing(double *C, int ldc, const double *Lp, const double *Up, int su, int nrow, int ncol, int t) const {
    const int nw = T >> 6, w = t >> 6, lane = t & 63;
    const int nchunk = (ncol + 63) >> 6;
    for (int pr = w; pr < nrow * nchunk; pr += nw) {
      const int i = pr / nchunk, j = (pr - i * nchunk) * 64 + lane;
      if (j >= ncol) continue;
      const double *l = Lp + (size_t)i * SPICEY_LPLD;
      double acc = C[(size_t)i * ldc + j];
      for (int k = 0; k < SPICEY_FB; k++) acc = fma(-l[k], Up[(size_t)k * su + j], acc);
      C[(size_t)i * ldc + j] = acc;
    }
  }

  // ---- backward substitution of one front -----------------------------------------------------------------
  SPICEY_HD void solve(const SpiceyFront &F) const {
    const double *A = FW + F.off;
    double *lds = ex.lds();
    double *xs = lds, *tt = xs + F.Mp, *part = tt + F.Pp, *Db = part + (size_t)F.Pp * 4;  // Db: 16 x 16 diagonal block
    const uint32_t *bnd = P.fr_bnd + F.bnd0;
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int j = t; j < F.q; j += T) xs[F.Pp + j] = W[(size_t)P.nLU + bnd[j]];
      SPICEY_NOUNROLL
      for (int i = t; i < F.Pp; i += T) tt[i] = A[(size_t)i * F.ld + F.Mp];
    });
    ex.wg_phase([&](int t) {  // t = y_P - U_PB x_B: four partial sums per row, combined in a fixed order
      SPICEY_NOUNROLL
      for (int it = t; it < F.p * 4; it += T) {
        const int i = it >> 2, sg = it & 3;
        const double *row = A + (size_t)i * F.ld + F.Pp;
        double s = 0.0;
        for (int j = sg; j < F.q; j += 4) s = fma(row[j], xs[F.Pp + j], s);
        part[it] = s;
      }
    });
    const int b_last = F.Pp - SPICEY_FB;
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int i = t; i < F.p; i += T) tt[i] -= (part[4 * i] + part[4 * i + 1]) + (part[4 * i + 2] + part[4 * i + 3]);
      for (int e = t; e < SPICEY_FB * SPICEY_FB; e += T) Db[e] = A[(size_t)(b_last + (e >> 4)) * F.ld + b_last + (e & 15)];
    });
    for (int b0 = b_last; b0 >= 0; b0 -= SPICEY_FB) {
      // one wave, 16 lanes in lockstep: step s solves x of row k = 15 - s and removes it from the rows above it
      ex.wave_lockstep(SPICEY_FB, SPICEY_FB, [&](int lane, int s) {
        const int k = SPICEY_FB - 1 - s;
        if (lane > k) return;
        const double x = tt[b0 + k] * Db[k * SPICEY_FB + k];
        if (lane == k) xs[b0 + k] = x;
        else tt[b0 + lane] = fma(-Db[lane * SPICEY_FB + k], x, tt[b0 + lane]);
      });
      if (b0 == 0) break;
      ex.wg_phase([&](int t) {  // rows above the block lose its 16 solved unknowns; next diagonal block -> LDS
        SPICEY_NOUNROLL
        for (int i = t; i < b0; i += T) {
          const double *row = A + (size_t)i * F.ld + b0;
          double s = tt[i];
          for (int k = 0; k < SPICEY_FB; k++) s = fma(-row[k], xs[b0 + k], s);
          tt[i] = s;
        }
        const int nb = b0 - SPICEY_FB;  // (the lockstep that read Db has ended with a workgroup barrier)
        for (int e = t; e < SPICEY_FB * SPICEY_FB; e += T) Db[e] = A[(size_t)(nb + (e >> 4)) * F.ld + nb + (e & 15)];
      });
    }
    ex.wg_phase([&](int t) {
      SPICEY_NOUNROLL
      for (int i = t; i < F.p; i += T) W[(size_t)P.nLU + F.k0 + i] = xs[i];
    });
  }

  // ---- the two sweeps over this workgroup's share of the front tree -------------------------------------------
  SPICEY_HD void forward(unsigned int epoch) const {
    const int w = ex.wg();
    for (uint32_t s = R.fs_first[w]; s < R.fs_first[w + 1]; s++) {
      const uint32_t f = R.fs_list[s];
      const SpiceyFront F = P.fr[f];
      for (uint32_t ci = 0; ci < F.child_n; ci++) {
        const uint32_t c = P.fr_child[F.child0 + ci];
        if (foreign(c)) ex.front_wait(fl + c, epoch);
      }
      assemble(F);
      factor(F);
      if (F.parent >= 0 && foreign((uint32_t)F.parent)) ex.front_post(fl + f, epoch);
    }
  }
  SPICEY_HD void backward(unsigned int epoch) const {
    const int w = ex.wg();
    for (uint32_t s = R.fs_first[w + 1]; s > R.fs_first[w]; s--) {
      const uint32_t f = R.fs_list[s - 1];
      const SpiceyFront F = P.fr[f];
      if (F.parent >= 0 && foreign((uint32_t)F.parent)) ex.front_wait(fl + P.nFronts + F.parent, epoch);
      solve(F);
      bool any = false;
      for (uint32_t ci = 0; ci < F.child_n; ci++) any = any || foreign(P.fr_child[F.child0 + ci]);
      if (any) ex.front_post(fl + P.nFronts + f, epoch);
    }
  }
};
