// program.h — the device-side "program": everything the symbolic phase hands to the kernels.
//
// The reference re-discovers the matrix structure every iteration (dense zero fill,
// simulateTRAN.ts:152-153; pivot search, solveReal.ts:15-34).  Here the structure is compiled
// ONCE per topology into flat index arrays; the per-timestep kernel only interprets them.
//
// Index spaces
//   W index      unified workspace: [0, nLU) = entries of L+U (row-major CSR order of the permuted
//                matrix), [nLU, nLU+n) = right-hand side / solution x' in pivot order
//   u index      source vector: [0,nC) vPrev | [nC,nC+nL) iPrev | [..,+nV) V(t) | [..,+nD) diode ieq
//   gdyn index   [0,nS) switch conductances | [nS,nS+nD) diode gd
//   gstat index  [0,nR) 1/R | [nR,nR+nC) C/dt | [..,+nL) dt/L | last = 1.0 (voltage-source +-1)
// Signs are folded into the top bit of an index (SPICEY_NEG).
#pragma once
#include <stdint.h>

#define SPICEY_NEG 0x80000000u
#define SPICEY_IDX(x) ((x)&0x7fffffffu)
#define SPICEY_EPS 1e-15
#define SPICEY_VT300 0.02585
#define SPICEY_MAX_ITER 20

// flags in the top bits of an update-task target
#define SPICEY_TGT_RECIP 0x80000000u  // diagonal becomes final in this task: store 1/pivot, check singularity
#define SPICEY_TGT_PAD 0x7fffffffu    // padding lane of a slice

// flags of a 16-bit task record (meta bits 8..15)
#define SPICEY_R16_VALID 0x80u
#define SPICEY_R16_RECIP 0x02u
#define SPICEY_R16_K 0x01u
#define SPICEY_R16_FUSED 0x04u  // a ROW record of two 16-byte halves (fus16, below)

// Dense front of the upper elimination tree (large instances, fronts_exec.h): a supernode — `p` consecutive pivots
// whose rows share one structure — with its `q` boundary unknowns, stored as a dense (Mp x ld) row-major block of the
// per-instance front workspace.  Local index i in [0, p): pivot k0 + i; [p, Pp): identity padding; Pp + j: boundary
// element j; column Mp: right-hand side.  Pp, Mp are multiples of 16 (panel / MFMA tile width), ld = Mp + 16.
struct SpiceyFront {
  int32_t k0, p, q;
  int32_t Pp, Mp, ld;
  int32_t parent;    // front id or -1
  uint32_t rel0;     // fr_rel[rel0 + j]: local index IN THE PARENT of boundary element j (extend-add map)
  uint32_t off;      // offset (doubles) of the block inside the front workspace
  uint32_t asm0, asm_n;  // fr_asm[2 (asm0 + t)] = W index, [.. + 1] = local row << 16 | local column: entries the front takes over from W
  uint32_t bnd0;     // fr_bnd[bnd0 + j]: pivot position of boundary element j
  uint32_t child0, child_n;  // fr_child[child0 + t]: child fronts in assembly order
  uint32_t pad0_, pad1_;
};

struct SpiceySlice {
  uint32_t off;  // offset (in uint32 words) of the slice's index block inside `pairs`
  uint32_t len;  // longest task of the slice (number of products)
};

// All pointers are device pointers (or host pointers inside the test emulator).
struct SpiceyProg {
  int32_t n;       // unknowns = n_nodes + nV
  int32_t nLU;     // entries of L+U
  int32_t nRestore; // entries [0, nRestore) are dynamic or update targets and are re-stamped every step; the rest never change
  int32_t nW;      // nLU + n
  int32_t nLevels; // etree height
  int32_t nR, nC, nL, nV, nS, nD;
  int32_t nU;      // nC+nL+nV+nD
  int32_t nGdyn;   // nS+nD
  int32_t nGstat;  // nR+nC+nL+1
  int32_t nOut, nCur;

  // --- prologue: static part of every L+U entry (CSR over entries, gstat index | sign)
  const uint32_t *stat_ptr;  // [nLU+1]
  const uint32_t *stat_idx;  // [...]
  const uint8_t *ent_flag;   // [nLU] bit0: leaf diagonal (no update reaches it) ; bit1: has dynamic stamps

  // --- phase B: dynamic stamps (switches, diodes) and right-hand side
  int32_t nDynEnt;
  const uint32_t *dyn_ent;  // [nDynEnt] entry id | SPICEY_TGT_RECIP if leaf diagonal
  const uint32_t *dyn_ptr;  // [nDynEnt+1]
  const uint32_t *dyn_idx;  // gdyn index | sign
  const uint32_t *rhs_ptr;  // [n+1] per pivot-order row
  const uint32_t *rhs_idx;  // [nRhsIdx] u index | sign
  const uint32_t *rhs_cof;  // [nRhsIdx] gstat index of the coefficient (C/dt for capacitors, the 1.0 slot otherwise)
  int32_t nRhsIdx;

  // --- factor (fused forward elimination): per level, wave-sized slices of update tasks
  const uint32_t *lvl_slice;    // [nLevels+1] first slice of each level (last level has none)
  const SpiceySlice *upd_slice; // [nUpdSlices]
  const uint32_t *upd_tgt;      // [nUpdSlices*64] W index | SPICEY_TGT_RECIP, or SPICEY_TGT_PAD
  const uint32_t *upd_cnt;      // [nUpdSlices*64]
  const uint32_t *upd_pairs;    // per slice: [len][3][64]  (L entry, pivot diagonal, U entry)
  int32_t nUpdSlices;

  // --- backward substitution, column-oriented (v1): per level (executed top level first) the rows below get
  //     y[r] -= (y[k] * dinv[k]) * U[r][k]; same slice format as the factor tasks; then x[i] = y[i] * W[bk_d[i]]
  const uint32_t *bk_lvl_slice; // [nLevels+1]
  const SpiceySlice *bk_slice;  // [nBkSlices]
  const uint32_t *bk_x;         // [nBkSlices*64] W index of y[r], or SPICEY_TGT_PAD
  const uint32_t *bk_d;         // [n] W index of pivot i's (reciprocal) diagonal
  const uint32_t *bk_cnt;       // [nBkSlices*64]
  const uint32_t *bk_pairs;     // per slice: [len][3][64]  (y[k], dinv[k], U[r][k])
  int32_t nBkSlices;

  // --- compact 16-bit task records (LDS path: nW < 65536 always holds there) ---------------------
  // One 16-byte record per update / backward task; phases in execution order:
  //   phase p in [0, nLevels)          = factor level p          (U task)
  //   phase p in [nLevels, 2 nLevels)  = backward level 2 nLevels-1-p  (K task)
  // w0 = tgt | meta<<16, meta = cnt (8 bits) | flags<<8 (SPICEY_R16_*).
  //   U, cnt<=2: w1 = l0|d0<<16, w2 = u0|l1<<16, w3 = d1|u1<<16 ; cnt>2: w3 = offset of cnt triplets in ovf16
  //   K        : w1 = d|u0<<16,  w2 = x0|u1<<16, w3 = x1        ; cnt>2: w3 = offset of cnt pairs in ovf16
  // Per-entry dynamic-stamp descriptor (v2 B phase): bits 0-13 idx0+1, bit 14 sign0, bits 15-28 idx1+1,
  // bit 29 sign1 (gdyn indices; 0 = none), bit 30 leaf diagonal (store reciprocal), bit 31 overflow: the
  // entry has > 2 dynamic stamps or an index > 16382 and is handled through dynx_* instead.
  const uint32_t *ent_dd;    // [nLU]
  const uint32_t *dynx_ent;  // [nDynX] entry | SPICEY_TGT_RECIP
  const uint32_t *dynx_ptr;  // [nDynX+1]
  const uint32_t *dynx_idx;  // gdyn index | sign
  int32_t nDynX;
  // Per-row right-hand-side descriptor (v2): four 16-bit fields (u index + 1) | sign << 15, 0 = none;
  // row_desc[r][1] == 0xFFFFFFFF: more than 4 contributions, row listed in rowx and summed from rhs_ptr/rhs_idx
  const uint32_t *row_desc;  // [n][2]
  const uint32_t *rowx;      // [nRowX]
  int32_t nRowX;
  // element terminals packed a | b << 16 as W indices, 0xFFFF = ground
  const uint32_t *R_ab, *C_ab, *L_ab, *D_ab;
  const uint32_t *rec16;     // [nRec16][4]
  const uint16_t *ovf16;
  const uint32_t *ph_first;  // [2 nLevels]
  const uint32_t *ph_cnt;    // [2 nLevels]
  int32_t nRec16;
  int32_t has16;             // 0 when nW >= 65536 (global-workspace path only)
  // Row records (second encoding of a factor phase, used where the phase is STREAMED).  On ladders / chains a target row i
  // of a level receives from at most two pivots k, each with at most one other neighbour o_k: its four targets
  // (a_ii, y_i, a_i,o1, a_i,o2) share the multipliers -(L_ik d_k) and most operands, yet the generic encoding runs them
  // as four tasks (22 operand reads, ~160 instructions against 14 and ~70).  fus16 holds, per factor phase p with
  // fus_pairs[p] > 0: fus_gen[p] generic 16-byte records (the tasks of rows that do not fit the pattern; the first
  // fus_rhs[p] are the right-hand-side ones) followed by fus_pairs[p] 32-byte row records, starting at 16-byte unit
  // fus_first[p].  Row record, 16 u16: [0] a_ii  [1] meta = pivots (1|2) | has_o1 << 4 | has_o2 << 5 | flags << 8
  // (VALID | FUSED | RECIP)  [2] y_i  then per pivot k: L_ik, d_k, U_ki, y_k, U_k,o, a_i,o  [15] spare.
  // Same products in the same order as the generic tasks: bit-identical results.
  const uint32_t *fus16;
  const uint32_t *fus_first, *fus_gen, *fus_rhs, *fus_pairs;  // [nLevels]
  // Tridiagonal top (16-bit records only): when the <= 64 pivots of level >= pcr_level see each other, after the levels
  // below are eliminated, only along a PATH (ladders, chains, lines), their Schur complement is tridiagonal and one wave
  // solves it by parallel cyclic reduction (log2 steps, no factor / backward levels above pcr_level: their phases hold no
  // records).  pcr_tab[i] = W indices {sub-diagonal (t_i, t_i-1), diagonal, super-diagonal (t_i, t_i+1), right-hand side}
  // of row i in path order, 0xFFFF = none.
  int32_t pcr_n, pcr_level;
  const uint16_t *pcr_tab;   // [pcr_n][4]

  // --- dense fronts (nFronts > 0): pivots of elimination-tree level >= front_cut are factored front by front
  //     (multifrontal: assemble from W + children's contribution blocks, blocked dense LU in LDS, trailing update),
  //     the levels below keep the task lists above.  bk level `front_cut` then holds the INTERFACE tasks: every row
  //     below the cut receives its products with all upper unknowns in one phase.  one_slot: W index of a constant 1.0
  //     (the "reciprocal diagonal" the interface / scaling tasks see for upper pivots, whose W[nLU + k] is x itself).
  int32_t nFronts, front_cut;
  int32_t one_slot, max_front_mp;
  int64_t front_ws;  // doubles per instance
  const SpiceyFront *fr;
  const uint32_t *fr_asm, *fr_bnd, *fr_child, *fr_rel;
  //     Levels below the cut are SUBTREE-LOCAL (nBins > 0): the elimination subtrees hanging below the cut are dealt
  //     into nBins bins of equal work; a task whose target has its smaller pivot below the cut touches only entries its
  //     own subtree writes, so the slices of level l are stored bin by bin (bin_upd[l * (nBins + 1) + b] = first slice of
  //     bin b at level l, [.. + nBins] = lvl_slice[l + 1]) and workgroup g of a group of G walks bins g, g + G, ... over
  //     ALL levels below the cut with workgroup barriers only.  Targets above the cut (front entries, upper right-hand
  //     sides) collect their products of every level below the cut, in (level, pivot) order, in the one all-workgroup
  //     phase stored as factor level `front_cut`.  bin_bk: the same for the backward levels below the cut (targets = rows
  //     of the bin's subtrees) AND for the interface tasks (backward level `front_cut`: their rows lie below the cut as
  //     well and the upper unknowns they read were published by the barrier behind the fronts): no group barrier between
  //     the interface and the levels below it either.
  int32_t nBins, pad_bins_;
  const uint32_t *bin_upd, *bin_bk;  // [front_cut][nBins + 1], [front_cut + 1][nBins + 1]

  // --- natural (reference) numbering of what the workspace holds, for the AC sweep's dense partial-pivoting fallback
  //     (ac_exec.h): entry id -> row / column of A as simulateAC.ts builds it (node - 1, branches behind the nodes);
  //     pivot position -> natural row (right-hand side W[nLU + r]) and natural column (solution W[nLU + k])
  const int32_t *ent_ro, *ent_co;   // [nLU]
  const int32_t *pos_row, *pos_col; // [n]

  // --- hybrid workspace (16-bit interpreter, circuits a little too large for the LDS of one CU): the entries that the LEAVES of
  //     the elimination tree own (level-0 pivots: their diagonals, L and U entries — about half of L+U under nested
  //     dissection) live in a global array G[nLU] indexed by entry id (L2-resident, read by ONE phase of the factorisation
  //     and one of the backward sweep: two L2 round trips per solve), so do the element vectors u / gd; the LDS holds the
  //     rest of L+U and the right-hand side / solution.  With the slot-major numbering the leaf-owned entries are the two id
  //     ranges [0, hyb_g0) (dynamic class) and [nRestore, nRestore + hyb_g2) (never-modified class); every W index in the
  //     16-bit records, element terminals, outputs and the tridiagonal-top table is then an LDS index
  //     (id - hyb_g0 below nRestore, id - hyb_g0 - hyb_g2 above), except the operand fields of factor phase 0 and of the
  //     last backward phase, which keep entry ids and are read from G.  xoff = LDS index of the right-hand side of pivot 0
  //     (nLU without the hybrid layout).
  int32_t hybrid, hyb_g0, hyb_g2, xoff;

  // --- diagnostics (SpiceyOptions.diagnostics bit 0): the structural entries of A column by column in the reference's
  //     numbering — col_ent[col_ptr[c] .. col_ptr[c + 1]) = entry id, | SPICEY_TGT_RECIP where the workspace holds the
  //     entry's reciprocal after phase B (leaf diagonals).  Fill entries (zero in A) are not listed.
  const uint32_t *col_ptr, *col_ent;  // [n + 1], [nnzA]

  // --- elements: terminal positions in W (x' slots), -1 = ground
  const int32_t *R_a, *R_b, *C_a, *C_b, *L_a, *L_b, *S_a, *S_b, *S_cp, *S_cn, *D_a, *D_b;
  const int32_t *V_x;   // [nV] W index of the branch current
  const int32_t *out_x; // [nOut] W index of each recorded node voltage
};

// Register-resident part of the program, built for one workgroup size T (spicey_build_resident):
// res[slot][T] records live in VGPRs for the whole run; every (wave, slot) chunk belongs to ONE
// phase (res_phase[wave][slot], -1 = unused) so the per-phase dispatch is wave-uniform.  Phases that
// did not fit stay streamed: st_first/st_cnt index rec16.
struct SpiceyResident {
  const uint32_t *st_rhs;  // [2L] leading right-hand-side tasks of a streamed factor phase (the only ones a reused factorisation runs)
  const uint32_t *st_fus;  // [2L] 1: this streamed factor phase runs from its row-record encoding (SpiceyProg::fus16)
  // everything a streamed phase needs to find its records, in ONE 32-byte descriptor per phase (one scalar load instead of a
  // chain of dependent ones in front of the first record fetch): {rows (0 | 1), first, count, rhs_count, rem_first, rem_count,
  // rem_rhs, 0} — first / rem_first in 16-byte units of rec16 (rows = 0) or fus16 (rows = 1: `first` = the row pairs,
  // `rem_*` = the generic remainder); *_rhs = the leading right-hand-side records a reused factorisation runs
  const uint32_t *st_desc; // [2L][8]
  const uint32_t *res;        // [RMAX][T][4]
  const int32_t *res_phase;   // [T/64][RMAX]
  const uint32_t *st_first;   // [2 nLevels]
  const uint32_t *st_cnt;     // [2 nLevels]
  int32_t rmax;
  int32_t T;
  // "tail": the run of consecutive phases [tail_first, tail_first + tail_n) at the top of the elimination tree
  // that have <= 64 tasks each.  ONE wave executes them back to back inside a single barrier phase (records in
  // LDS), which removes tail_n - 1 workgroup barriers and phase dispatches from every solve.
  int32_t tail_first;
  int32_t tail_n;
  // tridiagonal top: the first backward phase below it (<= 64 rows, each needs unknowns of the top only) is resident in the
  // slots of WAVE 0 and runs there right behind the last stage of the top, inside the same barrier phase (0: none)
  int32_t k_merge;
};

// Per-run, per-instance data (device pointers; instance-major arrays)
struct SpiceyRun {
  int32_t n_inst;
  int32_t want_currents;
  int32_t debug_empty_phases;  // diagnostics: extra empty barrier phases per solve (0 in production)
  int32_t no_reuse;            // diagnostics: refactor every step even when the matrix cannot change (linear circuits)
  int64_t steps;
  double dt;
  // parameters
  const double *R_val, *C_val, *L_val;             // [n_inst][n*]
  const double *S_ron, *S_roff, *S_von, *S_voff;   // [n_inst][nS]
  const double *D_is, *D_n;                        // [n_inst][nD]
  // state (in: entering the run, out: leaving it)
  double *C_vprev, *L_iprev, *D_vdprev;            // [n_inst][n*]
  int32_t *S_ison;                                 // [n_inst][nS]
  // per-run workspaces written by the prologue
  double *gstat;  // [n_inst][nGstat]
  double *statv;  // [n_inst][nLU]  static part of every entry (pre-inverted for static leaf diagonals)
  double *rcoef;  // [n_inst][nRhsIdx] coefficient of every right-hand-side contribution
  double *dpar;   // [n_inst][nD][2] per-diode {1/(N VT), Is/(N VT)}
  // global-memory fallback for W/u/gdyn when LDS is too small: [n_workgroups][...]
  double *gW;
  // io
  const double *src;  // [steps+1][nV]
  double *out_v;      // [n_inst][steps+1][nOut]
  double *out_i;      // [n_inst][steps+1][nCur] or null
  int32_t *iters;     // [n_inst][steps+1] or null
  // status: [n_workgroups][4] = {code, inst, step, iter}; solve counts [n_workgroups]
  int32_t *status;
  unsigned long long *solves;
  // multi-workgroup-per-instance mode (large circuits): [n_groups][SPICEY_GRP_SYNC_WORDS] = {flat counter, abort flag, timeout note [2..7],
  // stale-poll count [8], -, XCD census / arrivals / top / generation [16..]}, zeroed before every launch, and [n_groups][4] int32 flags (the WgCtx flags live in global memory there)
  unsigned int *grp_sync;
  int32_t *grp_flags;
  int32_t wgs_per_group;
  unsigned long long *prof;  // optional [n_workgroups][SPICEY_PH_SLOTS] shader-clock cycles per phase kind (diagnostics)
  // dense fronts: workspace [n_groups][front_ws], per-workgroup schedule (front ids in postorder; fs_first[G + 1]),
  // done flags [n_groups][2 * nFronts] (forward | backward; value = solve sequence number, zeroed before every launch)
  double *front_ws;
  const uint32_t *fs_first, *fs_list, *fs_owner;  // fs_owner[nFronts]: workgroup (of the group) that runs a front
  unsigned int *front_flags;
  int32_t front_right_looking; // experiments / tests: staged fronts take the right-looking sweep (a trailing update of the workspace per panel) even where the left-looking one fits
  int32_t front_lds_doubles;  // LDS scratch per workgroup (fronts that fit live there whole; tests shrink it to force the staged path)
  int32_t force_abort;        // tests: group mode raises its abort word at start-up (exercises the host's one relaunch)
  // group mode: longest single cross-workgroup wait, in ticks of the chip-wide 100 MHz counter, before the launch aborts
  unsigned long long grp_timeout_ticks;
  // profiling: per front {forward: children assembled, forward: done; backward: parent's unknowns there, backward: done},
  // 100 MHz ticks since the owner entered the forward sweep of that solve, summed over the solves [n_groups][nFronts][4]
  unsigned long long *front_ticks;
  // diagnostics (null = off; none of them feeds back into the solve)
  //   skip_risk [n_inst]: (solve, column) pairs whose stamped matrix column holds a nonzero entry below 1e-15 x the column's
  //     largest — where the reference's partial pivoting makes `|f| < EPS` (solveReal.ts:45) drop a row update this build performs
  //   lin_vd [n_inst][nD]: junction voltage every diode was linearised at for the solve in progress (simulateTRAN.ts:85);
  //   lin_err [n_inst][steps+1]: max over the diodes of |vd(x) - lin_vd| after the step's last solve, as the bit pattern of a
  //     non-negative double (combined with integer atomic max; zeroed by the host before the launch)
  unsigned long long *skip_risk;
  double *lin_vd;
  unsigned long long *lin_err;
  // hybrid workspace (SpiceyProg::hybrid): leaf-owned entries [n_inst][nLU] and the element vectors u | gd [n_inst][nU + nGdyn]
  double *hyb_G, *hyb_ug;
};
