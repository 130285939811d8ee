// symbolic.h — host symbolic phase: topology -> device program (see symbolic.cpp).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/spicey_hip.h"
#include "program.h"

struct HostProgram {
  SpiceyProg hdr{};  // counts filled; pointers filled by bind()
  bool structurally_singular = false;
  int32_t nnzA = 0;
  int64_t n_products = 0;  // multiply-adds of one factorisation (incl. fused forward elimination)
  int64_t n_bk_products = 0;

  std::vector<int32_t> cpos, rpos;  // original column / row -> pivot position
  std::vector<int32_t> level;       // per pivot position
  std::vector<int32_t> parent;      // etree

  std::vector<uint32_t> stat_ptr, stat_idx;
  std::vector<uint8_t> ent_flag;
  std::vector<uint32_t> dyn_ent, dyn_ptr, dyn_idx;
  std::vector<uint32_t> rhs_ptr, rhs_idx, rhs_cof;
  std::vector<uint32_t> lvl_slice, upd_tgt, upd_cnt, upd_pairs;
  std::vector<SpiceySlice> upd_slice;
  std::vector<uint32_t> bk_lvl_slice, bk_x, bk_d, bk_cnt, bk_pairs;
  std::vector<SpiceySlice> bk_slice;
  std::vector<uint32_t> rec16, ph_first, ph_cnt;  // compact records (has16)
  std::vector<uint32_t> ph_rhs;  // per factor phase: its leading right-hand-side tasks (host only)
  std::vector<uint16_t> ovf16;
  std::vector<uint32_t> fus16, fus_first, fus_gen, fus_rhs, fus_pairs;  // row-record encoding of the factor phases (program.h)
  std::vector<uint16_t> pcr_tab;  // tridiagonal top in path order (hdr.pcr_n rows of 4 W indices), see program.h
  std::vector<uint32_t> ent_dd, dynx_ent, dynx_ptr, dynx_idx, row_desc, rowx, R_ab, C_ab, L_ab, D_ab;
  std::vector<int32_t> ent_ro, ent_co, pos_row, pos_col;  // natural numbering of entries / pivot positions (program.h)
  std::vector<int32_t> R_a, R_b, C_a, C_b, L_a, L_b, S_a, S_b, S_cp, S_cn, D_a, D_b, V_x, out_x;
  // dense fronts above the cut (empty when hdr.nFronts == 0)
  std::vector<SpiceyFront> fronts;
  std::vector<uint32_t> fr_asm, fr_bnd, fr_child, fr_rel;
  std::vector<uint32_t> bin_upd, bin_bk;  // subtree-local levels below the cut (program.h)
  std::vector<uint32_t> col_ptr, col_ent;  // structural entries of A by natural column (diagnostics, program.h)
  std::vector<double> front_work;  // multiply-adds of each front's partial factorisation (for the schedule)

  // Serialise all arrays into one blob (16-byte aligned sections) and return a SpiceyProg whose
  // pointers are `base + offset`.  `base` may be a device address: the blob is then memcpy'd there.
  std::vector<uint8_t> blob;
  void pack();
  SpiceyProg bind(const void *base) const;
  std::vector<size_t> offsets;  // section offsets in pack() order
};

// Builds the program for `d`'s topology.  Returns SPICEY_OK or SPICEY_ERR_BAD_DESC (err filled).
// A structurally singular matrix is not an error here: hp.structurally_singular is set and the
// run reports SPICEY_ERR_SINGULAR, like the reference throws at the first solve.
// front_cut: elimination-tree level from which pivots are factored as dense fronts (fronts_exec.h); 0 = no fronts,
// -1 = automatic (large nonlinear circuits only).
// pcr_top: let the 16-bit records stop below a tridiagonal top that one wave solves by parallel cyclic reduction.
// hybrid: lay the program out for the hybrid workspace (program.h: leaf-owned entries in global memory) — slot-major numbering
// only, no fronts; hp.hdr.hybrid stays 0 when the circuit has no 16-bit records or fewer than 3 levels.
int32_t spicey_build_program(const SpiceyDesc *d, HostProgram &hp, std::string &err, bool bank_aware = true, int front_cut = 0, bool pcr_top = true, bool hybrid = false);

// Front schedule for G cooperating workgroups (proportional mapping of the front tree: a subtree's workgroup range is
// split among its children by work; a front runs on the first workgroup of its range once its children are done).
// first[G + 1], list[nFronts]: workgroup w executes list[first[w] .. first[w + 1]) in that order (global postorder).
void spicey_build_front_schedule(const HostProgram &hp, int G, std::vector<uint32_t> &first, std::vector<uint32_t> &list);

// Diagnostics / tests: the operand-read LDS cycles of the compact records per solve vs the conflict-free minimum
// (spicey_build_program's `bank_aware` = false keeps the plain CSR numbering of the entries).
void spicey_bank_cost(const HostProgram &hp, int64_t *cycles, int64_t *ideal);

// Resident (register) layout of the compact records for a workgroup of T threads with `rmax` slots per
// thread.  Blob sections: res[rmax][T][4] u32, res_phase[T/64][rmax] i32, st_first[2L] u32, st_cnt[2L] u32.
struct HostResident {
  std::vector<uint32_t> res, st_first, st_cnt, st_rhs, st_fus, st_desc;
  std::vector<int32_t> res_phase;
  int rmax = 0, T = 0, tail_first = 0, tail_n = 0, k_merge = 0;
  int64_t resident_tasks = 0, streamed_tasks = 0;
  std::vector<uint8_t> blob;
  std::vector<size_t> offsets;
  void pack();
  SpiceyResident bind(const void *base) const;
};
// row_records: streamed factor phases that have a row-record encoding (program.h: fus16) run from it
void spicey_build_resident(const HostProgram &hp, int T, int rmax, HostResident &out, int max_tail = 0, bool row_records = true);

// SURVEY.md §8(d) algorithmic bytes per solve.
int64_t spicey_algorithmic_bytes(const SpiceyDesc *d, int32_t nnzA, int32_t nnzLU);
