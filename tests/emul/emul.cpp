// tests/emul/emul.cpp — CPU emulation of the device program (TEST INFRASTRUCTURE ONLY).
//
// Runs the SAME symbolic phase (spicey_amd/csrc/symbolic.cpp) and the SAME phase interpreter
// (spicey_amd/csrc/tran_exec.h) as the HIP kernel, with `phase(f)` executed as a sequential loop
// over the thread ids.  It lets the CPU test-suite check ordering, scheduling and the interpreter
// numerics against the oracle without a GPU, and detects intra-phase races by running the threads
// of every phase in reverse order too.  It is NOT part of the product and is never loaded by
// spicey_amd/: libspicey_hip.so has no CPU path.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../spicey_amd/csrc/symbolic.h"
#include "../../spicey_amd/csrc/ac_exec.h"
#include "../../spicey_amd/csrc/tran_exec.h"

namespace {
struct SeqExec {
  static constexpr bool keep_root = true;  // (fronts_exec.h: a root front stays in LDS between the sweeps)
  int T;
  bool reverse;
  void *rr = nullptr;  // per-thread "registers" of the v2 interpreter: std::vector<Regs>*
  int threads() const { return T; }
  int atomic_inc(int32_t *p) { return (*p)++; }
  bool failed() const { return false; }
  // group-mode emulation: `chain` makes the backward levels run as a serial chain over the first T / 2 threads only
  // (what workgroup 0 of a two-workgroup group does on the GPU)
  bool chain = false;
  bool serial_chain() const { return chain; }
  // dense fronts: the emulator is ONE workgroup (every front in postorder, no hand-offs)
  std::vector<double> lds_buf;
  int wg() const { return 0; }
  // work the GPU splits over a group's workgroups without a barrier in between: the emulator plays `virt_wgs` workgroups
  // one after the other, each through ALL its phases (a dependence between two of them would change the result)
  unsigned long long ticks_now() const { return 0; }
  void add_ticks(unsigned long long *, unsigned long long) const {}
  int virt_wgs = 1;
  template <class F>
  void for_each_wg(F f) { for (int g = 0; g < virt_wgs; g++) f(g, virt_wgs); }
  double *lds() { return lds_buf.data(); }
  template <class F>
  void wg_phase(F f) { local_phase(f); }
  template <class F>
  void wave_lockstep(int nlanes, int nsteps, F f) {
    for (int s = 0; s < nsteps; s++) {
      if (!reverse)
        for (int t = 0; t < nlanes; t++) f(t, s);
      else
        for (int t = nlanes - 1; t >= 0; t--) f(t, s);
    }
  }
  // the same with 4 doubles per lane that survive from step to step (registers on the GPU)
  template <class F>
  void wave_lockstep_keep(int nlanes, int nsteps, F f) {
    std::vector<double> keep((size_t)nlanes * 4, 0.0);
    for (int s = 0; s < nsteps; s++) {
      if (!reverse)
        for (int t = 0; t < nlanes; t++) f(t, s, &keep[(size_t)t * 4]);
      else
        for (int t = nlanes - 1; t >= 0; t--) f(t, s, &keep[(size_t)t * 4]);
    }
  }
  void front_post(unsigned int *, unsigned int) {}
  void front_wait(unsigned int *, unsigned int) {}
  void mark(int) {}
  template <class F>
  void phase_marked(int slot, F f) { phase(slot, f); }
  template <class X>
  X fresh(const X &x) const { return x; }
  int local_threads() const { return chain ? T / 2 : T; }
  void sync() {}
  template <class F>
  void local_phase(F f) {
    const int n = local_threads();
    if (!reverse)
      for (int t = 0; t < n; t++) f(t);
    else
      for (int t = n - 1; t >= 0; t--) f(t);
  }
  template <class Regs>
  Regs &regs(int tid) {
    return (*static_cast<std::vector<Regs> *>(rr))[tid];
  }
  // one wave, `nlev` dependent levels in lockstep: every lane finishes level l before any lane starts l + 1
  template <class L, class F>
  void tail_phase(int, int nlev, L load, F f) {
    for (int l = 0; l < nlev; l++) {
      uint32_t r[4];
      if (!reverse)
        for (int t = 0; t < 64; t++) { load(t, l, r); f(t, l, r); }
      else
        for (int t = 63; t >= 0; t--) { load(t, l, r); f(t, l, r); }
    }
  }
  template <class F>
  void phase(int, F f) {
    if (!reverse)
      for (int t = 0; t < T; t++) f(t);
    else
      for (int t = T - 1; t >= 0; t--) f(t);
  }
};

template <int K>
void run_groups(const HostProgram &hp, const SpiceyProg &P, SpiceyRun &R, int T, bool reverse, int rmax, bool chain = false, bool row_records = true,
                int virt_wgs = 1) {
  const int ngroups = (R.n_inst + K - 1) / K;
  std::vector<double> W((size_t)P.nW * K), u((size_t)(P.nU + 1) * K), gd((size_t)(P.nGdyn + 1) * K);
  std::vector<int32_t> ison((size_t)(P.nS + 1) * K), flags(4);
  for (int g = 0; g < ngroups; g++) {
    WgCtx<K> c;
    c.W = W.data(); c.u = u.data(); c.gd = gd.data(); c.ison = ison.data(); c.flags = flags.data(); c.tail = nullptr;
    c.G = nullptr;
    std::vector<double> Wl;  // hybrid: the LDS part only, sized exactly (the sanitizer build then catches an index into the part that moved out)
    if (P.hybrid) {
      Wl.assign((size_t)(P.nW - P.hyb_g0 - P.hyb_g2), 0.0);
      c.W = Wl.data();
      c.G = R.hyb_G + (size_t)g * (size_t)P.nLU;
      c.u = R.hyb_ug + (size_t)g * (size_t)(P.nU + P.nGdyn);
      c.gd = c.u + P.nU;
    }
    for (int k = 0; k < K; k++) {
      int in = g * K + k;
      c.valid[k] = in < R.n_inst;
      c.inst[k] = in < R.n_inst ? in : R.n_inst - 1;
    }
    SeqExec ex{T, reverse};
    ex.chain = chain;
    ex.virt_wgs = virt_wgs;
    if (P.nFronts > 0) ex.lds_buf.assign((size_t)SPICEY_FRONT_LDS_DOUBLES, 0.0);
    if (rmax < 0) {
      spicey_tran_run<K, true>(ex, P, R, c, g);
    } else {
      HostResident hr;
      spicey_build_resident(hp, T, rmax, hr, 24, row_records);
      SpiceyResident Q = hr.bind(hr.blob.data());
      std::vector<uint32_t> tail((size_t)(hr.tail_n + 6) * 64 * 4);  // (+ the cyclic-reduction buffers of a tridiagonal top)
      c.tail = tail.data();
      // NSV = 2 resident entries per thread: small on purpose so that tests also cover the streamed remainder
      // NEL = 2 resident elements / rows per thread: with the small T the tests use, the remainder loops run too.
      // rmax <= 8 takes the static-dispatch code path (RMAX = 8), larger the indexed-register path (RMAX = 16).
      if (P.hybrid) {
        if constexpr (K == 1) {
          if (rmax == 6) {  // the 1024-thread build's shape: 6 slots, ONE element per thread, beyond-resident loops two at a time
            std::vector<ResRegs<K, 6, 2, 1>> regs(T);
            ex.rr = &regs;
            spicey_tran_run_v2<K, 6, 2, 1, true>(ex, P, Q, R, c, g);
          } else if (rmax <= 8) {
            std::vector<ResRegs<K, 8, 2, 2>> regs(T);
            ex.rr = &regs;
            spicey_tran_run_v2<K, 8, 2, 2, true>(ex, P, Q, R, c, g);
          } else {
            std::vector<ResRegs<K, 16, 2, 2>> regs(T);
            ex.rr = &regs;
            spicey_tran_run_v2<K, 16, 2, 2, true>(ex, P, Q, R, c, g);
          }
        }
      } else if (rmax <= 8) {
        std::vector<ResRegs<K, 8, 2, 2>> regs(T);
        ex.rr = &regs;
        spicey_tran_run_v2<K, 8, 2, 2>(ex, P, Q, R, c, g);
      } else {
        std::vector<ResRegs<K, 16, 2, 2>> regs(T);
        ex.rr = &regs;
        spicey_tran_run_v2<K, 16, 2, 2>(ex, P, Q, R, c, g);
      }
    }
  }
}
}  // namespace

// diagnostics outputs of the NEXT spicey_emul_run (SpiceyOptions.diagnostics of the product): skip-risk counters [n_inst]
// and per-step linearisation error [n_inst][steps + 1]; both optional, consumed by one run
static int g_hybrid = 0;  // next runs: build the program with the hybrid workspace layout (v2 interpreter only)
extern "C" void spicey_emul_set_hybrid(int32_t on) { g_hybrid = on; }
static unsigned long long *g_diag_skip = nullptr;
static double *g_diag_linerr = nullptr;
extern "C" void spicey_emul_set_diag(unsigned long long *skip_risk, double *lin_err) { g_diag_skip = skip_risk; g_diag_linerr = lin_err; }

extern "C" int32_t spicey_emul_run(const SpiceyDesc *d, int32_t K, int32_t T, int64_t steps, double dt, const double *src,
                                   double *out_v, double *out_i, int32_t *iters, double *C_vprev, double *L_iprev,
                                   double *D_vdprev, int32_t *S_ison, int32_t reverse, SpiceyInfo *info, int32_t *err4,
                                   int64_t *solves_out, int32_t rmax /* <0: v1 interpreter, else v2 with rmax resident slots */) {
  HostProgram hp;
  std::string err;
  const int front_cut = reverse >> 8;  // bits 8..: elimination-tree level from which pivots are factored as dense fronts
  // bit 4: no tridiagonal top (the 16-bit records then cover every level); interleaved instances (K > 1) never use it
  // bit 5: streamed factor phases keep the generic records even where a row-record encoding exists
  const bool want_hyb = g_hybrid && rmax >= 0 && K == 1 && front_cut == 0;
  int32_t rc = spicey_build_program(d, hp, err, true, front_cut, !(reverse & 16) && K == 1, want_hyb);
  if (rc != SPICEY_OK) return rc;
  SpiceyProg P = hp.bind(hp.blob.data());
  if (want_hyb && !P.hybrid) return SPICEY_ERR_BAD_DESC;
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_var = P.n; info->nnz_a = hp.nnzA; info->nnz_lu = P.nLU; info->n_levels = P.nLevels;
    info->threads = T; info->inst_per_wg = K; info->n_cur = P.nCur; info->n_out = P.nOut;
    info->program_bytes = (int64_t)hp.blob.size();
    info->algorithmic_bytes_solve = spicey_algorithmic_bytes(d, hp.nnzA, P.nLU);
    info->n_workgroups = (d->n_inst + K - 1) / K;
    info->pcr_rows = (rmax >= 0 && K == 1) ? P.pcr_n : 0;
    info->pcr_level = info->pcr_rows ? P.pcr_level : 0;
    info->hybrid_entries = P.hybrid ? P.hyb_g0 + P.hyb_g2 : 0;
  }
  if (hp.structurally_singular) {
    if (err4) { err4[0] = 1; err4[1] = 0; err4[2] = 0; err4[3] = 0; }
    return SPICEY_ERR_SINGULAR;
  }
  const int ni = d->n_inst;
  SpiceyRun R{};
  R.n_inst = ni; R.want_currents = out_i != nullptr; R.steps = steps; R.dt = dt;
  R.no_reuse = (reverse >> 1) & 1;  // bit 1 of `reverse`: refactor every step (to check the reuse path against)
  R.R_val = d->R_val; R.C_val = d->C_val; R.L_val = d->L_val;
  R.S_ron = d->S_ron; R.S_roff = d->S_roff; R.S_von = d->S_von; R.S_voff = d->S_voff;
  R.D_is = d->D_is; R.D_n = d->D_n;
  R.C_vprev = C_vprev; R.L_iprev = L_iprev; R.D_vdprev = D_vdprev; R.S_ison = S_ison;
  std::vector<double> gstat((size_t)ni * P.nGstat), statv((size_t)ni * P.nLU), rcoef((size_t)ni * (P.nRhsIdx + 1));
  std::vector<double> dpar((size_t)ni * (P.nD + 1) * 2);
  R.gstat = gstat.data(); R.statv = statv.data(); R.rcoef = rcoef.data(); R.dpar = dpar.data();
  R.src = src; R.out_v = out_v; R.out_i = out_i; R.iters = iters;
  const int ngroups = (ni + K - 1) / K;
  std::vector<int32_t> status((size_t)ngroups * 4);
  std::vector<unsigned long long> solves(ngroups);
  R.status = status.data(); R.solves = solves.data();
  // dense fronts: one workspace per group, the G = 1 schedule (every front on the one emulated workgroup, postorder)
  std::vector<double> front_ws;
  std::vector<uint32_t> fs_first, fs_list, fs_owner;
  std::vector<unsigned int> front_flags;
  if (P.nFronts > 0) {
    if (K != 1) return SPICEY_ERR_BAD_DESC;
    front_ws.assign((size_t)ngroups * (size_t)P.front_ws, 0.0);
    spicey_build_front_schedule(hp, 1, fs_first, fs_list);
    fs_owner.assign(P.nFronts, 0u);
    front_flags.assign((size_t)ngroups * 2 * P.nFronts, 0u);
    R.front_ws = front_ws.data(); R.fs_first = fs_first.data(); R.fs_list = fs_list.data(); R.fs_owner = fs_owner.data();
    R.front_flags = front_flags.data();
    R.front_lds_doubles = (reverse & 8) ? 6144 : SPICEY_FRONT_LDS_DOUBLES;
    R.front_right_looking = getenv("SPICEY_FRONT_RIGHT_LOOKING") != nullptr ? 1 : 0;  // bit 3: force the staged path for fronts above 64 rows
    if (info) info->tail_levels = P.nFronts;  // (diagnostic: number of fronts)
  }
  std::vector<double> hybG, hybUG;
  if (P.hybrid) {
    hybG.assign((size_t)ni * (size_t)P.nLU, 0.0);
    hybUG.assign((size_t)ni * (size_t)(P.nU + P.nGdyn + 1), 0.0);
    R.hyb_G = hybG.data(); R.hyb_ug = hybUG.data();
  }
  std::vector<double> lin_vd((size_t)ni * (P.nD + 1), 0.0);
  if (g_diag_skip) { memset(g_diag_skip, 0, sizeof(unsigned long long) * (size_t)ni); R.skip_risk = g_diag_skip; }
  if (g_diag_linerr) {
    memset(g_diag_linerr, 0, sizeof(double) * (size_t)ni * (size_t)(steps + 1));
    R.lin_err = (unsigned long long *)g_diag_linerr;
    R.lin_vd = P.nD > 0 ? lin_vd.data() : nullptr;
  }
  g_diag_skip = nullptr; g_diag_linerr = nullptr;
  if (T <= 0 || (T & 63)) return SPICEY_ERR_BAD_DESC;
  if ((reverse & 4) && T < 128) return SPICEY_ERR_BAD_DESC;  // the chain runs on T / 2 threads: at least one wave
  if (rmax > 16 || (rmax >= 0 && (!P.has16 || K > 2))) return SPICEY_ERR_BAD_DESC;  // v2 supports K <= 2
  // bits 6, 7: the subtree-local levels below a front cut are played as 3 / 7 (both: 21) workgroups, one after the other
  const int virt_wgs = ((reverse & 64) ? 3 : 1) * ((reverse & 128) ? 7 : 1);
  switch (K) {
    case 1: run_groups<1>(hp, P, R, T, (reverse & 1) != 0, rmax, (reverse & 4) != 0, !(reverse & 32), virt_wgs); break;
    case 2: run_groups<2>(hp, P, R, T, (reverse & 1) != 0, rmax, (reverse & 4) != 0, !(reverse & 32)); break;
    case 4: run_groups<4>(hp, P, R, T, (reverse & 1) != 0, rmax, (reverse & 4) != 0, !(reverse & 32)); break;
    default: return SPICEY_ERR_BAD_DESC;
  }
  int64_t tot = 0;
  for (int g = 0; g < ngroups; g++) tot += (int64_t)solves[g];
  if (solves_out) *solves_out = tot;
  for (int g = 0; g < ngroups; g++)
    if (status[(size_t)g * 4]) {
      if (err4) memcpy(err4, &status[(size_t)g * 4], 16);
      return SPICEY_ERR_SINGULAR;
    }
  return SPICEY_OK;
}

// Expose ordering details for structural tests.
extern "C" int32_t spicey_emul_symbolic(const SpiceyDesc *d, int32_t *cpos, int32_t *rpos, int32_t *level, SpiceyInfo *info,
                                        int64_t *products) {
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(d, hp, err);
  if (rc != SPICEY_OK) return rc;
  const int n = hp.hdr.n;
  if (cpos) memcpy(cpos, hp.cpos.data(), sizeof(int32_t) * n);
  if (rpos) memcpy(rpos, hp.rpos.data(), sizeof(int32_t) * n);
  if (level) memcpy(level, hp.level.data(), sizeof(int32_t) * n);
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_var = n; info->nnz_a = hp.nnzA; info->nnz_lu = hp.hdr.nLU; info->n_levels = hp.hdr.nLevels;
    info->n_cur = hp.hdr.nCur; info->n_out = hp.hdr.nOut; info->program_bytes = (int64_t)hp.blob.size();
    info->algorithmic_bytes_solve = spicey_algorithmic_bytes(d, hp.nnzA, hp.hdr.nLU);
  }
  if (products) { products[0] = hp.n_products; products[1] = hp.n_bk_products; }
  return hp.structurally_singular ? SPICEY_ERR_SINGULAR : SPICEY_OK;
}

// Resident-layout introspection for structural tests: phase of every (wave, slot), per-phase counts and tail.
extern "C" int32_t spicey_emul_resident(const SpiceyDesc *d, int32_t T, int32_t rmax, int32_t max_tail, int32_t *res_phase /*[T/64][rmax]*/,
                                        uint32_t *res_valid /*[rmax][T] 1 if the slot holds a task*/, uint32_t *ph_cnt /*[2L]*/,
                                        uint32_t *st_cnt /*[2L]*/, int32_t *meta /*[6]: nLevels, tail_first, tail_n, has16, pcr_n, pcr_level*/,
                                        int32_t pcr_top /* bit 0: tridiagonal top; bit 1: row records */) {
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(d, hp, err, true, 0, (pcr_top & 1) != 0);
  if (rc != SPICEY_OK) return rc;
  HostResident hr;
  spicey_build_resident(hp, T, rmax, hr, max_tail, (pcr_top & 2) != 0);
  const int nPh = (int)hp.ph_cnt.size();
  memcpy(res_phase, hr.res_phase.data(), sizeof(int32_t) * hr.res_phase.size());
  for (int s = 0; s < rmax; s++)
    for (int t = 0; t < T; t++) res_valid[(size_t)s * T + t] = (hr.res[((size_t)s * T + t) * 4] >> 16 & (SPICEY_R16_VALID << 8)) ? 1u : 0u;
  for (int p = 0; p < nPh; p++) { ph_cnt[p] = hp.ph_cnt[p]; st_cnt[p] = hr.st_cnt[p]; }
  meta[0] = hp.hdr.nLevels; meta[1] = hr.tail_first; meta[2] = hr.tail_n; meta[3] = hp.hdr.has16;
  meta[4] = hp.hdr.pcr_n; meta[5] = hp.hdr.pcr_level;
  return SPICEY_OK;
}

// Row records per factor level (program.h: fus16), for structural tests.
extern "C" int32_t spicey_emul_row_records(const SpiceyDesc *d, uint32_t *pairs /*[cap]*/, int32_t cap) {
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(d, hp, err);
  if (rc != SPICEY_OK) return -rc;
  const int nL = (int)hp.fus_pairs.size();
  for (int l = 0; l < nL && l < cap; l++) pairs[l] = hp.fus_pairs[l];
  return nL;
}

// AC sweep through the same phase code as the HIP kernel (spicey_amd/csrc/ac_exec.h), one (instance, frequency) at a time.
extern "C" int32_t spicey_emul_ac(const SpiceyDesc *d, int32_t T, int64_t n_freq, const double *freqs, const double *vph, double *out_v,
                                  double *out_i, int32_t reverse, SpiceyInfo *info) {
  SpiceyDesc dd = *d;
  dd.nS = 0;
  dd.nD = 0;
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(&dd, hp, err, true, 0, false);
  if (rc != SPICEY_OK) return rc;
  SpiceyProg P = hp.bind(hp.blob.data());
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_var = P.n; info->nnz_a = hp.nnzA; info->nnz_lu = P.nLU; info->n_levels = P.nLevels; info->threads = T;
    info->n_cur = P.nR + P.nC + P.nL + P.nV; info->n_out = P.nOut;
  }
  if (hp.structurally_singular) return SPICEY_ERR_SINGULAR;
  if (T <= 0 || (T & 63)) return SPICEY_ERR_BAD_DESC;
  const size_t slots = (size_t)d->n_inst * (size_t)n_freq;
  std::vector<int32_t> status(slots + 1);
  SpiceyAcRun R{};
  std::vector<double> rinv((size_t)d->n_inst * (size_t)d->nR);
  for (size_t i2 = 0; i2 < rinv.size(); i2++) rinv[i2] = 1.0 / d->R_val[i2];
  R.R_inv = rinv.data(); R.C_val = d->C_val; R.L_val = d->L_val;
  R.freqs = freqs; R.vph = vph; R.out_v = out_v; R.out_i = out_i; R.gW = nullptr; R.status = status.data();
  R.n_freq = n_freq; R.n_inst = d->n_inst;
  std::vector<SpiceyCx> W((size_t)P.nW + 1);
  int32_t flags[2] = {0, 0};
  if (reverse & 2) {
    // resident sweep (bit 1): one emulated workgroup per (instance, residue class of 3 frequencies); small register
    // capacities (8 record slots, 2 entries per thread) so that the streamed / beyond-capacity paths run too
    if (!P.has16) return SPICEY_ERR_BAD_DESC;
    HostResident hr;
    spicey_build_resident(hp, T, 8, hr, 0, false);
    SpiceyResident Q = hr.bind(hr.blob.data());
    for (int in = 0; in < d->n_inst; in++)
      for (int64_t f0 = 0; f0 < 3 && f0 < n_freq; f0++) {
        SeqExec ex{T, (reverse & 1) != 0};
        std::vector<AcResRegs<8, 2>> regs(T);
        ex.rr = &regs;
        spicey_ac_sweep_resident<8, 2>(ex, P, Q, R, W.data(), flags, (size_t)in, f0, 3);
      }
  } else
  for (size_t s = 0; s < slots; s++) {
    SeqExec ex{T, (reverse & 1) != 0};
    spicey_ac_solve(ex, P, R, W.data(), flags, (int64_t)s);
  }
  // solves that tripped a pivot guard are repeated with partial pivoting (bit 2 of `reverse`: off)
  if (!(reverse & 4)) {
    std::vector<SpiceyCx> A((size_t)P.n * ((size_t)P.n + 1));
    std::vector<double> sd((size_t)T + 2 * (size_t)P.n + 2);
    std::vector<int32_t> si((size_t)T + (size_t)P.n + 4);
    for (size_t s = 0; s < slots; s++)
      if (status[s]) {
        SeqExec ex{T, (reverse & 1) != 0};
        spicey_ac_dense_solve(ex, P, R, W.data(), A.data(), sd.data(), si.data(), flags, (int64_t)s);
        if (info) info->tail_levels++;  // (diagnostic: number of dense fallback solves)
      }
  }
  for (size_t s = 0; s < slots; s++)
    if (status[s]) return status[s] == 1 ? SPICEY_ERR_SINGULAR : SPICEY_ERR_COMPLEX_DIV;
  return SPICEY_OK;
}

// LDS bank cost of the compiled records with and without the bank-aware numbering pass (symbolic.cpp).
extern "C" int32_t spicey_emul_bank_cost(const SpiceyDesc *d, int64_t *out4) {
  for (int pass = 0; pass < 2; pass++) {
    HostProgram hp;
    std::string err;
    int32_t rc = spicey_build_program(d, hp, err, pass == 1);
    if (rc != SPICEY_OK) return rc;
    spicey_bank_cost(hp, &out4[pass * 2], &out4[pass * 2 + 1]);
  }
  return SPICEY_OK;
}

// per-level slice counts of the factor / backward lists and pivots per level (diagnostics for tools/)
extern "C" int32_t spicey_emul_level_stats(const SpiceyDesc *d, int32_t cap, int32_t *n_levels, int32_t *upd_slices, int32_t *bk_slices, int32_t *pivots) {
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(d, hp, err);
  if (rc != SPICEY_OK) return rc;
  *n_levels = hp.hdr.nLevels;
  for (int l = 0; l < hp.hdr.nLevels && l < cap; l++) {
    upd_slices[l] = (int32_t)(hp.lvl_slice[l + 1] - hp.lvl_slice[l]);
    bk_slices[l] = (int32_t)(hp.bk_lvl_slice[l + 1] - hp.bk_lvl_slice[l]);
    pivots[l] = 0;
  }
  for (int k = 0; k < hp.hdr.n; k++)
    if (hp.level[k] < cap) pivots[hp.level[k]]++;
  return SPICEY_OK;
}

// Subtree-local levels below a front cut (program.h, nBins): {bins, cut, factor slices below the cut, slices of the phase
// that updates the targets above the cut, backward slices below the cut, widest bin's slices over all factor levels}
extern "C" int32_t spicey_emul_bin_stats(const SpiceyDesc *d, int32_t front_cut, int32_t *out6) {
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(d, hp, err, true, front_cut);
  if (rc != SPICEY_OK) return rc;
  const int Lc = hp.hdr.front_cut, nb = hp.hdr.nBins;
  out6[0] = nb; out6[1] = Lc;
  out6[2] = Lc > 0 ? (int32_t)hp.lvl_slice[Lc] : 0;
  out6[3] = Lc > 0 ? (int32_t)(hp.lvl_slice[Lc + 1] - hp.lvl_slice[Lc]) : 0;
  out6[4] = Lc > 0 ? (int32_t)hp.bk_lvl_slice[Lc] : 0;
  int widest = 0;
  for (int b = 0; b < nb; b++) {
    int sum = 0;
    for (int l = 0; l < Lc; l++) sum += (int)(hp.bin_upd[(size_t)l * (nb + 1) + b + 1] - hp.bin_upd[(size_t)l * (nb + 1) + b]);
    widest = std::max(widest, sum);
  }
  out6[5] = widest;
  if (getenv("SPICEY_BIN_HIST") && Lc > 0) {
    for (int pass = 0; pass < 2; pass++) {
      const std::vector<uint32_t> &ls = pass ? hp.bk_lvl_slice : hp.lvl_slice;
      const std::vector<SpiceySlice> &sl = pass ? hp.bk_slice : hp.upd_slice;
      const std::vector<uint32_t> &cn = pass ? hp.bk_cnt : hp.upd_cnt;
      long tot = 0, padded = 0; int hist[12] = {0};
      for (uint32_t s = ls[Lc]; s < ls[Lc + 1]; s++) {
        padded += (long)sl[s].len * 64;
        for (int l = 0; l < 64; l++) { uint32_t c = cn[(size_t)s * 64 + l]; tot += c; int b = 0; while ((1u << b) < c && b < 11) b++; hist[b]++; }
      }
      fprintf(stderr, "%s interface: %u slices, %ld products (%ld padded), longest %u; log2 histogram:", pass ? "backward" : "factor", ls[Lc + 1] - ls[Lc], tot, padded, ls[Lc + 1] > ls[Lc] ? sl[ls[Lc]].len : 0);
      for (int b = 0; b < 12; b++) fprintf(stderr, " %d", hist[b]);
      fprintf(stderr, "\n");
    }
  }
  return SPICEY_OK;
}

// Front tree introspection (tools/front_stats.py, structural tests): per front k0, p, q, parent, owner under a G-workgroup schedule.
extern "C" int32_t spicey_emul_front_stats(const SpiceyDesc *d, int32_t front_cut, int32_t G, int32_t cap, int32_t *meta /*[4]: nFronts, cut, maxMp, nLevels*/,
                                           int64_t *ws_doubles, int32_t *k0, int32_t *p, int32_t *q, int32_t *parent, int32_t *owner, int32_t *seq) {
  HostProgram hp;
  std::string err;
  int32_t rc = spicey_build_program(d, hp, err, true, front_cut);
  if (rc != SPICEY_OK) return rc;
  meta[0] = hp.hdr.nFronts; meta[1] = hp.hdr.front_cut; meta[2] = hp.hdr.max_front_mp; meta[3] = hp.hdr.nLevels;
  *ws_doubles = hp.hdr.front_ws;
  std::vector<uint32_t> first, list;
  spicey_build_front_schedule(hp, G, first, list);
  for (int w = 0; w < G; w++)
    for (uint32_t s = first[w]; s < first[w + 1]; s++)
      if ((int)list[s] < cap) { owner[list[s]] = w; seq[list[s]] = (int32_t)(s - first[w]); }
  for (int f = 0; f < hp.hdr.nFronts && f < cap; f++) {
    k0[f] = hp.fronts[f].k0; p[f] = hp.fronts[f].p; q[f] = hp.fronts[f].q; parent[f] = hp.fronts[f].parent;
  }
  return SPICEY_OK;
}
