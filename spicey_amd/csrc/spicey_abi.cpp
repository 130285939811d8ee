// spicey_abi.cpp — the C-ABI of include/spicey_hip.h: handle management, uploads, launches.
// Host code only (HIP runtime API); the kernels are in kernels.hip, the symbolic phase in
// symbolic.cpp.  There is NO CPU solve path in this library: without a HIP device every entry
// point that would compute returns SPICEY_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spicey_hip.h"
#include "kernels.h"
#include "symbolic.h"
#include "fronts_exec_consts.h"

struct SpiceyHandle {
  HostProgram hp;
  HostResident hres;
  SpiceyProg dprog{};
  SpiceyResident dres{};
  int interp = 1;
  bool packed = false;  // two 512-thread workgroups per CU
  int G = 1;            // workgroups per instance group (group mode: global workspace only)
  unsigned int *d_gsync = nullptr;
  int32_t *d_gflags = nullptr;
  // dense fronts: workspace [grid][front_ws], schedule of the G workgroups, done flags [grid][2 nFronts]
  double *d_front_ws = nullptr;
  uint32_t *d_fs = nullptr;  // first[G + 1] | list[nFronts] | owner[nFronts]
  unsigned int *d_front_flags = nullptr;
  void *d_res = nullptr;
  // v2 kernels take their argument structs from device memory (scalar loads per phase instead of ~110 pointers in SGPRs)
  SpiceyProg *d_Pstruct = nullptr;
  SpiceyResident *d_Qstruct = nullptr;
  SpiceyRun *d_Rstruct = nullptr;
  SpiceyRun run_args{};  // host copy of the last launch's SpiceyRun (source of the asynchronous upload)
  SpiceyOptions opt{};
  int n_inst = 0, n_nodes = 0;
  int K = 1, T = 256, grid = 1;
  bool lds = true;
  size_t lds_bytes = 0;
  int device = 0;
  int64_t algo_bytes = 0;
  // device memory
  void *d_blob = nullptr;
  double *d_R = nullptr, *d_C = nullptr, *d_L = nullptr, *d_Sron = nullptr, *d_Sroff = nullptr, *d_Svon = nullptr, *d_Svoff = nullptr,
         *d_Dis = nullptr, *d_Dn = nullptr;
  double *d_Cv = nullptr, *d_Li = nullptr, *d_Dv = nullptr;
  int32_t *d_Son = nullptr;
  double *d_Cv0 = nullptr, *d_Li0 = nullptr, *d_Dv0 = nullptr;  // the descriptor's state, for spicey_reset_state
  // group mode: the state as it entered the launch in flight (a launch that ends in the bounded-spin abort is repeated once)
  double *d_Cv_s = nullptr, *d_Li_s = nullptr, *d_Dv_s = nullptr;
  int32_t *d_Son_s = nullptr;
  SpiceyRun grp_R{};      // that launch's arguments
  int group_retries = 0;  // launches repeated so far (spicey_group_retries)
  // diagnostics (SpiceyOptions.diagnostics): skip-risk counters [n_inst], linearisation points [n_inst][nD], per-step
  // linearisation error [n_inst][steps + 1] of the last run (grown on demand)
  double *d_hybG = nullptr, *d_hybUG = nullptr;  // hybrid workspace: leaf-owned entries [n_inst][nLU], u | gd [n_inst][nU + nGdyn]
  unsigned long long *d_skip = nullptr, *d_linerr = nullptr;
  double *d_linvd = nullptr;
  size_t linerr_cap = 0;
  int64_t last_steps = -1;
  int64_t stale_polls = 0;  // group mode: waits that only the read-modify-write poll saw satisfied (spicey_group_stale_polls)
  bool gated = false;     // this handle is counted in its device's group-mode handles (DeviceGate)
  int32_t *d_Son0 = nullptr;
  double *d_gstat = nullptr, *d_statv = nullptr, *d_rcoef = nullptr, *d_gW = nullptr, *d_dpar = nullptr;
  int32_t *d_status = nullptr;
  unsigned long long *d_solves = nullptr;
  unsigned long long *d_prof = nullptr;
  hipStream_t stream = nullptr;  // owned stream for spicey_run
  hipStream_t last_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool pending = false;
  int64_t last_solves = 0;
  double last_ms = 0.0;
  std::string err;
};

static thread_local std::string g_err;  // message of the calling thread's last failed spicey_create (no handle to hang it on)

// roctx ranges around the host-side phases (SURVEY §5 tracing hook): `rocprofv3 --marker-trace` shows symbolic phase, uploads,
// kernel and result copies as named ranges.  The marker library is looked up at run time (no link dependency: without it,
// or outside a profiler, the ranges cost one null check).
namespace {
struct Roctx {
  typedef int (*push_t)(const char *);
  typedef int (*pop_t)();
  static push_t push_fn() {
    static push_t f = []() -> push_t {
      for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"})
        if (void *hnd = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL))
          if (void *sym = dlsym(hnd, "roctxRangePushA")) { pop_fn_ref() = (pop_t)dlsym(hnd, "roctxRangePop"); return (push_t)sym; }
      return nullptr;
    }();
    return f;
  }
  static pop_t &pop_fn_ref() { static pop_t p = nullptr; return p; }
  bool on;
  explicit Roctx(const char *name) : on(false) {
    if (push_t f = push_fn()) { f(name); on = pop_fn_ref() != nullptr; }
  }
  ~Roctx() { if (on) pop_fn_ref()(); }
};
}  // namespace

// ---- launch admission per device (include/spicey_hip.h, "Launch admission") ------------------------------------------
// A group-mode launch (G > 1 workgroups per instance that wait for one another inside the kernel) needs all its workgroups
// resident.  The host sizes it to at most one workgroup per CU, so that holds on an otherwise idle device; what must not
// happen is a second launch of this library taking CUs away while the group is starting, or two groups each holding some
// CUs and waiting for the rest.  Per device: `ev_group` = completion of the last group-mode launch; `ring` = completions of
// the recent launches that are not group-mode.  A group-mode launch first makes its stream wait for ev_group and for every
// ring event; any other launch waits for ev_group only (and for the ring slot it is about to reuse, which keeps "waiting
// for the ring" = "waiting for every earlier launch" when more than RING launches are in flight).  All of it is
// stream-ordered (hipStreamWaitEvent): nothing blocks on the host, so one thread may hold several launches in flight.
// The bookkeeping is skipped while no group-mode handle exists on the device; the first one to be created drains the
// device once (hipDeviceSynchronize) so that earlier, unrecorded launches are known to have finished.
namespace {
struct DeviceGate {
  static const int RING = 16;
  std::mutex mu;
  int group_handles = 0;
  hipEvent_t ev_group = nullptr;
  bool group_recorded = false;
  hipEvent_t ring[RING] = {};
  bool ring_used[RING] = {};
  int ring_next = 0;
};
DeviceGate &device_gate(int device) {
  static std::mutex m;
  static std::map<int, DeviceGate *> gates;  // (never freed: a gate may be touched by a handle destroyed at process exit)
  std::lock_guard<std::mutex> lk(m);
  DeviceGate *&g = gates[device];
  if (!g) g = new DeviceGate();
  return *g;
}
// with g.mu held and the device current; `st` is the launch stream.  Returns a HIP error or hipSuccess.
hipError_t gate_before_launch(DeviceGate &g, bool group, hipStream_t st, int *slot) {
  *slot = -1;
  if (!group && g.group_handles == 0) return hipSuccess;
  hipError_t e;
  if (g.group_recorded && (e = hipStreamWaitEvent(st, g.ev_group, 0)) != hipSuccess) return e;
  if (group) {
    for (int i = 0; i < DeviceGate::RING; i++)
      if (g.ring_used[i] && (e = hipStreamWaitEvent(st, g.ring[i], 0)) != hipSuccess) return e;
    return hipSuccess;
  }
  const int i = g.ring_next;
  if (!g.ring[i] && (e = hipEventCreateWithFlags(&g.ring[i], hipEventDisableTiming)) != hipSuccess) return e;
  if (g.ring_used[i] && (e = hipStreamWaitEvent(st, g.ring[i], 0)) != hipSuccess) return e;
  *slot = i;
  return hipSuccess;
}
hipError_t gate_after_launch(DeviceGate &g, bool group, hipStream_t st, int slot) {
  hipError_t e;
  if (group) {
    if (!g.ev_group && (e = hipEventCreateWithFlags(&g.ev_group, hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(g.ev_group, st)) != hipSuccess) return e;
    g.group_recorded = true;
    for (int i = 0; i < DeviceGate::RING; i++) g.ring_used[i] = false;  // (this launch waited for all of them)
    return hipSuccess;
  }
  if (slot < 0) return hipSuccess;
  if ((e = hipEventRecord(g.ring[slot], st)) != hipSuccess) return e;
  g.ring_used[slot] = true;
  g.ring_next = (slot + 1) % DeviceGate::RING;
  return hipSuccess;
}
}  // namespace

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) {                                                            \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
      return SPICEY_ERR_HIP;                                                            \
    }                                                                                   \
  } while (0)

template <class T>
static int32_t upload(SpiceyHandle *h, T **dst, const T *src, size_t count) {
  *dst = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
  HIPCHK(h, hipMalloc((void **)dst, bytes));
  if (count && src) HIPCHK(h, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  else HIPCHK(h, hipMemset(*dst, 0, bytes));
  return SPICEY_OK;
}

extern "C" const char *spicey_version(void) { return "spicey_hip abi2 gfx950 (persistent LDS-resident sparse-LU transient kernel)"; }

extern "C" const char *spicey_last_error(SpiceyHandle *h) { return h ? h->err.c_str() : g_err.c_str(); }

extern "C" void spicey_destroy(SpiceyHandle *h) {
  if (!h) return;
  if (h->pending && h->last_stream) (void)hipStreamSynchronize(h->last_stream);
  if (h->gated) {
    DeviceGate &g = device_gate(h->device);
    std::lock_guard<std::mutex> lk(g.mu);
    g.group_handles--;
  }
  void *ptrs[] = {h->d_res, h->d_blob, h->d_R, h->d_C, h->d_L, h->d_Sron, h->d_Sroff, h->d_Svon, h->d_Svoff, h->d_Dis, h->d_Dn, h->d_Cv,
                  h->d_Li, h->d_Dv, h->d_Son, h->d_Cv0, h->d_Li0, h->d_Dv0, h->d_Son0, h->d_Cv_s, h->d_Li_s, h->d_Dv_s, h->d_Son_s, h->d_gstat, h->d_statv, h->d_rcoef, h->d_gW, h->d_dpar, h->d_Pstruct, h->d_Qstruct, h->d_Rstruct, h->d_gsync, h->d_gflags, h->d_front_ws, h->d_fs, h->d_front_flags, h->d_status, h->d_solves, h->d_prof, h->d_skip, h->d_linerr, h->d_linvd, h->d_hybG, h->d_hybUG};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int pick_threads(const HostProgram &hp, bool v2, int K) {
  const SpiceyProg &P = hp.hdr;
  if (v2) {
    // smallest workgroup in which the whole program is register-resident: all factor/backward tasks in the
    // RMAX slots, one right-hand-side row, one element of each kind and NSV re-stamped entries per thread
    int64_t chunks = 0;  // 64-lane chunks of task records
    for (uint32_t c : hp.ph_cnt) chunks += (c + 63) / 64;
    const int widest = std::max(std::max(P.n, P.nOut), std::max(std::max(P.nR, P.nC), P.nD));
    // measured on diode_chain(1000): per-step time T=1024 < T=512 < T=256 (more waves hide the issue-bound
    // phases B/Z); small circuits take the smallest workgroup that holds everything
    const int tmax = spicey_v2_max_threads(K);
    for (int T = 64; T <= tmax; T *= 2) {
      const int rmax = spicey_v2_rmax(T), nsv = spicey_v2_nsv(T), nel = spicey_v2_nel(T);
      const bool fits = chunks <= (int64_t)rmax * (T / 64) && widest <= nel * T && P.nRestore <= nsv * T;
      if (fits && (T >= tmax || widest <= T)) return T;  // prefer one element per thread when a larger T offers it
    }
    return tmax;
  }
  const int n = P.n;
  if (n <= 48) return 64;
  if (n <= 160) return 128;
  if (n <= 400) return 256;
  if (n <= 4000) return 512;
  return 1024;
}

extern "C" int32_t spicey_create(const SpiceyDesc *desc, const SpiceyOptions *opt, SpiceyHandle **out) {
  if (!out) { g_err = "null out pointer"; return SPICEY_ERR_BAD_DESC; }
  *out = nullptr;
  SpiceyHandle *h = new SpiceyHandle();
  if (opt) h->opt = *opt;
  std::string err;
  Roctx range_create("spicey_create");
  // dense fronts: explicit level, or automatic for large nonlinear circuits that run one instance per workgroup (the
  // interleaved K > 1 layouts and forced interpreter 2 keep the task lists); -1 = never
  int front_cut = h->opt.front_cut > 0 ? h->opt.front_cut : (h->opt.front_cut == 0 ? -1 : 0);
  if (h->opt.inst_per_wg > 1 || h->opt.interpreter == 2) front_cut = 0;
  if (front_cut < 0 && desc && desc->n_inst >= 512) front_cut = 0;  // big batches fill the chip with interleaved instances instead
  // tridiagonal top by cyclic reduction (16-bit records, one instance per workgroup); diagnostics: bit 5 = never
  const bool pcr_top = !((h->opt.debug >> 5) & 1) && h->opt.inst_per_wg <= 1;
  int32_t rc = spicey_build_program(desc, h->hp, err, !((h->opt.debug >> 2) & 1), front_cut, pcr_top);  // diagnostics: bit 2 = plain CSR numbering
  if (rc != SPICEY_OK) {
    g_err = err;
    delete h;
    return rc;
  }
  h->n_inst = desc->n_inst;
  h->n_nodes = desc->n_nodes;
  h->algo_bytes = spicey_algorithmic_bytes(desc, h->hp.nnzA, h->hp.hdr.nLU);

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_err = "no HIP device: libspicey_hip has no CPU path";
    delete h;
    return SPICEY_ERR_NO_DEVICE;
  }
  h->device = h->opt.device;
  if (h->device < 0 || h->device >= ndev) {
    g_err = "device ordinal out of range";
    delete h;
    return SPICEY_ERR_BAD_DESC;
  }
  auto fail = [&](int32_t code) {
    g_err = h->err;
    spicey_destroy(h);
    return code;
  };
  if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(SPICEY_ERR_HIP); }
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);

  // ---- geometry: instances per workgroup, threads, LDS or global workspace --------------------
  const SpiceyProg &P = h->hp.hdr;
  int K = h->opt.inst_per_wg;
  const bool want_lds = !h->opt.force_global;
  if (K != 0 && K != 1 && K != 2 && K != 4) { h->err = "inst_per_wg must be 0, 1, 2 or 4"; return fail(SPICEY_ERR_BAD_DESC); }
  // diagnostics are compiled into the kernels with at most two interleaved instances and not into the two-workgroups-per-CU
  // geometry (tran_exec.h, DIAG): a handle with the option stays out of both
  const bool diag = h->opt.diagnostics != 0;
  if (diag && (K == 4 || h->opt.geometry == 2)) { h->err = "diagnostics need inst_per_wg <= 2 and geometry != 2"; return fail(SPICEY_ERR_BAD_DESC); }
  if (P.nS > 0) K = 1;  // the switch iteration count is per instance: no interleaving
  if (P.nFronts > 0) K = 1;  // dense fronts: one instance per workgroup (group)
  if (K == 0) {
    // LDS path: one instance per workgroup (measured faster than two interleaved ones: VGPR pressure in phase Z).
    // Global-workspace path (large circuits): once the batch exceeds the CUs, interleaving 2-4 instances shares
    // the index stream and fills more of every gathered cache line (rcd_mesh(50) x 1024: ~3x with K = 4).
    K = 1;
    if (!want_lds || spicey_lds_bytes(P, 1, true) > SPICEY_LDS_MAX) K = (h->n_inst >= 4 * ncu && !diag) ? 4 : (h->n_inst >= 2 * ncu ? 2 : 1);
  }
  if (K > h->n_inst) K = 1;
  // Hybrid workspace (program.h, SpiceyProg::hybrid): a circuit whose L+U no longer fits the LDS of one CU but whose upper
  // elimination tree does keeps the 16-bit register-resident interpreter — the entries the LEAVES own (half of L+U under
  // nested dissection) and the element vectors move to global memory, read by one factor phase and one backward phase.
  // Without it such a circuit falls to the 32-bit task lists on a global workspace (diode_chain(2600): 56 us per step on 16
  // cooperating workgroups against ~16 for the 2000-node chain that still fits).  One instance per workgroup; 1024 threads
  // (SpiceyOptions.threads = 512 selects the 512-thread build of the same kernel).
  if (want_lds && (h->opt.inst_per_wg == 0 || h->opt.inst_per_wg == 1) && h->opt.interpreter != 1 && h->opt.geometry != 2 && P.has16 && P.nFronts == 0 &&
      (h->opt.threads == 0 || h->opt.threads == 512 || h->opt.threads == 1024) && h->opt.wgs_per_inst <= 1 && !diag && !getenv("SPICEY_NO_HYBRID") &&
      spicey_lds_bytes(P, 1, true, 5) > SPICEY_LDS_MAX) {
    HostProgram hyb;
    std::string err2;
    if (spicey_build_program(desc, hyb, err2, true, 0, pcr_top, true) == SPICEY_OK && hyb.hdr.hybrid && !hyb.ph_cnt.empty() && hyb.ph_cnt[0] > 64 &&
        spicey_lds_bytes(hyb.hdr, 1, true, 5) <= SPICEY_LDS_MAX) {
      h->hp = std::move(hyb);
      K = 1;
    }
  }
  h->lds = want_lds && spicey_lds_bytes(P, K, true) <= SPICEY_LDS_MAX;
  if (!h->lds && want_lds && K > 1 && spicey_lds_bytes(P, 1, true) <= SPICEY_LDS_MAX) {
    K = 1;  // one instance fits LDS where K interleaved ones do not: LDS wins
    h->lds = true;
  }
  h->K = K;
  // interpreter: v2 needs the LDS workspace, 16-bit records and one instance per workgroup (the K = 2 build of the
  // register-resident kernel spilled vector registers whatever its geometry: kernels.hip)
  const bool v2_ok = h->lds && P.has16 && K == 1;
  if (h->opt.interpreter == 2 && !v2_ok) { h->err = "interpreter 2 needs the LDS workspace, < 65536 workspace entries and inst_per_wg = 1"; return fail(SPICEY_ERR_BAD_DESC); }
  h->interp = (h->opt.interpreter == 1 || !v2_ok) ? 1 : 2;
  h->T = h->opt.threads > 0 ? h->opt.threads : pick_threads(h->hp, h->interp == 2, K);
  if (P.hybrid) {
    if (h->interp != 2) { h->err = "internal: hybrid layout without the 16-bit interpreter"; return fail(SPICEY_ERR_BAD_DESC); }
    h->T = h->opt.threads == 512 ? 512 : 1024;  // (the two geometries the hybrid kernel is built for: kernels.hip, spicey_launch_tran_v2)
  }
  if (P.nFronts > 0 && h->T > 512) {  // kernels with the dense-front code are built for <= 512 threads (256 VGPRs)
    if (h->opt.threads > 512) { h->err = "front_cut needs threads <= 512"; return fail(SPICEY_ERR_BAD_DESC); }
    h->T = 512;
  }
  if (h->T > 1024 || (h->T & 63) || h->T < 64) { h->err = "threads must be a multiple of 64 in [64, 1024]"; return fail(SPICEY_ERR_BAD_DESC); }
  h->grid = (h->n_inst + K - 1) / K;
  h->lds_bytes = spicey_lds_bytes(P, K, h->lds);
  if (h->interp == 2) {
    // geometry: "throughput" packs two 512-thread workgroups on a CU (needs K = 1, half the LDS, and the per-thread
    // resident items of a 512-thread workgroup); chosen automatically once the batch can fill every CU twice
    const size_t base = spicey_lds_bytes(P, K, true, 0);
    const int widest = std::max(std::max(P.n, P.nOut), std::max(std::max(P.nR, P.nC), P.nD));
    const bool packable = K == 1 && base <= SPICEY_LDS_MAX / 2 && widest <= spicey_v2_nel(512, true) * 512 &&
                          P.nRestore <= spicey_v2_nsv(512, true) * 512 && (h->opt.threads == 0 || h->opt.threads == 512);
    if (h->opt.geometry == 2 && !packable) { h->err = "geometry 2 needs inst_per_wg = 1, <= 80 KB of LDS per instance and <= 1024 unknowns"; return fail(SPICEY_ERR_BAD_DESC); }
    if (h->opt.geometry < 0 || h->opt.geometry > 2) { h->err = "geometry must be 0, 1 or 2"; return fail(SPICEY_ERR_BAD_DESC); }
    h->packed = packable && !diag && !P.hybrid && (h->opt.geometry == 2 || (h->opt.geometry == 0 && h->n_inst >= 2 * ncu && h->opt.threads == 0));
    if (h->packed) { h->T = 512; h->grid = (h->n_inst + K - 1) / K; }
    // tail levels go to LDS: as many as fit beside the workspace (1 KB each), at most 24; the packed geometry
    // must leave room for a second workgroup on the CU
    const size_t lds_cap = h->packed ? SPICEY_LDS_MAX / 2 : SPICEY_LDS_MAX;
    int max_tail = (int)std::min<size_t>(24, base < lds_cap ? (lds_cap - base) / 1024 : 0);
    if (h->opt.debug & 1) max_tail = 0;  // diagnostics: disable the tail merge
    spicey_build_resident(h->hp, h->T, spicey_v2_rmax(h->T, h->packed, P.hybrid != 0), h->hres, max_tail, !((h->opt.debug >> 6) & 1));  // diagnostics: bit 6 = no row records
    h->lds_bytes = spicey_lds_bytes(P, K, true, h->hres.tail_n);
  }

  // ---- uploads -----------------------------------------------------------------------------------
  const size_t ni = (size_t)h->n_inst;
  if (hipMalloc(&h->d_blob, h->hp.blob.size()) != hipSuccess) { h->err = "hipMalloc(program) failed"; return fail(SPICEY_ERR_HIP); }
  if (hipMemcpy(h->d_blob, h->hp.blob.data(), h->hp.blob.size(), hipMemcpyHostToDevice) != hipSuccess) {
    h->err = "hipMemcpy(program) failed";
    return fail(SPICEY_ERR_HIP);
  }
  h->dprog = h->hp.bind(h->d_blob);
  if (h->interp == 2) {
    if (hipMalloc(&h->d_res, h->hres.blob.size()) != hipSuccess ||
        hipMemcpy(h->d_res, h->hres.blob.data(), h->hres.blob.size(), hipMemcpyHostToDevice) != hipSuccess) {
      h->err = "upload of the resident program failed";
      return fail(SPICEY_ERR_HIP);
    }
    h->dres = h->hres.bind(h->d_res);
    if (hipMalloc((void **)&h->d_Pstruct, sizeof(SpiceyProg)) != hipSuccess || hipMalloc((void **)&h->d_Qstruct, sizeof(SpiceyResident)) != hipSuccess ||
        hipMalloc((void **)&h->d_Rstruct, sizeof(SpiceyRun)) != hipSuccess ||
        hipMemcpy(h->d_Pstruct, &h->dprog, sizeof(SpiceyProg), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_Qstruct, &h->dres, sizeof(SpiceyResident), hipMemcpyHostToDevice) != hipSuccess) {
      h->err = "upload of the argument structs failed";
      return fail(SPICEY_ERR_HIP);
    }
  }
#define UP(dst, src, cnt) \
  if ((rc = upload(h, &h->dst, desc->src, (cnt))) != SPICEY_OK) return fail(rc)
  UP(d_R, R_val, ni * P.nR);
  UP(d_C, C_val, ni * P.nC);
  UP(d_L, L_val, ni * P.nL);
  UP(d_Sron, S_ron, ni * P.nS);
  UP(d_Sroff, S_roff, ni * P.nS);
  UP(d_Svon, S_von, ni * P.nS);
  UP(d_Svoff, S_voff, ni * P.nS);
  UP(d_Dis, D_is, ni * P.nD);
  UP(d_Dn, D_n, ni * P.nD);
  UP(d_Cv, C_vprev, ni * P.nC);
  UP(d_Li, L_iprev, ni * P.nL);
  UP(d_Dv, D_vdprev, ni * P.nD);
  UP(d_Son, S_ison, ni * P.nS);
  UP(d_Cv0, C_vprev, ni * P.nC);
  UP(d_Li0, L_iprev, ni * P.nL);
  UP(d_Dv0, D_vdprev, ni * P.nD);
  UP(d_Son0, S_ison, ni * P.nS);
#undef UP
  const double *nodbl = nullptr;
  if ((rc = upload(h, &h->d_gstat, nodbl, ni * P.nGstat)) != SPICEY_OK) return fail(rc);
  if ((rc = upload(h, &h->d_statv, nodbl, ni * P.nLU)) != SPICEY_OK) return fail(rc);
  if ((rc = upload(h, &h->d_rcoef, nodbl, ni * (size_t)(P.nRhsIdx + 1))) != SPICEY_OK) return fail(rc);
  if ((rc = upload(h, &h->d_dpar, nodbl, ni * (size_t)P.nD * 2)) != SPICEY_OK) return fail(rc);
  if (!h->lds) {
    if ((rc = upload(h, &h->d_gW, nodbl, (size_t)h->grid * spicey_gw_doubles_per_wg(P, K))) != SPICEY_OK) return fail(rc);
    // group mode: several CUs per instance when the batch leaves CUs idle and the circuit is large enough for the
    // cross-workgroup barrier (~3 us per phase) to pay; all workgroups must be co-resident: grid * G <= #CU
    int G = h->opt.wgs_per_inst;
    if (G < 0 || G > 256 || (G > 1 && K > 2)) { h->err = "wgs_per_inst must be in [0, 256] (and inst_per_wg <= 2 with it)"; return fail(SPICEY_ERR_BAD_DESC); }
    if (G == 0) {
      G = 1;
      // (with dense fronts the group barriers that are left belong to a dozen wide levels, and the front tree wants one
      // workgroup per subtree: up to 128 CUs for a single instance (measured on rcd_mesh(100): 32 / 48 / 64 / 96 / 128
      // workgroups = 0.78 / 0.71 / 0.70 / 0.69 / 0.68 ms per step); without them every one of ~600 barriers per step grows with G)
      const int gmax = P.nFronts > 0 ? 128 : 16;
      // (the workspace is in HBM / L2 here: from ~10 k entries on (what no longer fits LDS) the extra CUs pay for the group barriers also without
      // fronts — one diode_chain(4000) 68 -> 60 us per step, (8000) 119 -> 77, rc_ladder(8000) 80 -> 66 at G = 16)
      if (K <= 2 && (P.nLU >= 10000 || P.nFronts > 0))
        while (G * 2 <= gmax && h->grid * G * 2 <= ncu) G *= 2;
    }
    if (h->grid * G > ncu) G = std::max(1, ncu / h->grid);
    if (G > 1) {
      // residency: ask the runtime how many workgroups of THIS kernel (its LDS size, these threads) a CU holds; the group
      // is laid out for one per CU, so any answer >= 1 means grid * G <= #CU workgroups are co-resident on an idle device
      const int T_grp = (P.nFronts > 0 && h->T > 512) ? 512 : h->T;
      if (spicey_grp_blocks_per_cu(P, K, T_grp) < 1) {
        if (h->opt.wgs_per_inst > 1) { h->err = "wgs_per_inst: the group-mode kernel cannot be resident on this device (occupancy query says 0 workgroups per CU)"; return fail(SPICEY_ERR_HIP); }
        G = 1;
      }
    }
    h->G = G;
    if (G > 1) {
      {
        // the first group-mode handle on a device drains it once: launches enqueued before were not recorded (DeviceGate)
        DeviceGate &g = device_gate(h->device);
        std::lock_guard<std::mutex> lk(g.mu);
        if (g.group_handles++ == 0) (void)hipDeviceSynchronize();
        h->gated = true;
      }
      const unsigned int *nou = nullptr;
      if ((rc = upload(h, &h->d_gsync, nou, (size_t)h->grid * SPICEY_GRP_SYNC_WORDS)) != SPICEY_OK) return fail(rc);
      const int32_t *noi = nullptr;
      if ((rc = upload(h, &h->d_gflags, noi, (size_t)h->grid * 4)) != SPICEY_OK) return fail(rc);
      if ((rc = upload(h, &h->d_Cv_s, nodbl, ni * P.nC)) != SPICEY_OK) return fail(rc);
      if ((rc = upload(h, &h->d_Li_s, nodbl, ni * P.nL)) != SPICEY_OK) return fail(rc);
      if ((rc = upload(h, &h->d_Dv_s, nodbl, ni * P.nD)) != SPICEY_OK) return fail(rc);
      if ((rc = upload(h, &h->d_Son_s, noi, ni * P.nS)) != SPICEY_OK) return fail(rc);
    }
  }
  if (P.nFronts > 0) {
    std::vector<uint32_t> first, list, owner((size_t)P.nFronts, 0u);
    spicey_build_front_schedule(h->hp, h->G, first, list);
    for (int w = 0; w < h->G; w++)
      for (uint32_t s2 = first[w]; s2 < first[w + 1]; s2++) owner[list[s2]] = (uint32_t)w;
    std::vector<uint32_t> all(first);
    all.insert(all.end(), list.begin(), list.end());
    all.insert(all.end(), owner.begin(), owner.end());
    if ((rc = upload(h, &h->d_fs, all.data(), all.size())) != SPICEY_OK) return fail(rc);
    if ((rc = upload(h, &h->d_front_ws, nodbl, (size_t)h->grid * (size_t)P.front_ws)) != SPICEY_OK) return fail(rc);
    const unsigned int *nou2 = nullptr;
    if ((rc = upload(h, &h->d_front_flags, nou2, (size_t)h->grid * 2 * (size_t)P.nFronts)) != SPICEY_OK) return fail(rc);
  }
  const int32_t *noint = nullptr;
  if ((rc = upload(h, &h->d_status, noint, (size_t)h->grid * 4)) != SPICEY_OK) return fail(rc);
  const unsigned long long *noull = nullptr;
  if ((rc = upload(h, &h->d_solves, noull, (size_t)h->grid)) != SPICEY_OK) return fail(rc);
  if (h->opt.profile)
    // (+ per-front event times behind the per-workgroup section timers)
    if ((rc = upload(h, &h->d_prof, noull, (size_t)h->grid * h->G * 72 + (size_t)h->grid * 4 * (size_t)P.nFronts)) != SPICEY_OK) return fail(rc);
  if (P.hybrid) {
    if ((rc = upload(h, &h->d_hybG, nodbl, ni * (size_t)P.nLU)) != SPICEY_OK) return fail(rc);
    if ((rc = upload(h, &h->d_hybUG, nodbl, ni * (size_t)(P.nU + P.nGdyn))) != SPICEY_OK) return fail(rc);
  }
  if (h->opt.diagnostics & 1)
    if ((rc = upload(h, &h->d_skip, noull, ni)) != SPICEY_OK) return fail(rc);
  if ((h->opt.diagnostics & 2) && P.nD > 0)
    if ((rc = upload(h, &h->d_linvd, nodbl, ni * P.nD)) != SPICEY_OK) return fail(rc);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    h->err = "stream/event creation failed";
    return fail(SPICEY_ERR_HIP);
  }
  *out = h;
  return SPICEY_OK;
}

extern "C" int32_t spicey_get_info(SpiceyHandle *h, SpiceyInfo *info) {
  if (!h || !info) return SPICEY_ERR_BAD_DESC;
  memset(info, 0, sizeof(*info));
  info->n_var = h->hp.hdr.n;
  info->nnz_a = h->hp.nnzA;
  info->nnz_lu = h->hp.hdr.nLU;
  info->n_levels = h->hp.hdr.nLevels;
  info->threads = h->T;
  info->inst_per_wg = h->K;
  info->lds_bytes = h->lds ? (int32_t)h->lds_bytes : 0;
  info->n_cur = h->hp.hdr.nCur;
  info->n_out = h->hp.hdr.nOut;
  info->n_workgroups = h->grid;
  info->interpreter = h->interp;
  info->geometry = h->interp == 2 ? (h->packed ? 2 : 1) : 0;
  info->tail_levels = h->hres.tail_n;
  info->wgs_per_inst = h->G;
  info->resident_slots = h->interp == 2 ? h->hres.rmax : 0;
  info->resident_tasks = h->hres.resident_tasks;
  info->streamed_tasks = h->hres.streamed_tasks;
  info->program_bytes = (int64_t)h->hp.blob.size();
  info->algorithmic_bytes_solve = h->algo_bytes;
  info->factor_reuse = (h->hp.hdr.nD == 0 && h->hp.hdr.nS == 0 && h->hp.hdr.nDynEnt == 0 && !((h->opt.debug >> 1) & 1)) ? 1 : 0;
  info->n_fronts = h->hp.hdr.nFronts;
  info->front_cut = h->hp.hdr.front_cut;
  info->max_front = h->hp.hdr.max_front_mp;
  info->front_ws_bytes = h->hp.hdr.front_ws * (int64_t)sizeof(double);
  info->pcr_rows = (h->interp == 2 && h->K == 1) ? h->hp.hdr.pcr_n : 0;
  info->pcr_level = info->pcr_rows ? h->hp.hdr.pcr_level : 0;
  info->hybrid_entries = h->hp.hdr.hybrid ? h->hp.hdr.hyb_g0 + h->hp.hdr.hyb_g2 : 0;
  return SPICEY_OK;
}

// enqueue the kernel of a prepared launch behind its device's admission gate; ev0 / ev1 bracket the kernel alone
static int32_t enqueue_kernel(SpiceyHandle *h, const SpiceyRun &R, hipStream_t st) {
  DeviceGate &g = device_gate(h->device);
  std::lock_guard<std::mutex> lk(g.mu);  // (wait, launch and record are one step with respect to other launches)
  const bool group = h->G > 1;
  int slot = -1;
  HIPCHK(h, gate_before_launch(g, group, st, &slot));
  HIPCHK(h, hipEventRecord(h->ev0, st));  // (argument upload, flag resets and admission waits stay outside the timed kernel)
  if (h->interp == 2) {
    HIPCHK(h, spicey_launch_tran_v2(h->dprog, h->dres, h->d_Pstruct, h->d_Qstruct, h->d_Rstruct, h->K, h->grid, h->T, st, h->packed));
  } else if (group) {
    HIPCHK(h, spicey_launch_tran_grp(h->dprog, R, h->K, h->grid, h->T, st));
  } else {
    HIPCHK(h, spicey_launch_tran(h->dprog, R, h->K, h->lds, h->grid, h->T, st));
  }
  HIPCHK(h, hipEventRecord(h->ev1, st));
  HIPCHK(h, gate_after_launch(g, group, st, slot));
  return SPICEY_OK;
}

static size_t prof_words(const SpiceyHandle *h) { return (size_t)h->grid * h->G * 72 + (size_t)h->grid * 4 * (size_t)h->hp.hdr.nFronts; }

extern "C" int32_t spicey_run_device(SpiceyHandle *h, int64_t steps, double dt, const double *d_src_table, double *d_out_v,
                                     double *d_out_i, int32_t *d_iters, void *stream) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  if (steps < 0 || !d_out_v || (h->hp.hdr.nV > 0 && !d_src_table)) { h->err = "bad run arguments"; return SPICEY_ERR_BAD_DESC; }
  if (h->hp.structurally_singular) {
    h->err = "singular at inst 0 step 0 iter 0 (structurally singular matrix)";
    return SPICEY_ERR_SINGULAR;
  }
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  // a run still in flight on ANOTHER stream: this launch would reset status words, barrier words and front flags under
  // it — finish it first (its result is then reported here instead of by the next spicey_sync)
  if (h->pending && h->last_stream != st) {
    const int32_t rc0 = spicey_sync(h);
    if (rc0 != SPICEY_OK) return rc0;
  }
  SpiceyRun R{};
  R.n_inst = h->n_inst;
  R.want_currents = d_out_i != nullptr;
  R.debug_empty_phases = h->opt.debug >> 8;
  R.no_reuse = (h->opt.debug >> 1) & 1;
  R.steps = steps;
  R.dt = dt;
  R.R_val = h->d_R; R.C_val = h->d_C; R.L_val = h->d_L;
  R.S_ron = h->d_Sron; R.S_roff = h->d_Sroff; R.S_von = h->d_Svon; R.S_voff = h->d_Svoff;
  R.D_is = h->d_Dis; R.D_n = h->d_Dn;
  R.C_vprev = h->d_Cv; R.L_iprev = h->d_Li; R.D_vdprev = h->d_Dv; R.S_ison = h->d_Son;
  R.gstat = h->d_gstat; R.statv = h->d_statv; R.rcoef = h->d_rcoef; R.gW = h->d_gW; R.dpar = h->d_dpar;
  R.src = d_src_table; R.out_v = d_out_v; R.out_i = d_out_i; R.iters = d_iters;
  R.status = h->d_status; R.solves = h->d_solves; R.prof = h->d_prof;
  if (h->d_prof) HIPCHK(h, hipMemsetAsync(h->d_prof, 0, prof_words(h) * sizeof(unsigned long long), st));
  // diagnostics: counters and per-step maxima start from zero in every run
  h->last_steps = steps;
  if (h->d_skip) {
    HIPCHK(h, hipMemsetAsync(h->d_skip, 0, (size_t)h->n_inst * sizeof(unsigned long long), st));
    R.skip_risk = h->d_skip;
  }
  if (h->opt.diagnostics & 2) {
    const size_t need = (size_t)h->n_inst * (size_t)(steps + 1);
    if (need > h->linerr_cap) {
      if (h->pending) { const int32_t rc0 = spicey_sync(h); if (rc0 != SPICEY_OK) return rc0; }
      if (h->d_linerr) (void)hipFree(h->d_linerr);
      h->d_linerr = nullptr; h->linerr_cap = 0;
      HIPCHK(h, hipMalloc((void **)&h->d_linerr, need * sizeof(unsigned long long)));
      h->linerr_cap = need;
    }
    HIPCHK(h, hipMemsetAsync(h->d_linerr, 0, need * sizeof(unsigned long long), st));
    R.lin_err = h->d_linerr;
    R.lin_vd = h->d_linvd;  // (null without diodes: the error stays 0)
  }
  R.hyb_G = h->d_hybG; R.hyb_ug = h->d_hybUG;
  R.front_ticks = (h->d_prof && h->hp.hdr.nFronts > 0) ? h->d_prof + (size_t)h->grid * h->G * 72 : nullptr;
  R.wgs_per_group = h->G;
  R.grp_sync = h->d_gsync;
  R.grp_flags = h->d_gflags;
  if (h->hp.hdr.nFronts > 0) {
    R.front_ws = h->d_front_ws;
    R.fs_first = h->d_fs;
    R.fs_list = h->d_fs + (h->G + 1);
    R.fs_owner = R.fs_list + h->hp.hdr.nFronts;
    R.front_flags = h->d_front_flags;
    R.front_lds_doubles = (h->opt.debug & 8) ? 6144 : SPICEY_FRONT_LDS_DOUBLES;
    R.front_right_looking = getenv("SPICEY_FRONT_RIGHT_LOOKING") != nullptr ? 1 : 0;  // experiments: the round-2 sweep of staged fronts  // diagnostics: bit 3 = stage every front above 64 rows through panels
    HIPCHK(h, hipMemsetAsync(h->d_front_flags, 0, (size_t)h->grid * 2 * (size_t)h->hp.hdr.nFronts * sizeof(unsigned int), st));
  }
  if (h->G > 1) {
    HIPCHK(h, hipMemsetAsync(h->d_gsync, 0, (size_t)h->grid * SPICEY_GRP_SYNC_WORDS * sizeof(unsigned int), st));
    // longest single cross-workgroup wait, in ticks of the chip-wide 100 MHz counter
    int ms = h->opt.group_timeout_ms;
    if (ms <= 0) { const char *e = getenv("SPICEY_GROUP_TIMEOUT_MS"); ms = e ? atoi(e) : 0; }
    if (ms <= 0) ms = 5000;
    R.grp_timeout_ticks = (unsigned long long)ms * 100000ull;
    if (h->opt.group_retry) {
      // the state entering this launch, for the one relaunch after a bounded-wait abort (spicey_sync)
      const SpiceyProg &P = h->hp.hdr;
      const size_t ni = (size_t)h->n_inst;
      if (P.nC) HIPCHK(h, hipMemcpyAsync(h->d_Cv_s, h->d_Cv, ni * P.nC * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (P.nL) HIPCHK(h, hipMemcpyAsync(h->d_Li_s, h->d_Li, ni * P.nL * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (P.nD) HIPCHK(h, hipMemcpyAsync(h->d_Dv_s, h->d_Dv, ni * P.nD * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (P.nS) HIPCHK(h, hipMemcpyAsync(h->d_Son_s, h->d_Son, ni * P.nS * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
    R.force_abort = getenv("SPICEY_TEST_FORCE_GROUP_ABORT") != nullptr ? 1 : 0;  // tests: the first attempt of every launch aborts
    h->grp_R = R;
  }
  if (h->interp == 2) {
    h->run_args = R;
    HIPCHK(h, hipMemcpyAsync(h->d_Rstruct, &h->run_args, sizeof(SpiceyRun), hipMemcpyHostToDevice, st));
  }
  const int32_t rc = enqueue_kernel(h, R, st);
  if (rc != SPICEY_OK) return rc;
  h->pending = true;
  h->last_stream = st;
  return SPICEY_OK;
}

extern "C" int32_t spicey_sync(SpiceyHandle *h) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  if (!h->pending) return SPICEY_OK;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<int32_t> status((size_t)h->grid * 4);
  std::vector<unsigned long long> solves((size_t)h->grid);
  int best = -1;
  std::string aborted;  // text of an aborted attempt that was repeated
  for (int attempt = 0;; attempt++) {
    HIPCHK(h, hipStreamSynchronize(h->last_stream));
    h->pending = false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_ms = ms;
    HIPCHK(h, hipMemcpy(status.data(), h->d_status, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(solves.data(), h->d_solves, solves.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    h->last_solves = 0;
    for (auto s : solves) h->last_solves += (int64_t)s;
    std::vector<unsigned int> gsync;
    if (h->G > 1 && h->d_gsync) {
      gsync.resize((size_t)h->grid * SPICEY_GRP_SYNC_WORDS);
      HIPCHK(h, hipMemcpy(gsync.data(), h->d_gsync, gsync.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
      for (int g = 0; g < h->grid; g++) h->stale_polls += (int64_t)gsync[(size_t)g * SPICEY_GRP_SYNC_WORDS + 8];
    }
    // earliest failure wins (the reference throws at the first singular solve)
    best = -1;
    for (int g = 0; g < h->grid; g++)
      if (status[(size_t)g * 4] != 0 && (best < 0 || status[(size_t)g * 4 + 2] < status[(size_t)best * 4 + 2])) best = g;
    if (best < 0 || status[(size_t)best * 4] != 3) break;
    // (the first workgroup that gave up left a note in its group's barrier words, GpuGroupExec::note_timeout)
    const unsigned int none[9] = {0};
    const unsigned int *note = gsync.empty() ? none : gsync.data() + (size_t)best * SPICEY_GRP_SYNC_WORDS;
    const char *kind = note[2] == 2 ? "front hand-over" : note[2] == 3 ? "census barrier" : note[2] == 1 ? "group barrier" : note[2] == 4 ? "XCD census does not add up" : "abort word raised";
    char buf[384];
    snprintf(buf, sizeof(buf), "cross-workgroup wait timed out (group mode) at step %d: %s, group %d of %d, workgroup %u of %d (XCD %u), %s %u, waited for %u, saw %u; %d threads, %d fronts, timeout %.0f ms",
             status[(size_t)best * 4 + 2], kind, best, h->grid, note[3], h->G, note[7], note[2] == 2 ? "front flag" : "barrier", note[4], note[5], note[6], h->T,
             h->hp.hdr.nFronts, (double)h->grp_R.grp_timeout_ticks / 1e5);
    h->err = buf;
    fprintf(stderr, "spicey: %s%s\n", buf, (attempt == 0 && h->opt.group_retry) ? " -- repeating the launch once (SpiceyOptions.group_retry)" : "");
    if (attempt > 0 || h->G <= 1 || !h->opt.group_retry) return SPICEY_ERR_HIP;
    // The bounded wait turned what would have been a hang into an abort; nothing of the aborted launch is kept.  On request
    // the launch is repeated ONCE from the state it started with (the kernel writes state only in its last step, but that
    // step may be the one that aborted): same arguments, same stream, fresh barrier words and front flags.
    h->group_retries++;
    aborted = buf;
    const SpiceyProg &P = h->hp.hdr;
    const size_t ni = (size_t)h->n_inst;
    hipStream_t st = h->last_stream;
    if (P.nC) HIPCHK(h, hipMemcpyAsync(h->d_Cv, h->d_Cv_s, ni * P.nC * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (P.nL) HIPCHK(h, hipMemcpyAsync(h->d_Li, h->d_Li_s, ni * P.nL * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (P.nD) HIPCHK(h, hipMemcpyAsync(h->d_Dv, h->d_Dv_s, ni * P.nD * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (P.nS) HIPCHK(h, hipMemcpyAsync(h->d_Son, h->d_Son_s, ni * P.nS * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemsetAsync(h->d_gsync, 0, (size_t)h->grid * SPICEY_GRP_SYNC_WORDS * sizeof(unsigned int), st));
    if (P.nFronts > 0) HIPCHK(h, hipMemsetAsync(h->d_front_flags, 0, (size_t)h->grid * 2 * (size_t)P.nFronts * sizeof(unsigned int), st));
    if (h->d_prof) HIPCHK(h, hipMemsetAsync(h->d_prof, 0, prof_words(h) * sizeof(unsigned long long), st));
    h->grp_R.force_abort = 0;
    const int32_t rc = enqueue_kernel(h, h->grp_R, st);
    if (rc != SPICEY_OK) return rc;
    h->pending = true;
  }
  // (the text of an aborted attempt stays readable although its repetition went through)
  if (!aborted.empty() && best < 0) h->err = "recovered: " + aborted;
  if (best >= 0) {
    char buf[160];
    snprintf(buf, sizeof(buf), "singular at inst %d step %d iter %d", status[(size_t)best * 4 + 1], status[(size_t)best * 4 + 2],
             status[(size_t)best * 4 + 3]);
    h->err = buf;
    return SPICEY_ERR_SINGULAR;
  }
  return SPICEY_OK;
}

extern "C" int32_t spicey_group_retries(const SpiceyHandle *h) { return h ? h->group_retries : 0; }
extern "C" int64_t spicey_group_stale_polls(const SpiceyHandle *h) { return h ? h->stale_polls : 0; }

extern "C" int32_t spicey_run(SpiceyHandle *h, int64_t steps, double dt, const double *src_table, double *out_v, double *out_i,
                              int32_t *iters) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  if (steps < 0 || !out_v || (h->hp.hdr.nV > 0 && !src_table)) { h->err = "bad run arguments"; return SPICEY_ERR_BAD_DESC; }
  if (h->hp.structurally_singular) {
    h->err = "singular at inst 0 step 0 iter 0 (structurally singular matrix)";
    return SPICEY_ERR_SINGULAR;
  }
  HIPCHK(h, hipSetDevice(h->device));
  Roctx range_run("spicey_run");
  const SpiceyProg &P = h->hp.hdr;
  const size_t np = (size_t)steps + 1, ni = (size_t)h->n_inst;
  double *d_src = nullptr, *d_v = nullptr, *d_i = nullptr;
  int32_t *d_it = nullptr;
  int32_t rc = SPICEY_OK;
  auto cleanup = [&]() {
    if (d_src) (void)hipFree(d_src);
    if (d_v) (void)hipFree(d_v);
    if (d_i) (void)hipFree(d_i);
    if (d_it) (void)hipFree(d_it);
  };
#define TRY(call)                                                        \
  do {                                                                   \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) {                                             \
      h->err = std::string(#call) + ": " + hipGetErrorString(e__);       \
      cleanup();                                                         \
      return SPICEY_ERR_HIP;                                             \
    }                                                                    \
  } while (0)
  TRY(hipMalloc((void **)&d_src, std::max<size_t>(np * P.nV, 1) * sizeof(double)));
  if (P.nV) TRY(hipMemcpyAsync(d_src, src_table, np * P.nV * sizeof(double), hipMemcpyHostToDevice, h->stream));
  TRY(hipMalloc((void **)&d_v, std::max<size_t>(ni * np * P.nOut, 1) * sizeof(double)));
  if (out_i) TRY(hipMalloc((void **)&d_i, std::max<size_t>(ni * np * P.nCur, 1) * sizeof(double)));
  if (iters) TRY(hipMalloc((void **)&d_it, ni * np * sizeof(int32_t)));
  {
    Roctx range_kernel("spicey_run:kernel");
    rc = spicey_run_device(h, steps, dt, d_src, d_v, d_i, d_it, h->stream);
    if (rc == SPICEY_OK) rc = spicey_sync(h);
  }
  if (rc == SPICEY_OK) {
    Roctx range_copy("spicey_run:results");
    TRY(hipMemcpy(out_v, d_v, ni * np * P.nOut * sizeof(double), hipMemcpyDeviceToHost));
    if (out_i) TRY(hipMemcpy(out_i, d_i, ni * np * P.nCur * sizeof(double), hipMemcpyDeviceToHost));
    if (iters) TRY(hipMemcpy(iters, d_it, ni * np * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
#undef TRY
  cleanup();
  return rc;
}

extern "C" int32_t spicey_get_state(SpiceyHandle *h, double *C_vprev, double *L_iprev, double *D_vdprev, int32_t *S_ison) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  int32_t rc = spicey_sync(h);
  if (rc != SPICEY_OK && rc != SPICEY_ERR_SINGULAR) return rc;
  const SpiceyProg &P = h->hp.hdr;
  const size_t ni = (size_t)h->n_inst;
  if (C_vprev && P.nC) HIPCHK(h, hipMemcpy(C_vprev, h->d_Cv, ni * P.nC * sizeof(double), hipMemcpyDeviceToHost));
  if (L_iprev && P.nL) HIPCHK(h, hipMemcpy(L_iprev, h->d_Li, ni * P.nL * sizeof(double), hipMemcpyDeviceToHost));
  if (D_vdprev && P.nD) HIPCHK(h, hipMemcpy(D_vdprev, h->d_Dv, ni * P.nD * sizeof(double), hipMemcpyDeviceToHost));
  if (S_ison && P.nS) HIPCHK(h, hipMemcpy(S_ison, h->d_Son, ni * P.nS * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SPICEY_OK;
}

extern "C" int32_t spicey_set_state(SpiceyHandle *h, const double *C_vprev, const double *L_iprev, const double *D_vdprev,
                                    const int32_t *S_ison) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  int32_t rc = spicey_sync(h);
  if (rc != SPICEY_OK && rc != SPICEY_ERR_SINGULAR) return rc;
  const SpiceyProg &P = h->hp.hdr;
  const size_t ni = (size_t)h->n_inst;
  HIPCHK(h, hipSetDevice(h->device));
  if (C_vprev && P.nC) HIPCHK(h, hipMemcpy(h->d_Cv, C_vprev, ni * P.nC * sizeof(double), hipMemcpyHostToDevice));
  if (L_iprev && P.nL) HIPCHK(h, hipMemcpy(h->d_Li, L_iprev, ni * P.nL * sizeof(double), hipMemcpyHostToDevice));
  if (D_vdprev && P.nD) HIPCHK(h, hipMemcpy(h->d_Dv, D_vdprev, ni * P.nD * sizeof(double), hipMemcpyHostToDevice));
  if (S_ison && P.nS) HIPCHK(h, hipMemcpy(h->d_Son, S_ison, ni * P.nS * sizeof(int32_t), hipMemcpyHostToDevice));
  return SPICEY_OK;
}

extern "C" int32_t spicey_reset_state(SpiceyHandle *h, void *stream) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  const SpiceyProg &P = h->hp.hdr;
  const size_t ni = (size_t)h->n_inst;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  if (P.nC) HIPCHK(h, hipMemcpyAsync(h->d_Cv, h->d_Cv0, ni * P.nC * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (P.nL) HIPCHK(h, hipMemcpyAsync(h->d_Li, h->d_Li0, ni * P.nL * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (P.nD) HIPCHK(h, hipMemcpyAsync(h->d_Dv, h->d_Dv0, ni * P.nD * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (P.nS) HIPCHK(h, hipMemcpyAsync(h->d_Son, h->d_Son0, ni * P.nS * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return SPICEY_OK;
}

extern "C" int64_t spicey_last_solve_count(SpiceyHandle *h) { return h ? h->last_solves : 0; }

extern "C" int64_t spicey_last_skip_risk(SpiceyHandle *h, int64_t *per_inst) {
  if (!h || !h->d_skip) return -1;
  if (spicey_sync(h) == SPICEY_ERR_HIP) return -1;
  std::vector<unsigned long long> tmp((size_t)h->n_inst);
  if (hipSetDevice(h->device) != hipSuccess || hipMemcpy(tmp.data(), h->d_skip, tmp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int64_t tot = 0;
  for (size_t i = 0; i < tmp.size(); i++) { tot += (int64_t)tmp[i]; if (per_inst) per_inst[i] = (int64_t)tmp[i]; }
  return tot;
}

extern "C" int32_t spicey_get_lin_err(SpiceyHandle *h, double *out) {
  if (!h || !out) return SPICEY_ERR_BAD_DESC;
  if (!(h->opt.diagnostics & 2) || !h->d_linerr || h->last_steps < 0) { h->err = "spicey_get_lin_err needs SpiceyOptions.diagnostics bit 1 and a finished run"; return SPICEY_ERR_BAD_DESC; }
  const int32_t rc = spicey_sync(h);
  if (rc != SPICEY_OK && rc != SPICEY_ERR_SINGULAR) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  // (bit patterns of non-negative doubles: a plain copy)
  HIPCHK(h, hipMemcpy(out, h->d_linerr, (size_t)h->n_inst * (size_t)(h->last_steps + 1) * sizeof(double), hipMemcpyDeviceToHost));
  return SPICEY_OK;
}
extern "C" double spicey_last_kernel_ms(SpiceyHandle *h) { return h ? h->last_ms : 0.0; }

extern "C" int32_t spicey_debug_phase_cycles(SpiceyHandle *h, uint64_t *out, int32_t n) {
  if (!h || !out) return 0;
  for (int i = 0; i < n; i++) out[i] = 0;
  if (!h->d_prof) return 0;
  if (spicey_sync(h) == SPICEY_ERR_HIP) return 0;
  unsigned long long tmp[72];
  if (hipMemcpy(tmp, h->d_prof, sizeof(tmp), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  for (int i = 0; i < n && i < 72; i++) out[i] = tmp[i];
  return 72;
}

// Same for launched workgroup `wg` (group mode: wg = group * wgs_per_inst + index; the slots then hold 100 MHz wall ticks
// per SECTION: [1] B, [8] factor levels below the cut, [9] fronts forward, [10] fronts backward, [11] sync + publish,
// [12..20] inside the fronts (wait, assemble, panel load, diagonal block, triangular solves, trailing update, ...), [40] backward
// levels, [4] Z).
extern "C" int32_t spicey_debug_phase_cycles_wg(SpiceyHandle *h, int32_t wg, uint64_t *out, int32_t n) {
  if (!h || !out) return 0;
  for (int i = 0; i < n; i++) out[i] = 0;
  if (!h->d_prof || wg < 0 || wg >= h->grid * h->G) return 0;
  if (spicey_sync(h) == SPICEY_ERR_HIP) return 0;
  unsigned long long tmp[72];
  if (hipMemcpy(tmp, h->d_prof + (size_t)wg * 72, sizeof(tmp), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  for (int i = 0; i < n && i < 72; i++) out[i] = tmp[i];
  return 72;
}

// Per-front event times of group `grp` in the last run (profile option; program.h, SpiceyRun::front_ticks): out[f * 4 + e]
// = 100 MHz ticks since the owner entered the forward sweep, SUMMED over the solves; also the fronts' shape and owner
// (meta[f * 4 + {0: pivots, 1: boundary, 2: parent, 3: owning workgroup}]).  Returns the number of fronts.
extern "C" int32_t spicey_debug_front_ticks(SpiceyHandle *h, int32_t grp, uint64_t *out, int32_t *meta, int32_t cap_fronts) {
  if (!h || !out || !meta || !h->d_prof || grp < 0 || grp >= h->grid) return 0;
  const int nf = h->hp.hdr.nFronts;
  if (nf <= 0 || cap_fronts < nf) return 0;
  if (spicey_sync(h) == SPICEY_ERR_HIP) return 0;
  std::vector<unsigned long long> tmp((size_t)nf * 4);
  if (hipMemcpy(tmp.data(), h->d_prof + (size_t)h->grid * h->G * 72 + (size_t)grp * 4 * nf, tmp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  for (size_t i = 0; i < tmp.size(); i++) out[i] = tmp[i];
  std::vector<uint32_t> first, list;
  spicey_build_front_schedule(h->hp, h->G, first, list);
  for (int w = 0; w < h->G; w++)
    for (uint32_t s2 = first[w]; s2 < first[w + 1]; s2++) meta[(size_t)list[s2] * 4 + 3] = w;
  for (int f = 0; f < nf; f++) {
    meta[(size_t)f * 4 + 0] = h->hp.fronts[f].p; meta[(size_t)f * 4 + 1] = h->hp.fronts[f].q; meta[(size_t)f * 4 + 2] = h->hp.fronts[f].parent;
  }
  return nf;
}

// ---------------------------------------------------------------------------------------------------------------------
// Several devices behind one handle: instance shards, one SpiceyHandle per shard, host threads around the blocking runs.
#include <thread>

struct SpiceyMulti {
  struct Shard { SpiceyHandle *h = nullptr; int device = 0, first = 0, count = 0; };
  std::vector<Shard> shards;
  int n_inst = 0;
  int nC = 0, nL = 0, nD = 0, nS = 0, nOut = 0, nCur = 0;
  int64_t last_solves = 0;
  double last_ms = 0.0;
  std::string err;
};

extern "C" const char *spicey_multi_last_error(SpiceyMulti *m) { return m ? m->err.c_str() : g_err.c_str(); }

extern "C" void spicey_destroy_multi(SpiceyMulti *m) {
  if (!m) return;
  for (auto &s : m->shards) spicey_destroy(s.h);
  delete m;
}

extern "C" int32_t spicey_create_multi(const SpiceyDesc *desc, const SpiceyOptions *opt, const int32_t *devices, int32_t n_dev, SpiceyMulti **out) {
  if (!out) { g_err = "null out pointer"; return SPICEY_ERR_BAD_DESC; }
  *out = nullptr;
  if (!desc || !devices || n_dev < 1) { g_err = "spicey_create_multi needs a descriptor and a list of >= 1 devices"; return SPICEY_ERR_BAD_DESC; }
  if (desc->n_inst < 1) { g_err = "negative count or n_inst < 1"; return SPICEY_ERR_BAD_DESC; }
  for (int d = 0; d < n_dev; d++)
    if (devices[d] < 0) { g_err = "device ordinal out of range"; return SPICEY_ERR_BAD_DESC; }
  SpiceyMulti *m = new SpiceyMulti();
  m->n_inst = desc->n_inst;
  const int ni = desc->n_inst;
  for (int d = 0; d < n_dev; d++) {
    // block partition: shard d = instances [ceil(d ni / n_dev), ceil((d + 1) ni / n_dev))
    const int lo = (int)(((int64_t)d * ni + n_dev - 1) / n_dev), hi = (int)(((int64_t)(d + 1) * ni + n_dev - 1) / n_dev);
    if (hi <= lo) continue;
    SpiceyDesc sd = *desc;
    sd.n_inst = hi - lo;
    auto adv = [&](const double *p, int n) { return p ? p + (size_t)lo * (size_t)n : p; };
    sd.R_val = adv(desc->R_val, desc->nR);
    sd.C_val = adv(desc->C_val, desc->nC); sd.C_vprev = adv(desc->C_vprev, desc->nC);
    sd.L_val = adv(desc->L_val, desc->nL); sd.L_iprev = adv(desc->L_iprev, desc->nL);
    sd.S_ron = adv(desc->S_ron, desc->nS); sd.S_roff = adv(desc->S_roff, desc->nS);
    sd.S_von = adv(desc->S_von, desc->nS); sd.S_voff = adv(desc->S_voff, desc->nS);
    sd.S_ison = desc->S_ison ? desc->S_ison + (size_t)lo * (size_t)desc->nS : nullptr;
    sd.D_is = adv(desc->D_is, desc->nD); sd.D_n = adv(desc->D_n, desc->nD); sd.D_vdprev = adv(desc->D_vdprev, desc->nD);
    SpiceyOptions so{};
    if (opt) so = *opt;
    so.device = devices[d];
    SpiceyMulti::Shard s;
    s.device = devices[d]; s.first = lo; s.count = hi - lo;
    const int32_t rc = spicey_create(&sd, &so, &s.h);
    if (rc != SPICEY_OK) {
      char buf[64];
      snprintf(buf, sizeof(buf), " (shard %d on device %d)", d, devices[d]);
      g_err += buf;
      spicey_destroy_multi(m);
      return rc;
    }
    m->shards.push_back(s);
  }
  const SpiceyProg &P = m->shards[0].h->hp.hdr;
  m->nC = P.nC; m->nL = P.nL; m->nD = P.nD; m->nS = P.nS; m->nOut = P.nOut; m->nCur = P.nCur;
  *out = m;
  return SPICEY_OK;
}

extern "C" int32_t spicey_run_multi(SpiceyMulti *m, int64_t steps, double dt, const double *src_table, double *out_v, double *out_i, int32_t *iters) {
  if (!m) return SPICEY_ERR_BAD_DESC;
  if (steps < 0 || !out_v) { m->err = "bad run arguments"; return SPICEY_ERR_BAD_DESC; }
  const size_t np = (size_t)steps + 1;
  std::vector<int32_t> rcs(m->shards.size(), SPICEY_OK);
  std::vector<std::thread> th;
  for (size_t i = 0; i < m->shards.size(); i++) {
    th.emplace_back([&, i]() {
      const SpiceyMulti::Shard &s = m->shards[i];
      rcs[i] = spicey_run(s.h, steps, dt, src_table, out_v + (size_t)s.first * np * (size_t)m->nOut,
                          out_i ? out_i + (size_t)s.first * np * (size_t)m->nCur : nullptr, iters ? iters + (size_t)s.first * np : nullptr);
    });
  }
  for (auto &t : th) t.join();
  m->last_solves = 0;
  m->last_ms = 0.0;
  int32_t rc = SPICEY_OK;
  for (size_t i = 0; i < m->shards.size(); i++) {
    const SpiceyMulti::Shard &s = m->shards[i];
    if (rcs[i] != SPICEY_OK && rc == SPICEY_OK) {  // first failing shard in instance order; its instance number made global
      rc = rcs[i];
      char buf[96];
      snprintf(buf, sizeof(buf), " (shard %d: instances %d..%d on device %d)", (int)i, s.first, s.first + s.count - 1, s.device);
      m->err = std::string(spicey_last_error(s.h)) + buf;
    }
    m->last_solves += spicey_last_solve_count(s.h);
    m->last_ms = std::max(m->last_ms, spicey_last_kernel_ms(s.h));
  }
  return rc;
}

extern "C" int32_t spicey_get_state_multi(SpiceyMulti *m, double *C_vprev, double *L_iprev, double *D_vdprev, int32_t *S_ison) {
  if (!m) return SPICEY_ERR_BAD_DESC;
  for (auto &s : m->shards) {
    const int32_t rc = spicey_get_state(s.h, C_vprev ? C_vprev + (size_t)s.first * m->nC : nullptr, L_iprev ? L_iprev + (size_t)s.first * m->nL : nullptr,
                                        D_vdprev ? D_vdprev + (size_t)s.first * m->nD : nullptr, S_ison ? S_ison + (size_t)s.first * m->nS : nullptr);
    if (rc != SPICEY_OK) { m->err = spicey_last_error(s.h); return rc; }
  }
  return SPICEY_OK;
}

extern "C" int32_t spicey_multi_get_shard(SpiceyMulti *m, int32_t shard, SpiceyInfo *info, int32_t *device, int32_t *first_inst, int32_t *n_inst) {
  if (!m || shard < 0 || shard >= (int32_t)m->shards.size()) return SPICEY_ERR_BAD_DESC;
  const SpiceyMulti::Shard &s = m->shards[shard];
  if (info) spicey_get_info(s.h, info);
  if (device) *device = s.device;
  if (first_inst) *first_inst = s.first;
  if (n_inst) *n_inst = s.count;
  return SPICEY_OK;
}

extern "C" int32_t spicey_multi_group_retries(SpiceyMulti *m) {
  int32_t n = 0;
  if (m) for (auto &s : m->shards) n += spicey_group_retries(s.h);
  return n;
}
extern "C" int64_t spicey_multi_group_stale_polls(SpiceyMulti *m) {
  int64_t n = 0;
  if (m) for (auto &s : m->shards) n += spicey_group_stale_polls(s.h);
  return n;
}
extern "C" int64_t spicey_multi_last_solve_count(SpiceyMulti *m) { return m ? m->last_solves : 0; }
extern "C" double spicey_multi_last_kernel_ms(SpiceyMulti *m) { return m ? m->last_ms : 0.0; }
