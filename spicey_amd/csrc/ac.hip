// ac.hip — AC sweep on the GPU: kernel and C-ABI (spicey_ac_* of include/spicey_hip.h).
//
// One workgroup per (instance, frequency) pair runs a whole complex MNA solve (ac_exec.h): the pairs are
// independent (simulateAC.ts:80 `for (const f of freqs)`), so the sweep is one launch of n_inst * n_freq workgroups.
// Workspace: nW complex entries = 16 bytes each, in LDS when it fits (<= ~10 000 entries), else one slice of a global
// buffer per workgroup.  The schedule ("program") is the transient one of symbolic.cpp built from the descriptor
// without diodes and switches.  No CPU path: without a HIP device spicey_ac_create returns SPICEY_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spicey_hip.h"
#include "ac_exec.h"
#include "kernels.h"
#include "symbolic.h"

namespace {

struct GpuAcExec {
  __device__ __forceinline__ int threads() const { return (int)blockDim.x; }
  __device__ __forceinline__ int atomic_inc(int32_t *p) { return atomicAdd(p, 1); }
  template <class F>
  __device__ __forceinline__ void phase(int, F f) {
    f((int)threadIdx.x);
    __syncthreads();
  }
};

template <class Regs>
struct GpuAcExecRes {
  Regs rr;
  __device__ __forceinline__ int threads() const { return (int)blockDim.x; }
  template <class R2>
  __device__ __forceinline__ R2 &regs(int) { return rr; }
  template <class F>
  __device__ __forceinline__ void phase(int, F f) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));  // (keeps per-thread addresses from being hoisted out of the frequency loop)
    f(tid);
    __syncthreads();
  }
};

// Resident sweep: blockIdx.x = instance * n_chunk + c; the workgroup runs frequencies c, c + n_chunk, ... of its instance
// with the task records and the frequency-independent stamp parts in registers (ac_exec.h).
template <int RMAX, int NSE>
__global__ void __launch_bounds__(512) spicey_ac_kernel_res(SpiceyProg P, SpiceyResident Q, SpiceyAcRun R, int n_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int32_t flags[2];
  GpuAcExecRes<AcResRegs<RMAX, NSE>> ex;
  spicey_ac_sweep_resident<RMAX, NSE>(ex, P, Q, R, (SpiceyCx *)smem, flags, (size_t)(blockIdx.x / (unsigned)n_chunk), (int64_t)(blockIdx.x % (unsigned)n_chunk),
                                      (int64_t)n_chunk);
}

template <bool LDS>
__global__ void __launch_bounds__(1024) spicey_ac_kernel(SpiceyProg P, SpiceyAcRun R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int32_t flags[2];
  const int64_t slot = R.slot_base + (int64_t)blockIdx.x;
  SpiceyCx *W = LDS ? (SpiceyCx *)smem : (SpiceyCx *)R.gW + (size_t)blockIdx.x * (size_t)P.nW;
  GpuAcExec ex;
  spicey_ac_solve(ex, P, R, W, flags, slot);
}

// Dense partial-pivoting fallback (ac_exec.h): blockIdx.x = index into the list of (instance, frequency) slots whose solve
// tripped a pivot guard; A | b and the sparse scratch of each in global memory, reduction scratch in LDS.
__global__ void __launch_bounds__(1024) spicey_ac_dense_kernel(const SpiceyProg *__restrict__ Pp, const SpiceyAcRun *__restrict__ Rp, const int64_t *slots,
                                                                SpiceyCx *Ws_all, SpiceyCx *A_all) {
  // (the two argument structs by pointer: by value their ~110 fields are all live SGPRs and some spill)
  const SpiceyProg &P = *Pp;
  const SpiceyAcRun &R = *Rp;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int32_t flags[2];
  const int T = (int)blockDim.x, n = P.n;
  double *sd = (double *)smem;
  int32_t *si = (int32_t *)(sd + (size_t)T + 2 * (size_t)n + 2);
  GpuAcExec ex;
  spicey_ac_dense_solve(ex, P, R, Ws_all + (size_t)blockIdx.x * (size_t)P.nW, A_all + (size_t)blockIdx.x * (size_t)n * ((size_t)n + 1), sd, si, flags,
                        slots[blockIdx.x]);
}

}  // namespace

// resident sweep geometry: <= 512 threads (256 VGPRs, no spills): 12 task records, 10 entries' stamp parts per thread
#define SPICEY_AC_RMAX 12
#define SPICEY_AC_NSE 10

struct SpiceyAcHandle {
  HostProgram hp;
  HostResident hres;       // resident layout of the 16-bit records (batched sweeps)
  SpiceyResident dres{};
  void *d_res = nullptr;
  bool resident_ok = false;
  int ncu = 256;
  int last_mode = 0;       // 1 = one workgroup per (instance, frequency), 2 = resident sweep
  int Tres = 512;          // threads of the resident sweep's workgroups
  SpiceyProg dprog{};
  SpiceyOptions opt{};
  int n_inst = 0, T = 256, device = 0;
  bool lds = true;
  size_t lds_bytes = 0;
  void *d_blob = nullptr;
  double *d_R = nullptr, *d_C = nullptr, *d_L = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0.0;
  int64_t last_dense = 0;  // solves of the last run that went through the dense partial-pivoting fallback
  std::string err;
};

static thread_local std::string g_ac_err;  // message of the calling thread's last failed spicey_ac_create

extern "C" const char *spicey_ac_last_error(SpiceyAcHandle *h) { return h ? h->err.c_str() : g_ac_err.c_str(); }

extern "C" void spicey_ac_destroy(SpiceyAcHandle *h) {
  if (!h) return;
  void *ptrs[] = {h->d_blob, h->d_R, h->d_C, h->d_L, h->d_res};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

#define ACCHK(h, call)                                                 \
  do {                                                                 \
    hipError_t e__ = (call);                                           \
    if (e__ != hipSuccess) {                                           \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);   \
      return SPICEY_ERR_HIP;                                           \
    }                                                                  \
  } while (0)

static int32_t ac_upload(SpiceyAcHandle *h, double **dst, const double *src, size_t count) {
  *dst = nullptr;
  ACCHK(h, hipMalloc((void **)dst, (count ? count : 1) * sizeof(double)));
  if (count && src) ACCHK(h, hipMemcpy(*dst, src, count * sizeof(double), hipMemcpyHostToDevice));
  return SPICEY_OK;
}

extern "C" int32_t spicey_ac_create(const SpiceyDesc *desc, const SpiceyOptions *opt, SpiceyAcHandle **out) {
  if (!out) { g_ac_err = "null out pointer"; return SPICEY_ERR_BAD_DESC; }
  *out = nullptr;
  if (!desc) { g_ac_err = "null descriptor"; return SPICEY_ERR_BAD_DESC; }
  SpiceyAcHandle *h = new SpiceyAcHandle();
  if (opt) h->opt = *opt;
  SpiceyDesc d = *desc;  // simulateAC.ts:38-59 stamps R, C, L and V only
  d.nS = 0;
  d.nD = 0;
  std::string err;
  int32_t rc = spicey_build_program(&d, h->hp, err, true, 0, false);  // (task records for every level: the real-valued cyclic reduction of a tridiagonal top is the transient kernel's)
  if (rc != SPICEY_OK) {
    g_ac_err = err;
    delete h;
    return rc;
  }
  h->n_inst = desc->n_inst;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_ac_err = "no HIP device: libspicey_hip has no CPU path";
    delete h;
    return SPICEY_ERR_NO_DEVICE;
  }
  h->device = h->opt.device;
  auto fail = [&](int32_t code) {
    g_ac_err = h->err;
    spicey_ac_destroy(h);
    return code;
  };
  if (h->device < 0 || h->device >= ndev) { h->err = "device ordinal out of range"; return fail(SPICEY_ERR_BAD_DESC); }
  if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(SPICEY_ERR_HIP); }
  const SpiceyProg &P = h->hp.hdr;
  h->lds_bytes = (size_t)P.nW * sizeof(SpiceyCx);
  h->lds = !h->opt.force_global && h->lds_bytes + 64 <= SPICEY_LDS_MAX;
  const int n = P.n;
  // measured on rc_ladder(1000) x 201 frequencies: 256 / 512 / 1024 threads = 62 / 43 / 35 us per sweep (one wave of workgroups)
  h->T = h->opt.threads > 0 ? h->opt.threads : (n <= 48 ? 64 : n <= 160 ? 128 : n <= 400 ? 256 : 1024);
  if (h->T > 1024 || (h->T & 63) || h->T < 64) { h->err = "threads must be a multiple of 64 in [64, 1024]"; return fail(SPICEY_ERR_BAD_DESC); }
  if (hipMalloc(&h->d_blob, h->hp.blob.size()) != hipSuccess ||
      hipMemcpy(h->d_blob, h->hp.blob.data(), h->hp.blob.size(), hipMemcpyHostToDevice) != hipSuccess) {
    h->err = "upload of the program failed";
    return fail(SPICEY_ERR_HIP);
  }
  h->dprog = h->hp.bind(h->d_blob);
  (void)hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, h->device);
  // resident sweep for batches that outnumber the CUs: needs the LDS workspace, 16-bit records, and every entry in the
  // NSE register slots of a thread
  h->Tres = std::min(h->T, 512);
  if (h->lds && P.has16 && P.nLU <= SPICEY_AC_NSE * h->Tres && (int)h->hp.ph_cnt.size() <= 254) {
    spicey_build_resident(h->hp, h->Tres, SPICEY_AC_RMAX, h->hres, 0, false);  // (the complex executor knows generic records only)
    if (hipMalloc(&h->d_res, h->hres.blob.size()) == hipSuccess &&
        hipMemcpy(h->d_res, h->hres.blob.data(), h->hres.blob.size(), hipMemcpyHostToDevice) == hipSuccess) {
      h->dres = h->hres.bind(h->d_res);
      h->resident_ok = true;
    }
  }
  const size_t ni = (size_t)h->n_inst;
  {
    std::vector<double> rinv(ni * (size_t)P.nR);
    for (size_t i = 0; i < rinv.size(); i++) rinv[i] = 1.0 / desc->R_val[i];
    if ((rc = ac_upload(h, &h->d_R, rinv.data(), rinv.size())) != SPICEY_OK) return fail(rc);
  }
  if ((rc = ac_upload(h, &h->d_C, desc->C_val, ni * P.nC)) != SPICEY_OK) return fail(rc);
  if ((rc = ac_upload(h, &h->d_L, desc->L_val, ni * P.nL)) != SPICEY_OK) return fail(rc);
  if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    h->err = "stream/event creation failed";
    return fail(SPICEY_ERR_HIP);
  }
  *out = h;
  return SPICEY_OK;
}

extern "C" int32_t spicey_ac_get_info(SpiceyAcHandle *h, SpiceyInfo *info) {
  if (!h || !info) return SPICEY_ERR_BAD_DESC;
  memset(info, 0, sizeof(*info));
  info->n_var = h->hp.hdr.n;
  info->nnz_a = h->hp.nnzA;
  info->nnz_lu = h->hp.hdr.nLU;
  info->n_levels = h->hp.hdr.nLevels;
  info->threads = h->T;
  info->inst_per_wg = 1;
  info->lds_bytes = h->lds ? (int32_t)h->lds_bytes : 0;
  info->n_cur = h->hp.hdr.nR + h->hp.hdr.nC + h->hp.hdr.nL + h->hp.hdr.nV;
  info->n_out = h->hp.hdr.nOut;
  info->interpreter = h->last_mode == 2 ? 2 : 1;  // 2 = the last run used the resident sweep
  info->resident_slots = h->resident_ok ? SPICEY_AC_RMAX : 0;
  info->resident_tasks = h->hres.resident_tasks;
  info->streamed_tasks = h->hres.streamed_tasks;
  info->wgs_per_inst = 1;
  info->program_bytes = (int64_t)h->hp.blob.size();
  info->tail_levels = (int32_t)h->last_dense;  // (AC handles: solves of the last run repeated with partial pivoting)
  return SPICEY_OK;
}

extern "C" double spicey_ac_last_kernel_ms(SpiceyAcHandle *h) { return h ? h->last_ms : 0.0; }

extern "C" int32_t spicey_ac_run(SpiceyAcHandle *h, int64_t n_freq, const double *freqs, const double *vph, double *out_v, double *out_i) {
  if (!h) return SPICEY_ERR_BAD_DESC;
  const SpiceyProg &P = h->hp.hdr;
  if (n_freq < 0 || (n_freq > 0 && (!freqs || !out_v)) || (P.nV > 0 && !vph)) { h->err = "bad run arguments"; return SPICEY_ERR_BAD_DESC; }
  if (n_freq == 0) return SPICEY_OK;
  if (h->hp.structurally_singular) { h->err = "Singular matrix (complex): structurally singular"; return SPICEY_ERR_SINGULAR; }
  const size_t slots = (size_t)h->n_inst * (size_t)n_freq;
  if (slots > 0x7fffffffull) { h->err = "n_inst * n_freq exceeds the grid limit"; return SPICEY_ERR_BAD_DESC; }
  ACCHK(h, hipSetDevice(h->device));
  const int nCur = P.nR + P.nC + P.nL + P.nV;
  double *d_f = nullptr, *d_ph = nullptr, *d_ov = nullptr, *d_oi = nullptr, *d_gW = nullptr;
  int32_t *d_status = nullptr;
  std::vector<int32_t> status(slots);
  int32_t rc = SPICEY_OK;
  auto body = [&]() -> int32_t {
    ACCHK(h, hipMalloc((void **)&d_f, (size_t)n_freq * sizeof(double)));
    ACCHK(h, hipMalloc((void **)&d_ph, std::max<size_t>(1, (size_t)h->n_inst * P.nV * 2) * sizeof(double)));
    ACCHK(h, hipMalloc((void **)&d_ov, std::max<size_t>(1, slots * (size_t)P.nOut * 2) * sizeof(double)));
    if (out_i) ACCHK(h, hipMalloc((void **)&d_oi, std::max<size_t>(1, slots * (size_t)nCur * 2) * sizeof(double)));
    ACCHK(h, hipMalloc((void **)&d_status, slots * sizeof(int32_t)));
    // global workspace: one slice per workgroup of a launch; sweeps whose slices would exceed 16 GiB run in chunks
    const size_t slice = (size_t)P.nW * sizeof(SpiceyCx);
    const size_t chunk = h->lds ? slots : std::min(slots, std::max<size_t>(1, ((size_t)16 << 30) / slice));
    if (!h->lds) ACCHK(h, hipMalloc((void **)&d_gW, chunk * slice));
    ACCHK(h, hipMemcpyAsync(d_f, freqs, (size_t)n_freq * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (P.nV > 0) ACCHK(h, hipMemcpyAsync(d_ph, vph, (size_t)h->n_inst * P.nV * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    SpiceyAcRun R{};
    R.R_inv = h->d_R; R.C_val = h->d_C; R.L_val = h->d_L;
    R.freqs = d_f; R.vph = d_ph; R.out_v = d_ov; R.out_i = d_oi; R.gW = d_gW; R.status = d_status;
    R.n_freq = n_freq; R.n_inst = h->n_inst;
    ACCHK(h, hipEventRecord(h->ev0, h->stream));
    // batches that outnumber the CUs (one workgroup per CU at this LDS size): persistent workgroups, ~2 per CU, each
    // keeping its share of the program in registers across its frequencies
    const bool resident = h->resident_ok && !(h->opt.debug & 16) && slots > (size_t)2 * (size_t)h->ncu;
    h->last_mode = resident ? 2 : 1;
    if (resident) {
      const int n_chunk = (int)std::min<int64_t>(n_freq, std::max<int64_t>(1, ((int64_t)2 * h->ncu + h->n_inst - 1) / h->n_inst));
      auto kern = spicey_ac_kernel_res<SPICEY_AC_RMAX, SPICEY_AC_NSE>;
      if (h->lds_bytes > 48 * 1024)
        ACCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
      R.slot_base = 0;
      hipLaunchKernelGGL(kern, dim3((unsigned)(h->n_inst * n_chunk)), dim3(h->Tres), h->lds_bytes, h->stream, h->dprog, h->dres, R, n_chunk);
      ACCHK(h, hipGetLastError());
    } else
    for (size_t base = 0; base < slots; base += chunk) {
      const unsigned grid = (unsigned)std::min(chunk, slots - base);
      R.slot_base = (int64_t)base;
      if (h->lds) {
        auto kern = spicey_ac_kernel<true>;
        if (h->lds_bytes > 48 * 1024)
          ACCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(h->T), h->lds_bytes, h->stream, h->dprog, R);
      } else {
        hipLaunchKernelGGL(spicey_ac_kernel<false>, dim3(grid), dim3(h->T), 0, h->stream, h->dprog, R);
      }
      ACCHK(h, hipGetLastError());
    }
    ACCHK(h, hipEventRecord(h->ev1, h->stream));
    ACCHK(h, hipMemcpyAsync(status.data(), d_status, slots * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    ACCHK(h, hipStreamSynchronize(h->stream));
    // Solves that tripped a pivot guard of the static order (a diagonal cancelling at a resonance) are repeated with
    // partial pivoting, dense, the way the reference solves every frequency; whatever fails there fails in the reference
    // too.  (diagnostics: SpiceyOptions.debug bit 7 = off; circuits beyond 4096 unknowns keep the error)
    h->last_dense = 0;
    if (!((h->opt.debug >> 7) & 1) && P.n <= 4096) {
      std::vector<int64_t> bad;
      for (size_t s2 = 0; s2 < slots; s2++)
        if (status[s2] != 0) bad.push_back((int64_t)s2);
      if (!bad.empty()) {
        const size_t per = (size_t)P.n * ((size_t)P.n + 1) * sizeof(SpiceyCx);
        const size_t nb = std::min(bad.size(), std::max<size_t>(1, ((size_t)1 << 30) / per));
        int64_t *d_slots = nullptr;
        SpiceyCx *d_A = nullptr, *d_Ws = nullptr;
        SpiceyProg *d_P = nullptr;
        SpiceyAcRun *d_R = nullptr;
        auto freed = [&]() {
          void *ps[] = {d_slots, d_A, d_Ws, d_P, d_R};
          for (void *q : ps)
            if (q) (void)hipFree(q);
        };
        if (hipMalloc((void **)&d_slots, nb * sizeof(int64_t)) != hipSuccess || hipMalloc((void **)&d_A, nb * per) != hipSuccess ||
            hipMalloc((void **)&d_Ws, nb * (size_t)P.nW * sizeof(SpiceyCx)) != hipSuccess || hipMalloc((void **)&d_P, sizeof(SpiceyProg)) != hipSuccess ||
            hipMalloc((void **)&d_R, sizeof(SpiceyAcRun)) != hipSuccess ||
            hipMemcpy(d_P, &h->dprog, sizeof(SpiceyProg), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_R, &R, sizeof(SpiceyAcRun), hipMemcpyHostToDevice) != hipSuccess) {
          freed();
          h->err = "allocation of the dense fallback workspace failed";
          return SPICEY_ERR_HIP;
        }
        const int Td = 1024;
        const size_t lds = ((size_t)Td + 2 * (size_t)P.n + 2) * sizeof(double) + ((size_t)Td + (size_t)P.n + 4) * sizeof(int32_t);
        hipError_t e2 = hipSuccess;
        if (lds > 48 * 1024)
          e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(spicey_ac_dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (size_t b0 = 0; b0 < bad.size() && e2 == hipSuccess; b0 += nb) {
          const size_t cnt = std::min(nb, bad.size() - b0);
          e2 = hipMemcpyAsync(d_slots, bad.data() + b0, cnt * sizeof(int64_t), hipMemcpyHostToDevice, h->stream);
          if (e2 != hipSuccess) break;
          hipLaunchKernelGGL(spicey_ac_dense_kernel, dim3((unsigned)cnt), dim3(Td), lds, h->stream, d_P, d_R, d_slots, d_Ws, d_A);
          e2 = hipGetLastError();
          if (e2 == hipSuccess) e2 = hipStreamSynchronize(h->stream);
        }
        freed();
        if (e2 != hipSuccess) { h->err = std::string("dense fallback: ") + hipGetErrorString(e2); return SPICEY_ERR_HIP; }
        h->last_dense = (int64_t)bad.size();
        ACCHK(h, hipMemcpyAsync(status.data(), d_status, slots * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
      }
    }
    ACCHK(h, hipMemcpyAsync(out_v, d_ov, slots * (size_t)P.nOut * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (out_i) ACCHK(h, hipMemcpyAsync(out_i, d_oi, slots * (size_t)nCur * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    ACCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_ms = ms;
    return SPICEY_OK;
  };
  rc = body();
  void *tmp[] = {d_f, d_ph, d_ov, d_oi, d_status, d_gW};
  for (void *p : tmp)
    if (p) (void)hipFree(p);
  if (rc != SPICEY_OK) return rc;
  // the reference stops at the first frequency that throws (simulateAC.ts:80-83): report the first failing slot
  for (size_t s = 0; s < slots; s++)
    if (status[s] != 0) {
      const bool sing = status[s] == 1;
      h->err = std::string(sing ? "Singular matrix (complex)" : "Complex divide by ~0") + " at inst " + std::to_string(s / (size_t)n_freq) +
               " frequency index " + std::to_string(s % (size_t)n_freq);
      return sing ? SPICEY_ERR_SINGULAR : SPICEY_ERR_COMPLEX_DIV;
    }
  return SPICEY_OK;
}
