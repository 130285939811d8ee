// kernels.hip — gfx950 kernels of the transient solver.
//
// spicey_tran_kernel<K, LDS>: ONE persistent workgroup per K instances runs the complete transient
// (/root/reference/lib/analysis/simulateTRAN.ts:146-238) — time loop, switch iteration, stamping,
// sparse LU with fused forward elimination, backward substitution, recording and state update —
// with the matrix, right-hand side and element state resident in LDS (160 KB/CU on MI355X).
// HBM sees only the result stream (coalesced 8*n_out / 8*n_cur bytes per step) and the L2-resident
// program + per-instance static values.  The phase bodies live in tran_exec.h.
//
// LDS = false is the capacity fallback (circuits whose L+U does not fit 160 KB): same program, the
// workspace lives in global memory (L2).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "tran_exec.h"

namespace {

// one wave, `nlanes` lanes in lockstep through `nsteps` dependent steps (LDS operations of one wave execute in order, so
// step s + 1 sees what step s wrote without a workgroup barrier); the other waves wait at the closing barrier
template <class F>
__device__ __forceinline__ void gpu_wave_lockstep(int nlanes, int nsteps, F f) {
  int tid = (int)threadIdx.x;
  asm volatile("" : "+v"(tid));
  if (tid < 64) {
    for (int s = 0; s < nsteps; s++) {
      if (tid < nlanes) f(tid, s);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
}

// issue priorities of the phases of a kernel that shares its CU with a second workgroup (see GpuExecV2)
#define SPICEY_PRIO_TOP 3
#define SPICEY_PRIO_NARROW 2
#define SPICEY_PRIO_WIDE_U 1
template <class F>
__device__ __forceinline__ void gpu_wave_lockstep_keep(int nlanes, int nsteps, F f, bool prio = false) {
  int tid = (int)threadIdx.x;
  asm volatile("" : "+v"(tid));
  if (tid < 64) {
    // the one wave on the critical path of its workgroup: ahead of the co-resident workgroup's waves in the issue arbitration
    if (prio) __builtin_amdgcn_s_setprio(SPICEY_PRIO_TOP);
    double keep[4] = {0.0, 0.0, 0.0, 0.0};  // per-lane values that live in registers from step to step
    for (int s = 0; s < nsteps; s++) {
      if (tid < nlanes) f(tid, s, keep);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();
}

struct GpuExec {
  // (fronts_exec.h: a root front stays in LDS between the sweeps — group kernel only: in the one-workgroup kernels with an LDS
  // workspace the same code sends this hipcc into "Illegal instruction detected: Operand has incorrect register class")
  static constexpr bool keep_root = false;
  // profiling clock (100 MHz, the same counter on every CU) and an accumulator bump by one thread
  __device__ __forceinline__ unsigned long long ticks_now() const { return (unsigned long long)wall_clock64(); }
  __device__ __forceinline__ void add_ticks(unsigned long long *dst, unsigned long long t0) const {
    if (threadIdx.x == 0) *dst += (unsigned long long)wall_clock64() - t0;
  }

  unsigned long long *prof;  // null unless phase profiling was requested
  double *lds_;              // scratch for the dense fronts (null when the program has none)
  __device__ __forceinline__ int wg() const { return 0; }
  __device__ __forceinline__ double *lds() const { return lds_; }
  template <class F>
  __device__ __forceinline__ void wg_phase(F f) { phase(0, f); }
  template <class F>
  __device__ __forceinline__ void for_each_wg(F f) { f(0, 1); }
  template <class F>
  __device__ __forceinline__ void phase_marked(int slot, F f) { phase(slot, f); }
  template <class F>
  __device__ __forceinline__ void wave_lockstep(int nlanes, int nsteps, F f) { gpu_wave_lockstep(nlanes, nsteps, f); }
  __device__ __forceinline__ void front_post(unsigned int *, unsigned int) {}
  __device__ __forceinline__ void front_wait(unsigned int *, unsigned int) {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ int threads() const { return (int)blockDim.x; }
  __device__ __forceinline__ bool failed() const { return false; }
  __device__ __forceinline__ bool serial_chain() const { return false; }
  __device__ __forceinline__ int local_threads() const { return (int)blockDim.x; }
  __device__ __forceinline__ void sync() {}
  template <class F>
  __device__ __forceinline__ void local_phase(F f) { phase(0, f); }
  template <class F>
  __device__ __forceinline__ void phase(int tag, F f) {
    long long t0 = 0;
    if (prof && threadIdx.x == 0) t0 = clock64();
    // The thread id is made opaque per phase: otherwise hipcc hoists every `base + tid * 8` address (a VGPR
    // pair per array and instance) out of the time loop and the register-resident program spills.
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    f(tid);
    __syncthreads();
    if (prof && threadIdx.x == 0) prof[tag] += (unsigned long long)(clock64() - t0);
  }
};

template <int K, bool LDS, bool FRONTS>
__global__ void __launch_bounds__(FRONTS ? 512 : 1024) spicey_tran_kernel(SpiceyProg P, SpiceyRun R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WgCtx<K> c;
  const int wg = (int)blockIdx.x;
  const size_t nW = (size_t)P.nW * K, nU = (size_t)P.nU * K, nG = (size_t)P.nGdyn * K;
  if (LDS) {
    c.W = (double *)smem;
    c.u = c.W + nW;
    c.gd = c.u + nU;
    c.ison = (int32_t *)(c.gd + nG);
    c.flags = c.ison + (size_t)P.nS * K;
  } else {
    const size_t stride = nW + nU + nG + (((size_t)P.nS * K + 1) >> 1);
    c.W = R.gW + (size_t)wg * stride;
    c.u = c.W + nW;
    c.gd = c.u + nU;
    c.ison = (int32_t *)(c.gd + nG);
    c.flags = (int32_t *)smem;
  }
  c.tail = nullptr;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int in = wg * K + k;
    c.valid[k] = in < R.n_inst;
    c.inst[k] = in < R.n_inst ? in : R.n_inst - 1;
  }
  // (front scratch behind the workspace / the flags: offsets from `smem` keep the LDS address space)
  const size_t lds_off = LDS ? (((nW + nU + nG) * sizeof(double) + ((size_t)P.nS * K + 4) * sizeof(int32_t) + 15) & ~(size_t)15) : 64;
  GpuExec ex{R.prof ? R.prof + (size_t)wg * SPICEY_PH_SLOTS : nullptr, (double *)(smem + lds_off)};
  spicey_tran_run<K, FRONTS>(ex, P, R, c, wg);
}

// Group mode: G workgroups (G CUs) cooperate on ONE large instance whose workspace lives in global memory / L2.
// A phase is the same body run by G*blockDim threads, followed by a barrier across the G workgroups: every storing
// wave drains its stores, the workgroup barriers, lane 0 releases at agent scope and bumps a monotonic counter, polls
// it relaxed, then acquires at agent scope before the workgroup's threads read what the others wrote (cdna guide
// Guideline 16).  All G workgroups are co-resident by construction (the host launches at most one per CU) and every
// spin is bounded: on a timeout the abort word is set, every workgroup leaves, and the run reports an error.
struct GpuGroupExec {
  static constexpr bool keep_root = true;
  // profiling clock (100 MHz, the same counter on every CU) and an accumulator bump by one thread
  __device__ __forceinline__ unsigned long long ticks_now() const { return (unsigned long long)wall_clock64(); }
  __device__ __forceinline__ void add_ticks(unsigned long long *dst, unsigned long long t0) const {
    if (threadIdx.x == 0) *dst += (unsigned long long)wall_clock64() - t0;
  }

  int G, wgi;
  unsigned int *counter, *abortf;
  unsigned int epoch;
  bool bad;
  double *lds_;
  unsigned long long *prof;  // per-workgroup section timers (100 MHz wall ticks since the previous mark), or null
  unsigned long long last;
  __device__ __forceinline__ void mark(int slot) {
    if (prof && threadIdx.x == 0) {
      const unsigned long long now = (unsigned long long)wall_clock64();
      prof[slot] += now - last;
      last = now;
    }
  }
  __device__ __forceinline__ int wg() const { return wgi; }
  __device__ __forceinline__ double *lds() const { return lds_; }
  // work that is split over the workgroups of the group without a barrier between them: f(this workgroup, G)
  template <class F>
  __device__ __forceinline__ void for_each_wg(F f) { f(wgi, G); }
  // a phase of THIS workgroup alone (every workgroup of the group runs its own): workgroup barrier only
  template <class F>
  __device__ __forceinline__ void wg_phase(F f) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    f(tid);
    __syncthreads();
  }
  template <class F>
  __device__ __forceinline__ void wave_lockstep(int nlanes, int nsteps, F f) { gpu_wave_lockstep(nlanes, nsteps, f); }
  // the first workgroup whose wait runs out leaves a note for the host's error text: counter[2..7] = {kind (1 group barrier,
  // 2 front hand-over, 3 census barrier, 4 census does not add up), workgroup, barrier number / front flag index, value
  // waited for, value seen, XCD}; counter[8] counts waits that only a read-modify-write poll saw satisfied (below)
  __device__ __forceinline__ void note_timeout(unsigned int kind, unsigned int what, unsigned int want, unsigned int seen) {
    if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
        __hip_atomic_exchange(counter + 2, kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      counter[3] = (unsigned int)wgi; counter[4] = what; counter[5] = want; counter[6] = seen; counter[7] = my_x;
    }
  }
  // ONE lane waits until *word has reached `want` (wrap-safe signed distance).  Every wait is bounded in TIME: `limit` ticks
  // of the chip-wide 100 MHz counter (SpiceyRun::grp_timeout_ticks; the clock is looked at every 1024 polls, the first time
  // ~0.15 ms into the wait, which starts the measurement).  At the same cadence the word is also read by an atomic
  // read-modify-write (+0), which is served at the memory side and cannot return a stale cached line: should the plain
  // `sc1` poll ever fail to see a value that is there, the wait ends at the next such read instead of running into the
  // deadline, and counter[8] counts the case when a plain load issued after it still shows the old value
  // (spicey_group_stale_polls).  Returns false when the launch is aborting.
  unsigned long long limit;
  __device__ __forceinline__ bool spin_until(unsigned int *word, unsigned int want, unsigned int kind, unsigned int what) {
    unsigned int spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
      unsigned int v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(v - want) >= 0) return true;
      if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if ((++spins & 1023u) == 0u) {
        v = __hip_atomic_fetch_add(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - want) >= 0) {
          // (the value may simply have arrived between the plain load above and this read: that happens about once in a
          // thousand long waits and means nothing.  A plain load issued AFTER the read-modify-write has returned the value
          // and STILL showing the old one does: the load path serves a stale copy — counted)
          const unsigned int again = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(again - want) < 0) __hip_atomic_fetch_add(counter + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return true;
        }
        const unsigned long long now = (unsigned long long)wall_clock64();
        if (t0 == 0) t0 = now;
        else if (now - t0 > limit) {
          note_timeout(kind, what, want, v);
          __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return false;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  // what lane 0 learned about the abort word reaches every wave through one LDS word, so that `bad` — and with it every
  // branch on failed() — is uniform across the workgroup (each lane reading the abort word for itself could see the moment
  // it is raised differently)
  unsigned int *s_ab;
  __device__ __forceinline__ void publish_abort() {  // lane 0, before the workgroup barrier that ends a wait
    *s_ab = __hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // front hand-off between two workgroups of the group: the producer's stores are drained by every wave, the workgroup
  // meets, lane 0 releases at agent scope and publishes the solve number; the consumer polls relaxed, acquires, and
  // holds its workgroup's barrier until the invalidate has completed (MI355X_MICROARCH.md, valid forms)
  __device__ __forceinline__ void front_post(unsigned int *flag, unsigned int value) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __device__ __forceinline__ void front_wait(unsigned int *flag, unsigned int value) {
    if (threadIdx.x == 0) {
      (void)spin_until(flag, value, 2u, (unsigned int)(flag - front_flags0));
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish_abort();
    }
    __syncthreads();
    if (*s_ab != 0u) bad = true;
  }
  unsigned int *front_flags0;  // this group's first front flag (error text: which flag a hand-over waited for)
  __device__ __forceinline__ int threads() const { return G * (int)blockDim.x; }
  __device__ __forceinline__ bool failed() const { return bad; }
  __device__ __forceinline__ bool serial_chain() const { return G > 1; }
  __device__ __forceinline__ int local_threads() const { return (int)blockDim.x; }
  __device__ __forceinline__ void sync() { barrier(); }
  // a phase of workgroup 0 alone (the others fall through to the next group barrier): data it reads was published by
  // the last group barrier, data it writes is read by its own waves (one CU, one L1) until the next group barrier
  template <class F>
  __device__ __forceinline__ void local_phase(F f) {
    if (wgi == 0) {
      int tid = (int)threadIdx.x;
      asm volatile("" : "+v"(tid));
      f(tid);
      __syncthreads();
    }
  }
  // XCD-hierarchical form (MI355X_MICROARCH.md, price list row "barrier-xcd"): the L2 is shared inside an XCD, so ONE
  // write-back per XCD publishes the stores of all its workgroups.  Each workgroup drains its stores and arrives on its
  // XCD's counter; the last arriver of an XCD releases (buffer_wbl2) and arrives on the top counter; the last XCD
  // publishes the generation; everybody polls the generation, then acquires.  Which workgroups share an XCD is read
  // from HW_REG_XCC_ID by a census at kernel start (census()): nothing is assumed about placement.
  unsigned int *hb;       // this group's hierarchical-barrier words: [x * 16] members of XCD x, [128 + x * 16] arrivals, [256] top, [272] generation
  unsigned int hep, my_x, n_mine, n_xcd;
  __device__ __forceinline__ void census() {
    unsigned int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    my_x = x & 7u;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(hb + my_x * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flat_barrier();
    n_mine = 0; n_xcd = 0;
    unsigned int total = 0;
    for (unsigned int i = 0; i < 8; i++) {
      const unsigned int m = __hip_atomic_load(hb + i * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      n_xcd += m != 0u ? 1u : 0u;
      total += m;
      if (i == my_x) n_mine = m;
    }
    n_mine = (unsigned int)__builtin_amdgcn_readfirstlane((int)n_mine);
    n_xcd = (unsigned int)__builtin_amdgcn_readfirstlane((int)n_xcd);
    total = (unsigned int)__builtin_amdgcn_readfirstlane((int)total);
    hep = 0;
    // every later barrier counts on these numbers: a census that does not add up to G (a workgroup that read the member
    // counts before all of them had landed) would leave a barrier one arrival short for ever — refuse to run on it
    if (total != (unsigned int)G || n_mine == 0u) {
      if (threadIdx.x == 0 && !bad) {
        note_timeout(4u, total, (unsigned int)G, n_mine);
        __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      bad = true;
    }
  }
  __device__ __forceinline__ void barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    hep++;
    if (threadIdx.x == 0) {
      const unsigned int a = __hip_atomic_fetch_add(hb + 128 + my_x * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a + 1u == hep * n_mine) {  // last workgroup of this XCD: its write-back covers the whole XCD's L2
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(hb + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1u == hep * n_xcd) __hip_atomic_store(hb + 272, hep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      (void)spin_until(hb + 272, hep, 1u, hep);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish_abort();
    }
    __syncthreads();
    if (*s_ab != 0u) bad = true;
  }
  // the flat form (one counter, every workgroup releases): used once, for the census
  __device__ __forceinline__ void flat_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    epoch++;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      (void)spin_until(counter, epoch * (unsigned int)G, 3u, epoch);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish_abort();
    }
    __syncthreads();
    if (*s_ab != 0u) bad = true;
  }
  // thread ids of a group phase: wave v of workgroup g is wave v * G + g of the group, so that consecutive slices of a
  // task list (sorted longest first) go to different workgroups
  __device__ __forceinline__ int group_tid() const {
    return ((((int)threadIdx.x >> 6) * G + wgi) << 6) | ((int)threadIdx.x & 63);
  }
  template <class F>
  __device__ __forceinline__ void phase(int, F f) {
    int tid = group_tid();
    asm volatile("" : "+v"(tid));
    f(tid);
    barrier();
  }
  // the same with a section mark between this workgroup's share of the work and the group barrier (profiling)
  template <class F>
  __device__ __forceinline__ void phase_marked(int slot, F f) {
    int tid = group_tid();
    asm volatile("" : "+v"(tid));
    f(tid);
    if (prof) { __syncthreads(); mark(slot); }
    barrier();
  }
};

template <int K, bool FRONTS>
__global__ void __launch_bounds__(FRONTS ? 512 : 1024) spicey_tran_kernel_grp(SpiceyProg P, SpiceyRun R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // scratch of the dense fronts
  WgCtx<K> c;
  const int G = R.wgs_per_group;
  // Workgroups are dealt to the 8 XCDs round-robin by block index, and each XCD has its own L2.  The schedules (front tree by
  // proportional mapping, bins of the subtree-local levels) give neighbouring parts of the elimination tree to neighbouring
  // LOGICAL workgroups — so the logical index is laid out XCD-major: the G / 8 workgroups of one XCD are consecutive, a
  // subtree of up to G / 8 workgroups hands its fronts on inside one L2.  (Nothing else depends on the mapping: the barrier
  // takes its XCD membership from a census of HW_REG_XCC_ID.)
  const int grp = (int)blockIdx.x / G, pb = (int)blockIdx.x % G;
  const int wgi = (G & 7) == 0 ? (pb & 7) * (G >> 3) + (pb >> 3) : pb;
  const size_t nW = (size_t)P.nW * K, nU = (size_t)P.nU * K, nG = (size_t)P.nGdyn * K;
  const size_t stride = nW + nU + nG + (((size_t)P.nS * K + 1) >> 1);
  c.W = R.gW + (size_t)grp * stride;
  c.u = c.W + nW;
  c.gd = c.u + nU;
  c.ison = (int32_t *)(c.gd + nG);
  c.flags = R.grp_flags + (size_t)grp * 4;
  c.tail = nullptr;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int in = grp * K + k;
    c.valid[k] = in < R.n_inst;
    c.inst[k] = in < R.n_inst ? in : R.n_inst - 1;
  }
  unsigned int *gs = R.grp_sync + (size_t)grp * SPICEY_GRP_SYNC_WORDS;
  __shared__ unsigned int s_abort;  // lane 0's view of the abort word, for the whole workgroup (GpuGroupExec::publish_abort)
  GpuGroupExec ex;
  ex.G = G; ex.wgi = wgi;
  ex.counter = gs; ex.abortf = gs + 1;
  ex.epoch = 0u; ex.bad = false;
  ex.lds_ = (double *)smem;
  ex.prof = R.prof ? R.prof + ((size_t)grp * G + wgi) * SPICEY_PH_SLOTS : nullptr;  // (by logical workgroup)
  ex.last = (unsigned long long)wall_clock64();
  ex.limit = R.grp_timeout_ticks;
  ex.s_ab = &s_abort;
  ex.front_flags0 = R.front_flags ? R.front_flags + (size_t)grp * 2 * (size_t)P.nFronts : gs;
  ex.hb = gs + 16; ex.hep = 0u; ex.my_x = 0u; ex.n_mine = 1u; ex.n_xcd = 1u;
  if (threadIdx.x == 0) s_abort = 0u;
  __syncthreads();
  ex.census();
  if (R.force_abort && wgi == 0 && threadIdx.x == 0) __hip_atomic_store(gs + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (tests)
  spicey_tran_run<K, FRONTS>(ex, P, R, c, grp);
}

// v2: register-resident program (LDS workspace only; 16-bit records)
// PRIO: the kernel is built for two workgroups per CU (the packed geometry), which run different phases at the same moment.
// The issue arbitration then favours the phase that has less to hide its latency behind: 3 for the tridiagonal top (one
// wave, a dependent chain), 2 for the narrow levels (backward levels, factor levels from the third on: one short task per
// lane or none), 1 for the two wide factor levels, 0 for stamping and the update (wide, throughput-bound).  Measured on the
// headline batch: 3.91e7 -> 4.05e7 solves/s; a lone workgroup per CU gains nothing (9.86 -> 10.1 us), so it is off there.
template <class Regs, bool PRIO = false>
struct GpuExecV2 {
  unsigned long long *prof;  // LDS accumulators [SPICEY_PH_SLOTS] when profiling, else null
  long long t_last = 0;      // profiling: end of the previous phase (slot 7 accumulates the time BETWEEN phases)
  Regs rr;
  __device__ __forceinline__ int threads() const { return (int)blockDim.x; }
  // the argument structs live in global memory; a phase sees them through an address the compiler cannot trace back, so
  // the fields it uses are scalar-loaded inside the phase instead of being kept (and spilled) around the whole time loop
  // (returned BY VALUE: a copy through a constant-address-space pointer, so that the loads are scalar (s_load, scalar cache);
  // after inlining only the fields the phase touches are fetched, the rest of the copy is dead)
  template <class X>
  __device__ __forceinline__ X fresh(const X &x) const {
    typedef const X __attribute__((address_space(4))) *cptr;
    cptr p = (cptr)(&x);
    asm volatile("" : "+s"(p));
    X v;
    __builtin_memcpy(&v, p, sizeof(X));
    return v;
  }
  template <class R2>
  __device__ __forceinline__ R2 &regs(int) {
    return rr;
  }
  // tail: wave 0 runs `nlev` dependent levels back to back.  LDS operations of one wave execute in order, so a
  // level's ds_writes are seen by the next level's ds_reads without any workgroup barrier; the fences only stop
  // the compiler from moving or caching LDS accesses across levels.
  template <class F>
  __device__ __forceinline__ void wave_lockstep(int nlanes, int nsteps, F f) {
    long long t0 = 0;
    if (prof && threadIdx.x == 0) t0 = clock64();
    gpu_wave_lockstep(nlanes, nsteps, f);
    if (prof && threadIdx.x == 0) atomicAdd(&prof[SPICEY_PH_U0 + 31], (unsigned long long)(clock64() - t0));  // (the slot of the tail it replaces)
  }
  template <class F>
  __device__ __forceinline__ void wave_lockstep_keep(int nlanes, int nsteps, F f) {
    long long t0 = 0;
    if (prof && threadIdx.x == 0) { t0 = clock64(); if (t_last) atomicAdd(&prof[7], (unsigned long long)(t0 - t_last)); }
    gpu_wave_lockstep_keep(nlanes, nsteps, f, PRIO);
    if (prof && threadIdx.x == 0) { t_last = clock64(); atomicAdd(&prof[SPICEY_PH_U0 + 31], (unsigned long long)(t_last - t0)); }
  }
  template <class L, class F>
  __device__ __forceinline__ void tail_phase(int tag, int nlev, L load, F f) {
    long long t0 = 0;
    if (prof && threadIdx.x == 0) t0 = clock64();
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (tid < 64) {
      uint32_t cur[4], nxt[4] = {0u, 0u, 0u, 0u};
      load(tid, 0, cur);
      for (int l = 0; l < nlev; l++) {
        if (l + 1 < nlev) load(tid, l + 1, nxt);  // records are read-only: safe to fetch ahead of level l's writes
        f(tid, l, cur);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; cur[3] = nxt[3];
      }
    }
    __syncthreads();
    if (prof && threadIdx.x == 0) atomicAdd(&prof[tag], (unsigned long long)(clock64() - t0));
  }
  template <class F>
  __device__ __forceinline__ void phase(int tag, F f) {
    long long t0 = 0;
    if (prof && threadIdx.x == 0) { t0 = clock64(); if (t_last) atomicAdd(&prof[7], (unsigned long long)(t0 - t_last)); }
    // The thread id is made opaque per phase: otherwise hipcc hoists every `base + tid * 8` address (a VGPR
    // pair per array and instance) out of the time loop and the register-resident program spills.
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int pr = !PRIO ? 0 : tag >= SPICEY_PH_U0 + 2 ? SPICEY_PRIO_NARROW : tag >= SPICEY_PH_U0 ? SPICEY_PRIO_WIDE_U : 0;  // (uniform: an SGPR)
    if (pr == SPICEY_PRIO_NARROW) __builtin_amdgcn_s_setprio(SPICEY_PRIO_NARROW);
    if (pr == SPICEY_PRIO_WIDE_U) __builtin_amdgcn_s_setprio(SPICEY_PRIO_WIDE_U);
    f(tid);
    if (pr) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (prof && threadIdx.x == 0) { t_last = clock64(); atomicAdd(&prof[tag], (unsigned long long)(t_last - t0)); }  // LDS: no stall
  }
};

template <int K, int RMAX, int NSV, int NEL, int MAXT, int MINW, bool HYB = false>
__global__ void __launch_bounds__(MAXT, MINW) spicey_tran_kernel_v2(const SpiceyProg *__restrict__ Pg, const SpiceyResident *__restrict__ Qg,
                                                                    const SpiceyRun *__restrict__ Rg) {
  const SpiceyProg &P = *Pg;
  const SpiceyResident &Q = *Qg;
  const SpiceyRun &R = *Rg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WgCtx<K> c;
  const int wg = (int)blockIdx.x;
  // LDS: the workspace (hybrid layout: without the leaf-owned entries, which live in R.hyb_G, and without the element
  // vectors u / gd, which live in R.hyb_ug), then switch states, flags, profiling slots, tail / tridiagonal-top buffers
  const size_t nW = HYB ? (size_t)(P.nW - P.hyb_g0 - P.hyb_g2) * K : (size_t)P.nW * K;
  const size_t nU = HYB ? 0 : (size_t)P.nU * K, nG = HYB ? 0 : (size_t)P.nGdyn * K;
  c.W = (double *)smem;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int in = wg * K + k;
    c.valid[k] = in < R.n_inst;
    c.inst[k] = in < R.n_inst ? in : R.n_inst - 1;
  }
  if (HYB) {
    c.G = R.hyb_G + (size_t)c.inst[0] * (size_t)P.nLU;
    c.u = R.hyb_ug + (size_t)c.inst[0] * (size_t)(P.nU + P.nGdyn);
    c.gd = c.u + P.nU;
    c.ison = (int32_t *)(c.W + nW);
  } else {
    c.G = nullptr;
    c.u = c.W + nW;
    c.gd = c.u + nU;
    c.ison = (int32_t *)(c.gd + nG);
  }
  c.flags = c.ison + (size_t)P.nS * K;
  GpuExecV2<ResRegs<K, RMAX, NSV, NEL>, (MINW * 256 >= 2 * MAXT)> ex;  // two workgroups per CU
  // profiling accumulators live in LDS behind the flags (576 B, reserved by spicey_lds_bytes)
  // (offsets from `smem`, no integer casts: the pointers must keep their LDS address space, or every access
  // becomes a flat_load)
  const size_t off_prof = ((nW + nU + nG) * sizeof(double) + ((size_t)P.nS * K + 4) * sizeof(int32_t) + 15) & ~(size_t)15;
  unsigned long long *lprof = (unsigned long long *)(smem + off_prof);
  c.tail = (uint32_t *)(smem + off_prof + SPICEY_PH_SLOTS * sizeof(unsigned long long));
  ex.prof = R.prof ? lprof : nullptr;
#if SPICEY_EXP & 64
  c.zprof = R.prof ? lprof + 56 : nullptr;
#endif
  if (R.prof) {
    for (int i = (int)threadIdx.x; i < SPICEY_PH_SLOTS; i += (int)blockDim.x) lprof[i] = 0;
    __syncthreads();
    // start stamps wait in their own slots (not in registers: nothing may stay live around the whole run)
    if (threadIdx.x == 0) { lprof[5] = (unsigned long long)clock64(); lprof[6] = (unsigned long long)wall_clock64(); }
  }
  spicey_tran_run_v2<K, RMAX, NSV, NEL, HYB>(ex, P, Q, R, c, wg);
  if (R.prof) {
    __syncthreads();
    if (threadIdx.x == 0) {  // slots 5/6: whole-run shader cycles and 100 MHz wall ticks -> effective clock
      lprof[5] = (unsigned long long)clock64() - lprof[5];
      lprof[6] = (unsigned long long)wall_clock64() - lprof[6];
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < SPICEY_PH_SLOTS; i += (int)blockDim.x) R.prof[(size_t)wg * SPICEY_PH_SLOTS + i] = lprof[i];
  }
}

template <int K, int RMAX, int NSV, int NEL, int MAXT, int MINW, bool HYB = false>
hipError_t launch_v2_t(const SpiceyProg *P, const SpiceyResident *Q, const SpiceyRun *R, int grid, int threads, size_t lds, hipStream_t st) {
  auto kern = spicey_tran_kernel_v2<K, RMAX, NSV, NEL, MAXT, MINW, HYB>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, P, Q, R);
  return hipGetLastError();
}

template <int K, bool LDS, bool FRONTS = false>
hipError_t launch_t(const SpiceyProg &P, const SpiceyRun &R, int grid, int threads, size_t lds, hipStream_t st) {
  auto kern = spicey_tran_kernel<K, LDS, FRONTS>;
  if (FRONTS && threads > 512) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, P, R);
  return hipGetLastError();
}

}  // namespace

// LDS scratch of the dense fronts: the widest panel (U rows 16 x ld, L rows (Mp - 16) x 17, the 16 x 16 block of L and
// the reciprocal pivots), which also covers the backward solve's vectors
size_t spicey_front_lds_bytes(const SpiceyProg &P) {
  if (P.nFronts <= 0) return 0;
  return (size_t)SPICEY_FRONT_LDS_DOUBLES * sizeof(double);  // fronts of up to 128 rows live here whole; larger ones stage panels (max_front_mp <= 448)
}

size_t spicey_lds_bytes(const SpiceyProg &P, int K, bool lds, int tail_n) {
  if (!lds) return 64 + spicey_front_lds_bytes(P);
  size_t b = ((size_t)P.nW + P.nU + P.nGdyn) * K * sizeof(double) + ((size_t)P.nS * K + 4) * sizeof(int32_t);
  if (P.hybrid)  // hybrid workspace: leaf-owned entries and the element vectors are in global memory
    b = ((size_t)P.nW - P.hyb_g0 - P.hyb_g2) * K * sizeof(double) + ((size_t)P.nS * K + 4) * sizeof(int32_t);
  b = ((b + 15) & ~size_t(15)) + SPICEY_PH_SLOTS * sizeof(unsigned long long);  // + profiling accumulators
  if (P.pcr_n > 0 && tail_n < 5) tail_n = 5;                                         // tridiagonal top: two 2 KB row buffers + its index table
  b = ((b + 15) & ~size_t(15)) + (size_t)tail_n * 64 * 16;                          // + tail task records
  b += spicey_front_lds_bytes(P);                                                    // + dense-front scratch (32-bit interpreter only)
  return (b + 15) & ~size_t(15);
}

size_t spicey_gw_doubles_per_wg(const SpiceyProg &P, int K) {
  return ((size_t)P.nW + P.nU + P.nGdyn) * K + (((size_t)P.nS * K + 1) >> 1);
}

template <int K, bool FRONTS>
static hipError_t launch_grp_t(const SpiceyProg &P, const SpiceyRun &R, int grid, int threads, size_t lds, hipStream_t st, int *blocks_per_cu = nullptr) {
  auto kern = spicey_tran_kernel_grp<K, FRONTS>;
  if (FRONTS && threads > 512) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (blocks_per_cu)  // residency query only (spicey_grp_blocks_per_cu)
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(kern), threads, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, P, R);
  return hipGetLastError();
}

static hipError_t grp_dispatch(const SpiceyProg &P, const SpiceyRun &R, int K, int grid, int threads, hipStream_t st, int *blocks_per_cu) {
  const size_t lds = spicey_front_lds_bytes(P);
  if (P.nFronts > 0) return K == 1 ? launch_grp_t<1, true>(P, R, grid, threads, lds, st, blocks_per_cu) : hipErrorInvalidValue;
  switch (K) {
    case 1: return launch_grp_t<1, false>(P, R, grid, threads, 0, st, blocks_per_cu);
    case 2: return launch_grp_t<2, false>(P, R, grid, threads, 0, st, blocks_per_cu);
  }
  return hipErrorInvalidValue;
}

hipError_t spicey_launch_tran_grp(const SpiceyProg &P, const SpiceyRun &R, int K, int n_groups, int threads, hipStream_t st) {
  return grp_dispatch(P, R, K, n_groups * R.wgs_per_group, threads, st, nullptr);
}

// Workgroups of the group-mode kernel for this program that one CU can hold at `threads` threads (the runtime's answer for
// the very kernel and LDS size that will be launched); the host sizes a group so that grid * G workgroups fit the chip at
// ONE per CU and refuses to run when the answer is 0.
int spicey_grp_blocks_per_cu(const SpiceyProg &P, int K, int threads) {
  int nb = 0;
  SpiceyRun none{};
  if (grp_dispatch(P, none, K, 1, threads, nullptr, &nb) != hipSuccess) return 0;
  return nb;
}

hipError_t spicey_launch_tran(const SpiceyProg &P, const SpiceyRun &R, int K, bool lds, int grid, int threads, hipStream_t st) {
  const size_t bytes = spicey_lds_bytes(P, K, lds, 0);
  if (P.nFronts > 0) {
    if (K != 1) return hipErrorInvalidValue;
    return lds ? launch_t<1, true, true>(P, R, grid, threads, bytes, st) : launch_t<1, false, true>(P, R, grid, threads, bytes, st);
  }
  if (lds) {
    switch (K) {
      case 1: return launch_t<1, true>(P, R, grid, threads, bytes, st);
      case 2: return launch_t<2, true>(P, R, grid, threads, bytes, st);
      case 4: return launch_t<4, true>(P, R, grid, threads, bytes, st);
    }
  } else {
    switch (K) {
      case 1: return launch_t<1, false>(P, R, grid, threads, bytes, st);
      case 2: return launch_t<2, false>(P, R, grid, threads, bytes, st);
      case 4: return launch_t<4, false>(P, R, grid, threads, bytes, st);
    }
  }
  return hipErrorInvalidValue;
}

// v2 geometry per workgroup size.  MINW (waves per SIMD the register budget is cut for) is chosen so that
// NO variant spills: ROCm 7.2's hipcc places spill stores of values defined in divergent loops where EXEC can
// be zero (the store is lost and a later reload returns a previous kernel's scratch) — observed as stale vPrev
// registers; the Makefile therefore fails the build if any kernel reports a non-zero ScratchSize.
//   T <= 256 : 28 slots, 12 entries, 4 elements per thread (one wave per SIMD, 256 VGPRs, no spills to AGPRs)
//   (K = 2 interleaved instances: every variant tried — 16 to 32 slots — spilled 9 vector registers to AGPRs; the 16-bit
//   interpreter is therefore built for K = 1 only and interleaved instances run on interpreter 1)
//   T <= 512 : 16 slots,  8 entries, 2 elements per thread (<= 256 VGPRs)
//   T <= 1024:  8 slots,  4 entries, 1 element  per thread (<= 128 VGPRs)
// `packed` = the two-workgroups-per-CU geometry: 512 threads, <= 128 VGPRs; only 4 slots stay resident (the small,
// latency-critical phases), the wide bottom levels are streamed from L2 with the records prefetched in batches.
#define SPICEY_V2_RMAX256 28  // (32 slots spilled 6 vector registers to AGPRs: refused by check_no_spills.py)
// (hybrid workspace at 1024 threads: 4 slots — with 8 the build spills 10 registers at the 128-register cap)
int spicey_v2_rmax(int threads, bool packed, bool hybrid) { return packed ? 4 : (threads <= 256 ? SPICEY_V2_RMAX256 : (threads <= 512 ? 16 : (hybrid ? 4 : 8))); }
int spicey_v2_nsv(int threads, bool packed) { return packed ? 6 : (threads <= 256 ? 12 : (threads <= 512 ? 8 : 4)); }
int spicey_v2_nel(int threads, bool packed) { return packed ? 2 : (threads <= 256 ? 4 : (threads <= 512 ? 2 : 1)); }
int spicey_v2_max_threads(int K) { return K == 1 ? 1024 : 0; }

hipError_t spicey_launch_tran_v2(const SpiceyProg &Ph, const SpiceyResident &Qh, const SpiceyProg *P, const SpiceyResident *Q, const SpiceyRun *R, int K, int grid,
                                 int threads, hipStream_t st, bool packed) {
  const size_t bytes = spicey_lds_bytes(Ph, K, true, Qh.tail_n);
  if (packed) {
    if (K == 1 && threads == 512) return launch_v2_t<1, 4, 6, 2, 512, 4>(P, Q, R, grid, threads, bytes, st);
    return hipErrorInvalidValue;
  }
  if (Ph.hybrid) {
    // hybrid workspace: 1024 threads (4 slots, 4 entries, 1 element per thread, loads of the beyond-resident loops two at a
    // time: what fits 128 registers) or 512 threads (16 slots, 8 entries, 2 elements, four at a time)
    if (K == 1 && threads == 512) return launch_v2_t<1, 16, 8, 2, 512, 2, true>(P, Q, R, grid, threads, bytes, st);
    if (K == 1 && threads == 1024) return launch_v2_t<1, 4, 4, 1, 1024, 4, true>(P, Q, R, grid, threads, bytes, st);
    return hipErrorInvalidValue;
  }
  if (threads <= 256) {
    if (K == 1) return launch_v2_t<1, SPICEY_V2_RMAX256, 12, 4, 256, 1>(P, Q, R, grid, threads, bytes, st);
  } else if (threads <= 512) {
    if (K == 1) return launch_v2_t<1, 16, 8, 2, 512, 2>(P, Q, R, grid, threads, bytes, st);
  } else if (K == 1) {
    return launch_v2_t<1, 8, 4, 1, 1024, 4>(P, Q, R, grid, threads, bytes, st);
  }
  return hipErrorInvalidValue;
}
