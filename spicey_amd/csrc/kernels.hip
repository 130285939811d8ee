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

struct GpuExec {
  unsigned long long *prof;  // null unless phase profiling was requested
  __device__ __forceinline__ int threads() const { return (int)blockDim.x; }
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

template <int K, bool LDS>
__global__ void __launch_bounds__(1024) spicey_tran_kernel(SpiceyProg P, SpiceyRun R) {
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
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int in = wg * K + k;
    c.valid[k] = in < R.n_inst;
    c.inst[k] = in < R.n_inst ? in : R.n_inst - 1;
  }
  GpuExec ex{R.prof ? R.prof + (size_t)wg * SPICEY_PH_SLOTS : nullptr};
  spicey_tran_run<K>(ex, P, R, c, wg);
}

// v2: register-resident program (LDS workspace only; 16-bit records)
template <class Regs>
struct GpuExecV2 {
  unsigned long long *prof;
  Regs rr;
  __device__ __forceinline__ int threads() const { return (int)blockDim.x; }
  template <class R2>
  __device__ __forceinline__ R2 &regs(int) {
    return rr;
  }
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

template <int K, int RMAX, int NSV, int MAXT>
__global__ void __launch_bounds__(MAXT) spicey_tran_kernel_v2(SpiceyProg P, SpiceyResident Q, SpiceyRun R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WgCtx<K> c;
  const int wg = (int)blockIdx.x;
  const size_t nW = (size_t)P.nW * K, nU = (size_t)P.nU * K, nG = (size_t)P.nGdyn * K;
  c.W = (double *)smem;
  c.u = c.W + nW;
  c.gd = c.u + nU;
  c.ison = (int32_t *)(c.gd + nG);
  c.flags = c.ison + (size_t)P.nS * K;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int in = wg * K + k;
    c.valid[k] = in < R.n_inst;
    c.inst[k] = in < R.n_inst ? in : R.n_inst - 1;
  }
  GpuExecV2<ResRegs<K, RMAX, NSV>> ex;
  ex.prof = R.prof ? R.prof + (size_t)wg * SPICEY_PH_SLOTS : nullptr;
  spicey_tran_run_v2<K, RMAX, NSV>(ex, P, Q, R, c, wg);
}

template <int K, int RMAX, int NSV, int MAXT>
hipError_t launch_v2_t(const SpiceyProg &P, const SpiceyResident &Q, const SpiceyRun &R, int grid, int threads, size_t lds, hipStream_t st) {
  auto kern = spicey_tran_kernel_v2<K, RMAX, NSV, MAXT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, P, Q, R);
  return hipGetLastError();
}

template <int K, bool LDS>
hipError_t launch_t(const SpiceyProg &P, const SpiceyRun &R, int grid, int threads, size_t lds, hipStream_t st) {
  auto kern = spicey_tran_kernel<K, LDS>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, P, R);
  return hipGetLastError();
}

}  // namespace

size_t spicey_lds_bytes(const SpiceyProg &P, int K, bool lds) {
  if (!lds) return 64;
  size_t b = ((size_t)P.nW + P.nU + P.nGdyn) * K * sizeof(double) + ((size_t)P.nS * K + 4) * sizeof(int32_t);
  return (b + 15) & ~size_t(15);
}

size_t spicey_gw_doubles_per_wg(const SpiceyProg &P, int K) {
  return ((size_t)P.nW + P.nU + P.nGdyn) * K + (((size_t)P.nS * K + 1) >> 1);
}

hipError_t spicey_launch_tran(const SpiceyProg &P, const SpiceyRun &R, int K, bool lds, int grid, int threads, hipStream_t st) {
  const size_t bytes = spicey_lds_bytes(P, K, lds);
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

int spicey_v2_rmax(int threads) { return threads <= 512 ? 16 : 8; }

hipError_t spicey_launch_tran_v2(const SpiceyProg &P, const SpiceyResident &Q, const SpiceyRun &R, int K, int grid, int threads,
                                 hipStream_t st) {
  const size_t bytes = spicey_lds_bytes(P, K, true);
  if (threads <= 512) {
    switch (K) {
      case 1: return launch_v2_t<1, 16, 8, 512>(P, Q, R, grid, threads, bytes, st);
      case 2: return launch_v2_t<2, 16, 8, 512>(P, Q, R, grid, threads, bytes, st);
    }
  } else {
    switch (K) {
      case 1: return launch_v2_t<1, 8, 4, 1024>(P, Q, R, grid, threads, bytes, st);
      case 2: return launch_v2_t<2, 8, 4, 1024>(P, Q, R, grid, threads, bytes, st);
    }
  }
  return hipErrorInvalidValue;
}
