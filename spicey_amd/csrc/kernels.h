// kernels.h — host-callable launchers of kernels.hip
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>

#include "program.h"

#define SPICEY_GRP_SYNC_WORDS 320  // uint32 words of barrier state per group: [0] flat counter, [1] abort, [2..7] timeout note, [8] stale polls, [16..] XCD census / arrivals / top / generation
#define SPICEY_LDS_MAX 163840  // 160 KiB per CU on MI355X (MI355X_MICROARCH.md "Chip-level parameters")

size_t spicey_lds_bytes(const SpiceyProg &P, int K, bool lds, int tail_n = 0);
size_t spicey_front_lds_bytes(const SpiceyProg &P);
size_t spicey_gw_doubles_per_wg(const SpiceyProg &P, int K);
hipError_t spicey_launch_tran(const SpiceyProg &P, const SpiceyRun &R, int K, bool lds, int grid, int threads, hipStream_t st);

// v2 (register-resident program): slots per thread for a workgroup size, and the launcher (K in {1, 2})
int spicey_v2_rmax(int threads, bool packed = false, bool hybrid = false);
int spicey_v2_nsv(int threads, bool packed = false);
int spicey_v2_nel(int threads, bool packed = false);
int spicey_v2_max_threads(int K);
// (Ph / Qh: host copies for the launch geometry; P / Q / R: the same structs in DEVICE memory — the kernel reads them by scalar loads)
hipError_t spicey_launch_tran_v2(const SpiceyProg &Ph, const SpiceyResident &Qh, const SpiceyProg *P, const SpiceyResident *Q, const SpiceyRun *R, int K, int grid,
                                 int threads, hipStream_t st, bool packed = false);

// group mode: R.wgs_per_group workgroups per K instances, workspace in global memory (large circuits)
hipError_t spicey_launch_tran_grp(const SpiceyProg &P, const SpiceyRun &R, int K, int n_groups, int threads, hipStream_t st);
int spicey_grp_blocks_per_cu(const SpiceyProg &P, int K, int threads);  // occupancy of that kernel (0: cannot run)
