/*
 * spicey_hip.h — C-ABI of the MI355X-native transient MNA solver that replaces the body of
 * spicey's simulateTRAN().
 *
 * Boundary: the reference has no plugin/FFI layer; the boundary is cut INSIDE
 *   simulateTRAN(ckt)            /root/reference/lib/analysis/simulateTRAN.ts:130-252
 * The host (TypeScript via bun:ffi, or the Python mirror in spicey_amd/) keeps parsing,
 * computeEffectiveTimeStep (:14-19), waveform pre-evaluation (:67, closures cannot cross FFI),
 * flattening ParsedCircuit (parseNetlist.ts:85-105) into the POD arrays below, result re-keying
 * and state write-back.  The native side runs the whole `for step … for iter …` nest
 * (:146-238): stamping (:25-102, lib/stamping/stamp{Admittance,Current,VoltageSource}Real.ts), the linear solve that replaces
 * solveReal (lib/math/solveReal.ts:3-73), the switch iteration (:108-128,:151-162), result
 * recording (:164-219) and the state update (:221-237).
 *
 * Everything is plain C: POD structs, raw pointers and sizes, int32 status codes; no torch or
 * C++ types.  A handle owns its device memory and is not thread-safe; distinct handles may be
 * used from distinct threads.  The library never calls abort()/exit().
 *
 * Launch admission (one rule per DEVICE, enforced inside the library).  A large instance may run on several cooperating
 * workgroups that wait for one another inside the kernel ("group mode", SpiceyInfo.wgs_per_inst > 1): such a launch only
 * makes progress while ALL its workgroups are resident, one per CU.  Therefore (a) a group is sized from the runtime's
 * occupancy answer for the very kernel and LDS size it launches, at most one workgroup per CU; (b) a group-mode launch
 * starts only after every transient launch this library has enqueued on that device before it has finished, and no
 * later transient launch of the library starts before the group-mode launch has finished — whatever handles, streams
 * and host threads they come from (stream-ordered event waits, nothing blocks on the host; launches that are not
 * group-mode stay concurrent with each other); (c) every cross-workgroup wait is bounded in time
 * (SpiceyOptions.group_timeout_ms): a launch whose wait runs out aborts as a whole with SPICEY_ERR_HIP and the waiter's
 * position in spicey_last_error() — it never hangs.  What the library cannot see are kernels of OTHER code on the same
 * device: keep long-running foreign kernels off the device while a group-mode run is in flight, or raise the timeout.
 *
 * Conventions
 *   node ids      0 = ground, 1..n_nodes = non-ground nodes (NodeIndex.ts:28-31: row = id-1)
 *   unknowns      x[0..n_nodes-1] node voltages, x[n_nodes+k] = branch current of source k
 *                 (parseNetlist.ts:455-460)
 *   instances     n_inst circuits sharing ONE topology (node ids, element order) with
 *                 per-instance element values and state: every `double` array below is
 *                 instance-major, [n_inst][n<kind>]
 *   outputs       step-major: out_v[inst][step][n_out], out_i[inst][step][n_cur] with
 *                 n_cur = nR+nC+nL+nV+nS+nD in the reference's recording order R,C,L,V,S,D
 *                 (simulateTRAN.ts:173-219)
 */
#ifndef SPICEY_HIP_H
#define SPICEY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPICEY_ABI_VERSION 2 /* 2: SpiceyOptions grew (group_retry, group_timeout_ms, diagnostics) */

/* status codes (SURVEY.md §8(b) "Errors") */
#define SPICEY_OK 0
#define SPICEY_ERR_SINGULAR 1 /* host maps to Error("Singular matrix (real)"), solveReal.ts:28 */
#define SPICEY_ERR_BAD_DESC 2
#define SPICEY_ERR_HIP 3
#define SPICEY_ERR_NO_DEVICE 4 /* the product path has no CPU fallback: no GPU -> this error */
#define SPICEY_ERR_COMPLEX_DIV 5 /* AC only: host maps to Error("Complex divide by ~0"), math/Complex.ts:40-42 */

/* Flat circuit descriptor: the ParsedCircuit of parseNetlist.ts:85-105 as SoA arrays.
 * All pointers are HOST pointers, read during spicey_create only. */
typedef struct SpiceyDesc {
  int32_t abi_version; /* SPICEY_ABI_VERSION */
  int32_t n_nodes;     /* ckt.nodes.count() - 1 */
  int32_t n_inst;      /* >= 1 */
  int32_t nR, nC, nL, nV, nS, nD;

  /* resistors  (ParsedResistor, parseNetlist.ts:12) */
  const int32_t *R_n1, *R_n2; /* [nR] */
  const double *R_val;        /* [n_inst][nR] ohms */
  /* capacitors (ParsedCapacitor :13-19); C_vprev = state entering the run (vPrev) */
  const int32_t *C_n1, *C_n2;
  const double *C_val, *C_vprev; /* [n_inst][nC] */
  /* inductors (ParsedInductor :20-26); L_iprev = iPrev */
  const int32_t *L_n1, *L_n2;
  const double *L_val, *L_iprev; /* [n_inst][nL] */
  /* independent voltage sources (ParsedVoltageSource :34-43); values come per step in src_table */
  const int32_t *V_n1, *V_n2; /* [nV] */
  /* voltage-controlled switches (ParsedSwitch :63-72 + ParsedVSwitchModel :45-51) */
  const int32_t *S_n1, *S_n2, *S_cp, *S_cn;          /* [nS] */
  const double *S_ron, *S_roff, *S_von, *S_voff;     /* [n_inst][nS] */
  const int32_t *S_ison;                             /* [n_inst][nS] 0/1, state entering the run */
  /* diodes (ParsedDiode :53-61 + ParsedDiodeModel :28-32); D_vdprev = vdPrev */
  const int32_t *D_np, *D_nm;                        /* [nD] */
  const double *D_is, *D_n, *D_vdprev;               /* [n_inst][nD] */

  /* recorded node voltages: out_nodes[n_out] are node ids (1-based); n_out = 0 / NULL -> all
   * nodes in id order (simulateTRAN.ts:164-171; .PRINT filtering :240-249 done before writing) */
  int32_t n_out;
  const int32_t *out_nodes;
} SpiceyDesc;

typedef struct SpiceyOptions {
  int32_t device;        /* HIP device ordinal */
  int32_t threads;       /* workgroup size, 0 = auto (a circuit that takes the hybrid workspace, SpiceyInfo.hybrid_entries > 0, runs on
                            1024 threads; 512 selects the 512-thread build of that kernel: the same results bit for bit) */
  int32_t inst_per_wg;   /* instances interleaved in one workgroup's LDS, 0 = auto */
  int32_t want_currents; /* 1: record element currents (out_i) */
  int32_t force_global;  /* 1: keep the LU workspace in HBM/L2 even if it fits LDS (testing) */
  int32_t profile;       /* 1: accumulate per-phase shader-clock cycles (spicey_debug_phase_cycles) */
  int32_t interpreter;   /* 0 auto; 1 = v1 (32-bit sliced task lists from L2); 2 = v2 (register-resident 16-bit records) */
  int32_t geometry;      /* v2 only. 0 auto; 1 = latency: one workgroup per CU, whole program in registers;
                            2 = throughput: two 512-thread workgroups per CU (<= 128 VGPRs, wide levels streamed) */
  int32_t debug;         /* diagnostics: bit 0 = no tail merge; bit 1 = refactor every step even for linear circuits;
                            bit 2 = plain CSR numbering of the L+U entries (no LDS-bank-aware slot-major numbering);
                            bit 3 = dense fronts above 64 rows take the staged (global-memory) path even if they fit LDS;
                            bit 4 = AC: never use the resident sweep (one workgroup per (instance, frequency) always);
                            bit 5 = no tridiagonal top (interpreter 2 keeps its task lists for the top levels of a chain);
                            bit 6 = no row records (streamed factor levels of a chain keep one task per target entry);
                            bit 7 = AC: no dense partial-pivoting fallback (a solve whose static pivot order hits a cancelled
                                    diagonal reports "Complex divide by ~0" / "Singular matrix (complex)" instead);
                            bits 8.. = extra empty phases per solve */
  int32_t wgs_per_inst;  /* global-workspace path: workgroups (CUs) cooperating on one instance; 0 auto, 1 = none */
  int32_t front_cut;     /* dense fronts (large instances): pivots of elimination-tree level >= front_cut are factored as
                            dense supernodal fronts (LDS-staged panels, MFMA trailing updates) instead of one
                            barrier-separated level per pivot.  0 = auto (large nonlinear circuits), -1 = never, > 0 = this level */
  int32_t group_retry;   /* group mode: 1 = a launch that ends in the bounded-wait abort is repeated ONCE from the state it
                            started with (spicey_sync); 0 (default) = the abort is reported as SPICEY_ERR_HIP */
  int32_t group_timeout_ms; /* group mode: longest single cross-workgroup wait before the launch aborts; 0 = 5000 */
  int32_t diagnostics;   /* bit 0: count the solves whose stamped matrix has a column with 0 < |a_ik| < 1e-15 max_j |a_jk| — the
                                   situation in which the reference's `if (Math.abs(f) < EPS) continue` (solveReal.ts:45) drops a
                                   row update that this library performs (spicey_last_skip_risk);
                            bit 1: record per step the one-shot linearisation error max_d |vd_new - vd_lin| over the diodes
                                   (spicey_get_lin_err); diagnostic only, never changes an iteration count */
} SpiceyOptions;

typedef struct SpiceyInfo {
  int32_t n_var;       /* n_nodes + nV */
  int32_t nnz_a;       /* structural nonzeros of A */
  int32_t nnz_lu;      /* nonzeros of L+U (incl. fill) under the chosen ordering */
  int32_t n_levels;    /* elimination-tree height = barrier-separated factor phases */
  int32_t threads;
  int32_t inst_per_wg;
  int32_t lds_bytes;   /* dynamic LDS per workgroup; 0 = global workspace */
  int32_t n_cur;       /* element-current columns */
  int32_t n_out;       /* recorded node-voltage columns */
  int32_t n_workgroups;
  int32_t interpreter;      /* 1 or 2, see SpiceyOptions */
  int32_t geometry;         /* 1 or 2 (v2), see SpiceyOptions */
  int32_t tail_levels;      /* v2: elimination-tree levels merged into the single-wave tail phase */
  int32_t wgs_per_inst;     /* workgroups cooperating on one instance (group mode), else 1 */
  int32_t resident_slots;   /* v2: 16-byte task records per thread kept in VGPRs */
  int64_t resident_tasks;   /* v2: factor/backward tasks held in registers */
  int64_t streamed_tasks;   /* v2: tasks still fetched from L2 every step */
  int64_t program_bytes;            /* device-side schedule ("program") size */
  int64_t algorithmic_bytes_solve;  /* SURVEY.md §8(d) formula */
  int32_t factor_reuse;     /* 1: no diodes / switches -> the factors of step 0 are reused, later steps solve only */
  int32_t n_fronts;         /* dense fronts of the upper elimination tree (0 = none) */
  int32_t front_cut;        /* first elimination-tree level handled by fronts (0 = none) */
  int32_t max_front;        /* rows of the largest front (padded to 16) */
  int64_t front_ws_bytes;   /* front workspace per instance */
  int32_t pcr_rows;         /* interpreter 2: rows of the tridiagonal top solved by one wave with parallel cyclic reduction (0 = none) */
  int32_t pcr_level;        /* first elimination-tree level of that top */
  int32_t hybrid_entries;   /* interpreter 2, hybrid workspace: entries of L+U (those the leaves of the elimination tree own) kept in
                               global memory because the whole L+U does not fit the LDS of one CU; 0 = everything in LDS */
} SpiceyInfo;

typedef struct SpiceyHandle SpiceyHandle;

/* Symbolic phase (MNA pattern, zero-free-diagonal row matching, nested-dissection ordering,
 * symbolic LU, level schedule) + upload.  Replaces the per-iteration dense allocation and
 * pivot search of simulateTRAN.ts:152-153 / solveReal.ts:15-34. */
int32_t spicey_create(const SpiceyDesc *desc, const SpiceyOptions *opt, SpiceyHandle **out);

/* One transient run of `steps`+1 points (step = 0..steps inclusive, t = step*dt;
 * simulateTRAN.ts:146-147) for all instances, continuing from the handle's current state
 * (a second run continues like the reference does).
 *   src_table   [steps+1][nV] source values at t = step*dt, shared by all instances
 *   out_v       [n_inst][steps+1][n_out]
 *   out_i       [n_inst][steps+1][n_cur] or NULL
 *   iters       [n_inst][steps+1] iterations executed per step (1..20) or NULL
 * HOST buffers; blocking. */
int32_t spicey_run(SpiceyHandle *h, int64_t steps, double dt, const double *src_table,
                   double *out_v, double *out_i, int32_t *iters);

/* Same with DEVICE buffers, enqueued on `stream` (a hipStream_t, NULL = default stream) without
 * synchronising: the error word is checked by spicey_sync(). */
int32_t spicey_run_device(SpiceyHandle *h, int64_t steps, double dt, const double *d_src_table,
                          double *d_out_v, double *d_out_i, int32_t *d_iters, void *stream);
/* Wait for enqueued runs; returns SPICEY_ERR_SINGULAR etc. like spicey_run.
 * Group mode (several workgroups per instance): every cross-workgroup wait is bounded in time; a launch whose wait runs
 * out aborts as a whole (nothing of it is kept): SPICEY_ERR_HIP with the first waiter's position (which wait, which
 * workgroup on which XCD, the value it waited for and the value it saw) in spicey_last_error() and on stderr.  Only with
 * SpiceyOptions.group_retry = 1 is the launch repeated ONCE from the state it started with, on the same stream; the
 * text of the aborted attempt then stays in spicey_last_error() (prefixed "recovered: ") although the call returns OK. */
int32_t spicey_sync(SpiceyHandle *h);
/* Number of launches this handle has repeated that way (0 in a healthy run; each is also reported on stderr). */
int32_t spicey_group_retries(const SpiceyHandle *h);
/* Group mode, summed over this handle's launches: waits that only the read-modify-write poll saw satisfied, i.e. where the
 * plain sc1 load poll kept returning an older value (kernels.hip, spin_until).  0 in a healthy run. */
int64_t spicey_group_stale_polls(const SpiceyHandle *h);

/* Final state after the last run (write-back to ckt: simulateTRAN.ts:221-237,122-124).
 * Any pointer may be NULL.  Arrays are [n_inst][n<kind>]. */
int32_t spicey_get_state(SpiceyHandle *h, double *C_vprev, double *L_iprev, double *D_vdprev,
                         int32_t *S_ison);

/* State entering the NEXT run (the counterpart of spicey_get_state; the reference keeps this state on the caller's
 * `ckt`, parseNetlist.ts:316,327,439,422, and a caller may rewrite it between two simulateTRAN calls).  HOST arrays
 * [n_inst][n<kind>]; a NULL pointer leaves that kind as it is.  Blocking. */
int32_t spicey_set_state(SpiceyHandle *h, const double *C_vprev, const double *L_iprev, const double *D_vdprev,
                         const int32_t *S_ison);
/* Back to the state the descriptor of spicey_create carried (kept in device memory): enqueued on `stream` (a
 * hipStream_t, NULL = default stream) as device-to-device copies, no synchronisation — every spicey_run_device after
 * it repeats the same transient instead of continuing the previous one. */
int32_t spicey_reset_state(SpiceyHandle *h, void *stream);

/* Total solves (= sum of iterations) executed by the last run, all instances. */
int64_t spicey_last_solve_count(SpiceyHandle *h);
/* Diagnostics (SpiceyOptions.diagnostics bit 0).  The reference eliminates with partial pivoting and skips a row update whose
 * multiplier a_ik / pivot is below 1e-15 (`if (Math.abs(f) < EPS) continue`, solveReal.ts:45): a nonzero coupling dropped —
 * floor conductances (diode gd 1e-12 S, switch 1/Roff) next to a clamped diode, a milliohm resistor or a large C/dt.  A sparse
 * static pivot order cannot reproduce that entry for entry; this library PERFORMS those updates (the physically consistent
 * answer) and says when the situation occurs: the number of (solve, column) pairs of the last run in which the stamped
 * matrix column held a nonzero entry below 1e-15 x the column's largest magnitude (per instance in per_inst[n_inst] if not
 * NULL; the return value is the sum; -1 without the option).  0 means the reference took no such shortcut on the stamped
 * matrix and the two results agree to the 1e-9 parity bar; > 0 means the reference's own result may differ from this one by
 * up to |v_k| * |a_ik| / a_ii per flagged coupling ("Where the reference skips row updates"). */
int64_t spicey_last_skip_risk(SpiceyHandle *h, int64_t *per_inst);
/* Diagnostics (SpiceyOptions.diagnostics bit 1): out[n_inst][steps+1] = per step the largest |vd(x) - vd_lin| over the diodes,
 * vd_lin being the junction voltage the step's LAST solve was linearised at (vdPrev on iteration 0, the previous iterate
 * afterwards: simulateTRAN.ts:81-85).  The reference iterates only on switch flips and never looks at this quantity; it is
 * reported, not acted on: iteration counts and results are identical with and without the option. */
int32_t spicey_get_lin_err(SpiceyHandle *h, double *out);
/* Duration in ms of the last run's kernel, measured with HIP events on the launch stream. */
double spicey_last_kernel_ms(SpiceyHandle *h);

int32_t spicey_get_info(SpiceyHandle *h, SpiceyInfo *info);
/* Human-readable description of the last error ("singular at inst 0 step 3 iter 0", the
 * hipGetErrorString text, …).  Valid until the next call on the handle. NULL handle -> global. */
const char *spicey_last_error(SpiceyHandle *h);
void spicey_destroy(SpiceyHandle *h);

/* Diagnostics: shader-clock cycles spent per phase kind by workgroup 0 during the last run (needs
 * SpiceyOptions.profile = 1).  out[72]: [0] prologue, [1] B stamp+rhs, [2] S switches, [3] A re-linearise,
 * [4] Z record/next-eval, [8+l] factor level l, [40+l] backward level l.  Returns the slot count. */
int32_t spicey_debug_phase_cycles(SpiceyHandle *h, uint64_t *out, int32_t n);
/* The same slots of launched workgroup `wg` (group mode: wg = group * wgs_per_inst + index).  In group mode the slots hold
 * 100 MHz wall-clock ticks per SECTION of the step: [1] B, [8] factor levels below the front cut, [9] fronts forward,
 * [10] fronts backward, [11] publish + group barrier, [12..20] inside the fronts, [40] backward levels, [4] Z. */
int32_t spicey_debug_phase_cycles_wg(SpiceyHandle *h, int32_t wg, uint64_t *out, int32_t n);
/* Diagnostics (SpiceyOptions.profile, circuits with dense fronts): per front of group `grp` four event times of the last
 * run — forward: children assembled / done, backward: parent's unknowns there / done — in 100 MHz ticks since the owning
 * workgroup entered the forward sweep, SUMMED over the solves (out[f*4+e]); meta[f*4+{0,1,2,3}] = pivots, boundary rows,
 * parent front, owning workgroup.  Returns the number of fronts (0: nothing recorded / cap_fronts too small). */
int32_t spicey_debug_front_ticks(SpiceyHandle *h, int32_t grp, uint64_t *out, int32_t *meta, int32_t cap_fronts);

/* ---------------------------------------------------------------------------------------------------------------
 * Several devices behind one handle (SURVEY.md §8(b) "device ordinal(s)", §8(e)): the n_inst instances of the descriptor
 * are block-partitioned over the listed devices (instance i -> devices[floor(i * n_dev / n_inst)], the partition of
 * spicey_amd/dist.py), one SpiceyHandle + stream per device inside THIS process; spicey_run_multi launches every shard
 * from its own host thread and each shard's results land directly in its slice of the caller's single host buffers
 * (that is the gather).  No data-path exchange between devices: instances are independent (simulateTRAN.ts:130 is one
 * circuit, one thread).  A device may be listed more than once (it then gets several shards; shards in group mode then
 * run one after the other on it — "Launch admission" at the top of this file).  opt->device is ignored.
 *   devices   [n_dev] HIP device ordinals, n_dev >= 1; n_dev > n_inst leaves the surplus devices idle
 * Errors: the first failing shard's status; spicey_multi_last_error names the device. */
typedef struct SpiceyMulti SpiceyMulti;
int32_t spicey_create_multi(const SpiceyDesc *desc, const SpiceyOptions *opt, const int32_t *devices, int32_t n_dev, SpiceyMulti **out);
/* Same buffers as spicey_run, for ALL instances: out_v [n_inst][steps+1][n_out], out_i, iters likewise or NULL. Blocking. */
int32_t spicey_run_multi(SpiceyMulti *m, int64_t steps, double dt, const double *src_table, double *out_v, double *out_i, int32_t *iters);
int32_t spicey_get_state_multi(SpiceyMulti *m, double *C_vprev, double *L_iprev, double *D_vdprev, int32_t *S_ison);
/* Shard `shard` (0 .. n_shards-1): its SpiceyInfo, first instance and instance count; returns SPICEY_ERR_BAD_DESC past the end. */
int32_t spicey_multi_get_shard(SpiceyMulti *m, int32_t shard, SpiceyInfo *info, int32_t *device, int32_t *first_inst, int32_t *n_inst);
int64_t spicey_multi_last_solve_count(SpiceyMulti *m);  /* all shards */
int32_t spicey_multi_group_retries(SpiceyMulti *m);      /* spicey_group_retries summed over the shards */
int64_t spicey_multi_group_stale_polls(SpiceyMulti *m);  /* spicey_group_stale_polls summed over the shards */
double spicey_multi_last_kernel_ms(SpiceyMulti *m);     /* the slowest shard's kernel */
const char *spicey_multi_last_error(SpiceyMulti *m);    /* NULL handle -> the calling thread's last failed create */
void spicey_destroy_multi(SpiceyMulti *m);

/* Library build info: "spicey_hip <abi> gfx950 …" */
const char *spicey_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * AC sweep: replaces the per-frequency body of
 *   simulateAC(ckt)              /root/reference/lib/analysis/simulateAC.ts:64-130
 * i.e. buildLinearSystemForAC (:25-62, lib/stamping/stamp{Admittance,VoltageSource}Complex.ts), solveComplex
 * (lib/math/solveComplex.ts:4-73) and the recording (:84-126), for every (instance, frequency) pair in one launch.
 * The host keeps the frequency list (buildFrequencyArray :9-23, utils/logspace.ts — Math.pow is engine-defined), the
 * source phasors (Complex.fromPolar, math/Complex.ts:16-19) and the `R <name> must be > 0` check (:39).
 * Diodes and switches of the descriptor are ignored, like the reference's AC analysis ignores them.
 *   freqs   [n_freq] Hz
 *   vph     [n_inst][nV][2] source phasors (re, im)
 *   out_v   [n_inst][n_freq][n_out][2]   complex node voltages (re, im)
 *   out_i   [n_inst][n_freq][nR+nC+nL+nV][2] complex currents in the reference's recording order R, C, L, V, or NULL
 * HOST buffers; blocking.  Status: SPICEY_ERR_SINGULAR -> Error("Singular matrix (complex)") (solveComplex.ts:28),
 * SPICEY_ERR_COMPLEX_DIV -> Error("Complex divide by ~0") (a pivot with |z|^2 < 1e-15, Complex.ts:40-42); the message of
 * spicey_ac_last_error names the first failing (instance, frequency), the one at which the reference would throw. */
typedef struct SpiceyAcHandle SpiceyAcHandle;
int32_t spicey_ac_create(const SpiceyDesc *desc, const SpiceyOptions *opt /* device, threads, force_global */, SpiceyAcHandle **out);
int32_t spicey_ac_run(SpiceyAcHandle *h, int64_t n_freq, const double *freqs, const double *vph, double *out_v, double *out_i);
int32_t spicey_ac_get_info(SpiceyAcHandle *h, SpiceyInfo *info);
double spicey_ac_last_kernel_ms(SpiceyAcHandle *h);
const char *spicey_ac_last_error(SpiceyAcHandle *h);
void spicey_ac_destroy(SpiceyAcHandle *h);

/* ---------------------------------------------------------------------------------------------------------------
 * Result formatting fast path (host code, no GPU): the CSV text of
 *   formatTranResult(tran)       /root/reference/lib/formatting/formatTranResult.ts:1-23
 * straight from the typed arrays spicey_run filled.  Every number is Number.prototype.toPrecision(6) exactly as
 * ECMA-262 defines it (ties to the larger digit string, exponential notation for e < -6 or e >= 6).
 *   times    [n_points]
 *   values   [n_points][stride]; series j is column cols[j] (the caller applies the JS key order / probe filter)
 *   header   first line, e.g. "t(s), 1:V, 2:V"
 * Returns the byte length of the text (lines joined by "\n", no trailing newline) and writes it when out_cap
 * suffices (call with out = NULL to size the buffer); -1 on bad arguments. */
int64_t spicey_format_tran(int64_t n_points, int32_t n_series, const double *times, const double *values, int64_t stride,
                           const int32_t *cols, const char *header, char *out, int64_t out_cap);
/* toPrecision(6) of one double into dst (>= 32 bytes, not NUL-terminated); returns the length. */
int32_t spicey_to_precision6(double x, char *dst32);

#ifdef __cplusplus
}
#endif
#endif /* SPICEY_HIP_H */
