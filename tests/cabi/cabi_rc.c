/* tests/cabi/cabi_rc.c — the C-ABI of include/spicey_hip.h used from plain C (no Python, no C++): an RC low-pass
 * driven by a 1 V step, checked against the backward-Euler recurrence v[k] = (v[k-1] + a) / (1 + a), a = dt / (R C),
 * and an AC sweep of the same circuit against |H| = 1 / sqrt(1 + (w R C)^2).  Exit codes: 0 ok, 4 no GPU (what
 * spicey_create reports without a device), anything else = failure.  TEST INFRASTRUCTURE. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spicey_hip.h"

int main(void) {
  const double R = 1e3, Cc = 1e-6, dt = 1e-5;
  const int steps = 200;
  int32_t r_n1[1] = {1}, r_n2[1] = {2}, c_n1[1] = {2}, c_n2[1] = {0}, v_n1[1] = {1}, v_n2[1] = {0};
  double r_val[1] = {R}, c_val[1] = {Cc}, c_vprev[1] = {0.0};
  SpiceyDesc d;
  memset(&d, 0, sizeof d);
  d.abi_version = SPICEY_ABI_VERSION;
  d.n_nodes = 2; d.n_inst = 1;
  d.nR = 1; d.nC = 1; d.nV = 1;
  d.R_n1 = r_n1; d.R_n2 = r_n2; d.R_val = r_val;
  d.C_n1 = c_n1; d.C_n2 = c_n2; d.C_val = c_val; d.C_vprev = c_vprev;
  d.V_n1 = v_n1; d.V_n2 = v_n2;
  SpiceyOptions opt;
  memset(&opt, 0, sizeof opt);
  opt.want_currents = 1;
  SpiceyHandle *h = NULL;
  int32_t rc = spicey_create(&d, &opt, &h);
  if (rc == SPICEY_ERR_NO_DEVICE) { printf("no device: %s\n", spicey_last_error(NULL)); return 4; }
  if (rc != SPICEY_OK) { printf("create failed %d: %s\n", rc, spicey_last_error(NULL)); return 10; }
  double *src = malloc(sizeof(double) * (steps + 1)), *ov = malloc(sizeof(double) * (steps + 1) * 2), *oi = malloc(sizeof(double) * (steps + 1) * 3);
  int32_t *iters = malloc(sizeof(int32_t) * (steps + 1));
  for (int k = 0; k <= steps; k++) src[k] = 1.0;
  rc = spicey_run(h, steps, dt, src, ov, oi, iters);
  if (rc != SPICEY_OK) { printf("run failed %d: %s\n", rc, spicey_last_error(h)); return 11; }
  const double a = dt / (R * Cc);
  double v = 0.0, worst = 0.0;
  for (int k = 0; k <= steps; k++) {
    v = (v + a) / (1.0 + a);
    const double e = fabs(ov[2 * k + 1] - v) / (1e-9 * fabs(v) + 1e-12);
    if (e > worst) worst = e;
    if (fabs(ov[2 * k] - 1.0) > 1e-12 || iters[k] != 1) { printf("step %d: source node %g iters %d\n", k, ov[2 * k], iters[k]); return 12; }
    /* currents: R, C, V; the source delivers what the resistor carries */
    if (fabs(oi[3 * k] + oi[3 * k + 2]) > 1e-12 * fabs(oi[3 * k]) + 1e-18) { printf("step %d: KCL at the source\n", k); return 13; }
  }
  double cv = 0.0;
  if (spicey_get_state(h, &cv, NULL, NULL, NULL) != SPICEY_OK || fabs(cv - ov[2 * steps + 1]) > 0) { printf("state mismatch\n"); return 14; }
  if (spicey_last_solve_count(h) != steps + 1) { printf("solve count\n"); return 15; }
  spicey_destroy(h);
  if (worst > 1.0) { printf("transient parity %g x tolerance\n", worst); return 16; }

  SpiceyAcHandle *ah = NULL;
  rc = spicey_ac_create(&d, &opt, &ah);
  if (rc != SPICEY_OK) { printf("ac create failed %d: %s\n", rc, spicey_ac_last_error(NULL)); return 20; }
  double freqs[5] = {1.0, 50.0, 159.15494309189535, 1e3, 1e5}, vph[2] = {1.0, 0.0}, av[5 * 2 * 2], ai[5 * 3 * 2];
  rc = spicey_ac_run(ah, 5, freqs, vph, av, ai);
  if (rc != SPICEY_OK) { printf("ac run failed %d: %s\n", rc, spicey_ac_last_error(ah)); return 21; }
  double worst_ac = 0.0;
  for (int k = 0; k < 5; k++) {
    const double w = 2 * 3.141592653589793 * freqs[k], mag = hypot(av[(k * 2 + 1) * 2], av[(k * 2 + 1) * 2 + 1]);
    const double want = 1.0 / sqrt(1.0 + (w * R * Cc) * (w * R * Cc));
    const double e = fabs(mag - want) / (1e-9 * want + 1e-12);
    if (e > worst_ac) worst_ac = e;
  }
  spicey_ac_destroy(ah);
  if (worst_ac > 1.0) { printf("ac parity %g x tolerance\n", worst_ac); return 22; }
  char text[64];
  int32_t n = spicey_to_precision6(100000.5, text);
  text[n] = 0;
  if (strcmp(text, "100001") != 0) { printf("toPrecision tie: %s\n", text); return 30; }
  printf("cabi ok: transient %.3g x tol, ac %.3g x tol, %s\n", worst, worst_ac, spicey_version());
  free(src); free(ov); free(oi); free(iters);
  return 0;
}
