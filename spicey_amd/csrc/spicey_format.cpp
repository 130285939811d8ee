// spicey_format.cpp — native result formatting: formatTranResult straight from the typed
// result arrays.  Host code, no GPU involved.
//
// Replaces /root/reference/lib/formatting/formatTranResult.ts:1-23 for large runs: after the solve is fast, turning
// 10^7 doubles into `Number.prototype.toPrecision(6)` text dominates end-to-end time (9 s in numpy for the 10 001 x
// 1 001 table of BASELINE config 3; the kernel takes 0.12 s).  toPrecision(6) per ECMA-262 (Number.prototype.
// toPrecision, steps 10-13): n = the 6-digit integer closest to x / 10^(e-5), ties to the LARGER n (round half up on
// the exact binary value — C's printf rounds ties to even), exponential notation when e < -6 or e >= 6.
//
// Fast path: scale by a power of ten in x87 extended precision (64-bit significand) and round; whenever the scaled
// value is within 1e-5 of a rounding boundary (where the ~1e-12 scaling error could matter, and where exact ties
// live) the exact path takes over: 41 digits of glibc's exact "%.40e" expansion plus the tie rule.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spicey_hip.h"

namespace {

long double g_pow10[700];  // 10^(k - 350)
std::once_flag g_pow10_once;
void init_pow10() {
  std::call_once(g_pow10_once, [] {
    for (int k = 0; k < 700; k++) g_pow10[k] = powl(10.0L, (long double)(k - 350));
  });
}

// digits (6 significant, as an integer in [100000, 999999]) and decimal exponent e of |x| by the exact route
void exact_digits(double ax, int &n, int &e) {
  char buf[96];
  snprintf(buf, sizeof buf, "%.40e", ax);  // 41 significant digits of the exact binary value (glibc: correctly rounded)
  // No double lies within 1e-24 (relative) of a 6-digit midpoint without sitting on it (the midpoints are far too
  // sparse), so with 41 digits the 7th digit alone decides: '5' means "on the midpoint or above" -> the larger n
  // (ECMA-262: ties pick the larger n; printf would pick the even one).
  char *ep = strchr(buf, 'e');
  e = atoi(ep + 1);
  char dig[64];
  int nd = 0;
  for (char *p = buf; p < ep; p++)
    if (*p >= '0' && *p <= '9') dig[nd++] = *p;
  n = 0;
  for (int i = 0; i < 6; i++) n = n * 10 + (dig[i] - '0');
  if (dig[6] >= '5') n++;
  if (n == 1000000) { n = 100000; e++; }
}

// writes toPrecision(6) of x at dst (at most 32 bytes), returns the length
int to_precision6(double x, char *dst) {
  if (x != x) { memcpy(dst, "NaN", 3); return 3; }
  char *p = dst;
  if (x < 0) { *p++ = '-'; x = -x; }  // sign of -0 is dropped like in JS (x < 0 is false)
  if (x == INFINITY) { memcpy(p, "Infinity", 8); return (int)(p - dst) + 8; }
  if (x == 0) { memcpy(p, "0.00000", 7); return (int)(p - dst) + 7; }
  int e2;
  frexp(x, &e2);
  int e = (int)floor((e2 - 1) * 0.30102999566398119521);  // floor(log10(x)) or one less
  int n;
  long double s = (long double)x * g_pow10[350 + 5 - e];
  if (s >= 1000000.0L) { e++; s = (long double)x * g_pow10[350 + 5 - e]; }
  const long double fl = floorl(s);
  const long double fr = s - fl;
  if (e < -340 || e > 340 || fabsl(fr - 0.5L) < 1e-5L || s < 100000.0L || s >= 1000000.0L) {
    exact_digits(x, n, e);
  } else {
    n = (int)fl + (fr > 0.5L ? 1 : 0);
    if (n == 1000000) { n = 100000; e++; }
  }
  char d[6];
  for (int i = 5; i >= 0; i--) { d[i] = (char)('0' + n % 10); n /= 10; }
  if (e < -6 || e >= 6) {  // d.ddddde+x
    *p++ = d[0]; *p++ = '.';
    memcpy(p, d + 1, 5); p += 5;
    *p++ = 'e'; *p++ = e < 0 ? '-' : '+';
    int ae = e < 0 ? -e : e;
    char eb[8]; int ne = 0;
    do { eb[ne++] = (char)('0' + ae % 10); ae /= 10; } while (ae);
    while (ne) *p++ = eb[--ne];
  } else if (e >= 0) {  // e + 1 integer digits, 5 - e decimals
    memcpy(p, d, (size_t)e + 1); p += e + 1;
    if (e < 5) { *p++ = '.'; memcpy(p, d + e + 1, (size_t)(5 - e)); p += 5 - e; }
  } else {  // 0.000ddddd
    *p++ = '0'; *p++ = '.';
    for (int i = 0; i < -e - 1; i++) *p++ = '0';
    memcpy(p, d, 6); p += 6;
  }
  return (int)(p - dst);
}

}  // namespace

extern "C" int32_t spicey_to_precision6(double x, char *dst32) {
  init_pow10();
  return to_precision6(x, dst32);
}

// CSV body of formatTranResult: for every point k one line `t, v0, v1, ...` (", " separated, toPrecision(6)),
// lines joined by "\n" after `header` (no trailing newline).  Returns the number of bytes of the full text; writes
// it when out_cap is large enough (call once with out = NULL to size the buffer).
extern "C" int64_t spicey_format_tran(int64_t n_points, int32_t n_series, const double *times, const double *values, int64_t stride,
                                      const int32_t *cols, const char *header, char *out, int64_t out_cap) {
  init_pow10();
  if (n_points < 0 || n_series < 0 || !header || (n_points > 0 && !times) || (n_series > 0 && (!values || !cols))) return -1;
  const size_t hl = strlen(header);
  unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
  if (n_points * (int64_t)(n_series + 1) < 200000) nt = 1;
  nt = (unsigned)std::min<int64_t>(nt, std::max<int64_t>(1, n_points));
  std::vector<std::string> parts(nt);
  auto work = [&](unsigned t) {
    const int64_t k0 = n_points * t / nt, k1 = n_points * (t + 1) / nt;
    std::string &s = parts[t];
    s.reserve((size_t)(k1 - k0) * (size_t)(n_series + 1) * 11 + 16);
    char buf[40];
    for (int64_t k = k0; k < k1; k++) {
      s.push_back('\n');
      s.append(buf, (size_t)to_precision6(times[k], buf));
      const double *row = values + k * stride;
      for (int32_t j = 0; j < n_series; j++) {
        s.append(", ", 2);
        s.append(buf, (size_t)to_precision6(row[cols[j]], buf));
      }
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  int64_t total = (int64_t)hl;
  for (auto &s : parts) total += (int64_t)s.size();
  if (out && out_cap >= total) {
    memcpy(out, header, hl);
    char *p = out + hl;
    for (auto &s : parts) { memcpy(p, s.data(), s.size()); p += s.size(); }
  }
  return total;
}
