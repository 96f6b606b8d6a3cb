/* oracle/oracle_fmath.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the two table-driven functions of herumi fmath that the reference's
 * Raccess uses: fmath::expd(double) (fmath.hpp:439-479) and fmath::log(float)
 * (fmath.hpp:738-752), with their tables (fmath.hpp:148-216).  Both are pure IEEE
 * arithmetic + one table lookup, so an implementation that performs the same operations
 * in the same order is bit-exact.
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

#define EXPD_SBIT 11
#define EXPD_N (1 << EXPD_SBIT)
#define LOG_LEN 11
#define LOG_N (1 << LOG_LEN)

static uint64_t g_expd_tbl[EXPD_N];
static double g_expd_a, g_expd_ra;
static float g_log_tbl[2 * LOG_N]; /* {app, rev} pairs */
static float g_c_log2;
static int g_ready = 0;

void orc_fmath_init(void) {
  if (g_ready) return;
  /* ExpdVar ctor, fmath.hpp:148-181 */
  g_expd_a = (double)EXPD_N / log(2.0);
  g_expd_ra = 1 / g_expd_a;
  for (int i = 0; i < EXPD_N; i++) {
    double v = pow(2.0, i * (1.0 / EXPD_N));
    uint64_t b;
    memcpy(&b, &v, 8);
    g_expd_tbl[i] = b & ((1ULL << 52) - 1);
  }
  /* LogVar ctor, fmath.hpp:183-216 */
  g_c_log2 = logf(2.0f) / (1 << 23);
  const double e = 1 / (double)(1 << 24);
  const double h = 1 / (double)(1 << LOG_LEN);
  for (int i = 0; i < LOG_N; i++) {
    double x = 1 + (double)i / LOG_N;
    double a = log(x);
    g_log_tbl[2 * i] = (float)a;
    if (i < LOG_N - 1) {
      double b = log(x + h - e);
      g_log_tbl[2 * i + 1] = (float)((b - a) / ((h - e) * (1 << 23)));
    } else {
      g_log_tbl[2 * i + 1] = (float)(1 / (x * (1 << 23)));
    }
  }
  g_ready = 1;
}

const uint64_t *orc_expd_table(void) { orc_fmath_init(); return g_expd_tbl; }
const float *orc_log_table(void) { orc_fmath_init(); return g_log_tbl; }
void orc_fmath_consts(double *a, double *ra, float *c_log2) {
  orc_fmath_init();
  *a = g_expd_a;
  *ra = g_expd_ra;
  *c_log2 = g_c_log2;
}

/* fmath.hpp:439-466 (the SSE2 branch, scalar double arithmetic) */
double orc_expd(double x) {
  if (x <= -708.39641853226408) return 0;
  if (x >= 709.78271289338397) return INFINITY;
  const double C1 = 1.0, C2 = 0.16666666685227835064, C3 = 3.0000000027955394;
  const double b = (double)(3ULL << 51);
  double d = x * g_expd_a + b;
  uint64_t dbits;
  memcpy(&dbits, &d, 8);
  /* low 32 bits taken as a signed int, then widened (fmath.hpp:449) */
  uint64_t di = (uint64_t)(int64_t)(int32_t)(uint32_t)dbits;
  uint64_t iax = g_expd_tbl[di & (EXPD_N - 1)];
  double t = (d - b) * g_expd_ra - x;
  const uint64_t adj = (1ULL << (EXPD_SBIT + 10)) - (1ULL << EXPD_SBIT);
  uint64_t u = ((di + adj) >> EXPD_SBIT) << 52;
  double y = (C3 - t) * (t * t) * C2 - t + C1;
  u |= iax;
  double did;
  memcpy(&did, &u, 8);
  return y * did;
}

/* fmath.hpp:738-752 */
float orc_logf(float x) {
  uint32_t i;
  memcpy(&i, &x, 4);
  int a = (int)(i & (0xFFu << 23));
  uint32_t b1 = i & (((1u << LOG_LEN) - 1) << (23 - LOG_LEN));
  uint32_t b2 = i & ((1u << (23 - LOG_LEN)) - 1);
  int idx = (int)(b1 >> (23 - LOG_LEN));
  float f = (float)(a - (127 << 23)) * g_c_log2 + g_log_tbl[2 * idx] +
            (float)b2 * g_log_tbl[2 * idx + 1];
  return f;
}
