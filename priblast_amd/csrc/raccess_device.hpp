// Device-side building blocks of the Raccess kernels (gfx950): the table-driven exp/log the
// reference computes with, the order-preserving logsumexp fold, and the loop energies.
//
// Numerics contract: every function performs the same IEEE operations in the
// same order as the reference (compiled without FMA contraction: this translation unit is
// built with -ffp-contract=off), so results are bit-identical to the strict CPU build.
//   expd  <-> fmath::expd(double)   fmath.hpp:439-466
//   logf  <-> fmath::log(float)     fmath.hpp:738-752
//   lse   <-> Raccess::logsumexp    raccess.cpp:414-419
//   loop_energy / hairpin_energy <-> raccess.cpp:773-832
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "energy.hpp"

namespace prb {

constexpr double kNegInf = -1000000.0; // exact sentinel "-INF" of raccess.cpp (energy_par.hpp:8)

// LDS image shared by the waves of a workgroup.
struct RaLds {
  uint64_t expd_tbl[2048];
  float log_tbl[4096];
  double small[RaSmallLayout::kCount];
  double expd_a, expd_ra;
  float c_log2;
  unsigned char bp_pair[32]; // [a*5+b]
};

struct RaConst { // global-memory copies, uploaded once per context
  const uint64_t *expd_tbl;
  const float *log_tbl;
  const double *small;
  const double *big;
  double expd_a, expd_ra;
  float c_log2;
  unsigned char bp_pair[25];
};

__device__ __forceinline__ void ra_load_lds(RaLds &s, const RaConst &c) {
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s.expd_tbl[i] = c.expd_tbl[i];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s.log_tbl[i] = c.log_tbl[i];
  for (int i = threadIdx.x; i < RaSmallLayout::kCount; i += blockDim.x) s.small[i] = c.small[i];
  if (threadIdx.x < 25) s.bp_pair[threadIdx.x] = c.bp_pair[threadIdx.x];
  if (threadIdx.x == 0) {
    s.expd_a = c.expd_a;
    s.expd_ra = c.expd_ra;
    s.c_log2 = c.c_log2;
  }
  __syncthreads();
}

__device__ __forceinline__ double ra_expd(const RaLds &s, double x) {
  if (x <= -708.39641853226408) return 0.0;
  if (x >= 709.78271289338397) return __builtin_huge_val();
  const double b = (double)(3ULL << 51);
  double d = x * s.expd_a + b;
  long long di = (long long)(int)(unsigned int)__double_as_longlong(d); // low word, sign-extended
  unsigned long long iax = s.expd_tbl[di & 2047];
  double t = (d - b) * s.expd_ra - x;
  unsigned long long u = (((unsigned long long)di + 2095104ULL) >> 11) << 52; // adj = 2^21 - 2^11
  double y = (3.0000000027955394 - t) * (t * t) * 0.16666666685227835064 - t + 1.0;
  return y * __longlong_as_double((long long)(u | iax));
}

__device__ __forceinline__ float ra_logf(const RaLds &s, float x) {
  unsigned int i = __float_as_uint(x);
  int a = (int)(i & 0x7F800000u);
  unsigned int b2 = i & 0xFFFu;
  unsigned int idx = (i >> 12) & 0x7FFu;
  float2 e = *reinterpret_cast<const float2 *>(&s.log_tbl[2 * idx]);
  return (float)(a - (127 << 23)) * s.c_log2 + e.x + (float)b2 * e.y;
}

__device__ __forceinline__ double ra_lse(const RaLds &s, double x, double y) {
  double hi = x > y ? x : y;
  double lo = x > y ? y : x;
  return hi + (double)ra_logf(s, (float)(ra_expd(s, lo - hi) + 1.0));
}

__device__ __forceinline__ int ra_bp(const RaLds &s, int a, int b) { return s.bp_pair[a * 5 + b]; }
__device__ __forceinline__ int ra_rtype(int t) { return t == 0 ? 0 : ((t - 1) ^ 1) + 1; } // energy_par.hpp:26

// Loop energy for the loop closed by pair `type` with inner pair `type2` (already rtype'd),
// u1/u2 unpaired bases on the two sides; b_i1 = s[i+1], b_j1 = s[j-1], b_p1 = s[p-1],
// b_q1 = s[q+1] in the reference's LoopEnergy(type, type2, i, j, p, q) convention.
__device__ __forceinline__ double ra_loop_energy(const RaLds &s, const double *__restrict__ big, int type,
                                                 int type2, int u1, int u2, int b_i1, int b_j1, int b_p1,
                                                 int b_q1) {
  using S = RaSmallLayout;
  using B = RaBigLayout;
  double z;
  if (u1 == 0 && u2 == 0) {
    z = s.small[S::kStack + type * 7 + type2];
  } else if (u1 == 0 || u2 == 0) {
    int u = u1 == 0 ? u2 : u1; // <= 30 on every path that reaches here (MAXLOOP window)
    z = s.small[S::kBulge + u];
    if (u == 1) {
      z += s.small[S::kStack + type * 7 + type2];
    } else {
      if (type > 2) z += s.small[S::kTermAU];
      if (type2 > 2) z += s.small[S::kTermAU];
    }
  } else if (u1 + u2 == 2) {
    z = big[B::kInt11 + ((type * 8 + type2) * 5 + b_i1) * 5 + b_j1];
  } else if (u1 == 1 && u2 == 2) {
    z = big[B::kInt21 + (((type * 8 + type2) * 5 + b_i1) * 5 + b_q1) * 5 + b_j1];
  } else if (u1 == 2 && u2 == 1) {
    z = big[B::kInt21 + (((type2 * 8 + type) * 5 + b_q1) * 5 + b_i1) * 5 + b_p1];
  } else if (u1 == 2 && u2 == 2) {
    z = big[B::kInt22 + ((((type * 8 + type2) * 5 + b_i1) * 5 + b_p1) * 5 + b_q1) * 5 + b_j1];
  } else {
    z = s.small[S::kInternal + u1 + u2] + s.small[S::kMismatchI + (type * 5 + b_i1) * 5 + b_j1] +
        s.small[S::kMismatchI + (type2 * 5 + b_q1) * 5 + b_p1];
    int du = u1 - u2;
    z += s.small[S::kNinio + (du < 0 ? -du : du)];
  }
  return z;
}

// The same loop energy without branches, for callers whose lanes look at loops of different classes (every
// class would otherwise be walked through by the whole wavefront, each with its own table latency): one
// look-up in the big table (1x1, 2x1, 1x2, 2x2 loops), four in the LDS table, and the class decides which of
// them are added - in the order of the branches above, so the sum is bit-identical.
__device__ __forceinline__ double ra_loop_energy_bf(const RaLds &s, const double *__restrict__ big, int type, int type2, int u1,
                                                    int u2, int b_i1, int b_j1, int b_p1, int b_q1) {
  using S = RaSmallLayout;
  using B = RaBigLayout;
  const bool z1 = u1 == 0, z2 = u2 == 0;
  const bool stack = z1 && z2, bulge = z1 != z2;
  const int u = z1 ? u2 : u1;
  const bool c11 = u1 == 1 && u2 == 1, c12 = u1 == 1 && u2 == 2, c21 = u1 == 2 && u2 == 1, c22 = u1 == 2 && u2 == 2;
  const bool special = c11 || c12 || c21 || c22;
  const bool generic = !stack && !bulge && !special;
  const int tt = type * 8 + type2;
  int ib = B::kInt11 + (tt * 5 + b_i1) * 5 + b_j1;
  if (c12) ib = B::kInt21 + ((tt * 5 + b_i1) * 5 + b_q1) * 5 + b_j1;
  if (c21) ib = B::kInt21 + (((type2 * 8 + type) * 5 + b_q1) * 5 + b_i1) * 5 + b_p1;
  if (c22) ib = B::kInt22 + (((tt * 5 + b_i1) * 5 + b_p1) * 5 + b_q1) * 5 + b_j1;
  const double vb = big[ib]; // (always a valid index: unconditional, so that a batch of terms has its look-ups in flight together)
  const int st = S::kStack + type * 7 + type2;
  const int i1 = stack ? st : bulge ? S::kBulge + u : S::kInternal + u1 + u2;
  const int i2 = bulge ? (u == 1 ? st : S::kTermAU) : S::kMismatchI + (type * 5 + b_i1) * 5 + b_j1;
  const int i3 = bulge ? S::kTermAU : S::kMismatchI + (type2 * 5 + b_q1) * 5 + b_p1;
  const int du = u1 - u2;
  const double t1 = s.small[i1], t2 = s.small[i2], t3 = s.small[i3], t4 = s.small[S::kNinio + (du < 0 ? -du : du)];
  const bool use2 = bulge ? (u == 1 || type > 2) : generic;
  const bool use3 = bulge ? (u != 1 && type2 > 2) : generic;
  double z = special ? vb : t1;
  if (use2) z += t2;
  if (use3) z += t3;
  if (generic) z += t4;
  return z;
}

// HairpinEnergy(type, i, j): d = j-i-1 unpaired, b_i1 = s[i+1], b_j1 = s[j-1].
__device__ __forceinline__ double ra_hairpin_energy(const RaLds &s, int type, int d, int b_i1, int b_j1) {
  using S = RaSmallLayout;
  double q = s.small[S::kHairpin + d];
  if (d != 3) {
    q += s.small[S::kMismatchH + (type * 5 + b_i1) * 5 + b_j1];
  } else if (type > 2) {
    q += s.small[S::kTermAU];
  }
  return q;
}

} // namespace prb
