// Device helpers shared by the interaction-search kernels (search_kernels.hip, gapped_lds.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "search_kernels.hpp"

namespace prb {

#define US(x) ((int)(uint16_t)(x)) // Hit::GetQLength/GetDbLength are unsigned short (hit.hpp:62-64)

__device__ __forceinline__ int base_of(unsigned c) { return c <= 5 ? (int)c - 1 : (int)c - 5; } // codes 6..9 = soft-masked
__device__ __forceinline__ int get_char(const uint8_t *s, int64_t i) {                          // GetChar, gapped_extension.cpp:401-407
  if (i < 0) return 0;
  unsigned c = s[i];
  return c < 2 ? 0 : base_of(c);
}
__device__ __forceinline__ int bp_type(const SearchConst &sc, int a, int b) {
  return a == 0 ? 0 : (int)((sc.bp_rows >> (15 * (a - 1) + 3 * b)) & 7);
}
// GetBasePair's test (rna_interaction_search.cpp:371-385) for a position of the ungapped diagonal.
// The reference indexes BP_pair[5][5] with the raw codes minus 1, which for soft-masked codes
// (6..9, `db -r 1`) reads outside the table; here those are mapped to their bases like everywhere
// else in the extension code (ungapped_extension.cpp:68-69), see DESIGN.md section 2.
__device__ __forceinline__ bool diag_pairs(const SearchConst &sc, unsigned qc, unsigned dc) {
  return sc.bp_pair[base_of(qc) * 5 + base_of(dc)] != 0;
}
__device__ __forceinline__ int rtype_of(int t) { return t == 0 ? 0 : ((t - 1) ^ 1) + 1; } // energy_par.hpp:26
// z / 100.0 for an integer energy z (0.01 kcal/mol), correctly rounded without a division or
// a table look-up: one Newton step on z * fl(1/100) with fused multiply-adds.  Bit-identical to the
// IEEE quotient for every |z| <= 100000 (exhaustive check: tests/div100_check.c).
__device__ __forceinline__ double div100(const SearchConst &, int z) {
  const double zd = (double)z, r100 = 0.01;
  const double q0 = zd * r100;
  const double r = fma(-q0, 100.0, zd);
  return fma(r, r100, q0);
}


// GappedExtension::LoopEnergy, gapped_extension.cpp:426-473.  Every branch sums integers
// (units of 0.01 kcal/mol) and divides by 100.0 once.
__device__ __forceinline__ double loop_energy_gapped(const SearchConst &sc, int type, int type2, int i, int j, int p, int q,
                                                     const uint8_t *qs, const uint8_t *ds) {
  const int u1 = p - i - 1, u2 = q - j - 1;
  if (u1 == 0 && u2 == 0) return div100(sc, sc.stack37[type * 7 + type2]);
  if (u1 == 0 || u2 == 0) {
    const int u = u1 == 0 ? u2 : u1;
    if (u > 30) { // logarithmic extrapolation (:439); not reachable while -x <= 30
      double z = sc.bulge[u < 63 ? u : 63];
      if (type > 2) z += sc.terminal_au;
      if (type2 > 2) z += sc.terminal_au;
      return z / 100.0;
    }
    int z = (int)sc.bulge[u];
    if (u == 1) {
      z += sc.stack37[type * 7 + type2];
    } else {
      if (type > 2) z += sc.terminal_au;
      if (type2 > 2) z += sc.terminal_au;
    }
    return div100(sc, z);
  }
  const int a = base_of(qs[i + 1]), b = base_of(ds[j + 1]), c = base_of(qs[p - 1]), d = base_of(ds[q - 1]);
  int z;
  if (u1 + u2 == 2) z = sc.int11[((type * 8 + type2) * 5 + a) * 5 + b];
  else if (u1 == 1 && u2 == 2) z = sc.int21[(((type * 8 + type2) * 5 + a) * 5 + d) * 5 + b];
  else if (u1 == 2 && u2 == 1) z = sc.int21[(((type2 * 8 + type) * 5 + d) * 5 + a) * 5 + c];
  else if (u1 == 2 && u2 == 2) z = sc.int22[((((type * 8 + type2) * 5 + a) * 5 + c) * 5 + d) * 5 + b];
  else z = sc.internal37[u1 + u2] + sc.mismatchI37[(type * 5 + a) * 5 + b] + sc.mismatchI37[(type2 * 5 + d) * 5 + c];
  return div100(sc, z);
}

// The same loop energy from the loop sizes and the four neighbour bases
// a = q[i+1], b = db[j+1], c = q[p-1], d = db[q-1] (gapped_extension.cpp:449-456), written
// without branches: every loop class is a sum of at most three entries of one integer
// table (unused terms point at a zero entry), so the lanes of a group that look at loops of
// different classes stay on one instruction stream.  Requires loop sizes <= 30 (-x <= 30).
__device__ __forceinline__ double loop_energy_abcd(const SearchConst &sc, int type, int type2, int u1, int u2, int a, int b,
                                                   int c, int d) {
  using T = SearchTab;
  const int tt = type * 8 + type2;
  const int st = T::kStack + type * 7 + type2;
  const bool z1 = u1 == 0, z2 = u2 == 0;
  const int u = z1 ? u2 : u1;
  // interior classes
  const int i11 = T::kInt11 + (tt * 5 + a) * 5 + b;
  const int i21a = T::kInt21 + ((tt * 5 + a) * 5 + d) * 5 + b;
  const int i21b = T::kInt21 + (((type2 * 8 + type) * 5 + d) * 5 + a) * 5 + c;
  const int i22 = T::kInt22 + (((tt * 5 + a) * 5 + c) * 5 + d) * 5 + b;
  const bool c11 = u1 == 1 && u2 == 1, c12 = u1 == 1 && u2 == 2, c21 = u1 == 2 && u2 == 1, c22 = u1 == 2 && u2 == 2;
  const bool special = c11 || c12 || c21 || c22;
  int i1 = special ? (c11 ? i11 : c12 ? i21a : c21 ? i21b : i22) : T::kInternal + u1 + u2;
  int i2 = special ? T::kZero : T::kMismatchI + (type * 5 + a) * 5 + b;
  int i3 = special ? T::kZero : T::kMismatchI + (type2 * 5 + d) * 5 + c;
  // stack / bulge classes
  if (z1 || z2) {
    const bool stack = z1 && z2;
    i1 = stack ? st : T::kBulge + u;
    i2 = stack ? T::kZero : (u == 1 ? st : T::kTau + type);
    i3 = (stack || u == 1) ? T::kZero : T::kTau + type2;
  }
  return div100(sc, sc.tab[i1] + sc.tab[i2] + sc.tab[i3]);
}

// GetBPType, gapped_extension.cpp:321-338
__device__ __forceinline__ int ext_bp_type(const SearchConst &sc, int flag, const uint8_t *qs, const uint8_t *ds, int q_start,
                                           int64_t db_start, int i, int j, int x) {
  int qc, dc;
  if (flag == 0) {
    qc = get_char(qs, (int64_t)q_start - i - x);
    dc = get_char(ds, db_start - j - x);
  } else {
    qc = get_char(qs, (int64_t)q_start + i + x);
    dc = get_char(ds, db_start + j + x);
  }
  int t = bp_type(sc, qc, dc);
  if (flag == 1) t = rtype_of(t);
  return t;
}
__device__ __forceinline__ bool wobble(int t) { return t == 3 || t == 4; }

// CalcDangleEnergy, gapped_extension.cpp:366-399
__device__ __forceinline__ double dangle_energy_gapped(const SearchConst &sc, int q_pos, int64_t db_pos, int flag,
                                                       const uint8_t *qs, int qn, const uint8_t *ds, int64_t dn) {
  const int qc = get_char(qs, q_pos), dc = get_char(ds, db_pos);
  const int type = flag == 0 ? bp_type(sc, qc, dc) : bp_type(sc, dc, qc);
  const int q_length = qn - 1;
  int x = 0;
  if (type != 0) {
    if (flag == 0) {
      if (q_pos > 0) x += sc.dangle5[type * 5 + get_char(qs, q_pos - 1)];
      if (db_pos > 0 && ds[db_pos - 1] != 0) x += sc.dangle3[type * 5 + get_char(ds, db_pos - 1)];
      if ((db_pos == 0 || ds[db_pos - 1] == 0) && type > 2) x += sc.terminal_au;
    } else {
      if (db_pos < dn - 1 && ds[db_pos + 1] != 0) x += sc.dangle5[type * 5 + get_char(ds, db_pos + 1)];
      if (q_pos < q_length - 1) x += sc.dangle3[type * 5 + get_char(qs, q_pos + 1)];
      if ((db_pos == dn - 1 || ds[db_pos + 1] == 0) && type > 2) x += sc.terminal_au;
    }
  }
  return div100(sc, x);
}


struct HitState {
  int q_sp, db_sp, q_len, db_len, id_start;
  double e_tot, e_acc; // (hybridization energy = e_tot - e_acc)
};

} // namespace prb
