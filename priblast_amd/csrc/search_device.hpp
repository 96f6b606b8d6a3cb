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
__device__ __forceinline__ int bp_type(const SearchConst &sc, int a, int b) { return sc.bp_pair[a * 5 + b]; }


// GappedExtension::LoopEnergy, gapped_extension.cpp:426-473
__device__ __forceinline__ double loop_energy_gapped(const SearchConst &sc, int type, int type2, int i, int j, int p, int q,
                                                     const uint8_t *qs, const uint8_t *ds) {
  const int u1 = p - i - 1, u2 = q - j - 1;
  if (u1 == 0 && u2 == 0) return (double)sc.stack37[type * 7 + type2] / 100.0;
  if (u1 == 0 || u2 == 0) {
    const int u = u1 == 0 ? u2 : u1;
    double z = sc.bulge[u];
    if (u == 1) {
      z += sc.stack37[type * 7 + type2];
    } else {
      if (type > 2) z += sc.terminal_au;
      if (type2 > 2) z += sc.terminal_au;
    }
    return z / 100.0;
  }
  const int a = base_of(qs[i + 1]), b = base_of(ds[j + 1]), c = base_of(qs[p - 1]), d = base_of(ds[q - 1]);
  int z;
  if (u1 + u2 == 2) z = sc.int11[((type * 8 + type2) * 5 + a) * 5 + b];
  else if (u1 == 1 && u2 == 2) z = sc.int21[(((type * 8 + type2) * 5 + a) * 5 + d) * 5 + b];
  else if (u1 == 2 && u2 == 1) z = sc.int21[(((type2 * 8 + type) * 5 + d) * 5 + a) * 5 + c];
  else if (u1 == 2 && u2 == 2) z = sc.int22[((((type * 8 + type2) * 5 + a) * 5 + c) * 5 + d) * 5 + b];
  else z = sc.internal37[u1 + u2] + sc.mismatchI37[(type * 5 + a) * 5 + b] + sc.mismatchI37[(type2 * 5 + d) * 5 + c];
  return (double)z / 100.0;
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
  if (flag == 1) t = sc.rtype[t];
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
  return (double)x / 100.0;
}


struct HitState {
  int q_sp, db_sp, q_len, db_len, id_start;
  double e_tot, e_acc, e_hyb;
};

} // namespace prb
