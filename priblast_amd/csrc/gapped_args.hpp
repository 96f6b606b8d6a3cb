// What the gapped-extension kernels share (gapped_lds.hip: a group of lanes per hit, state in LDS or HBM
// scratch; gapped_front.hip: the kernel in front of them): the kernel arguments, the cell record of the LDS forms,
// the origin of a direction and the hand-over mark between kernels of the cascade.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "search_device.hpp"
#include "search_kernels.hpp"

namespace prb {

// A filled cell r is (i, j, pred = index of the predecessor cell, type = Stem::type, the bases at
// (i+1, j+1) that a later loop closing on this cell needs), packed into 32 bits.
struct Rec32 {
  using word = uint32_t; // i:7 | j:7 | pred:7 | type:3 | qa:3 | da:3 (the bases at (i+1, j+1) along the extension)
  static constexpr bool kBases = true;
  static __device__ __forceinline__ word pack(int i, int j, int pred, int type, int qa, int da) {
    return (word)i | ((word)j << 7) | ((word)pred << 14) | ((word)type << 21) | ((word)qa << 24) | ((word)da << 27);
  }
  static __device__ __forceinline__ int i(word v) { return v & 0x7F; }
  static __device__ __forceinline__ int j(word v) { return (v >> 7) & 0x7F; }
  static __device__ __forceinline__ int pred(word v) { return (v >> 14) & 0x7F; }
  static __device__ __forceinline__ int type(word v) { return (v >> 21) & 7; }
  static __device__ __forceinline__ int qa(word v) { return (v >> 24) & 7; }
  static __device__ __forceinline__ int da(word v) { return (v >> 27) & 7; }
};



// Where a direction starts (gapped_extension.cpp:88-128), from the hit as the direction found it:
// the outermost pair on that side, and the database sequence's accessibility indices.
struct DirOrigin {
  int q_start, id_start, id_end;
  int64_t db_start;
};
__device__ __forceinline__ DirOrigin dir_origin(const HitState &h, int flag) {
  DirOrigin o;
  if (flag == 0) {
    o.q_start = h.q_sp;
    o.db_start = h.db_sp;
  } else {
    o.q_start = h.q_sp + h.q_len - 1;
    o.db_start = (int64_t)h.db_sp + h.db_len - 1;
  }
  o.id_start = h.id_start;
  o.id_end = h.id_start + h.db_len - 1;
  return o;
}


struct GapArgs {
  HitSoA in, out;
  int64_t n;
  const uint32_t *subset;
  QBatchDev qb;
  PageDev pg;
  SearchConst sc;
  ExtOpts o;
  uint8_t *overflow;  // mode 0: overflow[w] = 1 if the capacities of this kernel did not suffice
  uint8_t *tier_out;  // mode 0: tier_out[x] = tier_id when hit x was completed here
  int tier_id;
  const uint8_t *first_flag;
  int32_t *bp_count; // mode 0: traced pairs of hit x, left | right << 16
  uint16_t *trace;   // mode 0 (LDS tiers): the first kTraceCap traced cells (i | j << 8) per direction of hit x
  const int64_t *bp_off;
  int32_t *bp_out;
  unsigned long long *next_work; // LDS kernels: work counter (zero at launch) behind the statically assigned first hits
  // mode 0: state dumps of hits that outgrow an LDS tier, for the next one to continue from
  // (slot[x] = -1: none): `rin` = what this kernel may continue, `rout` = where it leaves its own
  GapResume rin, rout;
  LongTrace lt;     // mode 0, the wavefront-per-hit kernel: where it leaves whole traceback chains (search_kernels.hpp)
  double *acc_scratch = nullptr; // LDS tiers 1 and 2: 2 x capacity doubles per resident group (gapped_acc_scratch_bytes)
  int period = 0;   // LDS tiers: lockstep iterations between the boundaries at which groups change direction / hit (0: the drop-out length)
  int early = 0;    // LDS tiers: a boundary also as soon as this many groups of the wavefront have finished their direction (0: never)
  int handover = 0; // mode 0, LDS tiers and the wavefront-per-hit kernel: stop behind a first direction that this kernel ran (kHandoverMark)
};


} // namespace prb
