// Host-visible interface of the Raccess kernels (raccess_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace prb {

struct RaConst; // raccess_device.hpp

// One sequence of a batch.  Offsets are in elements of the respective pool.
struct RaSeqDesc {
  int64_t band_off; // doubles: kRaBands tables of (L+2)*(W+2)
  int64_t vec_off;  // doubles: kRaVecs vectors of (L+2)
  int64_t code_off; // bytes: L+2 codes (0, s[1..L], 0)
  int64_t out_off;  // floats: L entries of acc / cond
  int32_t L;
  int32_t pad;
};

constexpr int kRaBands = 13;
constexpr int kRaVecs = 8;
constexpr int kRaMaxSpan = 255; // cells of a column / positions of a window: at most four passes of 64 lanes (raccess_kernels.hip)

struct RaBatch {
  const RaSeqDesc *desc;
  int32_t nseq;
  int32_t lmax; // the longest sequence of the batch (grids over positions)
  int32_t W, delta;
  double *band;
  double *vec;
  const unsigned char *codes;
  float *acc;  // out
  float *cond; // out
  // > 0: the LOGSUM branch of the bulge / interior-loop sums is computed by k_biloop_logsum, a wavefront per window of
  // 64 positions, this many windows per sequence (>= the longest sequence's); 0: by the ordered pass on the sequence's own wavefront
  int32_t logsum_windows = 0;
  // with logsum_windows > 0: the windows take the linear branch of every sequence as well (its classification pass and,
  // where that cannot decide, the ordered sums) - a launch with few sequences, where a sequence's own wavefront is the bottleneck
  int32_t windows_all = 0;
  // helper wavefronts per sequence for the big folds of the inside / outside passes (0, 1, 2; 3 = two helpers and a wavefront
  // that folds, beside the sequence's own, which keeps the other phases): a workgroup per sequence
  // instead of a wavefront - for the few sequences of a query batch, whose time is the latency of one wavefront's chain
  int32_t helpers = 0;
};

// Enqueue fill + inside + outside + biloop + accessibility for a batch on `stream`.
hipError_t ra_launch(const RaBatch &b, const RaConst &c, int64_t band_elems, int64_t vec_elems,
                     hipStream_t stream);

hipError_t ra_watchdog_tripped(int *tripped, hipStream_t stream);

// doubles / bytes needed for a sequence of length L
inline int64_t ra_band_elems(int L, int W) { return (int64_t)kRaBands * (L + 2) * (W + 2); }
inline int64_t ra_vec_elems(int L) { return (int64_t)kRaVecs * (L + 2); }

} // namespace prb
