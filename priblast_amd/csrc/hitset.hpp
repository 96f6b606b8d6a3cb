// prb_hitset (include/priblast_hip.h): the result of prb_search_page / prb_gather_hits, shared by
// capi_search.hip (which fills it) and capi_comm.hip (the final hit gather).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/priblast_hip.h"
#include "context.hpp"

namespace prb {
struct Drainer;

// pinned host staging buffer that only grows
struct PinnedBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return PRB_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      set_error("out of pinned host memory allocating " + std::to_string(want) + " bytes");
      return PRB_ERR_NOMEM;
    }
    cap = want;
    return PRB_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Device buffer that grows and keeps what it holds (the packed records of the sub-batches of a
// search, kept for the final hit gather).  `hint` = what the last search of the context ended up with: the first
// append allocates that much, so a steady stream of similar batches never grows.  A buffer that is outgrown is only
// retired, not freed, until the hit set goes: hipFree waits for the whole device - for the accessibility kernels of
// the next batch on the other stream as well (ten doublings per hit set cost 0.55 s per configs[2] step that way).
struct KeepBuf {
  DevBuf b;
  size_t used = 0, hint = 0;
  std::vector<DevBuf> retired;
  int append(const void *src_dev, size_t bytes, hipStream_t s) {
    if (used + bytes > b.cap) {
      DevBuf nb;
      int rc = nb.ensure(std::max({(used + bytes) * 2, hint + hint / 8, (size_t)1 << 20}));
      if (rc) return rc;
      if (used) {
        PRB_HIP(hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, s));
        PRB_HIP(hipStreamSynchronize(s));
      }
      if (b.p) retired.push_back(b);
      b = nb;
    }
    if (bytes) PRB_HIP(hipMemcpyAsync(static_cast<uint8_t *>(b.p) + used, src_dev, bytes, hipMemcpyDeviceToDevice, s));
    used += bytes;
    return PRB_OK;
  }
  void release() {
    b.release();
    for (DevBuf &r : retired) r.release();
    retired.clear();
    used = 0;
  }
};

} // namespace prb

struct prb_hitset {
  std::vector<prb_hit> hits;
  std::vector<int32_t> bp;
  // a gathered hit set (prb_gather_hits) borrows a pinned slot of its communicator instead, and gives it
  // back when it is freed
  const prb_hit *ext_hits = nullptr;
  const int32_t *ext_bp = nullptr;
  int64_t ext_nhits = 0, ext_bp_ints = 0;
  std::vector<int32_t> g_nq_of_rank, g_qlen; // gathered: batch size of every rank, unmasked lengths of all queries
  void (*ext_release)(void *owner, int slot) = nullptr;
  void *ext_owner = nullptr;
  int ext_slot = -1;
  // the same records as they were packed on the device (only with prb_ctx_keep_device_records)
  prb::KeepBuf d_hits, d_bp;
  bool on_device = false;
  int device = 0;
  ~prb_hitset() {
    if (ext_release) ext_release(ext_owner, ext_slot);
    if (d_hits.b.p || d_bp.b.p) {
      (void)hipSetDevice(device);
      d_hits.release();
      d_bp.release();
    }
  }
  int64_t counts[3] = {0, 0, 0};
  int64_t slow_hits = 0; // extensions that outgrew the LDS kernel
  // while prb_search_page runs: results of finished sub-batches are appended by a background
  // thread; the main thread only keeps the totals it needs for the offsets
  prb::Drainer *drain = nullptr;
  int64_t hits_total = 0, bp_ints_total = 0;
  int next_slot = 0;
};

