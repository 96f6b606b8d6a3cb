// prb_ctx: one GPU's state behind the C ABI (include/priblast_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "energy.hpp"
#include "raccess_device.hpp"

namespace prb {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what); // records the error text, returns PRB_ERR_HIP

#define PRB_HIP(call)                                   \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return prb::hip_fail(e_, #call); \
  } while (0)

// Device buffer that only grows.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes);
  void release();
  template <class T> T *as() const { return static_cast<T *>(p); }
};

struct StageTimer {
  double ms = 0;
  int64_t launches = 0;
};

} // namespace prb

struct prb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  prb::EnergyParams params;
  prb::RaccessTables ra_tables;
  prb::FmathTables fm_tables;
  prb::RaConst ra_const{};
  // int tables for the search stages (search_kernels.hip)
  void *search_const = nullptr;
  void *search_ws = nullptr;   // prb::SearchWs, capi_search.hip
  int64_t slow_hits = 0;       // extensions that went through the HBM-scratch fallback kernel
  int max_gap_caps = 128;      // largest diagonal capacity any gapped extension has needed
  bool keep_device_records = false; // final hit sets also keep their packed records in HBM (for prb_gather_hits)
  size_t keep_hint_hits = 0, keep_hint_bp = 0; // bytes the last such hit set ended up with (KeepBuf::hint)
  size_t host_hint_hits = 0, host_hint_bp = 0; // hits / pair ints of the last final hit set (its vectors' first reserve)
  prb::DevBuf d_expd, d_log, d_small, d_big;
  // Raccess workspaces
  prb::DevBuf ra_band, ra_vec, ra_codes, ra_desc, ra_acc, ra_cond;
  size_t ra_budget_bytes = 0;
  std::map<std::string, prb::StageTimer> timers;

  int time_begin();
  int time_end(const char *stage, int64_t launches);
};
