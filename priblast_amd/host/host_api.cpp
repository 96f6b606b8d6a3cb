// C ABI, host-only entry points (stage 2 of the seam: encode + suffix array; the CPU budget of the host thread pools).
#include "../../include/priblast_hip.h"
#include "cpu_budget.hpp"
#include "encoder.hpp"
#include "suffix_array.hpp"

extern "C" {

int prb_encode_query(const char *seq, int32_t len, int32_t repeat_flag, uint8_t *enc) {
  if (!seq || !enc || len < 0 || repeat_flag < 0 || repeat_flag > 2) return PRB_ERR_ARG;
  prb::Encoder(repeat_flag).encode_query(seq, len, enc);
  return PRB_OK;
}

int prb_suffix_array(const uint8_t *text, int32_t n, int32_t *sa) {
  if (!text || !sa || n < 0) return PRB_ERR_ARG;
  prb::suffix_array(text, n, sa);
  return PRB_OK;
}

int prb_cpu_budget(void) { return prb::cpu_budget(); }
int prb_host_threads_default(void) { return prb::default_host_threads(); }

} // extern "C"
