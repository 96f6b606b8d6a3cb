// Sequence encoder of the search stages (reference: encoder.hpp:36-79, encoder.cpp:27-44).
// A=2 C=3 G=4 U/T=5, other=1, sentinel=0; repeat_flag 1 keeps lower case as 6..9
// (soft masking), repeat_flag 2 folds lower case to upper (no masking).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace prb {

struct Encoder {
  uint8_t table[256];
  explicit Encoder(int repeat_flag);
  // query: forward, trailing sentinel (encoder.cpp:38-44); out has len+1 bytes
  void encode_query(const char *seq, int64_t len, uint8_t *out) const;
  // database side: each sequence REVERSED and followed by a sentinel (encoder.cpp:27-36)
  void append_db(const char *seq, int64_t len, std::vector<uint8_t> &out) const;
};

} // namespace prb
