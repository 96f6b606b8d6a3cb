#include "encoder.hpp"

#include <cstring>

namespace prb {

Encoder::Encoder(int repeat_flag) {
  std::memset(table, 1, sizeof table);
  table[(unsigned char)'A'] = 2;
  table[(unsigned char)'C'] = 3;
  table[(unsigned char)'G'] = 4;
  table[(unsigned char)'T'] = 5;
  table[(unsigned char)'U'] = 5;
  if (repeat_flag == 1) {
    table[(unsigned char)'a'] = 6;
    table[(unsigned char)'c'] = 7;
    table[(unsigned char)'g'] = 8;
    table[(unsigned char)'t'] = 9;
    table[(unsigned char)'u'] = 9;
  } else if (repeat_flag == 2) {
    table[(unsigned char)'a'] = 2;
    table[(unsigned char)'c'] = 3;
    table[(unsigned char)'g'] = 4;
    table[(unsigned char)'t'] = 5;
    table[(unsigned char)'u'] = 5;
  }
}

void Encoder::encode_query(const char *seq, int64_t len, uint8_t *out) const {
  for (int64_t i = 0; i < len; i++) out[i] = table[(unsigned char)seq[i]];
  out[len] = 0;
}

void Encoder::append_db(const char *seq, int64_t len, std::vector<uint8_t> &out) const {
  for (int64_t i = len - 1; i >= 0; i--) out.push_back(table[(unsigned char)seq[i]]);
  out.push_back(0);
}

} // namespace prb
