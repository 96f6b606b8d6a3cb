// Result lines of `ris` (SaveMyResults, rna_interaction_search.cpp:322-369): one text line per hit,
//   Id,qname,qlen,dbname,dblen,Eacc,Ehyb,Etotal,(q0-qN:db0-dbN)          -s 0
//   Id,qname,qlen,dbname,dblen,Eacc,Ehyb,Etotal,(q:db) (q:db) ...        -s 1
// doubles as `ostream << double` prints them (6 significant digits, "%g"), database coordinates
// turned from the reversed page text back into forward sequence coordinates (:352-363).
// Shared by the library (prb_write_lines) and the command line (`ris`, `txt`).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/priblast_hip.h"

namespace prb {

// What the output needs to know about the sequences of one database page.
struct SeqTable {
  std::vector<std::string> names;
  std::vector<int32_t> len, len_unmasked, start_pos;
};

// The hits of one batch against one page, and a batch: plain arrays, wherever they live (hit sets
// of the library, or a binary hit file read back).
struct PageHits {
  const prb_hit *h = nullptr;
  int64_t n = 0;
  const int32_t *bp = nullptr;
  int64_t nbp = 0; // pairs
};
struct BatchView {
  size_t nq = 0;
  const std::string *names = nullptr; // [nq]
  const int32_t *qlen_unmasked = nullptr;
  std::vector<PageHits> pages;
};

struct LineSink {
  int fd = -1; // -1: the lines are formatted and counted, not written
  int64_t lines = 0, bytes = 0;
};

// Lines of one batch, query by query and page by page as the reference groups them, numbered from
// `id0` on; returns the next id, or -1 when a write failed.  The hits of a page arrive grouped by
// query in ascending order, so a query's hits are one contiguous range per page; ranges are cut
// into pieces that host threads format in parallel.
int64_t format_batch(const BatchView &v, const std::vector<SeqTable> &tabs, int output_style, int64_t id0, LineSink &sink,
                     int threads);

int format_threads(); // PRB_HOST_THREADS, else min(32, hardware threads)

} // namespace prb
