// Result lines of `ris` (SaveMyResults, rna_interaction_search.cpp:322-369).  See output.hpp.
#include "output.hpp"

#include "cpu_budget.hpp"

#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace prb {

int format_threads() {
  // (its own knob first: with one process per GPU only rank 0 writes lines - for all ranks - while every rank runs a seed DFS)
  if (const char *e = std::getenv("PRB_FORMAT_THREADS")) return std::max(1, std::atoi(e));
  if (const char *e = std::getenv("PRB_HOST_THREADS")) return std::max(1, std::atoi(e));
  return default_host_threads();
}

namespace {

// A piece of one query's hits against one page: hits [i0, i1), the first of them numbered id.
struct Piece {
  uint32_t q, p;
  int64_t i0, i1, id;
};
constexpr int64_t kPieceLines = 8192;

// Text buffer that grows by doubling; the formatters write through raw pointers.
struct Buf {
  std::vector<char> v;
  size_t n = 0;
  char *room(size_t want) {
    if (n + want > v.size()) v.resize(std::max(v.size() * 2, n + want + 4096));
    return v.data() + n;
  }
};

inline char *put_int(char *p, long long x) { return std::to_chars(p, p + 24, x).ptr; }
// `ostream << double` with the default precision = printf("%g"): std::to_chars in the general
// format with precision 6 is specified as exactly that conversion (tests/test_host.py compares).
inline char *put_g(char *p, double x) { return std::to_chars(p, p + 40, x, std::chars_format::general, 6).ptr; }

void format_piece(const Piece &pc, const BatchView &v, const std::vector<SeqTable> &tabs, int output_style, Buf &b) {
  const PageHits &ph = v.pages[pc.p];
  const SeqTable &tab = tabs[pc.p];
  const std::string &qname = v.names[pc.q];
  const int32_t qlen = v.qlen_unmasked[pc.q];
  int64_t id = pc.id;
  for (int64_t i = pc.i0; i < pc.i1; i++) {
    const prb_hit &x = ph.h[i];
    const std::string &dname = tab.names[x.db_id];
    const int32_t len = tab.len[x.db_id], sp = tab.start_pos[x.db_id];
    const int32_t npairs = output_style == 1 ? x.bp_count : (x.bp_count > 0 ? 1 : 0);
    char *p = b.room(qname.size() + dname.size() + 200 + (size_t)npairs * 48);
    char *const p0 = p;
    p = put_int(p, id++);
    *p++ = ',';
    std::memcpy(p, qname.data(), qname.size());
    p += qname.size();
    *p++ = ',';
    p = put_int(p, qlen);
    *p++ = ',';
    std::memcpy(p, dname.data(), dname.size());
    p += dname.size();
    *p++ = ',';
    p = put_int(p, tab.len_unmasked[x.db_id]);
    *p++ = ',';
    p = put_g(p, x.e_acc);
    *p++ = ',';
    p = put_g(p, x.e_hyb);
    *p++ = ',';
    p = put_g(p, x.e_tot);
    *p++ = ',';
    const int32_t *pp = ph.bp + 2 * x.bp_offset;
    auto fwd = [&](int32_t dbpos) { return (len - 1) - (dbpos - sp); }; // reversed page text -> forward coordinate
    if (output_style == 1) {
      for (int32_t j = 0; j < x.bp_count; j++) {
        *p++ = '(';
        p = put_int(p, pp[2 * j]);
        *p++ = ':';
        p = put_int(p, fwd(pp[2 * j + 1]));
        *p++ = ')';
        *p++ = ' ';
      }
    } else if (x.bp_count > 0) {
      const int32_t l = x.bp_count - 1;
      *p++ = '(';
      p = put_int(p, pp[0]);
      *p++ = '-';
      p = put_int(p, pp[2 * l]);
      *p++ = ':';
      p = put_int(p, fwd(pp[1]));
      *p++ = '-';
      p = put_int(p, fwd(pp[2 * l + 1]));
      *p++ = ')';
      *p++ = ' ';
    }
    *p++ = '\n';
    b.n += (size_t)(p - p0);
  }
}

bool write_all(int fd, const char *p, size_t n) {
  while (n) {
    const ssize_t k = ::write(fd, p, n);
    if (k < 0) return false;
    p += k;
    n -= (size_t)k;
  }
  return true;
}

} // namespace

int64_t format_batch(const BatchView &v, const std::vector<SeqTable> &tabs, int output_style, int64_t id0, LineSink &sink,
                     int threads) {
  const size_t nq = v.nq, np = v.pages.size();
  std::vector<std::vector<int64_t>> first(np, std::vector<int64_t>(nq + 1, 0)); // first[p][q] = first hit of query q
  for (size_t p = 0; p < np; p++) {
    const int64_t n = v.pages[p].n;
    const prb_hit *h = v.pages[p].h;
    size_t q = 0;
    for (int64_t i = 0; i < n; i++)
      while (q < nq && (int64_t)q <= h[i].query) first[p][q++] = i;
    while (q <= nq) first[p][q++] = n;
  }
  std::vector<Piece> pieces;
  int64_t id = id0;
  for (size_t q = 0; q < nq; q++)
    for (size_t p = 0; p < np; p++)
      for (int64_t i = first[p][q]; i < first[p][q + 1]; i += kPieceLines) {
        const int64_t e = std::min(first[p][q + 1], i + kPieceLines);
        pieces.push_back(Piece{(uint32_t)q, (uint32_t)p, i, e, id});
        id += e - i;
      }
  // rounds of pieces: formatted in parallel, then written in order
  const size_t per_round = (size_t)std::max(1, threads) * 4;
  std::vector<Buf> bufs(std::min(per_round, std::max<size_t>(pieces.size(), 1)));
  bool ok = true;
  for (size_t r0 = 0; r0 < pieces.size() && ok; r0 += per_round) {
    const size_t r1 = std::min(pieces.size(), r0 + per_round);
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, threads))
    for (size_t k = r0; k < r1; k++) {
      Buf &b = bufs[k - r0];
      b.n = 0;
      format_piece(pieces[k], v, tabs, output_style, b);
    }
    for (size_t k = r0; k < r1 && ok; k++) {
      const Buf &b = bufs[k - r0];
      if (sink.fd >= 0 && b.n) ok = write_all(sink.fd, b.v.data(), b.n);
      sink.bytes += (int64_t)b.n;
      sink.lines += pieces[k].i1 - pieces[k].i0;
    }
  }
  return ok ? id : -1;
}

} // namespace prb
