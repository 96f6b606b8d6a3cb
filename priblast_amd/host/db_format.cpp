#include "db_format.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

namespace prb {

namespace {
struct File {
  FILE *f = nullptr;
  ~File() {
    if (f) fclose(f);
  }
  bool open(const std::string &p, const char *mode) {
    f = fopen(p.c_str(), mode);
    return f != nullptr;
  }
  bool rd(void *dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, f) == bytes; }
  bool wr(const void *src, size_t bytes) { return bytes == 0 || fwrite(src, 1, bytes, f) == bytes; }
};
} // namespace

std::string read_db(const std::string &prefix, DbHeader &hdr, std::vector<DbPage> &pages) {
  File bas, seq, acc, ind;
  if (!bas.open(prefix + ".bas", "rb")) return "Error: can't open " + prefix + ".bas";
  std::ifstream nam(prefix + ".nam");
  if (!seq.open(prefix + ".seq", "rb") || !acc.open(prefix + ".acc", "rb") || !ind.open(prefix + ".ind", "rb") || !nam)
    return "Error: can't open db_file";
  int32_t v[4];
  if (!bas.rd(v, sizeof v)) return "Error: truncated " + prefix + ".bas";
  hdr.hash_size = v[0];
  hdr.repeat_flag = v[1];
  hdr.maximal_span = v[2];
  hdr.min_accessible_length = v[3];
  if (hdr.hash_size < 1 || hdr.hash_size > 14) return "Error: implausible lookup table size in " + prefix + ".bas";
  pages.clear();
  for (;;) {
    int32_t nseq;
    if (!seq.rd(&nseq, 4)) break; // EOF on .seq ends the database (db_reader.cpp:73-82)
    if (nseq < 0 || nseq > (1 << 28)) return "Error: corrupt " + prefix + ".seq (sequence count)";
    pages.emplace_back();
    DbPage &pg = pages.back();
    pg.nseq = nseq;
    pg.seq_length.resize(nseq);
    if (!seq.rd(pg.seq_length.data(), 4 * (size_t)nseq)) return "Error: truncated " + prefix + ".seq";
    pg.start_pos.resize(nseq);
    int64_t t = 0;
    for (int i = 0; i < nseq; i++) {
      if (pg.seq_length[i] < 0) return "Error: corrupt " + prefix + ".seq (negative sequence length)";
      pg.start_pos[i] = (int32_t)t;
      t += pg.seq_length[i] + 1;
    }
    int32_t nchars;
    if (!seq.rd(&nchars, 4)) return "Error: truncated " + prefix + ".seq";
    // the text is the sequences, each followed by a 0 (db_construction.cpp:371-392): the kernels index it, the
    // suffix array and the k-mer table with these numbers, so they are checked once here
    if (nchars < 0 || (int64_t)nchars != t) return "Error: corrupt " + prefix + ".seq (text length does not match the sequence lengths)";
    pg.seqs.resize(nchars);
    if (!seq.rd(pg.seqs.data(), (size_t)nchars)) return "Error: truncated " + prefix + ".seq";
    {
      int32_t c = 0;
      for (int32_t i = 0; i < nchars; i++) {
        if (pg.seqs[i] == 0) {
          pg.seq_length_rep.push_back(c);
          c = 0;
        } else if (pg.seqs[i] >= 2 && pg.seqs[i] <= 5) {
          c++;
        }
      }
      pg.seq_length_rep.push_back(c);
    }
    pg.acc.assign((size_t)(t - nseq), 0.0f);
    pg.cond.assign((size_t)(t - nseq), 0.0f);
    std::vector<float> tmp;
    for (int i = 0; i < nseq; i++) {
      const int64_t base = pg.acc_base(i), L = pg.seq_length[i];
      int32_t n;
      if (!acc.rd(&n, 4)) return "Error: truncated " + prefix + ".acc";
      if (n < 0) n = 0;
      if (n > L + 1) return "Error: corrupt " + prefix + ".acc (more values than positions)";
      tmp.resize(n);
      if (!acc.rd(tmp.data(), 4 * (size_t)n)) return "Error: truncated " + prefix + ".acc";
      std::memcpy(pg.acc.data() + base, tmp.data(), 4 * (size_t)std::min<int64_t>(n, L));
      if (!acc.rd(&n, 4)) return "Error: truncated " + prefix + ".acc";
      if (n < 0) n = 0;
      if (n > L + 1) return "Error: corrupt " + prefix + ".acc (more values than positions)";
      tmp.resize(n);
      if (!acc.rd(tmp.data(), 4 * (size_t)n)) return "Error: truncated " + prefix + ".acc";
      std::memcpy(pg.cond.data() + base, tmp.data(), 4 * (size_t)std::min<int64_t>(n, L));
    }
    pg.names.resize(nseq);
    for (int i = 0; i < nseq; i++) std::getline(nam, pg.names[i]);
    int32_t nsa;
    if (!ind.rd(&nsa, 4)) return "Error: truncated " + prefix + ".ind";
    if (nsa != nchars) return "Error: corrupt " + prefix + ".ind (suffix array length does not match the text)";
    pg.sa.resize(nsa);
    if (!ind.rd(pg.sa.data(), 4 * (size_t)nsa)) return "Error: truncated " + prefix + ".ind";
    for (int32_t v : pg.sa)
      if (v < 0 || v >= nchars) return "Error: corrupt " + prefix + ".ind (suffix array entry out of range)";
    pg.start_hash.resize(hdr.hash_size);
    pg.end_hash.resize(hdr.hash_size);
    for (int pass = 0; pass < 2; pass++) {
      size_t n = 4;
      for (int i = 0; i < hdr.hash_size; i++, n *= 4) {
        auto &h = pass == 0 ? pg.start_hash[i] : pg.end_hash[i];
        h.resize(n);
        if (!ind.rd(h.data(), 4 * n)) return "Error: truncated " + prefix + ".ind";
      }
    }
    // every k-mer's [start, end] is an inclusive interval of the suffix array (the seed DFS and the kernels index
    // sa[start .. end]), or the empty one, stored as (1, 0)
    for (int i = 0; i < hdr.hash_size; i++)
      for (size_t k = 0; k < pg.start_hash[i].size(); k++) {
        const int32_t sp = pg.start_hash[i][k], ep = pg.end_hash[i][k];
        const bool empty = sp == 1 && ep == 0;
        if (!empty && (sp < 0 || ep >= nsa || sp > ep)) return "Error: corrupt " + prefix + ".ind (k-mer interval out of range)";
      }
  }
  return "";
}

void sa_narrow(const uint8_t *text, const int32_t *sa, int32_t *s, int32_t *e, uint8_t c, int32_t offset) {
  int32_t lo = *s, hi = *e;
  if (lo > hi) {
    *s = 1;
    *e = 0;
    return;
  }
  int32_t a = lo, b = hi + 1; // first suffix whose character is >= c
  while (a < b) {
    int32_t m = a + (b - a) / 2;
    if (text[sa[m] + offset] < c) a = m + 1;
    else b = m;
  }
  const int32_t first = a;
  b = hi + 1; // first suffix whose character is > c
  while (a < b) {
    int32_t m = a + (b - a) / 2;
    if (text[sa[m] + offset] <= c) a = m + 1;
    else b = m;
  }
  if (first >= a) {
    *s = 1;
    *e = 0;
  } else {
    *s = first;
    *e = a - 1;
  }
}

void build_kmer_table(const std::vector<uint8_t> &text, const std::vector<int32_t> &sa, int hash_size,
                      std::vector<std::vector<int32_t>> &start_hash, std::vector<std::vector<int32_t>> &end_hash) {
  start_hash.assign(hash_size, {});
  end_hash.assign(hash_size, {});
  size_t n = 4;
  for (int lvl = 0; lvl < hash_size; lvl++, n *= 4) {
    start_hash[lvl].resize(n);
    end_hash[lvl].resize(n);
    for (size_t j = 0; j < n; j++) {
      int32_t s, e;
      if (lvl == 0) {
        s = 0;
        e = (int32_t)sa.size() - 1;
      } else {
        s = start_hash[lvl - 1][j / 4];
        e = end_hash[lvl - 1][j / 4];
      }
      sa_narrow(text.data(), sa.data(), &s, &e, (uint8_t)(j % 4 + 2), lvl);
      start_hash[lvl][j] = s;
      end_hash[lvl][j] = e;
    }
  }
}

std::string DbWriter::open(const std::string &pfx, const DbHeader &hdr) {
  prefix = pfx;
  for (const char *ext : {".seq", ".acc", ".nam", ".ind"}) {
    File f;
    if (!f.open(prefix + ext, "wb")) return "Error: can't create " + prefix + ext;
  }
  File bas;
  if (!bas.open(prefix + ".bas", "wb")) return "Error: can't create " + prefix + ".bas";
  int32_t v[4] = {hdr.hash_size, hdr.repeat_flag, hdr.maximal_span, hdr.min_accessible_length};
  if (!bas.wr(v, sizeof v)) return "Error: can't write " + prefix + ".bas";
  return "";
}

std::string DbWriter::append_page(const DbPage &pg, int delta) {
  File seq, acc, nam, ind;
  if (!seq.open(prefix + ".seq", "ab") || !acc.open(prefix + ".acc", "ab") || !nam.open(prefix + ".nam", "ab") ||
      !ind.open(prefix + ".ind", "ab"))
    return "Error: can't append to database " + prefix;
  int32_t n = pg.nseq;
  bool ok = seq.wr(&n, 4) && seq.wr(pg.seq_length.data(), 4 * (size_t)n);
  n = (int32_t)pg.seqs.size();
  ok = ok && seq.wr(&n, 4) && seq.wr(pg.seqs.data(), pg.seqs.size());
  for (int i = 0; i < pg.nseq && ok; i++) {
    const int64_t base = pg.acc_base(i);
    const int32_t L = pg.seq_length[i];
    int32_t n1 = L - delta + 1; // raccess.cpp:449-450
    ok = ok && acc.wr(&n1, 4) && acc.wr(pg.acc.data() + base, 4 * (size_t)std::max(n1, 0));
    ok = ok && acc.wr(&L, 4) && acc.wr(pg.cond.data() + base, 4 * (size_t)L);
    ok = ok && nam.wr(pg.names[i].data(), pg.names[i].size()) && nam.wr("\n", 1);
  }
  n = (int32_t)pg.sa.size();
  ok = ok && ind.wr(&n, 4) && ind.wr(pg.sa.data(), 4 * pg.sa.size());
  for (auto &h : pg.start_hash) ok = ok && ind.wr(h.data(), 4 * h.size());
  for (auto &h : pg.end_hash) ok = ok && ind.wr(h.data(), 4 * h.size());
  return ok ? "" : "Error: short write to database " + prefix;
}

} // namespace prb
