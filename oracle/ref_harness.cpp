// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin driver (our code) around the UNMODIFIED reference classes.  It is compiled
// against the reference's object files (oracle/Makefile, target `ref`) inside the
// container that has /root/reference and writes golden vectors that tests/ compare
// the oracle restatement and the HIP path against.  Nothing from the reference is
// copied: the reference headers are included from where they lie.
//
// Sub-commands (all outputs little-endian binary, layouts documented in
// tests/golden/README.md and parsed by tests/refdump.py):
//   tables  <out>                         fmath expd/log tables (fmath.hpp:148-216)
//   raccess <fasta> <W> <delta> <out>     Raccess::Run in-memory overload per sequence
//                                         (raccess.cpp:42-50) -> acc[L], cond[L]
//   raccess_dbg <fasta> <W> <delta> <out> as above + the 12 DP tables and outer vectors
//   stages  <fasta> <dbprefix> <out> [-l -e -f -g -x -y -m as in ris]
//                                         per (query, page): hits after SearchSeed,
//                                         ExtendWithoutGap, ExtendWithGap
//                                         (rna_interaction_search.cpp:264-320)
//   sa      <fasta> <repeat_flag> <out>   Encoder::Encode + sais per sequence
#define private public
#define protected public
#include "raccess.hpp"
#include "rna_interaction_search.hpp"
#undef private
#undef protected

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "db_reader.hpp"
#include "encoder.hpp"
#include "fmath.hpp"
#include "sais.hpp"

static void read_fasta(const char *path, std::vector<std::string> &names,
                       std::vector<std::string> &seqs) {
  std::ifstream fp(path);
  if (!fp) {
    fprintf(stderr, "harness: cannot open %s\n", path);
    exit(1);
  }
  std::string line, cur;
  bool have = false;
  while (std::getline(fp, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n'))
      line.pop_back();
    if (!line.empty() && line[0] == '>') {
      if (have) seqs.push_back(cur);
      names.push_back(line.substr(1));
      cur.clear();
      have = true;
    } else {
      cur += line;
    }
  }
  if (have) seqs.push_back(cur);
}

template <class T> static void put(FILE *f, const T &v) { fwrite(&v, sizeof(T), 1, f); }

static void put_hits(FILE *f, const std::vector<Hit> &hits) {
  put<int32_t>(f, (int32_t)hits.size());
  for (const Hit &h : hits) {
    put<int32_t>(f, h.GetQSp());
    put<int32_t>(f, h.GetDbSp());
    put<int32_t>(f, h._q_length);
    put<int32_t>(f, h._db_length);
    put<int32_t>(f, h.GetDbSeqId());
    put<int32_t>(f, h.GetDbSeqIdStart());
    put<double>(f, h.GetAccessibilityEnergy());
    put<double>(f, h.GetHybridizationEnergy());
    put<double>(f, h.GetEnergy());
    put<int32_t>(f, h.GetBasePairLength());
    for (int i = 0; i < h.GetBasePairLength(); i++) {
      put<int32_t>(f, h.GetBasePairFirst(i));
      put<int32_t>(f, h.GetBasePairSecond(i));
    }
  }
}

static void put_table(FILE *f, const std::vector<std::vector<double>> &t) {
  for (const auto &row : t) fwrite(row.data(), sizeof(double), row.size(), f);
}

static int cmd_tables(const char *out) {
  FILE *f = fopen(out, "wb");
  const fmath::local::ExpdVar<> &e = fmath::local::C<>::expdVar;
  const fmath::local::LogVar<> &l = fmath::local::C<>::logVar;
  put<uint32_t>(f, 0x54424C31u); // "TBL1"
  put<double>(f, e.a);
  put<double>(f, e.ra);
  put<double>(f, e.C1[0]);
  put<double>(f, e.C2[0]);
  put<double>(f, e.C3[0]);
  for (int i = 0; i < 2048; i++) put<uint64_t>(f, (uint64_t)e.tbl[i]);
  put<float>(f, l.c_log2);
  for (int i = 0; i < 2048; i++) {
    put<float>(f, l.tbl[i].app);
    put<float>(f, l.tbl[i].rev);
  }
  // a few known-answer probes of the two functions themselves
  const double xs[] = {0.0, -1.0, 1.0, -0.5, -37.25, -700.0, -708.3, -708.4, -1e6, 88.7, 300.5, 709.7, 1e-9, -1e-9, 12.3456789};
  put<int32_t>(f, (int32_t)(sizeof(xs) / sizeof(xs[0])));
  for (double x : xs) {
    put<double>(f, x);
    put<double>(f, fmath::expd(x));
  }
  const float ys[] = {1.0f, 2.0f, 0.5f, 1.5f, 3.14159f, 1e-30f, 1e30f, 0.0f,
                      1e-45f, 3.4e38f, std::numeric_limits<float>::infinity(), 1.0000001f, 1.9999999f, 123456.78f};
  put<int32_t>(f, (int32_t)(sizeof(ys) / sizeof(ys[0])));
  for (float y : ys) {
    put<float>(f, y);
    put<float>(f, fmath::log(y));
  }
  fclose(f);
  return 0;
}

static int cmd_raccess(int argc, char **argv, bool dbg) {
  std::vector<std::string> names, seqs;
  read_fasta(argv[2], names, seqs);
  int W = atoi(argv[3]), delta = atoi(argv[4]);
  FILE *f = fopen(argv[5], "wb");
  put<uint32_t>(f, dbg ? 0x52414444u : 0x52414343u); // "RADD" / "RACC"
  put<int32_t>(f, (int32_t)seqs.size());
  put<int32_t>(f, W);
  put<int32_t>(f, delta);
  for (size_t s = 0; s < seqs.size(); s++) {
    Raccess r(W, delta);
    std::vector<float> acc, cond;
    int L = (int)seqs[s].size();
    put<int32_t>(f, L);
    if (!dbg) {
      r.Run(seqs[s], acc, cond);
    } else {
      r.Initiallize(seqs[s]);
      r.CalcInsideVariable();
      r.CalcOutsideVariable();
      fwrite(r._Alpha_outer.data(), sizeof(double), L + 1, f);
      fwrite(r._Beta_outer.data(), sizeof(double), L + 1, f);
      put_table(f, r._Alpha_stem);
      put_table(f, r._Alpha_stemend);
      put_table(f, r._Alpha_multi);
      put_table(f, r._Alpha_multibif);
      put_table(f, r._Alpha_multi1);
      put_table(f, r._Alpha_multi2);
      put_table(f, r._Beta_stem);
      put_table(f, r._Beta_stemend);
      put_table(f, r._Beta_multi);
      put_table(f, r._Beta_multibif);
      put_table(f, r._Beta_multi1);
      put_table(f, r._Beta_multi2);
      r.CalcAccessibility(acc, cond);
      r.Clear();
    }
    fwrite(acc.data(), sizeof(float), L, f);
    fwrite(cond.data(), sizeof(float), L, f);
  }
  fclose(f);
  return 0;
}

static int cmd_sa(int argc, char **argv) {
  std::vector<std::string> names, seqs;
  read_fasta(argv[2], names, seqs);
  int flag = atoi(argv[3]);
  FILE *f = fopen(argv[4], "wb");
  put<uint32_t>(f, 0x53415252u);
  put<int32_t>(f, (int32_t)seqs.size());
  for (auto &s : seqs) {
    Encoder enc(flag);
    std::vector<unsigned char> e;
    enc.Encode(s, e);
    std::vector<int> sa(e.size());
    sais(&e[0], &sa[0], (int)e.size());
    put<int32_t>(f, (int32_t)e.size());
    fwrite(e.data(), 1, e.size(), f);
    fwrite(sa.data(), sizeof(int), sa.size(), f);
  }
  fclose(f);
  return 0;
}

static int cmd_stages(int argc, char **argv) {
  // argv: stages <fasta> <dbprefix> <out> [ris options]
  std::vector<std::string> names, seqs;
  read_fasta(argv[2], names, seqs);
  std::string db = argv[3];
  const char *out = argv[4];

  RnaInteractionSearchParameters p;
  p._db_filename = db;
  for (int i = 5; i + 1 < argc; i += 2) {
    std::string o = argv[i];
    const char *v = argv[i + 1];
    if (o == "-l") p._max_seed_length = atoi(v);
    else if (o == "-e") p._hybrid_energy_threshold = atof(v);
    else if (o == "-f") p._interaction_energy_threshold = atof(v);
    else if (o == "-g") p._final_threshold = atof(v);
    else if (o == "-x") p._drop_out_length_w_gap = atoi(v);
    else if (o == "-y") p._drop_out_length_wo_gap = atoi(v);
    else if (o == "-m") p._min_helix_length = atoi(v);
    else { fprintf(stderr, "harness: unknown option %s\n", o.c_str()); return 1; }
  }
  p.SetDbParameters();

  RnaInteractionSearch ris;
  {
    DbReader rd(db, p.GetHashSize());
    ris._dbs = rd.LoadDatabases();
  }

  FILE *f = fopen(out, "wb");
  put<uint32_t>(f, 0x53544731u); // "STG1"
  put<int32_t>(f, (int32_t)seqs.size());
  put<int32_t>(f, (int32_t)ris._dbs.size());
  for (size_t qi = 0; qi < seqs.size(); qi++) {
    std::vector<float> acc, cond;
    std::vector<unsigned char> enc;
    std::vector<int> sa;
    ris.CalculateAccessibility(p, seqs[qi], acc, cond);
    ris.ConstructSuffixArray(p, seqs[qi], enc, sa);
    for (size_t pg = 0; pg < ris._dbs.size(); pg++) {
      std::vector<Hit> hits;
      put<int32_t>(f, (int32_t)qi);
      put<int32_t>(f, (int32_t)pg);
      ris.SearchSeed(p, hits, enc, sa, acc, cond, (int)pg);
      put_hits(f, hits);
      ris.ExtendWithoutGap(p, hits, enc, acc, cond, (int)pg);
      put_hits(f, hits);
      ris.ExtendWithGap(p, hits, enc, acc, cond, (int)pg);
      put_hits(f, hits);
    }
  }
  fclose(f);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: ref_harness tables|raccess|raccess_dbg|stages|sa ...\n");
    return 2;
  }
  std::string c = argv[1];
  if (c == "tables" && argc == 3) return cmd_tables(argv[2]);
  if (c == "raccess" && argc == 6) return cmd_raccess(argc, argv, false);
  if (c == "raccess_dbg" && argc == 6) return cmd_raccess(argc, argv, true);
  if (c == "sa" && argc == 5) return cmd_sa(argc, argv);
  if (c == "stages" && argc >= 5) return cmd_stages(argc, argv);
  fprintf(stderr, "harness: bad arguments\n");
  return 2;
}
