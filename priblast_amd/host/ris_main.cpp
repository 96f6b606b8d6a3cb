// pRIblast-hip: drop-in for the reference's `ris` sub-command on top of the C ABI.
//
// Command line, defaults, error texts and the output format follow the reference
// (main.cpp:36-111, 148-175; rna_interaction_search_parameters.cpp:33-114;
// rna_interaction_search.cpp:322-369, 445-476).  `-a` and `-p` are accepted and ignored: the
// MPI/OpenMP query schedulers are replaced by batched GPU stages; queries are dealt in
// batches to the GPUs named by PRB_DEVICES (default: device 0).
//
// Two additions to the reference's surface (the output path): `ris -b` writes
// the hits as binary records instead of text (no number formatting on the search path), and the
// `txt` sub-command turns such a file into exactly the text `ris` would have written.
#include <getopt.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/priblast_hip.h"
#include "fasta.hpp"
#include "output.hpp"

namespace {

using prb::BatchView;
using prb::LineSink;
using prb::PageHits;
using prb::SeqTable;

void usage() {
  std::puts("pRIblast-hip - RNA-RNA interaction search (ris step of pRIblast) on AMD Instinct GPUs\n"
            "\n"
            "pRIblast-hip ris -i InputFastaFile -o OutputFileName -d DatabaseFileName\n"
            "             [-l MaxSeedLength] [-e HybridizationEnergyThreshold] [-f InteractionEnergyThreshold]\n"
            "             [-x DropOutLengthInGappedExtension] [-y DropOutLengthInUngappedExtension]\n"
            "             [-g OutputEnergyThreshold] [-s OutputStyle] [-a ParallelAlgorithm] [-p TemporaryPath]\n"
            "\n"
            "  Options:\n"
            "(Required)\n"
            "    -i STR    RNA sequences in FASTA format\n"
            "    -d STR    Input database in pRIblast format\n"
            "    -o STR    Output file name\n"
            "\n"
            "(Optional)\n"
            "    -l INT    Max size of seed length [default:20]\n"
            "    -e DBL    Hybridization energy threshold for seed search [default: -6.0]\n"
            "    -f DBL    Interaction energy threshold for removal of the interaction candidate before gapped "
            "extension [default: -4.0]\n"
            "    -x INT    Dropout Length in gapped extension [default:16]\n"
            "    -y INT    Dropout Length in ungapped extension [default:5]\n"
            "    -g DBL    Energy threshold for output [default:-8.0]\n"
            "    -s INT    Designation of output format style. 0:simplified output, or 1:detailed output [default:0]\n"
            "    -m INT    Minimum helix length in gapped extension [default:3]\n"
            "    -a STR    accepted for compatibility (block, area, dynamic); ignored\n"
            "    -p STR    accepted for compatibility; ignored\n"
            "    -b        write binary hit records instead of text; `pRIblast-hip txt -i FILE -o TEXT` converts\n"
            "\n"
            "  Environment: PRB_DEVICES=0,1,..  GPUs to use;  PRB_BATCH=N  queries per batch [default 2048]");
}

struct Args {
  std::string in, out, db;
  prb_ris_opts o;
  bool binary = false;
};

[[noreturn]] void die(const std::string &msg) {
  std::fprintf(stderr, "%s\n", msg.c_str());
  std::exit(1);
}

struct Worker {
  int device;
  prb_ctx *ctx = nullptr;
  prb_db *db = nullptr;
};

// The hit sets of one batch (one per database page), waiting to be written.
struct BatchJob {
  size_t index = 0, b0 = 0, nq = 0;
  Worker *w = nullptr;
  std::vector<prb_hitset *> pages;
  std::vector<int32_t> qlen_unmasked;
};

// ---- binary hit file (little-endian, the layouts of include/priblast_hip.h) ----------------------
//   "PRBHITS\1" | i32 output_style | i32 npages | str header (the three text lines)
//   per page: i32 nseq, then per sequence i32 length, i32 length_unmasked, i32 start_pos, str name
//   blocks:   i64 'B' | i64 nq | per query: str name, i32 length_unmasked
//             | per page: i64 nhits, i64 npairs, prb_hit[nhits], int32[2 * npairs]
//   trailer:  i64 'E' | i64 total hits
//   str = i32 length + bytes.  prb_hit.bp_offset indexes the pair array of its own block and page.
constexpr char kMagic[8] = {'P', 'R', 'B', 'H', 'I', 'T', 'S', 1};
constexpr int64_t kBlock = 'B', kEnd = 'E';

void put(std::FILE *f, const void *p, size_t n) {
  if (n && std::fwrite(p, 1, n, f) != n) die("Error: can't write the output file");
}
template <class T> void put(std::FILE *f, T v) { put(f, &v, sizeof v); }
void put_str(std::FILE *f, const std::string &s) {
  put<int32_t>(f, (int32_t)s.size());
  put(f, s.data(), s.size());
}
void get(std::FILE *f, void *p, size_t n) {
  if (n && std::fread(p, 1, n, f) != n) die("Error: truncated binary hit file");
}
template <class T> T get(std::FILE *f) {
  T v;
  get(f, &v, sizeof v);
  return v;
}
std::string get_str(std::FILE *f) {
  const int32_t n = get<int32_t>(f);
  if (n < 0 || n > (1 << 28)) die("Error: corrupt binary hit file");
  std::string s((size_t)n, '\0');
  get(f, s.data(), s.size());
  return s;
}

void write_binary_head(std::FILE *f, int output_style, const std::string &header, const std::vector<SeqTable> &tabs) {
  put(f, kMagic, sizeof kMagic);
  put<int32_t>(f, output_style);
  put<int32_t>(f, (int32_t)tabs.size());
  put_str(f, header);
  for (const SeqTable &t : tabs) {
    put<int32_t>(f, (int32_t)t.names.size());
    for (size_t i = 0; i < t.names.size(); i++) {
      put<int32_t>(f, t.len[i]);
      put<int32_t>(f, t.len_unmasked[i]);
      put<int32_t>(f, t.start_pos[i]);
      put_str(f, t.names[i]);
    }
  }
}

int64_t write_binary_batch(const BatchView &v, std::FILE *f) {
  put<int64_t>(f, kBlock);
  put<int64_t>(f, (int64_t)v.nq);
  for (size_t q = 0; q < v.nq; q++) {
    put_str(f, v.names[q]);
    put<int32_t>(f, v.qlen_unmasked[q]);
  }
  int64_t total = 0;
  for (const PageHits &p : v.pages) {
    put<int64_t>(f, p.n);
    put<int64_t>(f, p.nbp);
    put(f, p.h, (size_t)p.n * sizeof(prb_hit));
    put(f, p.bp, (size_t)p.nbp * 2 * sizeof(int32_t));
    total += p.n;
  }
  return total;
}

// `txt` sub-command: binary hit file -> the text `ris` writes.
int txt_main(int argc, char **argv) {
  std::string in, out;
  int c;
  while ((c = getopt(argc, argv, "i:o:")) != -1) {
    switch (c) {
    case 'i': in = optarg; break;
    case 'o': out = optarg; break;
    default: die("Error: invalid argument");
    }
  }
  std::FILE *f = std::fopen(in.c_str(), "rb");
  if (!f) die("Error: can't open input_file: " + in);
  char magic[8];
  get(f, magic, sizeof magic);
  if (std::memcmp(magic, kMagic, sizeof kMagic)) die("Error: " + in + " is not a binary hit file");
  const int output_style = get<int32_t>(f);
  const int np = get<int32_t>(f);
  if (np < 0 || np > (1 << 24)) die("Error: corrupt binary hit file");
  const std::string header = get_str(f);
  std::vector<SeqTable> tabs((size_t)np);
  for (SeqTable &t : tabs) {
    const int32_t nseq = get<int32_t>(f);
    if (nseq < 0) die("Error: corrupt binary hit file");
    for (int32_t i = 0; i < nseq; i++) {
      t.len.push_back(get<int32_t>(f));
      t.len_unmasked.push_back(get<int32_t>(f));
      t.start_pos.push_back(get<int32_t>(f));
      t.names.push_back(get_str(f));
    }
  }
  std::FILE *o = std::fopen(out.c_str(), "w");
  if (!o) die("Error: can't open output_file: " + out);
  put(o, header.data(), header.size());
  if (std::fflush(o)) die("Error: can't write the output file");
  LineSink sink;
  sink.fd = fileno(o);
  int64_t id = 0;
  for (;;) {
    const int64_t tag = get<int64_t>(f);
    if (tag == kEnd) {
      if (get<int64_t>(f) != id) die("Error: corrupt binary hit file (hit count)");
      break;
    }
    if (tag != kBlock) die("Error: corrupt binary hit file");
    const int64_t nq = get<int64_t>(f);
    if (nq < 0 || nq > (1 << 28)) die("Error: corrupt binary hit file");
    std::vector<std::string> names((size_t)nq);
    std::vector<int32_t> qlen((size_t)nq);
    for (int64_t q = 0; q < nq; q++) {
      names[q] = get_str(f);
      qlen[q] = get<int32_t>(f);
    }
    std::vector<std::vector<prb_hit>> hits((size_t)np);
    std::vector<std::vector<int32_t>> pairs((size_t)np);
    BatchView v;
    v.nq = (size_t)nq;
    v.names = names.data();
    v.qlen_unmasked = qlen.data();
    for (int p = 0; p < np; p++) {
      const int64_t n = get<int64_t>(f), nbp = get<int64_t>(f);
      if (n < 0 || nbp < 0) die("Error: corrupt binary hit file");
      hits[p].resize((size_t)n);
      pairs[p].resize((size_t)nbp * 2);
      get(f, hits[p].data(), (size_t)n * sizeof(prb_hit));
      get(f, pairs[p].data(), (size_t)nbp * 2 * sizeof(int32_t));
      for (const prb_hit &x : hits[p])
        if (x.query < 0 || x.query >= nq || x.db_id < 0 || (size_t)x.db_id >= tabs[p].names.size() || x.bp_count < 0 ||
            x.bp_offset < 0 || x.bp_offset + x.bp_count > nbp)
          die("Error: corrupt binary hit file (record)");
      PageHits ph;
      ph.h = hits[p].data();
      ph.n = n;
      ph.bp = pairs[p].data();
      ph.nbp = nbp;
      v.pages.push_back(ph);
    }
    id = prb::format_batch(v, tabs, output_style, id, sink, prb::format_threads());
    if (id < 0) die("Error: can't write the output file");
  }
  std::fclose(f);
  if (std::fclose(o)) die("Error: can't write the output file");
  return 0;
}

// One batch through the GPU stages; the hit sets go to the writer.
void run_batch(Worker &w, const Args &a, const std::vector<std::string> &seqs, size_t b0, size_t b1, int W, int delta,
               int repeat_flag, int npages, BatchJob &job) {
  std::string cat;
  std::vector<int64_t> off(b1 - b0 + 1, 0);
  for (size_t i = b0; i < b1; i++) {
    cat += seqs[i];
    off[i - b0 + 1] = (int64_t)cat.size();
  }
  prb_qbatch *qb = nullptr;
  if (prb_qbatch_create(w.ctx, (int32_t)(b1 - b0), cat.data(), off.data(), repeat_flag, &qb)) die(prb_last_error());
  if (prb_qbatch_accessibility(w.ctx, qb, W, delta)) die(prb_last_error());
  job.b0 = b0;
  job.nq = b1 - b0;
  job.w = &w;
  job.qlen_unmasked.resize(job.nq);
  for (size_t q = 0; q < job.nq; q++) job.qlen_unmasked[q] = prb_qbatch_length_unmasked(qb, (int32_t)q);
  for (int page = 0; page < npages; page++) {
    prb_hitset *hs = nullptr;
    if (prb_search_page(w.ctx, qb, w.db, page, &a.o, 3, &hs)) die(prb_last_error());
    job.pages.push_back(hs);
  }
  prb_qbatch_destroy(qb);
}

int ris_main(int argc, char **argv) {
  Args a;
  prb_ris_opts_default(&a.o);
  int c;
  while ((c = getopt(argc, argv, "i:o:d:l:e:y:x:f:g:s:m:p:a:b")) != -1) {
    switch (c) {
    case 'i': a.in = optarg; break;
    case 'o': a.out = optarg; break;
    case 'd': a.db = optarg; break;
    case 'l': a.o.max_seed_length = std::atoi(optarg); break;
    case 'e': a.o.hybrid_threshold = std::atof(optarg); break;
    case 'f': a.o.interaction_threshold = std::atof(optarg); break;
    case 'g': a.o.final_threshold = std::atof(optarg); break;
    case 's': a.o.output_style = std::atoi(optarg); break;
    case 'x': a.o.drop_out_w_gap = std::atoi(optarg); break;
    case 'y': a.o.drop_out_wo_gap = std::atoi(optarg); break;
    case 'm': a.o.min_helix_length = std::atoi(optarg); break;
    case 'p': break;
    case 'b': a.binary = true; break;
    case 'a':
      if (std::strcmp(optarg, "block") && std::strcmp(optarg, "area") && std::strcmp(optarg, "dynamic"))
        die("Error: parallel algorithm not supported.");
      break;
    default: die("Error: invalid argument");
    }
  }
  std::vector<std::string> names, seqs;
  std::string err = prb::read_fasta(a.in, names, seqs);
  if (!err.empty()) die(err);

  std::vector<int> devices;
  if (const char *env = std::getenv("PRB_DEVICES")) {
    for (const char *p = env; *p;) {
      devices.push_back(std::atoi(p));
      while (*p && *p != ',') p++;
      if (*p == ',') p++;
    }
  }
  if (devices.empty()) devices.push_back(0);
  std::vector<Worker> workers(devices.size());
  int hash_size = 0, repeat_flag = 0, W = 0, delta = 0, npages = 0;
  for (size_t k = 0; k < devices.size(); k++) {
    workers[k].device = devices[k];
    if (prb_ctx_create(devices[k], nullptr, &workers[k].ctx)) die(std::string("Error: ") + prb_last_error());
    if (prb_db_open(workers[k].ctx, a.db.c_str(), &workers[k].db)) die(prb_last_error());
  }
  prb_db_info(workers[0].db, &hash_size, &repeat_flag, &W, &delta, &npages);

  std::vector<SeqTable> tabs((size_t)npages);
  for (int p = 0; p < npages; p++) {
    int32_t nseq = 0;
    int64_t nchars = 0;
    prb_db_page_info(workers[0].db, p, &nseq, &nchars);
    SeqTable &t = tabs[p];
    t.names.resize(nseq);
    t.len.resize(nseq);
    t.len_unmasked.resize(nseq);
    t.start_pos.resize(nseq);
    for (int32_t i = 0; i < nseq; i++) {
      t.names[i] = prb_db_seq_name(workers[0].db, p, i);
      prb_db_seq_lengths(workers[0].db, p, i, &t.len[i], &t.len_unmasked[i], &t.start_pos[i]);
    }
  }

  std::FILE *out = std::fopen(a.out.c_str(), a.binary ? "wb" : "w");
  if (!out) die("Error: can't open output_file: " + a.out);
  // MergeOutput header, rna_interaction_search.cpp:445-463
  std::string header = "RIblast ris result\n";
  {
    char buf[256];
    header += "input:" + a.in + ",database:" + a.db;
    std::snprintf(buf, sizeof buf,
                  ",RepeatFlag:%d,MaximalSpan:%d,MinAccessibleLength:%d,MaxSeedLength:%d,"
                  "InteractionEnergyThreshold:%g,HybridEnergyThreshold:%g,FinalThreshold:%g,DropOutLengthWoGap:%d,"
                  "DropOutLengthWGap:%d\n",
                  repeat_flag, W, delta, a.o.max_seed_length, a.o.interaction_threshold, a.o.hybrid_threshold,
                  a.o.final_threshold, a.o.drop_out_wo_gap, a.o.drop_out_w_gap);
    header += buf;
  }
  header += "Id,Query name, Query Length, Target name, Target Length, Accessibility Energy, Hybridization Energy, "
            "Interaction Energy, BasePair\n";
  if (a.binary) write_binary_head(out, a.o.output_style, header, tabs);
  else put(out, header.data(), header.size());
  if (std::fflush(out)) die("Error: can't write the output file");
  LineSink sink; // text lines go straight to the descriptor (nothing else is written through `out` meanwhile)
  sink.fd = fileno(out);

  const char *benv = std::getenv("PRB_BATCH");
  const size_t batch = std::max(1, benv ? std::atoi(benv) : 2048);
  const size_t nb = (seqs.size() + batch - 1) / batch;
  // GPU workers take batches from a counter; a writer thread turns finished batches into text in
  // batch order while the GPUs already work on the next ones (at most two finished batches per
  // worker wait for it).
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::condition_variable cv;
  std::map<size_t, BatchJob> done;
  size_t written = 0;
  int64_t total_hits = 0;
  std::thread writer([&] {
    int64_t id = 0;
    for (size_t b = 0; b < nb; b++) {
      BatchJob job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done.count(b) != 0; });
        job = std::move(done[b]);
        done.erase(b);
      }
      BatchView v;
      v.nq = job.nq;
      v.names = names.data() + job.b0;
      v.qlen_unmasked = job.qlen_unmasked.data();
      for (prb_hitset *hs : job.pages) {
        PageHits ph;
        ph.n = prb_hitset_size(hs);
        ph.h = prb_hitset_hits(hs);
        ph.bp = prb_hitset_basepairs(hs, &ph.nbp);
        v.pages.push_back(ph);
      }
      if (a.binary) id += write_binary_batch(v, out);
      else if ((id = prb::format_batch(v, tabs, a.o.output_style, id, sink, prb::format_threads())) < 0)
        die("Error: can't write the output file");
      for (prb_hitset *hs : job.pages) prb_hitset_free(hs);
      {
        std::lock_guard<std::mutex> lk(mu);
        written = b + 1;
      }
      cv.notify_all();
    }
    total_hits = id;
  });
  std::vector<std::thread> threads;
  for (auto &w : workers)
    threads.emplace_back([&, pw = &w] {
      for (;;) {
        const size_t b = next.fetch_add(1);
        if (b >= nb) break;
        {
          std::unique_lock<std::mutex> lk(mu); // bound the hit sets held in memory
          cv.wait(lk, [&] { return b < written + 2 * workers.size() + 1; });
        }
        BatchJob job;
        job.index = b;
        run_batch(*pw, a, seqs, b * batch, std::min(seqs.size(), (b + 1) * batch), W, delta, repeat_flag, npages, job);
        {
          std::lock_guard<std::mutex> lk(mu);
          done[b] = std::move(job);
        }
        cv.notify_all();
      }
    });
  for (auto &t : threads) t.join();
  writer.join();
  if (a.binary) {
    put<int64_t>(out, kEnd);
    put<int64_t>(out, total_hits);
  }
  if (std::fclose(out)) die("Error: can't write the output file");
  for (auto &w : workers) {
    prb_db_close(w.db);
    prb_ctx_destroy(w.ctx);
  }
  return 0;
}

// `db` sub-command (db_construction_parameters.cpp getopt string "i:o:r:s:w:d:c:a:p:",
// defaults db_construction_parameters.hpp:41-52): same files as the reference writes.
int db_main(int argc, char **argv) {
  std::string in, out;
  int repeat = 0, hash = 8, W = 70, delta = 5, chunk = 2147483647;
  int c;
  while ((c = getopt(argc, argv, "i:o:r:s:w:d:c:a:p:")) != -1) {
    switch (c) {
    case 'i': in = optarg; break;
    case 'o': out = optarg; break;
    case 'r': repeat = std::atoi(optarg); break;
    case 's': hash = std::atoi(optarg); break;
    case 'w': W = std::atoi(optarg); break;
    case 'd': delta = std::atoi(optarg); break;
    case 'c': chunk = std::atoi(optarg); break;
    case 'a': case 'p': break;
    default: die("Error: invalid argument");
    }
  }
  if (out.empty()) die("Error: -o option is required");
  if (delta <= 1) die("Error: -d option must be greater than 1");
  if (repeat < 0 || repeat > 2) die("Error: -r option must be 0, 1, or 2");
  std::vector<std::string> names, seqs;
  std::string err = prb::read_fasta(in, names, seqs);
  if (!err.empty()) die(err);
  std::string cat;
  std::vector<int64_t> off(seqs.size() + 1, 0);
  std::vector<const char *> np;
  for (size_t i = 0; i < seqs.size(); i++) {
    cat += seqs[i];
    off[i + 1] = (int64_t)cat.size();
    np.push_back(names[i].c_str());
  }
  prb_ctx *ctx = nullptr;
  const char *env = std::getenv("PRB_DEVICES");
  if (prb_ctx_create(env ? std::atoi(env) : 0, nullptr, &ctx)) die(std::string("Error: ") + prb_last_error());
  if (prb_db_build(ctx, out.c_str(), (int32_t)seqs.size(), np.data(), cat.data(), off.data(), repeat, hash, W, delta, chunk))
    die(std::string("Error: ") + prb_last_error());
  prb_ctx_destroy(ctx);
  return 0;
}

} // namespace

int main(int argc, char **argv) {
  if (argc == 1 || std::strcmp(argv[1], "-h") == 0) {
    usage();
    return 0;
  }
  if (std::strcmp(argv[1], "ris") == 0) return ris_main(argc - 1, argv + 1);
  if (std::strcmp(argv[1], "db") == 0) return db_main(argc - 1, argv + 1);
  if (std::strcmp(argv[1], "txt") == 0) return txt_main(argc - 1, argv + 1);
  std::puts("usage: pRIblast-hip [-h] {db | ris | txt} options");
  return 0;
}
