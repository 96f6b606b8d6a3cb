// pRIblast-hip: drop-in for the reference's `ris` sub-command on top of the C ABI.
//
// Command line, defaults, error texts and the output format follow the reference
// (main.cpp:36-111, 148-175; rna_interaction_search_parameters.cpp:33-114;
// rna_interaction_search.cpp:322-369, 445-476).  `-a` is accepted and ignored: the MPI/OpenMP query
// schedulers are replaced by batched GPU stages.  Queries are sorted by length, longest first, as the
// reference does before dealing them (utils.cpp:56-63, rna_interaction_search.cpp:145-153), cut into
// batches of PRB_BATCH, and
//   * one process: dealt from a counter to the workers named by PRB_DEVICES (default: device 0), one
//     host thread each; a worker computes the accessibilities of its next batch under a second
//     context while the current batch is searched; one writer thread prints finished batches in order;
//   * one process per GPU (WORLD_SIZE / RANK / LOCAL_RANK in the environment, as torchrun or mpirun
//     set them): batch b goes to rank b mod WORLD_SIZE, and after every round of batches the final
//     hits are gathered on rank 0 over RCCL (prb_gather_hits), which writes the lines - the
//     replacement of the reference's MPI token ring (rna_interaction_search.cpp:426-487).  The 128-byte
//     RCCL id travels through a file in the `-p` temporary directory (default: next to the output).
//
// Two additions to the reference's surface (the output path): `ris -b` writes
// the hits as binary records instead of text (no number formatting on the search path), and the
// `txt` sub-command turns such a file into exactly the text `ris` would have written.
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <numeric>
#include <condition_variable>
#include <deque>
#include <map>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
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
            "  Environment: PRB_DEVICES=0,1,..  GPUs (workers) of this process;  PRB_BATCH=N  queries per batch [default 2048];\n"
            "               WORLD_SIZE / RANK / LOCAL_RANK  one process per GPU, final hits gathered on rank 0 over RCCL");
}

struct Args {
  std::string in, out, db, tmp;
  prb_ris_opts o;
  bool binary = false;
};

[[noreturn]] void die(const std::string &msg) {
  std::fprintf(stderr, "%s\n", msg.c_str());
  std::exit(1);
}

struct Worker {
  int device;
  prb_ctx *ctx = nullptr, *prep_ctx = nullptr; // prep_ctx: accessibilities of the next batch, on a stream of its own
  prb_db *db = nullptr;
};

// The hit sets of one batch (one per database page), waiting to be written.
struct BatchJob {
  size_t index = 0, nq = 0;
  std::vector<prb_hitset *> pages;
  std::vector<std::string> names; // of its queries, in the order of their indices in the hit records
  std::vector<int32_t> qlen_unmasked;
};

// A batch with its accessibilities computed, ready to be searched.
struct Prepared {
  prb_qbatch *qb = nullptr;
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

// encode + suffix arrays + accessibilities of the queries `idx` under context c
Prepared prepare_batch(prb_ctx *c, const std::vector<std::string> &seqs, const std::vector<size_t> &idx, int W, int delta,
                       int repeat_flag, const prb_db *db, const prb_ris_opts *opts) {
  std::string cat;
  std::vector<int64_t> off(idx.size() + 1, 0);
  for (size_t k = 0; k < idx.size(); k++) {
    cat += seqs[idx[k]];
    off[k + 1] = (int64_t)cat.size();
  }
  Prepared p;
  if (prb_qbatch_create(c, (int32_t)idx.size(), cat.data(), off.data(), repeat_flag, &p.qb)) die(prb_last_error());
  // the seed DFS against the first page needs no GPU: it runs on host threads beside the accessibilities (and,
  // for a batch prepared ahead, beside the previous batch's search)
  if (db && prb_qbatch_seed_search_begin(c, p.qb, db, 0, opts)) die(prb_last_error());
  if (prb_qbatch_accessibility(c, p.qb, W, delta)) die(prb_last_error());
  p.qlen_unmasked.resize(idx.size());
  for (size_t q = 0; q < idx.size(); q++) p.qlen_unmasked[q] = prb_qbatch_length_unmasked(p.qb, (int32_t)q);
  return p;
}

// the search stages of a prepared batch against every page; the hit sets go to the writer
void search_batch(Worker &w, const Args &a, Prepared &p, int npages, std::vector<prb_hitset *> &pages) {
  for (int page = 0; page < npages; page++) {
    prb_hitset *hs = nullptr;
    if (prb_search_page(w.ctx, p.qb, w.db, page, &a.o, 3, &hs)) die(prb_last_error());
    pages.push_back(hs);
  }
  prb_qbatch_destroy(p.qb);
  p.qb = nullptr;
}

// The RCCL id of a multi-process run travels through files in the `-p` directory.  A file left behind by a run that
// died must never be taken for this run's (ncclCommInitRank would wait for ever on a stale id), and file times say
// nothing reliable about that, so the exchange is a handshake: every other rank writes a random nonce to
// `<path>.hello.<rank>`; rank 0 publishes `<path>` = the id followed by the nonces it has seen, and publishes again
// whenever a hello file changes (a stale hello of a dead run is simply overwritten by the live rank); a rank takes
// the id only from a file that carries ITS nonce.  Rank 0 keeps publishing until its communicator exists - which is
// when every rank has the id - and then removes the files.  A rank that fails kills the job under torchrun / mpirun;
// started by hand, the others give up after PRB_RENDEZVOUS_TIMEOUT seconds (default 300) instead of waiting for it.
struct Rendezvous {
  std::string path;
  int world = 1, rank = 0;
  char id[PRB_COMM_ID_BYTES] = {};
  std::thread publisher;
  std::atomic<bool> stop{false};

  static uint64_t nonce() {
    uint64_t v = 0;
    if (std::FILE *f = std::fopen("/dev/urandom", "rb")) {
      if (std::fread(&v, sizeof v, 1, f) != 1) v = 0;
      std::fclose(f);
    }
    return v ? v : ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  }
  static bool write_atomic(const std::string &p, const void *data, size_t n) {
    const std::string tmp = p + ".tmp";
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, n, f) == n;
    return (std::fclose(f) == 0) && ok && std::rename(tmp.c_str(), p.c_str()) == 0;
  }
  static bool read_exact(const std::string &p, void *data, size_t n) {
    struct stat st;
    if (stat(p.c_str(), &st) != 0 || (size_t)st.st_size != n) return false;
    std::FILE *f = std::fopen(p.c_str(), "rb");
    if (!f) return false;
    const bool ok = std::fread(data, 1, n, f) == n;
    std::fclose(f);
    return ok;
  }
  static int timeout_s() {
    const char *e = std::getenv("PRB_RENDEZVOUS_TIMEOUT");
    return e ? std::max(1, std::atoi(e)) : 300;
  }
  std::string hello(int r) const { return path + ".hello." + std::to_string(r); }

  void begin(const std::string &p, int world_, int rank_) {
    path = p;
    world = world_;
    rank = rank_;
    const size_t full = PRB_COMM_ID_BYTES + 8 * (size_t)(world - 1);
    if (rank == 0) {
      std::remove(path.c_str());
      std::remove((path + ".tmp").c_str());
      if (prb_comm_unique_id(id)) die(std::string("Error: ") + prb_last_error());
      if (world == 1) return;
      publisher = std::thread([this, full] {
        std::vector<uint64_t> seen((size_t)(world - 1), 0), now((size_t)(world - 1), 0);
        std::vector<char> buf(full);
        while (!stop.load()) {
          bool all = true;
          for (int r = 1; r < world; r++) all = read_exact(hello(r), &now[(size_t)r - 1], 8) && now[(size_t)r - 1] != 0 && all;
          if (all && now != seen) {
            std::memcpy(buf.data(), id, PRB_COMM_ID_BYTES);
            std::memcpy(buf.data() + PRB_COMM_ID_BYTES, now.data(), 8 * now.size());
            if (!write_atomic(path, buf.data(), full)) die("Error: can't write the rendezvous file " + path);
            seen = now;
          }
          std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
      });
      return;
    }
    const uint64_t mine = nonce();
    if (!write_atomic(hello(rank), &mine, 8)) die("Error: can't write the rendezvous file " + hello(rank));
    std::vector<char> buf(full);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      uint64_t got = 0;
      if (read_exact(path, buf.data(), full)) std::memcpy(&got, buf.data() + PRB_COMM_ID_BYTES + 8 * (size_t)(rank - 1), 8);
      if (got == mine) break;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s()))
        die("Error: no rendezvous file " + path + " from rank 0 (is it running? a rank that fails must end the whole job)");
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    std::memcpy(id, buf.data(), PRB_COMM_ID_BYTES);
  }
  // after prb_comm_create returned on this rank (on rank 0: every rank has joined the communicator, so has the id)
  void end() {
    if (rank != 0) return;
    stop.store(true);
    if (publisher.joinable()) publisher.join();
    std::remove(path.c_str());
    for (int r = 1; r < world; r++) std::remove(hello(r).c_str());
  }
};

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
    case 'p': a.tmp = optarg; break;
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
  // one process per GPU?  (torchrun / mpirun style environment)
  auto env_int = [](const char *a, const char *b, int dflt) {
    const char *e = std::getenv(a);
    if (!e && b) e = std::getenv(b);
    return e ? std::atoi(e) : dflt;
  };
  const int world = std::max(1, env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1));
  const int rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", 0);
  const bool rank_mode = world > 1 || std::getenv("PRB_FORCE_COMM") != nullptr;
  if (rank < 0 || rank >= world) die("Error: RANK outside WORLD_SIZE");
  if (rank_mode) {
    const int local = env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", rank);
    const int dev = devices.empty() ? local : devices[(size_t)local % devices.size()];
    devices.assign(1, dev);
    // The ranks of a node share its CPUs: every rank runs suffix arrays + a seed DFS, rank 0 alone writes the lines of all
    // of them.  Unless the user says otherwise: half of the CPUs split over the ranks for the former, the other half for
    // rank 0's formatting (WORLD_SIZE counts ranks on other nodes too - then this errs on the small side).
    const int budget = prb_cpu_budget();
    if (world > 1) {
      setenv("PRB_HOST_THREADS", std::to_string(std::max(2, std::min(32, budget / (2 * world)))).c_str(), 0);
      if (rank == 0) setenv("PRB_FORMAT_THREADS", std::to_string(std::max(2, std::min(32, budget / 2))).c_str(), 0);
    }
  }
  if (devices.empty()) devices.push_back(0);
  std::vector<Worker> workers(devices.size());
  int hash_size = 0, repeat_flag = 0, W = 0, delta = 0, npages = 0;
  const bool prefetch = !std::getenv("PRB_NO_PREFETCH");
  for (size_t k = 0; k < devices.size(); k++) {
    workers[k].device = devices[k];
    if (prb_ctx_create(devices[k], nullptr, &workers[k].ctx)) die(std::string("Error: ") + prb_last_error());
    if (prefetch && prb_ctx_create(devices[k], nullptr, &workers[k].prep_ctx)) die(std::string("Error: ") + prb_last_error());
    if (prb_db_open(workers[k].ctx, a.db.c_str(), &workers[k].db)) die(prb_last_error());
  }
  prb_comm *comm = nullptr;
  if (rank_mode) {
    // one file name per launch where the launcher names the launch (torchrun: TORCHELASTIC_RUN_ID; Open MPI: its job id;
    // or PRB_RUN_ID), else per MASTER_PORT
    std::string token;
    for (const char *k : {"PRB_RUN_ID", "TORCHELASTIC_RUN_ID", "OMPI_MCA_ess_base_jobid", "PMIX_NAMESPACE", "MASTER_PORT"})
      if (const char *e = std::getenv(k); e && *e && std::strcmp(e, "none") != 0) {
        token = e;
        break;
      }
    for (char &ch : token)
      if (!std::isalnum((unsigned char)ch) && ch != '-' && ch != '_') ch = '_';
    const std::string path = (a.tmp.empty() ? a.out : a.tmp + "/prb") + ".rccl_id." + (token.empty() ? "0" : token);
    Rendezvous rdv;
    rdv.begin(path, world, rank);
    if (prb_comm_create(workers[0].ctx, world, rank, rdv.id, &comm)) die(std::string("Error: ") + prb_last_error());
    rdv.end();
  }
  const bool writer_rank = rank == 0;
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

  std::FILE *out = std::fopen(writer_rank ? a.out.c_str() : "/dev/null", a.binary ? "wb" : "w");
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

  // Queries per batch: PRB_BATCH, else large batches (full launches of the small-list kernels) but at least four per
  // worker / rank, so that every GPU has work and the last round is short (the reference deals single queries)
  const char *benv = std::getenv("PRB_BATCH");
  const size_t consumers = rank_mode ? (size_t)world : devices.size();
  const size_t batch = benv ? (size_t)std::max(1, std::atoi(benv))
                            : std::max<size_t>(16, std::min<size_t>(2048, (seqs.size() + 4 * consumers - 1) / (4 * consumers)));
  // longest first (stable: equal lengths keep their FASTA order), then batches in that order
  std::vector<size_t> order(seqs.size());
  std::iota(order.begin(), order.end(), (size_t)0);
  std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return seqs[x].size() > seqs[y].size(); });
  const size_t nb = (seqs.size() + batch - 1) / batch;
  auto batch_idx = [&](size_t b) { return std::vector<size_t>(order.begin() + b * batch, order.begin() + std::min(seqs.size(), (b + 1) * batch)); };
  // The writer thread turns finished jobs into text in job order while the GPUs already work on the
  // next ones (a bounded number of finished jobs wait for it).  A job is a batch - or, with one process
  // per GPU, a round of WORLD_SIZE batches gathered on rank 0.
  const size_t njobs = rank_mode ? (nb + world - 1) / world : nb;
  std::mutex mu;
  std::condition_variable cv;
  std::map<size_t, BatchJob> done;
  size_t written = 0;
  int64_t total_hits = 0;
  std::thread writer([&] {
    int64_t id = 0;
    for (size_t b = 0; b < njobs && writer_rank; b++) {
      BatchJob job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done.count(b) != 0; });
        job = std::move(done[b]);
        done.erase(b);
      }
      BatchView v;
      v.nq = job.nq;
      v.names = job.names.data();
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
  auto submit = [&](size_t index, BatchJob &&job) {
    {
      std::lock_guard<std::mutex> lk(mu);
      done[index] = std::move(job);
    }
    cv.notify_all();
  };
  auto names_of = [&](const std::vector<size_t> &idx, std::vector<std::string> &dst) {
    for (size_t i : idx) dst.push_back(names[i]);
  };
  // A worker: its batches one after the other; the accessibilities of the next one are computed by a
  // helper thread under the worker's second context while this one is searched.
  auto work = [&](Worker &w, const std::function<bool(size_t &)> &next_batch, const std::function<void(size_t, Prepared *, std::vector<prb_hitset *> &)> &finish) {
    size_t b = 0, bn = 0;
    bool have = next_batch(b);
    Prepared cur;
    if (have) cur = prepare_batch(w.prep_ctx ? w.prep_ctx : w.ctx, seqs, batch_idx(b), W, delta, repeat_flag, w.db, &a.o);
    while (have) {
      const bool more = next_batch(bn);
      Prepared nxt;
      std::thread helper;
      if (more && w.prep_ctx) helper = std::thread([&] { nxt = prepare_batch(w.prep_ctx, seqs, batch_idx(bn), W, delta, repeat_flag, w.db, &a.o); });
      std::vector<prb_hitset *> pages;
      search_batch(w, a, cur, npages, pages);
      finish(b, &cur, pages);
      if (helper.joinable()) helper.join();
      else if (more) nxt = prepare_batch(w.ctx, seqs, batch_idx(bn), W, delta, repeat_flag, w.db, &a.o);
      cur = std::move(nxt);
      b = bn;
      have = more;
    }
  };
  if (!rank_mode) {
    std::atomic<size_t> next{0};
    std::vector<std::thread> threads;
    for (auto &w : workers)
      threads.emplace_back([&, pw = &w] {
        work(*pw,
             [&](size_t &b) {
               b = next.fetch_add(1);
               return b < nb;
             },
             [&](size_t b, Prepared *p, std::vector<prb_hitset *> &pages) {
               {
                 std::unique_lock<std::mutex> lk(mu); // bound the hit sets held in memory
                 cv.wait(lk, [&] { return b < written + 2 * workers.size() + 1; });
               }
               BatchJob job;
               job.index = b;
               job.pages = pages;
               job.qlen_unmasked = p->qlen_unmasked;
               names_of(batch_idx(b), job.names);
               job.nq = job.names.size();
               submit(b, std::move(job));
             });
      });
    for (auto &t : threads) t.join();
  } else {
    // round t: batch t * world + rank; every rank takes part in every round's gather, with or without a batch
    size_t t_next = 0;
    // The gathers run on a thread of their own, in round order, behind the search of the next batch (prb_gather_hits works
    // on the communicator's own stream): rank 0, which receives everybody's records and copies them to the host, then takes no
    // longer over a round than the others.  At most two rounds wait for it (their records stay in HBM until then).
    struct GatherJob {
      size_t t = 0;
      bool has = false;
      std::vector<int32_t> qlen;
      std::vector<prb_hitset *> pages;
    };
    std::mutex gmu;
    std::condition_variable gcv;
    std::deque<GatherJob> gq;
    bool gclosed = false;
    auto gather_round_now = [&](GatherJob &gj) {
      BatchJob job;
      job.index = gj.t;
      const size_t t = gj.t;
      std::vector<prb_hitset *> &pages = gj.pages;
      for (int page = 0; page < npages; page++) {
        prb_hitset *mine = pages.empty() ? nullptr : pages[(size_t)page], *all = nullptr;
        if (prb_gather_hits(comm, mine, gj.has ? (int32_t)gj.qlen.size() : 0, gj.has ? gj.qlen.data() : nullptr, 0, &all))
          die(std::string("Error: ") + prb_last_error());
        if (mine) prb_hitset_free(mine);
        if (all) job.pages.push_back(all);
      }
      if (!writer_rank) return;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return t < written + 3; });
      }
      int32_t nr = 0;
      const int32_t *nq_of = nullptr, *ql = nullptr;
      if (!job.pages.empty() && prb_hitset_gathered_queries(job.pages[0], &nr, &nq_of, &ql)) die(std::string("Error: ") + prb_last_error());
      for (int r = 0; r < nr; r++)
        if (nq_of[r]) names_of(batch_idx(t * (size_t)world + (size_t)r), job.names);
      job.nq = job.names.size();
      job.qlen_unmasked.assign(ql, ql + job.nq);
      submit(t, std::move(job));
    };
    std::thread gatherer([&] {
      for (;;) {
        GatherJob gj;
        {
          std::unique_lock<std::mutex> lk(gmu);
          gcv.wait(lk, [&] { return !gq.empty() || gclosed; });
          if (gq.empty()) return;
          gj = std::move(gq.front());
          gq.pop_front();
        }
        gcv.notify_all();
        gather_round_now(gj);
      }
    });
    auto gather_round = [&](size_t t, Prepared *p, std::vector<prb_hitset *> &pages) {
      GatherJob gj;
      gj.t = t;
      gj.has = p != nullptr;
      if (p) gj.qlen = p->qlen_unmasked;
      gj.pages = pages;
      std::unique_lock<std::mutex> lk(gmu);
      gcv.wait(lk, [&] { return gq.size() < 2; });
      gq.push_back(std::move(gj));
      lk.unlock();
      gcv.notify_all();
    };
    work(workers[0],
         [&](size_t &b) {
           b = t_next * (size_t)world + (size_t)rank;
           t_next++;
           return b < nb;
         },
         [&](size_t b, Prepared *p, std::vector<prb_hitset *> &pages) { gather_round(b / (size_t)world, p, pages); });
    // rounds in which this rank has no batch left (the last one, when nb is not a multiple of world)
    const size_t mine_rounds = nb > (size_t)rank ? (nb - (size_t)rank + (size_t)world - 1) / (size_t)world : 0;
    std::vector<prb_hitset *> none;
    for (size_t t = mine_rounds; t < njobs; t++) gather_round(t, nullptr, none);
    {
      std::lock_guard<std::mutex> lk(gmu);
      gclosed = true;
    }
    gcv.notify_all();
    gatherer.join();
  }
  writer.join();
  if (a.binary) {
    put<int64_t>(out, kEnd);
    put<int64_t>(out, total_hits);
  }
  if (std::fclose(out)) die("Error: can't write the output file");
  if (comm) prb_comm_destroy(comm);
  for (auto &w : workers) {
    prb_db_close(w.db);
    if (w.prep_ctx) prb_ctx_destroy(w.prep_ctx);
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
