// C ABI, part 2: database pages, query batches and the search stages
// (SearchSeed / ExtendWithoutGap / ExtendWithGap, rna_interaction_search.cpp:264-320).
#include <omp.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <numeric>
#include <thread>

#include <rocprim/rocprim.hpp>

#include "../../include/priblast_hip.h"
#include "context.hpp"
#include "cpu_budget.hpp"
#include "db_format.hpp"
#include "encoder.hpp"
#include "hitset.hpp"
#include "output.hpp"
#include "search_kernels.hpp"
#include "seed_dfs.hpp"
#include "suffix_array.hpp"

namespace prb {
int run_accessibility(prb_ctx *ctx, int32_t nseq, const char *seqs, const int64_t *in_off, const int32_t *lens,
                      const int64_t *out_off, int W, int delta, float *d_acc, float *d_cond);

struct SearchConstMem {
  DevBuf ints, bulge;
  SearchConst view{};
};

struct PageMem {
  DevBuf seqs, sa, sa_seq, blk_seq, start_pos, seq_length, acc, cond;
  PageDev view{};
};

// buffers reused across prb_search_page calls
struct SearchWs {
  DevBuf cands, row_count, row_off, hitsA, hitsB, hitsC, hitsTmp, kE, kL, kQ, kP, kTmp, kTmp2, idxA, idxB, sortTmp, endKey, pmax,
      state, keep, pending, surv, count, first, gapScratch, overflow, subset, subset2, cidx, ntrace, bpCount, bpOff, bpOut, bpEnds, bpCount2, bpOff2,
      scanTmp, tierOf, tierFin, listA, listB, trace, ntraceFin, packed, row_cand, seed_qacc, resumeSlot, resumePool, resumePool2, resumePool3, resumeCount, frontScratch, accScratch, listC, slowList, slowSlot, slowCnt, slowTrace, keptU, keptFirst, keptTier, keptNtrace, keptTrace;
  // The front of the one-pass seed path for a chunk of candidates - candidates and their pair offsets on the device,
  // query-side window sums, the pairs' keys and values, sorted - in buffers of its own, so that it can be issued for
  // the NEXT sub-batch, on a stream of its own, while this sub-batch is in its last, nearly idle stretch (search_range):
  // ~7 ms of bandwidth-bound work per configs[2] query beside the ~2 ms that the ~150 longest extensions of a query run
  // alone (twice) and the final sort / filter / copies, instead of standing in line behind them.
  struct FrontStage {
    DevBuf cands, seed_qacc, pair0, keyA, keyB, valA, valB, sortTmp;
    PinnedBuf pair0_pin;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool ahead = false;         // holds the first chunk of the sub-batch whose candidates are at `cd`, issued on `stream`
    const CandDev *cd = nullptr;
    int32_t nc = 0;
    int64_t np = 0;
    int init() {
      if (stream) return PRB_OK;
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // (lo = the numerically largest = least urgent)
      if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, lo) != hipSuccess ||
          hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
        set_error("hipStreamCreateWithPriority / hipEventCreate failed (seed front stage)");
        return PRB_ERR_HIP;
      }
      return PRB_OK;
    }
    void release() {
      if (stream) {
        (void)hipStreamSynchronize(stream);
        (void)hipStreamDestroy(stream);
        (void)hipEventDestroy(done);
        stream = nullptr;
      }
      for (DevBuf *b : {&cands, &seed_qacc, &pair0, &keyA, &keyB, &valA, &valB, &sortTmp}) b->release();
      pair0_pin.release();
      ahead = false;
    }
  } front;
  // results leave on a stream of their own: the next sub-batch does not queue behind 60 MB over PCIe
  hipStream_t copy_stream = nullptr;
  hipEvent_t packed_ready = nullptr, copy_done = nullptr;
  bool copy_pending = false;
  int copy_init() {
    if (copy_stream) return PRB_OK;
    if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&packed_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&copy_done, hipEventDisableTiming) != hipSuccess) {
      set_error("hipStreamCreate / hipEventCreate failed (result copies)");
      return PRB_ERR_HIP;
    }
    return PRB_OK;
  }
  PinnedBuf pinned, cand_pinned[2], tb_pinned, pin_hits[2], pin_bp[2];
  bool trim_next = false; // the last sub-batch had a giant list: its buffers are let go before the next one starts
  // every stage buffer over 256 MB (not the front stage's, which may hold the next sub-batch already, nor the packed results,
  // which may still be on their way to the host)
  void trim() {
    for (DevBuf *b : {&cands, &row_count, &row_off, &hitsA, &hitsB, &hitsC, &hitsTmp, &kE, &kL, &kQ, &kP, &kTmp, &kTmp2, &idxA, &idxB,
                      &sortTmp, &endKey, &pmax, &state, &keep, &surv, &first, &gapScratch, &overflow, &subset, &subset2, &cidx, &ntrace,
                      &bpCount, &bpOff, &bpOut, &bpEnds, &bpCount2, &bpOff2, &scanTmp, &tierOf, &tierFin, &listA, &listB, &trace, &ntraceFin,
                      &row_cand, &seed_qacc, &resumePool, &resumePool2, &resumePool3, &frontScratch, &listC, &slowList, &slowSlot, &slowCnt, &slowTrace, &keptU, &keptFirst, &keptTier,
                      &keptNtrace, &keptTrace})
      if (b->cap > ((size_t)256 << 20)) b->release();
    trim_next = false;
  }
  void release() {
    for (DevBuf *b : {&cands, &row_count, &row_off, &hitsA, &hitsB, &hitsC, &hitsTmp, &kE, &kL, &kQ, &kP, &kTmp, &kTmp2, &idxA, &idxB,
                      &sortTmp, &endKey, &pmax, &state, &keep, &pending, &surv, &count, &first, &gapScratch, &overflow,
                      &subset, &subset2, &cidx, &ntrace, &bpCount, &bpOff, &bpOut, &bpEnds, &bpCount2, &bpOff2, &scanTmp, &tierOf, &tierFin, &listA, &listB, &trace, &ntraceFin, &packed, &row_cand, &seed_qacc, &resumeSlot, &resumePool, &resumePool2, &resumePool3, &resumeCount, &frontScratch, &accScratch, &listC, &slowList, &slowSlot, &slowCnt, &slowTrace, &keptU, &keptFirst, &keptTier, &keptNtrace, &keptTrace})
      b->release();
    front.release();
    if (copy_stream) {
      (void)hipStreamSynchronize(copy_stream);
      (void)hipStreamDestroy(copy_stream);
      (void)hipEventDestroy(packed_ready);
      (void)hipEventDestroy(copy_done);
      copy_stream = nullptr;
    }
    pinned.release();
    cand_pinned[0].release();
    cand_pinned[1].release();
    tb_pinned.release();
    for (int i = 0; i < 2; i++) {
      pin_hits[i].release();
      pin_bp[i].release();
    }
  }
};

// Appends the results of finished sub-batches (pinned staging slots filled by asynchronous
// copies on the compute stream) to the hit set while the GPU already works on the next one.
struct Drainer {
  struct Job {
    int slot;
    int64_t nhits, nbp_ints;
  };
  std::vector<prb_hit> *hits;
  std::vector<int32_t> *bp;
  PinnedBuf *pin_hits, *pin_bp; // [2]
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job> jobs;
  bool busy[2] = {false, false}, stop = false, failed = false;
  // what the whole search is expected to deliver (extrapolated by the submitting thread from the queries done so far):
  // the vectors then grow once instead of doubling five times - each doubling of a list of 1e7 hits is a fresh 1 GB
  // mapping, page faults and a copy, on a thread the next staging slot waits for
  std::atomic<size_t> hint_hits{0}, hint_bp{0};

  Drainer(std::vector<prb_hit> *h, std::vector<int32_t> *b, PinnedBuf *ph, PinnedBuf *pb)
      : hits(h), bp(b), pin_hits(ph), pin_bp(pb) {}
  int start() {
    for (int i = 0; i < 2; i++)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return PRB_ERR_HIP;
    th = std::thread([this] { run(); });
    return PRB_OK;
  }
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;
        j = jobs.front();
        jobs.pop_front();
      }
      if (hipEventSynchronize(ev[j.slot]) != hipSuccess) failed = true;
      const prb_hit *src = static_cast<const prb_hit *>(pin_hits[j.slot].p);
      if (hits->capacity() < hits->size() + (size_t)j.nhits)
        hits->reserve(std::max({2 * hits->capacity(), hits->size() + (size_t)j.nhits, hint_hits.load()}));
      hits->insert(hits->end(), src, src + j.nhits);
      const int32_t *bsrc = static_cast<const int32_t *>(pin_bp[j.slot].p);
      if (bp->capacity() < bp->size() + (size_t)j.nbp_ints)
        bp->reserve(std::max({2 * bp->capacity(), bp->size() + (size_t)j.nbp_ints, hint_bp.load()}));
      bp->insert(bp->end(), bsrc, bsrc + j.nbp_ints);
      {
        std::lock_guard<std::mutex> lk(m);
        busy[j.slot] = false;
      }
      cv.notify_all();
    }
  }
  // blocks until the staging slot is no longer read by the background thread
  void acquire(int slot) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return !busy[slot]; });
    busy[slot] = true;
  }
  void submit(const Job &j) {
    {
      std::lock_guard<std::mutex> lk(m);
      jobs.push_back(j);
    }
    cv.notify_all();
  }
  int finish() { // everything submitted is in the hit set afterwards
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
    for (int i = 0; i < 2; i++)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    return failed ? PRB_ERR_HIP : PRB_OK;
  }
};

// wall-clock timer for host-side pieces, reported next to the device stage timers (pseudo-stage names "host_*")
struct HostTimer {
  prb_ctx *ctx;
  const char *name;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  HostTimer(prb_ctx *c, const char *n) : ctx(c), name(n) {}
  ~HostTimer() {
    auto &t = ctx->timers[name];
    t.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    t.launches++;
  }
};

static HitSoA carve_hits(DevBuf &b, int64_t n) {
  HitSoA h;
  uint8_t *p = b.as<uint8_t>();
  const size_t n8 = ((size_t)n + 1) & ~(size_t)1; // keep the double arrays 8-byte aligned
  h.e_acc = reinterpret_cast<double *>(p);
  h.e_hyb = h.e_acc + n8;
  h.e_tot = h.e_hyb + n8;
  int32_t *ip = reinterpret_cast<int32_t *>(h.e_tot + n8);
  h.q_sp = ip;
  h.db_sp = ip + n8;
  h.q_len = ip + 2 * n8;
  h.db_len = ip + 3 * n8;
  h.db_id = ip + 4 * n8;
  h.db_id_start = ip + 5 * n8;
  h.query = ip + 6 * n8;
  return h;
}
static size_t hits_bytes(int64_t n) { return (((size_t)n + 1) & ~(size_t)1) * kHitBytes + 64; }

} // namespace prb

using namespace prb;

struct prb_db {
  prb_ctx *ctx = nullptr;
  DbHeader hdr;
  std::vector<DbPage> pages;
  // Device residency (DbReader::LoadDatabases loads every page eagerly, db_reader.cpp:29-59; here a database
  // larger than HBM - or than the share of it one wants to give it - is streamed): `mem` are slots, at most
  // max_resident of them; a page that is searched is uploaded into a slot if it is not there (the least recently
  // used page makes room), and with two slots or more the NEXT page's upload runs on a copy stream of its own
  // while this one is searched (the host copies are page-locked for that).
  std::vector<PageMem> mem;
  std::vector<int> slot_of_page, page_in_slot;
  std::vector<uint64_t> slot_used; // "time" of the last search that used the slot
  std::vector<hipEvent_t> slot_ready;
  uint64_t clock = 0;
  hipStream_t copy_stream = nullptr;
  bool pinned = false;
  int64_t uploads = 0; // pages uploaded so far (tests)
  std::vector<SeqTable> tabs; // per page: what the result lines print about its sequences
};

// The seed search proper of a batch against one page (SeedSearch::Run's DFS, seed_search.cpp:153-295): per query,
// on host threads, in the background; a consumer waits for query q with wait_for(q).  It needs the encoded queries,
// their suffix arrays and the page's k-mer table only - not the accessibilities - so it can be started as soon as a
// batch exists (prb_qbatch_seed_search_begin), long before the batch is searched.
struct SeedPlan {
  const prb_db *db = nullptr;
  int32_t page = 0, nq = 0, max_seed_length = 0;
  double hybrid_threshold = 0;
  std::vector<std::vector<SeedCandidate>> per_q;
  std::vector<double> qpairs;
  std::vector<int64_t> qrows, qents;
  std::unique_ptr<std::atomic<int>[]> done;
  std::atomic<int32_t> next_query{0}; // queries are handed out strictly in order: the consumer needs the first ones first
  std::thread producer;
  double dfs_ms = 0;
  ~SeedPlan() {
    if (producer.joinable()) producer.join();
  }
  void wait_for(int32_t q) const {
    while (!done[q].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
};

struct prb_qbatch {
  prb_ctx *ctx = nullptr;
  int32_t nq = 0, repeat_flag = 0;
  std::vector<int64_t> off; // nq + 1; query q occupies [off[q], off[q] + len[q] + 1)
  std::vector<int32_t> len, len_unmasked;
  std::vector<char> seqs;   // same offsets, NUL after each query
  std::vector<uint8_t> enc;
  std::vector<int32_t> sa;
  DevBuf d_enc, d_sa, d_acc, d_cond, d_off, d_len;
  bool have_acc = false;
  int32_t W = 0, delta = 0;
  QBatchDev view{};
  std::unique_ptr<SeedPlan> plan; // a seed search started ahead of prb_search_page, if any
};

namespace prb {
// Host threads for the per-query host work (suffix arrays, seed DFS).  An explicit count,
// because launchers such as torchrun export OMP_NUM_THREADS=1: PRB_HOST_THREADS, else
// half of the CPUs the process may keep busy (cpu_budget.hpp), at most 32.
static int host_threads(int work_items) {
  static const int cap = [] {
    const char *e = getenv("PRB_HOST_THREADS");
    int n = e ? atoi(e) : default_host_threads();
    return std::max(1, n);
  }();
  return std::max(1, std::min(cap, work_items));
}
static SearchWs &ws_of(prb_ctx *ctx) {
  if (!ctx->search_ws) ctx->search_ws = new SearchWs();
  return *static_cast<SearchWs *>(ctx->search_ws);
}
} // namespace prb

extern "C" {

int prb_search_const_upload(prb_ctx *ctx) {
  auto *m = new SearchConstMem();
  ctx->search_const = m;
  const EnergyParams &p = ctx->params;
  std::vector<int32_t> ints;
  auto add = [&](const int *src, size_t n) {
    size_t at = ints.size();
    ints.insert(ints.end(), src, src + n);
    return at;
  };
  const size_t o_stack = add(&p.stack37[0][0], 49), o_int = add(p.internal37, 31), o_mm = add(&p.mismatchI37[0][0][0], 175),
               o_11 = add(&p.int11_37[0][0][0][0], 1600), o_21 = add(&p.int21_37[0][0][0][0][0], 8000),
               o_22 = add(&p.int22_37[0][0][0][0][0][0], 40000), o_d5 = add(&p.dangle5_37[0][0], 40),
               o_d3 = add(&p.dangle3_37[0][0], 40);
  {
    using T = SearchTab;
    if (o_stack != T::kStack || o_int != T::kInternal || o_mm != T::kMismatchI || o_11 != T::kInt11 || o_21 != T::kInt21 ||
        o_22 != T::kInt22 || o_d5 != T::kDangle5 || o_d3 != T::kDangle3) {
      set_error("internal error: SearchTab layout");
      return PRB_ERR_STATE;
    }
    add(p.bulge37, 31);
    int tau[8] = {0, 0, 0, p.terminal_au, p.terminal_au, p.terminal_au, p.terminal_au, 0};
    add(tau, 8);
    int zero = 0;
    add(&zero, 1);
  }
  std::vector<double> bulge(64);
  for (int u = 0; u < 64; u++) // gapped_extension.cpp:439
    bulge[u] = u <= 30 ? (double)p.bulge37[u] : p.bulge37[30] + p.lxc37 * std::log(u / 30.);
  int rc;
  if ((rc = m->ints.ensure(ints.size() * 4))) return rc;
  if ((rc = m->bulge.ensure(bulge.size() * 8))) return rc;
  PRB_HIP(hipMemcpy(m->ints.p, ints.data(), ints.size() * 4, hipMemcpyHostToDevice));
  PRB_HIP(hipMemcpy(m->bulge.p, bulge.data(), bulge.size() * 8, hipMemcpyHostToDevice));
  for (int t = 0; t < 7; t++)
    if (p.rtype[t] != (t == 0 ? 0 : ((t - 1) ^ 1) + 1)) {
      set_error("parameter file: rtype is not the expected pair-type involution");
      return PRB_ERR_ARG;
    }
  const int32_t *b = m->ints.as<int32_t>();
  SearchConst &v = m->view;
  v.tab = b;
  v.stack37 = b + o_stack;
  v.internal37 = b + o_int;
  v.mismatchI37 = b + o_mm;
  v.int11 = b + o_11;
  v.int21 = b + o_21;
  v.int22 = b + o_22;
  v.dangle5 = b + o_d5;
  v.dangle3 = b + o_d3;
  v.bulge = m->bulge.as<double>();
  v.bp_rows = 0;
  for (int a = 1; a < 5; a++)
    for (int c = 0; c < 5; c++) v.bp_rows |= (uint64_t)(p.bp_pair[a][c] & 7) << (15 * (a - 1) + 3 * c);
  v.terminal_au = p.terminal_au;
  for (int a = 0; a < 5; a++)
    for (int c = 0; c < 5; c++) v.bp_pair[a * 5 + c] = (unsigned char)p.bp_pair[a][c];
  v.pair_mask = v.wobble_mask = 0;
  for (int a = 0; a < 5; a++)
    for (int c = 0; c < 5; c++) {
      if (p.bp_pair[a][c] != 0) v.pair_mask |= 1u << (a * 5 + c);
      if (p.bp_pair[a][c] == 3 || p.bp_pair[a][c] == 4) v.wobble_mask |= 1u << (a * 5 + c);
    }
  return PRB_OK;
}

void prb_search_const_free(prb_ctx *ctx) {
  if (ctx->search_const) {
    auto *m = static_cast<SearchConstMem *>(ctx->search_const);
    m->ints.release();
    m->bulge.release();
    delete m;
    ctx->search_const = nullptr;
  }
  if (ctx->search_ws) {
    auto *w = static_cast<SearchWs *>(ctx->search_ws);
    w->release();
    delete w;
    ctx->search_ws = nullptr;
  }
}

// ------------------------------------------------------------------------ database
static int upload_page(const DbPage &pg, PageMem &m, hipStream_t stream) {
  int rc;
  auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
    if ((rc = b.ensure(std::max<size_t>(bytes, 16)))) return rc;
    if (bytes) PRB_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, stream));
    return PRB_OK;
  };
  if ((rc = up(m.seqs, pg.seqs.data(), pg.seqs.size()))) return rc;
  if ((rc = up(m.sa, pg.sa.data(), pg.sa.size() * 4))) return rc;
  if ((rc = up(m.start_pos, pg.start_pos.data(), pg.start_pos.size() * 4))) return rc;
  if ((rc = up(m.seq_length, pg.seq_length.data(), pg.seq_length.size() * 4))) return rc;
  if ((rc = up(m.acc, pg.acc.data(), pg.acc.size() * 4))) return rc;
  if ((rc = up(m.cond, pg.cond.data(), pg.cond.size() * 4))) return rc;
  if ((rc = m.sa_seq.ensure(std::max<size_t>(pg.sa.size() * 4, 16)))) return rc;
  if ((rc = m.blk_seq.ensure((size_t)blk_seq_entries((int64_t)pg.seqs.size()) * 4))) return rc;
  m.view.seqs = m.seqs.as<uint8_t>();
  m.view.sa = m.sa.as<int32_t>();
  m.view.sa_seq = m.sa_seq.as<int32_t>();
  m.view.blk_seq = m.blk_seq.as<int32_t>();
  m.view.start_pos = m.start_pos.as<int32_t>();
  m.view.seq_length = m.seq_length.as<int32_t>();
  m.view.acc = m.acc.as<float>();
  m.view.cond = m.cond.as<float>();
  m.view.nchars = (int32_t)pg.seqs.size();
  m.view.nseq = pg.nseq;
  PRB_HIP(launch_sa_seq(m.view, m.sa_seq.as<int32_t>(), stream));
  PRB_HIP(launch_blk_seq(m.view, m.blk_seq.as<int32_t>(), stream));
  return PRB_OK;
}

// Page `page` on the device: its slot, uploaded now if need be (on `stream`; the slot's event is recorded behind
// the upload).  `keep` = a page whose slot must not be taken (the one being searched), or -1.
static int page_slot(prb_ctx *user, prb_db *db, int page, int keep, hipStream_t stream, int *slot_out) {
  int slot = db->slot_of_page[page];
  if (slot < 0) {
    // a free slot, else the least recently used one
    for (size_t k = 0; k < db->page_in_slot.size() && slot < 0; k++)
      if (db->page_in_slot[k] < 0) slot = (int)k;
    if (slot < 0) {
      for (size_t k = 0; k < db->page_in_slot.size(); k++)
        if (db->page_in_slot[k] != keep && (slot < 0 || db->slot_used[k] < db->slot_used[(size_t)slot])) slot = (int)k;
      if (slot < 0) return PRB_ERR_STATE;
      // what still reads the slot's old page (a search on the context's stream, an earlier upload) must be over
      PRB_HIP(hipStreamSynchronize(user->stream));
      PRB_HIP(hipEventSynchronize(db->slot_ready[(size_t)slot]));
      db->slot_of_page[(size_t)db->page_in_slot[(size_t)slot]] = -1;
    }
    int rc = upload_page(db->pages[(size_t)page], db->mem[(size_t)slot], stream);
    if (rc) return rc;
    PRB_HIP(hipEventRecord(db->slot_ready[(size_t)slot], stream));
    db->page_in_slot[(size_t)slot] = page;
    db->slot_of_page[(size_t)page] = slot;
    db->uploads++;
  }
  db->slot_used[(size_t)slot] = ++db->clock;
  *slot_out = slot;
  return PRB_OK;
}

int prb_db_open_streaming(prb_ctx *ctx, const char *prefix, int32_t max_resident_pages, prb_db **out) {
  if (!ctx || !prefix || !out || max_resident_pages < 0) return PRB_ERR_ARG;
  *out = nullptr;
  auto *db = new prb_db();
  db->ctx = ctx;
  std::string err;
  try {
    err = read_db(prefix, db->hdr, db->pages);
  } catch (const std::exception &e) { // (no exception leaves the C ABI)
    err = std::string("Error: cannot load the database: ") + e.what();
  }
  if (!err.empty()) {
    set_error(err);
    delete db;
    return PRB_ERR_IO;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) {
    delete db;
    return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
  }
  const size_t np = db->pages.size();
  const size_t nslots = max_resident_pages == 0 ? np : std::min<size_t>(np, (size_t)max_resident_pages);
  db->mem.resize(nslots);
  db->page_in_slot.assign(nslots, -1);
  db->slot_used.assign(nslots, 0);
  db->slot_of_page.assign(np, -1);
  db->slot_ready.assign(nslots, nullptr);
  int rc = PRB_OK;
  for (size_t k = 0; k < nslots && rc == PRB_OK; k++)
    if (hipEventCreateWithFlags(&db->slot_ready[k], hipEventDisableTiming) != hipSuccess) rc = PRB_ERR_HIP;
  if (rc == PRB_OK && nslots < np) {
    // streaming: uploads of the next page run beside the search on a stream of their own, from page-locked memory
    if (hipStreamCreateWithFlags(&db->copy_stream, hipStreamNonBlocking) != hipSuccess) rc = PRB_ERR_HIP;
    for (DbPage &pg : db->pages) {
      if (rc != PRB_OK) break;
      auto pin = [&](void *p, size_t bytes) {
        if (bytes && hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) rc = PRB_ERR_HIP;
      };
      pin(pg.seqs.data(), pg.seqs.size());
      pin(pg.sa.data(), pg.sa.size() * 4);
      pin(pg.acc.data(), pg.acc.size() * 4);
      pin(pg.cond.data(), pg.cond.size() * 4);
    }
    db->pinned = rc == PRB_OK;
    if (rc != PRB_OK) set_error("prb_db_open: cannot set up page streaming (stream / page-locked host memory)");
  }
  // the first pages are resident from the start
  for (size_t i = 0; i < nslots && rc == PRB_OK; i++) {
    int slot = -1;
    rc = page_slot(ctx, db, (int)i, -1, ctx->stream, &slot);
  }
  if (rc == PRB_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = PRB_ERR_HIP;
  if (rc != PRB_OK) {
    prb_db_close(db);
    return rc;
  }
  db->tabs.resize(np);
  for (size_t i = 0; i < np; i++) {
    const DbPage &pg = db->pages[i];
    db->tabs[i] = SeqTable{pg.names, pg.seq_length, pg.seq_length_rep, pg.start_pos};
  }
  *out = db;
  return PRB_OK;
}

int prb_db_open(prb_ctx *ctx, const char *prefix, prb_db **out) {
  const char *e = getenv("PRB_DB_RESIDENT_PAGES"); // 0 / unset: every page resident
  return prb_db_open_streaming(ctx, prefix, e ? std::max(0, atoi(e)) : 0, out);
}

int64_t prb_db_page_uploads(const prb_db *db) { return db ? db->uploads : -1; }

void prb_db_close(prb_db *db) {
  if (!db) return;
  (void)hipSetDevice(db->ctx->device);
  (void)hipStreamSynchronize(db->ctx->stream);
  if (db->copy_stream) {
    (void)hipStreamSynchronize(db->copy_stream);
    (void)hipStreamDestroy(db->copy_stream);
  }
  for (hipEvent_t e : db->slot_ready)
    if (e) (void)hipEventDestroy(e);
  if (db->pinned)
    for (DbPage &pg : db->pages) {
      if (!pg.seqs.empty()) (void)hipHostUnregister(pg.seqs.data());
      if (!pg.sa.empty()) (void)hipHostUnregister(pg.sa.data());
      if (!pg.acc.empty()) (void)hipHostUnregister(pg.acc.data());
      if (!pg.cond.empty()) (void)hipHostUnregister(pg.cond.data());
    }
  for (auto &m : db->mem)
    for (DevBuf *b : {&m.seqs, &m.sa, &m.sa_seq, &m.blk_seq, &m.start_pos, &m.seq_length, &m.acc, &m.cond}) b->release();
  delete db;
}

int prb_db_info(const prb_db *db, int32_t *hash_size, int32_t *repeat_flag, int32_t *maximal_span,
                int32_t *min_accessible_length, int32_t *npages) {
  if (!db) return PRB_ERR_ARG;
  if (hash_size) *hash_size = db->hdr.hash_size;
  if (repeat_flag) *repeat_flag = db->hdr.repeat_flag;
  if (maximal_span) *maximal_span = db->hdr.maximal_span;
  if (min_accessible_length) *min_accessible_length = db->hdr.min_accessible_length;
  if (npages) *npages = (int32_t)db->pages.size();
  return PRB_OK;
}

int prb_db_page_info(const prb_db *db, int32_t page, int32_t *nseq, int64_t *nchars) {
  if (!db || page < 0 || page >= (int32_t)db->pages.size()) return PRB_ERR_ARG;
  if (nseq) *nseq = db->pages[page].nseq;
  if (nchars) *nchars = (int64_t)db->pages[page].seqs.size();
  return PRB_OK;
}

const char *prb_db_seq_name(const prb_db *db, int32_t page, int32_t id) {
  if (!db || page < 0 || page >= (int32_t)db->pages.size()) return nullptr;
  const DbPage &pg = db->pages[page];
  if (id < 0 || id >= pg.nseq) return nullptr;
  return pg.names[id].c_str();
}

int prb_db_seq_lengths(const prb_db *db, int32_t page, int32_t id, int32_t *length, int32_t *length_unmasked,
                       int32_t *start_pos) {
  if (!db || page < 0 || page >= (int32_t)db->pages.size()) return PRB_ERR_ARG;
  const DbPage &pg = db->pages[page];
  if (id < 0 || id >= pg.nseq) return PRB_ERR_ARG;
  if (length) *length = pg.seq_length[id];
  if (length_unmasked) *length_unmasked = pg.seq_length_rep[id];
  if (start_pos) *start_pos = pg.start_pos[id];
  return PRB_OK;
}

// DbConstruction::Run (db_construction.cpp:37-83) as a tool: accessibilities on the GPU
// (the same Raccess kernels as for queries), suffix array and k-mer table on the host.
int prb_db_build(prb_ctx *ctx, const char *prefix, int32_t nseq, const char *const *names, const char *seqs,
                 const int64_t *offsets, int32_t repeat_flag, int32_t hash_size, int32_t maximal_span,
                 int32_t min_accessible_length, int32_t page_size) {
  if (!ctx || !prefix || nseq <= 0 || !names || !seqs || !offsets || repeat_flag < 0 || repeat_flag > 2 ||
      hash_size < 1 || hash_size > 12 || page_size < 1) {
    set_error("prb_db_build: bad argument");
    return PRB_ERR_ARG;
  }
  const int64_t total = offsets[nseq] - offsets[0];
  std::vector<float> acc((size_t)std::max<int64_t>(total, 1)), cond((size_t)std::max<int64_t>(total, 1));
  int rc = prb_accessibility(ctx, nseq, seqs, offsets, maximal_span, min_accessible_length, acc.data(), cond.data());
  if (rc) return rc;
  DbHeader hdr{hash_size, repeat_flag, maximal_span, min_accessible_length};
  DbWriter w;
  std::string err = w.open(prefix, hdr);
  if (!err.empty()) {
    set_error(err);
    return PRB_ERR_IO;
  }
  Encoder enc(repeat_flag);
  for (int32_t first = 0; first < nseq; first += page_size) {
    const int32_t n = std::min(page_size, nseq - first);
    DbPage pg;
    pg.nseq = n;
    int64_t t = 0;
    for (int32_t k = 0; k < n; k++) {
      const int32_t i = first + k;
      const int64_t L = offsets[i + 1] - offsets[i];
      pg.seq_length.push_back((int32_t)L);
      pg.start_pos.push_back((int32_t)t);
      t += L + 1;
      enc.append_db(seqs + offsets[i], L, pg.seqs);
      pg.names.push_back(names[i]);
      // the file stores cond[0..delta) = 0 and the conditional value of window i at i+delta
      // (raccess.cpp:462-480): the in-memory layout of stage 1 already has that shape
      pg.acc.insert(pg.acc.end(), acc.begin() + (offsets[i] - offsets[0]), acc.begin() + (offsets[i + 1] - offsets[0]));
      pg.cond.insert(pg.cond.end(), cond.begin() + (offsets[i] - offsets[0]), cond.begin() + (offsets[i + 1] - offsets[0]));
    }
    if (t > INT32_MAX) {
      set_error("database page exceeds 2^31 characters: use a smaller page size");
      return PRB_ERR_ARG;
    }
    pg.sa.resize(pg.seqs.size());
    suffix_array(pg.seqs.data(), (int32_t)pg.seqs.size(), pg.sa.data());
    build_kmer_table(pg.seqs, pg.sa, hash_size, pg.start_hash, pg.end_hash);
    err = w.append_page(pg, min_accessible_length);
    if (!err.empty()) {
      set_error(err);
      return PRB_ERR_IO;
    }
  }
  // the band tables of a whole database build (tens of GB) are not what the query batches that follow need
  ctx->ra_band.release();
  ctx->ra_vec.release();
  return PRB_OK;
}

// -------------------------------------------------------------------- query batches
int prb_qbatch_create(prb_ctx *ctx, int32_t nq, const char *seqs, const int64_t *offsets, int32_t repeat_flag,
                      prb_qbatch **out) {
  if (!ctx || nq <= 0 || !seqs || !offsets || !out || repeat_flag < 0 || repeat_flag > 2) {
    set_error("prb_qbatch_create: bad argument");
    return PRB_ERR_ARG;
  }
  *out = nullptr;
  auto *qb = new prb_qbatch();
  qb->ctx = ctx;
  qb->nq = nq;
  qb->repeat_flag = repeat_flag;
  qb->off.resize(nq + 1);
  qb->len.resize(nq);
  qb->len_unmasked.resize(nq);
  int64_t t = 0;
  for (int32_t q = 0; q < nq; q++) {
    const int64_t L = offsets[q + 1] - offsets[q];
    if (L < 0 || L > (1 << 30)) {
      delete qb;
      set_error("prb_qbatch_create: bad offsets");
      return PRB_ERR_ARG;
    }
    qb->off[q] = t;
    qb->len[q] = (int32_t)L;
    t += L + 1;
  }
  qb->off[nq] = t;
  qb->seqs.assign((size_t)t, 0);
  qb->enc.assign((size_t)t, 0);
  qb->sa.assign((size_t)t, 0);
  Encoder enc(repeat_flag);
#pragma omp parallel for schedule(dynamic, 4) num_threads(host_threads(nq))
  for (int32_t q = 0; q < nq; q++) {
    const int64_t o = qb->off[q];
    const int32_t L = qb->len[q];
    std::memcpy(qb->seqs.data() + o, seqs + offsets[q], (size_t)L);
    enc.encode_query(seqs + offsets[q], L, qb->enc.data() + o);
    suffix_array(qb->enc.data() + o, L + 1, qb->sa.data() + o);
    int32_t c = 0;
    for (int32_t k = 0; k <= L; k++) c += qb->enc[o + k] >= 2 && qb->enc[o + k] <= 5; // rna_interaction_search.cpp:179-183
    qb->len_unmasked[q] = c;
  }
  if (hipError_t e = hipSetDevice(ctx->device); e != hipSuccess) {
    delete qb;
    return hip_fail(e, "hipSetDevice");
  }
  int rc = 0;
  const size_t n = (size_t)t;
  if ((rc = qb->d_enc.ensure(n)) || (rc = qb->d_sa.ensure(n * 4)) || (rc = qb->d_acc.ensure(n * 4)) ||
      (rc = qb->d_cond.ensure(n * 4)) || (rc = qb->d_off.ensure((size_t)(nq + 1) * 8)) ||
      (rc = qb->d_len.ensure((size_t)nq * 4))) {
    prb_qbatch_destroy(qb);
    return rc;
  }
  {
    hipError_t e = hipMemcpyAsync(qb->d_enc.p, qb->enc.data(), n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(qb->d_sa.p, qb->sa.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(qb->d_off.p, qb->off.data(), (size_t)(nq + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(qb->d_len.p, qb->len.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(qb->d_acc.p, 0, n * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(qb->d_cond.p, 0, n * 4, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      prb_qbatch_destroy(qb);
      return hip_fail(e, "prb_qbatch_create: upload");
    }
  }
  qb->view.enc = qb->d_enc.as<uint8_t>();
  qb->view.sa = qb->d_sa.as<int32_t>();
  qb->view.acc = qb->d_acc.as<float>();
  qb->view.cond = qb->d_cond.as<float>();
  qb->view.off = qb->d_off.as<int64_t>();
  qb->view.len = qb->d_len.as<int32_t>();
  qb->view.nq = nq;
  *out = qb;
  return PRB_OK;
}

void prb_qbatch_destroy(prb_qbatch *qb) {
  if (!qb) return;
  for (DevBuf *b : {&qb->d_enc, &qb->d_sa, &qb->d_acc, &qb->d_cond, &qb->d_off, &qb->d_len}) b->release();
  delete qb;
}

int prb_qbatch_accessibility(prb_ctx *ctx, prb_qbatch *qb, int32_t maximal_span, int32_t min_accessible_length) {
  if (!ctx || !qb || qb->ctx->device != ctx->device) return PRB_ERR_ARG;
  const size_t n = (size_t)qb->off[qb->nq];
  PRB_HIP(hipSetDevice(ctx->device));
  PRB_HIP(hipMemsetAsync(qb->d_acc.p, 0, n * 4, ctx->stream));
  PRB_HIP(hipMemsetAsync(qb->d_cond.p, 0, n * 4, ctx->stream));
  int rc = run_accessibility(ctx, qb->nq, qb->seqs.data(), qb->off.data(), qb->len.data(), qb->off.data(), maximal_span,
                             min_accessible_length, qb->d_acc.as<float>(), qb->d_cond.as<float>());
  if (rc) return rc;
  qb->have_acc = true;
  qb->W = maximal_span;
  qb->delta = min_accessible_length;
  return PRB_OK;
}

int prb_qbatch_get(prb_qbatch *qb, int32_t q, uint8_t *enc, int32_t *sa, float *acc, float *cond) {
  if (!qb || q < 0 || q >= qb->nq) return PRB_ERR_ARG;
  const int64_t o = qb->off[q];
  const int32_t L = qb->len[q];
  if (enc) std::memcpy(enc, qb->enc.data() + o, (size_t)L + 1);
  if (sa) std::memcpy(sa, qb->sa.data() + o, ((size_t)L + 1) * 4);
  if (acc || cond) {
    if (!qb->have_acc) {
      set_error("prb_qbatch_get: accessibilities not computed yet");
      return PRB_ERR_STATE;
    }
    PRB_HIP(hipSetDevice(qb->ctx->device));
    if (acc && L) PRB_HIP(hipMemcpy(acc, qb->d_acc.as<float>() + o, (size_t)L * 4, hipMemcpyDeviceToHost));
    if (cond && L) PRB_HIP(hipMemcpy(cond, qb->d_cond.as<float>() + o, (size_t)L * 4, hipMemcpyDeviceToHost));
  }
  return PRB_OK;
}

int32_t prb_qbatch_length_unmasked(const prb_qbatch *qb, int32_t q) {
  if (!qb || q < 0 || q >= qb->nq) return -1;
  return qb->len_unmasked[q];
}

int prb_qbatch_seed_search_begin(prb_ctx *ctx, prb_qbatch *qb, const prb_db *db, int32_t page, const prb_ris_opts *opts);

} // extern "C"

// ------------------------------------------------------------------------- search
namespace prb {

// Sorts `in` (n hits) into `out` by the reference's comparator made total:
// (query, db_sp asc, q_sp asc, db_len desc, q_len desc, energy asc, hybridization part asc, accessibility part asc,
// input order);
// LSD: one stable radix sort per key, least significant first.  perm_out[i] = index in `in`.
// Field bounds of the hits of one sub-batch, for the one-key sort
struct SortBounds {
  int32_t qmin = 0, qspan = 1, max_qlen = 0, max_dblen = 0, nchars = 0;
  int32_t eq_len_max = 0; // > 0: every hit of the list has q_len = db_len <= this (the one-pass seed path reports it)
};
static int bits_for(int64_t max_value) {
  int b = 1;
  while (b < 40 && (int64_t(1) << b) <= max_value) b++;
  return b;
}

static int sort_hits(prb_ctx *ctx, SearchWs &w, const HitRec *recs, HitSoA out, int64_t n, int nq, const SortBounds &sb,
                     uint32_t **perm_out) {
  int rc;
  const size_t N = (size_t)n;
  {
    // one stable radix sort over a packed key + a pass over the runs of identical coordinates
    PackedKeyInfo f;
    f.qmin = sb.qmin;
    f.one_len = sb.eq_len_max > 0;
    f.lmax = f.one_len ? sb.eq_len_max : std::max(sb.max_qlen, sb.max_dblen);
    f.bl = bits_for(f.lmax);
    f.bq = bits_for(sb.max_qlen);
    f.bd = bits_for(sb.nchars);
    const int total = (f.one_len ? 1 : 2) * f.bl + f.bq + f.bd + bits_for(sb.qspan - 1);
    if (total <= 64 && f.lmax <= 65535 && !getenv("PRB_SORT_FOUR_KEYS")) {
      if ((rc = w.kP.ensure(N * 8)) || (rc = w.kTmp2.ensure(N * 8)) || (rc = w.kE.ensure(N * 8)) || (rc = w.kTmp.ensure(N * 8)) ||
          (rc = w.idxA.ensure(N * 4)) || (rc = w.idxB.ensure(N * 4)) || (rc = w.pending.ensure(16)))
        return rc;
      PRB_HIP(launch_make_packed_keys_recs(recs, n, f, w.kP.as<uint64_t>(), w.kE.as<uint64_t>(), w.idxA.as<uint32_t>(), ctx->stream));
      size_t tmp = 0;
      PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, w.kP.as<uint64_t>(), w.kTmp2.as<uint64_t>(), w.idxA.as<uint32_t>(),
                                        w.idxB.as<uint32_t>(), N, 0, (unsigned)total, ctx->stream));
      if ((rc = w.sortTmp.ensure(tmp))) return rc;
      PRB_HIP(rocprim::radix_sort_pairs(w.sortTmp.p, tmp, w.kP.as<uint64_t>(), w.kTmp2.as<uint64_t>(), w.idxA.as<uint32_t>(),
                                        w.idxB.as<uint32_t>(), N, 0, (unsigned)total, ctx->stream));
      PRB_HIP(launch_gather_u64(w.kE.as<uint64_t>(), w.idxB.as<uint32_t>(), w.kTmp.as<uint64_t>(), n, ctx->stream));
      PRB_HIP(hipMemsetAsync(w.pending.p, 0, 4, ctx->stream));
      PRB_HIP(launch_fix_ties(w.kTmp2.as<uint64_t>(), w.kTmp.as<uint64_t>(), w.idxB.as<uint32_t>(), n, recs, w.idxA.as<uint32_t>(),
                              w.pending.as<int32_t>(), ctx->stream));
      int32_t too_long = 0;
      PRB_HIP(hipMemcpyAsync(&too_long, w.pending.p, 4, hipMemcpyDeviceToHost, ctx->stream));
      PRB_HIP(hipStreamSynchronize(ctx->stream));
      if (!too_long) {
        PRB_HIP(launch_gather_recs_to_hits(recs, w.idxA.as<uint32_t>(), out, n, ctx->stream));
        *perm_out = w.idxA.as<uint32_t>();
        return PRB_OK;
      }
    }
  }
  // the general form works on the fields as arrays
  if (getenv("PRB_DEBUG_MEM")) fprintf(stderr, "[mem] sort: the general form (%lld hits)\n", (long long)n);
  if ((rc = w.hitsTmp.ensure(hits_bytes(n)))) return rc;
  const HitSoA in = carve_hits(w.hitsTmp, n);
  PRB_HIP(launch_gather_recs_to_hits(recs, nullptr, in, n, ctx->stream));
  if ((rc = w.kE.ensure(N * 8)) || (rc = w.kL.ensure(N * 4)) || (rc = w.kQ.ensure(N * 4)) || (rc = w.kP.ensure(N * 8)) ||
      (rc = w.kTmp.ensure(N * 8)) || (rc = w.kTmp2.ensure(N * 8)) || (rc = w.idxA.ensure(N * 4)) ||
      (rc = w.idxB.ensure(N * 4)))
    return rc;
  PRB_HIP(launch_make_keys(in, n, w.kE.as<uint64_t>(), w.kL.as<uint32_t>(), w.kQ.as<uint32_t>(), w.kP.as<uint64_t>(),
                           w.idxA.as<uint32_t>(), ctx->stream));
  uint32_t *ia = w.idxA.as<uint32_t>(), *ib = w.idxB.as<uint32_t>();
  auto sort64 = [&](const uint64_t *keys, unsigned bits) -> int {
    size_t tmp = 0;
    PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys, w.kTmp2.as<uint64_t>(), ia, ib, N, 0, bits, ctx->stream));
    if ((rc = w.sortTmp.ensure(tmp))) return rc;
    PRB_HIP(rocprim::radix_sort_pairs(w.sortTmp.p, tmp, keys, w.kTmp2.as<uint64_t>(), ia, ib, N, 0, bits, ctx->stream));
    std::swap(ia, ib);
    return PRB_OK;
  };
  auto sort32 = [&](const uint32_t *keys, unsigned bits) -> int {
    size_t tmp = 0;
    PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys, w.kTmp2.as<uint32_t>(), ia, ib, N, 0, bits, ctx->stream));
    if ((rc = w.sortTmp.ensure(tmp))) return rc;
    PRB_HIP(rocprim::radix_sort_pairs(w.sortTmp.p, tmp, keys, w.kTmp2.as<uint32_t>(), ia, ib, N, 0, bits, ctx->stream));
    std::swap(ia, ib);
    return PRB_OK;
  };
  // least significant first: accessibility part, hybridization part, energy
  PRB_HIP(launch_order_keys(in.e_acc, n, w.kTmp.as<uint64_t>(), ctx->stream));
  if ((rc = sort64(w.kTmp.as<uint64_t>(), 64))) return rc; // keys in input order
  PRB_HIP(launch_order_keys(in.e_hyb, n, w.kTmp2.as<uint64_t>(), ctx->stream));
  PRB_HIP(launch_gather_u64(w.kTmp2.as<uint64_t>(), ia, w.kTmp.as<uint64_t>(), n, ctx->stream));
  if ((rc = sort64(w.kTmp.as<uint64_t>(), 64))) return rc;
  PRB_HIP(launch_gather_u64(w.kE.as<uint64_t>(), ia, w.kTmp.as<uint64_t>(), n, ctx->stream));
  if ((rc = sort64(w.kTmp.as<uint64_t>(), 64))) return rc;
  PRB_HIP(launch_gather_u32(w.kL.as<uint32_t>(), ia, w.kTmp.as<uint32_t>(), n, ctx->stream));
  if ((rc = sort32(w.kTmp.as<uint32_t>(), 32))) return rc;
  PRB_HIP(launch_gather_u32(w.kQ.as<uint32_t>(), ia, w.kTmp.as<uint32_t>(), n, ctx->stream));
  if ((rc = sort32(w.kTmp.as<uint32_t>(), 32))) return rc;
  PRB_HIP(launch_gather_u64(w.kP.as<uint64_t>(), ia, w.kTmp.as<uint64_t>(), n, ctx->stream));
  unsigned qbits = 1;
  while ((1 << qbits) < nq && qbits < 31) qbits++;
  if ((rc = sort64(w.kTmp.as<uint64_t>(), 32 + qbits))) return rc;
  PRB_HIP(launch_gather_hits(in, ia, out, n, ctx->stream));
  *perm_out = ia;
  return PRB_OK;
}

struct ToI64 {
  __host__ __device__ int64_t operator()(const int32_t &v) const { return (int64_t)v; }
};
struct MaxOp {
  __host__ __device__ int64_t operator()(const int64_t &a, const int64_t &b) const { return a > b ? a : b; }
};

// CheckRedundancy on the sorted list `h`; writes the indices of the survivors (ascending) to
// w.surv and returns their number.
static int filter_hits(prb_ctx *ctx, SearchWs &w, const HitSoA &h, int64_t n, double thr, int64_t *nsurv) {
  int rc;
  const size_t N = (size_t)n;
  *nsurv = 0;
  if (n == 0) return PRB_OK;
  if ((rc = w.endKey.ensure(N * 8)) || (rc = w.pmax.ensure(N * 8)) || (rc = w.state.ensure(N)) || (rc = w.keep.ensure(N)) ||
      (rc = w.pending.ensure(16)) || (rc = w.surv.ensure(N * 4)) || (rc = w.count.ensure(16)))
    return rc;
  PRB_HIP(launch_filter_init(h, n, thr, w.endKey.as<int64_t>(), w.state.as<uint8_t>(), ctx->stream));
  size_t tmp = 0;
  PRB_HIP(rocprim::inclusive_scan(nullptr, tmp, w.endKey.as<int64_t>(), w.pmax.as<int64_t>(), N, MaxOp(), ctx->stream));
  if ((rc = w.scanTmp.ensure(tmp))) return rc;
  PRB_HIP(rocprim::inclusive_scan(w.scanTmp.p, tmp, w.endKey.as<int64_t>(), w.pmax.as<int64_t>(), N, MaxOp(), ctx->stream));
  for (int round = 0;; round++) {
    PRB_HIP(hipMemsetAsync(w.pending.p, 0, 4, ctx->stream));
    PRB_HIP(launch_filter_round(h, n, w.pmax.as<int64_t>(), w.state.as<uint8_t>(), w.pending.as<int32_t>(), ctx->stream));
    int32_t pend = 0;
    PRB_HIP(hipMemcpyAsync(&pend, w.pending.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    if (!pend) break;
    if (round > 100000) {
      set_error("redundancy filter did not converge");
      return PRB_ERR_STATE;
    }
  }
  PRB_HIP(launch_filter_final(h, n, w.pmax.as<int64_t>(), w.state.as<uint8_t>(), w.keep.as<uint8_t>(), ctx->stream));
  tmp = 0;
  rocprim::counting_iterator<uint32_t> iota(0);
  PRB_HIP(rocprim::select(nullptr, tmp, iota, w.keep.as<uint8_t>(), w.surv.as<uint32_t>(), w.count.as<size_t>(), N,
                          ctx->stream));
  if ((rc = w.scanTmp.ensure(tmp))) return rc;
  PRB_HIP(rocprim::select(w.scanTmp.p, tmp, iota, w.keep.as<uint8_t>(), w.surv.as<uint32_t>(), w.count.as<size_t>(), N,
                          ctx->stream));
  size_t cnt = 0;
  PRB_HIP(hipMemcpyAsync(&cnt, w.count.p, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
  PRB_HIP(hipStreamSynchronize(ctx->stream));
  *nsurv = (int64_t)cnt;
  return PRB_OK;
}

// Drops the hits whose energy is above `thr` (they cannot survive CheckRedundancy nor influence
// it, see k_flag_not_above): recbuf <- the kept hits of in, in order, as records for the sort that
// follows; idxbuf[i] = index in `in`.
// `have` records already in recbuf are kept (the new ones are appended behind them).
// Room for `more` records behind the `have` records that `recbuf` holds (which are kept).
// PRB_DEBUG_MEM: free device memory at the stations of a search
static void mem_note(const char *where, const SearchWs &w) {
  static const bool on = getenv("PRB_DEBUG_MEM") != nullptr;
  if (!on) return;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  fprintf(stderr, "[mem] %-18s free %7zu MiB | hitsA %6zu hitsB %6zu hitsC %6zu hitsTmp %6zu MiB\n", where, free_b >> 20, w.hitsA.cap >> 20,
          w.hitsB.cap >> 20, w.hitsC.cap >> 20, w.hitsTmp.cap >> 20);
}

static int reserve_recs(prb_ctx *ctx, DevBuf &recbuf, int64_t have, int64_t more) {
  int rc;
  const size_t need = ((size_t)have + (size_t)more) * sizeof(HitRec);
  if (have > 0 && need > recbuf.cap) { // grow and keep what is there
    DevBuf bigger;
    // (with room to spare while memory allows: every growth is a copy of everything; a list of 1e9 records - 64 GB - must
    // still be able to grow next to its old copy)
    // (past 16 GB only a sixth: what is spare here is missing in the sort behind)
    if ((rc = need > ((size_t)16 << 30) ? PRB_ERR_NOMEM : bigger.ensure(need + need / 2)) && (rc = bigger.ensure(need + need / 6)) &&
        (rc = bigger.ensure(need)))
      return rc;
    PRB_HIP(hipMemcpyAsync(bigger.p, recbuf.p, (size_t)have * sizeof(HitRec), hipMemcpyDeviceToDevice, ctx->stream));
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    recbuf.release();
    recbuf = bigger;
    return PRB_OK;
  }
  return recbuf.ensure(need);
}

// `need` bytes in a buffer whose first `have` bytes are kept
static int grow_keep(prb_ctx *ctx, DevBuf &buf, size_t have, size_t need) {
  int rc;
  if (have > 0 && need > buf.cap) {
    DevBuf bigger;
    if ((rc = bigger.ensure(need + need / 2)) && (rc = bigger.ensure(need + need / 6)) && (rc = bigger.ensure(need))) return rc;
    PRB_HIP(hipMemcpyAsync(bigger.p, buf.p, have, hipMemcpyDeviceToDevice, ctx->stream));
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    buf.release();
    buf = bigger;
    return PRB_OK;
  }
  return buf.ensure(need);
}
// the hits from `first` on, as a list of their own
static HitSoA offset_hits(const HitSoA &h, int64_t first) {
  return HitSoA{h.q_sp + first, h.db_sp + first, h.q_len + first, h.db_len + first, h.db_id + first, h.db_id_start + first,
                h.query + first, h.e_acc + first, h.e_hyb + first, h.e_tot + first};
}

static int compact_below(prb_ctx *ctx, SearchWs &w, const HitSoA &in, int64_t n, double thr, DevBuf &idxbuf, DevBuf &recbuf,
                         int64_t *m, int64_t have = 0) {
  int rc;
  *m = 0;
  if (n == 0) return PRB_OK;
  const size_t N = (size_t)n;
  if ((rc = w.keep.ensure(N)) || (rc = idxbuf.ensure(N * 4)) || (rc = w.count.ensure(16))) return rc;
  PRB_HIP(launch_flag_not_above(in.e_tot, n, thr, w.keep.as<uint8_t>(), ctx->stream));
  size_t tmp = 0;
  rocprim::counting_iterator<uint32_t> iota(0);
  PRB_HIP(rocprim::select(nullptr, tmp, iota, w.keep.as<uint8_t>(), idxbuf.as<uint32_t>(), w.count.as<size_t>(), N, ctx->stream));
  if ((rc = w.scanTmp.ensure(tmp))) return rc;
  PRB_HIP(rocprim::select(w.scanTmp.p, tmp, iota, w.keep.as<uint8_t>(), idxbuf.as<uint32_t>(), w.count.as<size_t>(), N,
                          ctx->stream));
  size_t cnt = 0;
  PRB_HIP(hipMemcpyAsync(&cnt, w.count.p, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
  PRB_HIP(hipStreamSynchronize(ctx->stream));
  *m = (int64_t)cnt;
  if ((rc = reserve_recs(ctx, recbuf, have, (int64_t)std::max<size_t>(cnt, 1)))) return rc;
  PRB_HIP(launch_gather_hits_to_recs(in, idxbuf.as<uint32_t>(), recbuf.as<HitRec>() + have, *m, ctx->stream));
  return PRB_OK;
}

static int download_hits(prb_ctx *ctx, SearchWs &w, const HitSoA &h, int64_t n, std::vector<prb_hit> &out) {
  HostTimer ht(ctx, "host_download");
  const size_t base = out.size();
  if (n == 0) return PRB_OK;
  int rc;
  const size_t bytes = (size_t)n * sizeof(prb_hit);
  if ((rc = w.packed.ensure(bytes)) || (rc = w.pinned.ensure(bytes))) return rc;
  PRB_HIP(launch_pack_hits(h, n, nullptr, nullptr, -1, w.packed.p, ctx->stream));
  PRB_HIP(hipMemcpyAsync(w.pinned.p, w.packed.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (out.capacity() < base + (size_t)n) out.reserve(std::max(2 * out.capacity(), base + (size_t)n));
  PRB_HIP(hipStreamSynchronize(ctx->stream));
  const prb_hit *src = static_cast<const prb_hit *>(w.pinned.p);
  out.insert(out.end(), src, src + n);
  return PRB_OK;
}

// Candidates [c0, c1) of a sub-batch = its next chunk: as many as fit the pair budget (and the one-pass form's record)
static int32_t chunk_end(const CandDev *cd, int32_t ncand, int32_t c0, double chunk_pairs, double *pairs_out) {
  int32_t c1 = c0;
  double acc = 0;
  while (c1 < ncand) {
    const double pairs = (double)(cd[c1].ep_q - cd[c1].sp_q + 1) * (double)(cd[c1].ep_db - cd[c1].sp_db + 1);
    if (c1 > c0 && (acc + pairs > chunk_pairs || c1 - c0 >= kMaxFusedCands)) break;
    acc += pairs;
    c1++;
  }
  *pairs_out = acc;
  return c1;
}
struct SeedKnobs {
  double chunk_pairs;
  int row_shift;
  bool fused;
};
static SeedKnobs seed_knobs() {
  const char *cenv = getenv("PRB_SEARCH_CHUNK_PAIRS");
  const char *benv = getenv("PRB_SEARCH_PAIRS");
  const char *rsenv = getenv("PRB_SEED_ROW_SHIFT");
  const char *fenv = getenv("PRB_SEED_FUSED"); // 0: seeds written as a list, extended and thinned in separate passes
  return SeedKnobs{cenv ? atof(cenv) : (benv ? atof(benv) : 4.0e8), rsenv ? std::min(atoi(rsenv), 30) : 7, !(fenv && atoi(fenv) == 0)};
}

// The front of the one-pass seed path (SearchWs::FrontStage) for the chunk whose candidates - row0 / qoff relative to
// the chunk - are the nc entries at cd (page-locked): everything is issued on `s`.  *np_out = its pairs, or -1 when a
// candidate has more query entries than a pair's value has bits for (nothing is issued then: the list form takes it).
static int issue_front(SearchWs &w, const prb_qbatch *qb, const PageDev &pd, int delta, int row_shift, const CandDev *cd, int32_t nc,
                       int64_t cents, hipStream_t s, int64_t *np_out) {
  SearchWs::FrontStage &F = w.front;
  int rc;
  *np_out = -1;
  if ((rc = F.pair0_pin.ensure(((size_t)nc + 1) * 8))) return rc;
  int64_t *pair0 = static_cast<int64_t *>(F.pair0_pin.p);
  int64_t np = 0;
  for (int32_t c = 0; c < nc; c++) {
    const int64_t qw = cd[c].ep_q - cd[c].sp_q + 1;
    if (qw > kMaxFusedEntries) return PRB_OK;
    pair0[c] = np;
    np += qw * (int64_t)(cd[c].ep_db - cd[c].sp_db + 1);
  }
  pair0[nc] = np;
  const size_t NP = (size_t)np;
  const int qmin = cd[0].query;
  const int dbits = bits_for(std::max<int64_t>(1, ((int64_t)pd.nchars - 1) >> row_shift));
  const int kbits = dbits + bits_for(std::max<int64_t>(1, (int64_t)cd[nc - 1].query - qmin));
  const bool wide = kbits > 32;
  if ((rc = F.cands.ensure((size_t)nc * sizeof(CandDev))) || (rc = F.seed_qacc.ensure((size_t)std::max<int64_t>(cents, 1) * 8)) ||
      (rc = F.pair0.ensure(((size_t)nc + 1) * 8)) || (rc = F.keyA.ensure(NP * (wide ? 8 : 4))) || (rc = F.keyB.ensure(NP * (wide ? 8 : 4))) ||
      (rc = F.valA.ensure(NP * 8)) || (rc = F.valB.ensure(NP * 8)))
    return rc;
  PRB_HIP(hipMemcpyAsync(F.cands.p, cd, (size_t)nc * sizeof(CandDev), hipMemcpyHostToDevice, s));
  PRB_HIP(hipMemcpyAsync(F.pair0.p, pair0, ((size_t)nc + 1) * 8, hipMemcpyHostToDevice, s));
  PRB_HIP(launch_seed_qacc(F.cands.as<CandDev>(), nc, cents, qb->view, delta, F.seed_qacc.as<double>(), s));
  PRB_HIP(launch_pair_keys(F.cands.as<CandDev>(), F.pair0.as<int64_t>(), nc, np, pd, qmin, row_shift, dbits, wide, F.keyA.p,
                           F.valA.as<uint64_t>(), s));
  size_t tmp = 0;
  if (wide) {
    PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, F.keyA.as<uint64_t>(), F.keyB.as<uint64_t>(), F.valA.as<uint64_t>(),
                                      F.valB.as<uint64_t>(), NP, 0, (unsigned)kbits, s));
    if ((rc = F.sortTmp.ensure(tmp))) return rc;
    PRB_HIP(rocprim::radix_sort_pairs(F.sortTmp.p, tmp, F.keyA.as<uint64_t>(), F.keyB.as<uint64_t>(), F.valA.as<uint64_t>(),
                                      F.valB.as<uint64_t>(), NP, 0, (unsigned)kbits, s));
  } else {
    PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, F.keyA.as<uint32_t>(), F.keyB.as<uint32_t>(), F.valA.as<uint64_t>(),
                                      F.valB.as<uint64_t>(), NP, 0, (unsigned)kbits, s));
    if ((rc = F.sortTmp.ensure(tmp))) return rc;
    PRB_HIP(rocprim::radix_sort_pairs(F.sortTmp.p, tmp, F.keyA.as<uint32_t>(), F.keyB.as<uint32_t>(), F.valA.as<uint64_t>(),
                                      F.valB.as<uint64_t>(), NP, 0, (unsigned)kbits, s));
  }
  *np_out = np;
  return PRB_OK;
}

// One sub-batch of queries through the GPU stages.  cd (pinned host memory) = its seed candidates in
// query order, row0 filled in; nrows = their database SA entries in total.  `front_free` (may be empty) is called
// at most once, behind the LDS tiers of the gapped cascade (the front stage's buffers are long free by then): the
// caller's chance to issue the front of the NEXT sub-batch ahead (issue_front on SearchWs::front.stream, SearchWs::front.ahead set).
static int search_range(prb_ctx *ctx, prb_qbatch *qb, prb_db *db, int page, const prb_ris_opts &opts, int last_stage,
                        const CandDev *cd, int64_t ncand64, int64_t nrows, int64_t nqent, prb_hitset *hs,
                        const std::function<void()> &front_free) {
  SearchWs &w = ws_of(ctx);
  const SearchConst &sc = static_cast<SearchConstMem *>(ctx->search_const)->view;
  const DbPage &pg = db->pages[page];
  const PageDev &pd = db->mem[(size_t)db->slot_of_page[(size_t)page]].view;
  const int delta = db->hdr.min_accessible_length;
  ExtOpts eo{delta, opts.drop_out_wo_gap, opts.drop_out_w_gap, opts.min_helix_length};
  int rc;
  if (w.trim_next) {
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    w.trim();
  }
  // ---- seeds: one row per (candidate, db SA entry) ----
  if (ncand64 == 0) return PRB_OK;
  if (ncand64 > INT32_MAX) {
    set_error("too many seed candidates in one sub-batch: lower PRB_SEARCH_PAIRS, or build the database in smaller pages (db -c)");
    return PRB_ERR_NOMEM;
  }
  const int32_t ncand = (int32_t)ncand64;
  int max_qlen = 0;
  for (int32_t q = 0; q < qb->nq; q++) max_qlen = std::max(max_qlen, qb->len[q]);
  SortBounds sb;
  sb.qmin = cd[0].query; // the candidates are in query order
  sb.qspan = cd[ncand - 1].query - cd[0].query + 1;
  sb.max_qlen = max_qlen;
  for (int32_t L : pg.seq_length) sb.max_dblen = std::max(sb.max_dblen, L);
  sb.nchars = pd.nchars;
  // The seeds of the sub-batch are produced, extended without gaps and thinned to the hits under the -f threshold
  // (one seed in nine) in CHUNKS of consecutive candidates of at most `chunk_pairs` (query SA entry x database SA
  // entry) pairs each; the survivors of all chunks, in candidate order, are what the sort and the redundancy filter
  // then see.  So the seed pools are bounded by the pair budget whatever a single query brings - a 100 kb query
  // against a 100 M character page has 1e10 seeds - while the filter still runs over whole queries.
  // rows / pairs in (query, database position >> row_shift) order, see k_row_key; PRB_SEED_ROW_SHIFT = -1 keeps suffix-array order
  const SeedKnobs knobs = seed_knobs();
  const double chunk_pairs = knobs.chunk_pairs;
  const int row_shift = knobs.row_shift;
  const bool fused = knobs.fused;
  if ((rc = w.front.init())) return rc;
  CandDev *cdm = const_cast<CandDev *>(cd); // (the caller's staging buffer: each candidate is rebased once, for its chunk)
  int64_t m1 = 0, one_pass_maxlen = 0;
  bool all_one_pass = true; // every chunk through k_seed_extend: lengths known, q_len = db_len
  for (int32_t c0 = 0; c0 < ncand;) {
    double acc = 0;
    const int32_t c1 = chunk_end(cd, ncand, c0, chunk_pairs, &acc);
    const int32_t nc = c1 - c0;
    const int64_t row_base = cd[c0].row0, ent_base = cd[c0].qoff;
    const int64_t crows = (c1 < ncand ? cd[c1].row0 : nrows) - row_base, cents = (c1 < ncand ? cd[c1].qoff : nqent) - ent_base;
    for (int32_t c = c0; c < c1; c++) {
      cdm[c].row0 -= row_base;
      cdm[c].qoff -= ent_base;
    }
    // ---- seeds -> hits under the -f threshold in one pass over the sorted pairs (search_kernels.hip) ----
    int64_t np = -1;
    if (fused && last_stage != 1 && row_shift >= 0 && acc < 4.0e9) {
      if ((rc = ctx->time_begin())) return rc;
      SearchWs::FrontStage &F = w.front;
      if (c0 == 0 && F.ahead && F.cd == cd && F.nc == nc) { // issued while the last sub-batch was extended: wait for it
        np = F.np;
        PRB_HIP(hipStreamWaitEvent(ctx->stream, F.done, 0));
      } else {
        if (F.ahead) PRB_HIP(hipStreamSynchronize(F.stream)); // (not what was expected: its buffers are taken over)
        if ((rc = issue_front(w, qb, pd, delta, row_shift, cd + c0, nc, cents, ctx->stream, &np))) return rc;
      }
      F.ahead = false;
    }
    if (np >= 0) {
      SearchWs::FrontStage &F = w.front;
      if ((rc = w.count.ensure(16))) return rc;
      if ((rc = ctx->time_end("seed", 2))) return rc;
      if ((rc = ctx->time_begin())) return rc;
      const int64_t nsl = fused_slices(np);
      if ((rc = w.hitsA.ensure((size_t)nsl * kFusePairs * kSliceRecBytes)) || (rc = w.row_count.ensure((size_t)(nsl + 1) * 4)) ||
          (rc = w.row_off.ensure((size_t)(nsl + 1) * 8)))
        return rc;
      PRB_HIP(hipMemsetAsync(w.count.p, 0, 16, ctx->stream));
      PRB_HIP(hipMemsetAsync(w.row_count.as<int32_t>() + nsl, 0, 4, ctx->stream));
      PRB_HIP(launch_seed_extend(F.cands.as<CandDev>(), F.valB.as<uint64_t>(), np, qb->view, pd, sc, eo, F.seed_qacc.as<double>(),
                                 opts.interaction_threshold, max_qlen, w.hitsA.p, w.row_count.as<int32_t>(), w.count.as<uint64_t>(),
                                 ctx->stream));
      {
        size_t tmp2 = 0;
        auto in = rocprim::make_transform_iterator(w.row_count.as<int32_t>(), ToI64());
        PRB_HIP(rocprim::exclusive_scan(nullptr, tmp2, in, w.row_off.as<int64_t>(), (int64_t)0, (size_t)nsl + 1, rocprim::plus<int64_t>(),
                                        ctx->stream));
        if ((rc = w.scanTmp.ensure(tmp2))) return rc;
        PRB_HIP(rocprim::exclusive_scan(w.scanTmp.p, tmp2, in, w.row_off.as<int64_t>(), (int64_t)0, (size_t)nsl + 1,
                                        rocprim::plus<int64_t>(), ctx->stream));
      }
      uint64_t cnt[2] = {0, 0}, seeds_maxlen[2] = {0, 0};
      PRB_HIP(hipMemcpyAsync(seeds_maxlen, w.count.p, 16, hipMemcpyDeviceToHost, ctx->stream));
      PRB_HIP(hipMemcpyAsync(&cnt[1], w.row_off.as<int64_t>() + nsl, 8, hipMemcpyDeviceToHost, ctx->stream));
      PRB_HIP(hipStreamSynchronize(ctx->stream));
      cnt[0] = seeds_maxlen[0];
      one_pass_maxlen = std::max<int64_t>(one_pass_maxlen, (int64_t)seeds_maxlen[1]);
      if (cnt[1] > 0) {
        if ((rc = reserve_recs(ctx, w.hitsB, m1, (int64_t)cnt[1]))) return rc;
        PRB_HIP(launch_collect_slices(w.hitsA.p, w.row_count.as<int32_t>(), w.row_off.as<int64_t>(), nsl, w.hitsB.as<HitRec>() + m1,
                                      ctx->stream));
      }
      if ((rc = ctx->time_end("ungapped", 1))) return rc;
      hs->counts[0] += (int64_t)cnt[0];
      if (getenv("PRB_DEBUG_ROWS")) fprintf(stderr, "[pairs] cands %d pairs %lld seeds %lld kept %lld\n", nc, (long long)np, (long long)cnt[0], (long long)cnt[1]);
      m1 += (int64_t)cnt[1];
      c0 = c1;
      if (m1 > (int64_t)UINT32_MAX - 16) {
        set_error("more than 4e9 hits under the -f threshold in one sub-batch: build the database in smaller pages (db -c)");
        return PRB_ERR_NOMEM;
      }
      continue;
    }
    // ---- the list form: seeds counted, written, extended and thinned in passes of their own ----
    all_one_pass = false;
    if ((rc = w.cands.ensure((size_t)nc * sizeof(CandDev))) || (rc = w.row_count.ensure((size_t)(crows + 1) * 4)) ||
        (rc = w.row_off.ensure((size_t)(crows + 1) * 8)) || (rc = w.row_cand.ensure((size_t)(crows + 1) * 4)) ||
        (rc = w.seed_qacc.ensure((size_t)std::max<int64_t>(cents, 1) * 8)))
      return rc;
    PRB_HIP(hipMemcpyAsync(w.cands.p, cd + c0, (size_t)nc * sizeof(CandDev), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = ctx->time_begin())) return rc;
    // one extra zero entry so that the exclusive scan over crows+1 values also yields the total
    PRB_HIP(hipMemsetAsync(w.row_count.as<int32_t>() + crows, 0, 4, ctx->stream));
    PRB_HIP(launch_seed_qacc(w.cands.as<CandDev>(), nc, cents, qb->view, delta, w.seed_qacc.as<double>(), ctx->stream));
    // the rows in the order of (query, database position) - not for the seed-stage output, which keeps the reference's
    // emission order (candidate, database SA entry, query SA entry) - see k_row_key
    const uint32_t *row_perm = nullptr;
    if (last_stage != 1 && crows > 1 && crows < (int64_t)UINT32_MAX && row_shift >= 0) {
      const size_t NR = (size_t)crows;
      const int qmin = cd[c0].query;
      const int dbits = bits_for(std::max<int64_t>(1, ((int64_t)pd.nchars - 1) >> row_shift));
      const int kbits = dbits + bits_for(std::max<int64_t>(1, (int64_t)cd[c1 - 1].query - qmin));
      const bool wide = kbits > 32;
      if ((rc = w.kP.ensure(NR * 8)) || (rc = w.kTmp2.ensure(NR * 8)) || (rc = w.idxA.ensure(NR * 4)) || (rc = w.idxB.ensure(NR * 4)))
        return rc;
      PRB_HIP(launch_row_keys(w.cands.as<CandDev>(), nc, crows, pd, qmin, row_shift, dbits, wide, w.row_cand.as<int32_t>(), w.kP.p,
                              w.idxA.as<uint32_t>(), ctx->stream));
      size_t tmp = 0;
      if (wide) {
        PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, w.kP.as<uint64_t>(), w.kTmp2.as<uint64_t>(), w.idxA.as<uint32_t>(),
                                          w.idxB.as<uint32_t>(), NR, 0, (unsigned)kbits, ctx->stream));
        if ((rc = w.sortTmp.ensure(tmp))) return rc;
        PRB_HIP(rocprim::radix_sort_pairs(w.sortTmp.p, tmp, w.kP.as<uint64_t>(), w.kTmp2.as<uint64_t>(), w.idxA.as<uint32_t>(),
                                          w.idxB.as<uint32_t>(), NR, 0, (unsigned)kbits, ctx->stream));
      } else {
        PRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, w.kP.as<uint32_t>(), w.kTmp2.as<uint32_t>(), w.idxA.as<uint32_t>(),
                                          w.idxB.as<uint32_t>(), NR, 0, (unsigned)kbits, ctx->stream));
        if ((rc = w.sortTmp.ensure(tmp))) return rc;
        PRB_HIP(rocprim::radix_sort_pairs(w.sortTmp.p, tmp, w.kP.as<uint32_t>(), w.kTmp2.as<uint32_t>(), w.idxA.as<uint32_t>(),
                                          w.idxB.as<uint32_t>(), NR, 0, (unsigned)kbits, ctx->stream));
      }
      row_perm = w.idxB.as<uint32_t>();
    }
    PRB_HIP(launch_seed_count(w.cands.as<CandDev>(), nc, crows, qb->view, pd, delta, w.seed_qacc.as<double>(),
                              w.row_count.as<int32_t>(), w.row_cand.as<int32_t>(), row_perm, ctx->stream));
    {
      size_t tmp = 0;
      auto in = rocprim::make_transform_iterator(w.row_count.as<int32_t>(), ToI64());
      PRB_HIP(rocprim::exclusive_scan(nullptr, tmp, in, w.row_off.as<int64_t>(), (int64_t)0, (size_t)crows + 1,
                                      rocprim::plus<int64_t>(), ctx->stream));
      if ((rc = w.scanTmp.ensure(tmp))) return rc;
      PRB_HIP(rocprim::exclusive_scan(w.scanTmp.p, tmp, in, w.row_off.as<int64_t>(), (int64_t)0, (size_t)crows + 1,
                                      rocprim::plus<int64_t>(), ctx->stream));
    }
    int64_t nseed = 0;
    PRB_HIP(hipMemcpyAsync(&nseed, w.row_off.as<int64_t>() + crows, 8, hipMemcpyDeviceToHost, ctx->stream));
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    hs->counts[0] += nseed;
    if (getenv("PRB_DEBUG_ROWS")) fprintf(stderr, "[rows] cands %d rows %lld qents %lld pairs %.0f seeds %lld\n", nc, (long long)crows, (long long)cents, acc, (long long)nseed);
    c0 = c1;
    if (nseed == 0) {
      if ((rc = ctx->time_end("seed", 2))) return rc;
      continue;
    }
    if (nseed > (int64_t)UINT32_MAX - 16) {
      set_error("too many seed hits in one chunk of candidates (a single candidate with more than 4e9 seed hits?): lower "
                "PRB_SEARCH_CHUNK_PAIRS");
      return PRB_ERR_NOMEM;
    }
    if ((rc = w.hitsA.ensure(hits_bytes(nseed)))) return rc;
    HitSoA A = carve_hits(w.hitsA, nseed);
    PRB_HIP(launch_seed_emit(w.cands.as<CandDev>(), nc, crows, qb->view, pd, delta, w.seed_qacc.as<double>(),
                             w.row_cand.as<int32_t>(), w.row_off.as<int64_t>(), A, row_perm, ctx->stream));
    if ((rc = ctx->time_end("seed", 2))) return rc;
    if (last_stage == 1) {
      if ((rc = download_hits(ctx, w, A, nseed, hs->hits))) return rc;
      continue;
    }
    // ---- ungapped extension; hits above the -f threshold are dropped before the sort (they cannot survive the filter) ----
    if ((rc = ctx->time_begin())) return rc;
    PRB_HIP(launch_ungapped(A, nseed, qb->view, pd, sc, eo, max_qlen, ctx->stream));
    if ((rc = ctx->time_end("ungapped", 1))) return rc;
    int64_t mc = 0;
    if ((rc = ctx->time_begin())) return rc;
    if ((rc = compact_below(ctx, w, A, nseed, opts.interaction_threshold, w.cidx, w.hitsB, &mc, m1))) return rc;
    if ((rc = ctx->time_end("filter", 2))) return rc;
    m1 += mc;
    if (m1 > (int64_t)UINT32_MAX - 16) {
      set_error("more than 4e9 hits under the -f threshold in one sub-batch: build the database in smaller pages (db -c)");
      return PRB_ERR_NOMEM;
    }
  }
  if (last_stage == 1) return PRB_OK;
  // ---- sort, redundancy filter ----
  if (m1 == 0) return PRB_OK;
  // A list this long (one very long query against a large page: 1.2e9 hits pass -f for 45 kb against 100 M characters) needs
  // the memory that buffers of stages already over still hold: the seed pools now, the sort keys and the records behind the
  // sort.  (hipFree waits for the device: only where it is needed.)
  size_t big_bytes = (size_t)20 << 30;
  if (const char *e = getenv("PRB_BIG_LIST_BYTES")) big_bytes = (size_t)atof(e); // (tests: every list takes this path)
  const bool big_list = hits_bytes(m1) > big_bytes;
  mem_note("seed chunks done", w);
  if (big_list) {
    w.trim_next = true;
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    w.front.release();
    for (DevBuf *b : {&w.hitsA, &w.hitsTmp, &w.row_count, &w.row_off, &w.row_cand, &w.seed_qacc, &w.cands, &w.trace, &w.resumePool,
                      &w.resumePool2, &w.resumePool3, &w.keptU, &w.keptTrace})
      b->release();
    // (and this context's Raccess workspace, up to 48 GB when it last took a large batch: its launches are over)
    for (DevBuf *b : {&ctx->ra_band, &ctx->ra_vec, &ctx->ra_codes, &ctx->ra_desc}) b->release();
    if ((rc = w.front.init())) return rc;
  }
  mem_note("before the sort", w);
  if ((rc = w.hitsC.ensure(hits_bytes(m1)))) return rc;
  HitSoA B = carve_hits(w.hitsC, m1);
  uint32_t *perm = nullptr;
  if ((rc = ctx->time_begin())) return rc;
  SortBounds sb1 = sb; // hits extended without gaps have one length: the key is 11 + (11 - bits of the longest) bits shorter
  if (all_one_pass && one_pass_maxlen > 0 && !getenv("PRB_SORT_TWO_LENGTHS")) sb1.eq_len_max = (int32_t)one_pass_maxlen;
  if ((rc = sort_hits(ctx, w, w.hitsB.as<HitRec>(), B, m1, qb->nq, sb1, &perm))) return rc;
  if ((rc = ctx->time_end("sort", 9))) return rc;
  if (big_list) { // (the records and the sort's keys are dead)
    perm = nullptr;
    for (DevBuf *b : {&w.hitsB, &w.kE, &w.kL, &w.kQ, &w.kP, &w.kTmp, &w.kTmp2, &w.idxA, &w.idxB, &w.sortTmp}) b->release();
  }
  mem_note("sorted", w);
  int64_t nung = 0;
  if ((rc = ctx->time_begin())) return rc;
  if ((rc = filter_hits(ctx, w, B, m1, opts.interaction_threshold, &nung))) return rc;
  if ((rc = ctx->time_end("filter", 3))) return rc;
  hs->counts[1] += nung;
  if (nung == 0) return PRB_OK;
  // compact survivors into A (A's seed content is no longer needed; the last chunk's pool may be smaller than this list)
  if ((rc = w.hitsA.ensure(hits_bytes(nung)))) return rc;
  HitSoA U = carve_hits(w.hitsA, nung); // (the gapped stage below may narrow this view to a chunk of the list)
  PRB_HIP(launch_gather_hits(B, w.surv.as<uint32_t>(), U, nung, ctx->stream));
  if ((rc = w.first.ensure((size_t)nung))) return rc;
  PRB_HIP(launch_mark_first(U.query, nung, w.first.as<uint8_t>(), ctx->stream));
  if (last_stage == 2) {
    const size_t base = hs->hits.size();
    if ((rc = download_hits(ctx, w, U, nung, hs->hits))) return rc;
    // GetBasePair (rna_interaction_search.cpp:371-385): complementary positions of the diagonal
    for (size_t i = base; i < hs->hits.size(); i++) {
      prb_hit &h = hs->hits[i];
      const uint8_t *qs = qb->enc.data() + qb->off[h.query];
      h.bp_offset = (int64_t)hs->bp.size() / 2;
      const int len = (int)(uint16_t)h.q_len;
      // (soft-masked codes 6..9 mapped to their bases: the reference reads outside BP_pair for them)
      auto code_base = [](unsigned c) { return c <= 5 ? (int)c - 1 : (int)c - 5; };
      for (int j = 0; j < len; j++)
        if (ctx->params.bp_pair[code_base(qs[h.q_sp + j])][code_base(pg.seqs[h.db_sp + j])] != 0) {
          hs->bp.push_back(h.q_sp + j);
          hs->bp.push_back(h.db_sp + j);
          h.bp_count++;
        }
    }
    return PRB_OK;
  }

  // ---- gapped extension ----
  // The state of this stage is ~350 B per hit (the extended hit, work lists, trace slots, hand-over slots): a list
  // longer than PRB_GAPPED_CHUNK_HITS (a 45 kb query against a 100 M character page leaves 5e8 hits behind -f) goes through
  // it in chunks of that many - the list is sorted and filtered already, the extension of a hit depends on nothing but the
  // hit -, and only what is not above the -g threshold is kept of a chunk: the extended hit, the hit it came from, its
  // trace slot.  The final sort + filter then run over the union, as the reference's do over the whole list
  // (rna_interaction_search.cpp:302-320).
  int64_t gchunk = 120000000;
  if (const char *e = getenv("PRB_GAPPED_CHUNK_HITS")) gchunk = std::max<int64_t>(1, (int64_t)atof(e));
  const bool chunked = nung > gchunk;
  const int64_t nmax = chunked ? gchunk : nung;
  if ((rc = w.hitsC.ensure(hits_bytes(nmax)))) return rc;
  if ((rc = w.overflow.ensure((size_t)nmax)) || (rc = w.subset.ensure((size_t)nmax * 4)) ||
      (rc = w.ntrace.ensure((size_t)nmax * 4)) || (rc = w.tierOf.ensure((size_t)nmax)) ||
      (rc = w.listA.ensure((size_t)nmax * 4)) || (rc = w.listB.ensure((size_t)nmax * 4)) || (rc = w.count.ensure(16)) ||
      (rc = w.trace.ensure((size_t)nmax * 2 * kTraceCap * sizeof(uint16_t))))
    return rc;
  // the chunk the cascade below works on (the whole list, unless it is chunked): hits Uc -> Gc, per-hit arrays indexed from 0
  const HitSoA Uall = U;
  HitSoA G = carve_hits(w.hitsC, nmax);
  const uint8_t *firstc = w.first.as<uint8_t>();
  int64_t nch = nmax;
  // The cascade of kernels a hit goes through until one has the capacity for it: LDS tiers 0 and 1
  // (8 lanes per hit), tier 2 (16 lanes), tier 3 (a wavefront per hit), then the wave-per-hit kernel
  // with HBM scratch of any size.
  // PRB_GAPPED_FIRST_TIER exists for the tests: the cascade starts at that kernel (4 = the wave-per-hit
  // kernel alone), so the rarely taken kernels see every hit, from scratch.
  std::vector<int> cascade;
  {
    const char *e = getenv("PRB_GAPPED_FIRST_TIER");
    const int first = e ? std::min(std::max(atoi(e), 0), kWaveTier) : 0;
    for (int t = first; t <= kWaveTier; t++) cascade.push_back(t);
  }
  auto scratch_for = [&](int64_t n, int cap_diag, int cap_rec, GapScratch &gs) -> int {
    gs.cap_diag = cap_diag;
    gs.cap_rec = cap_rec;
    gs.bytes_per_thread = gapped_wave_scratch_bytes(cap_diag, cap_rec);
    if (gs.bytes_per_thread <= kGapWaveLdsBytes && !getenv("PRB_GAPPED_WAVE_HBM")) { // the state fits the LDS of a workgroup
      gs.base = nullptr;
      gs.nthreads = (int32_t)std::min<int64_t>(n, 4096);
      return PRB_OK;
    }
    int64_t nw = std::min<int64_t>(n, 4096);
    while (nw > 64 && (size_t)nw * gs.bytes_per_thread > ((size_t)4 << 30)) nw /= 2;
    gs.nthreads = (int32_t)nw;
    int r = w.gapScratch.ensure((size_t)nw * gs.bytes_per_thread);
    gs.base = w.gapScratch.as<uint8_t>();
    return r;
  };
  // next = the entries of the work list `cur` (nullptr: 0..m-1) whose overflow flag is set
  auto select_overflow = [&](const uint32_t *cur, int64_t m, uint32_t *next, int64_t *mout) -> int {
    size_t tmp = 0, cnt = 0;
    const uint8_t *flags = w.overflow.as<uint8_t>();
    if (cur) {
      PRB_HIP(rocprim::select(nullptr, tmp, cur, flags, next, w.count.as<size_t>(), (size_t)m, ctx->stream));
      if ((rc = w.scanTmp.ensure(tmp))) return rc;
      PRB_HIP(rocprim::select(w.scanTmp.p, tmp, cur, flags, next, w.count.as<size_t>(), (size_t)m, ctx->stream));
    } else {
      rocprim::counting_iterator<uint32_t> iota(0);
      PRB_HIP(rocprim::select(nullptr, tmp, iota, flags, next, w.count.as<size_t>(), (size_t)m, ctx->stream));
      if ((rc = w.scanTmp.ensure(tmp))) return rc;
      PRB_HIP(rocprim::select(w.scanTmp.p, tmp, iota, flags, next, w.count.as<size_t>(), (size_t)m, ctx->stream));
    }
    PRB_HIP(hipMemcpyAsync(&cnt, w.count.p, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    *mout = (int64_t)cnt;
    return PRB_OK;
  };
  // Wave-per-hit kernel with its state in HBM scratch, for the device list `cur` (indices into U;
  // nullptr = all) of m hits.  mode 0 writes G and retries hits that still overflow with a 4x
  // larger scratch; mode 2 writes base pairs at off_dev (indexed by list position).
  // (long traces, search_kernels.hpp: not when the stage runs in chunks - the kept lists renumber the hits -, not for a
  // list that is every hit - PRB_GAPPED_FIRST_TIER=4 -; PRB_TRACE_NO_LONG / PRB_TRACE_LONG_CAP are for the tests)
  LongTrace lt;
  int32_t lt_used = 0;
  auto run_wave = [&](int mode, const uint32_t *cur, int64_t m, uint32_t *spare, const int64_t *off_dev, int handover = 0) -> int {
    if (mode == 0 && !chunked && cur == nullptr && m <= 65536 && !getenv("PRB_TRACE_NO_LONG")) { // (every hit, as a list)
      if ((rc = w.slowList.ensure((size_t)m * 4))) return rc;
      PRB_HIP(launch_iota_u32(w.slowList.as<uint32_t>(), m, ctx->stream));
      cur = w.slowList.as<uint32_t>();
    }
    if (mode == 0 && !chunked && cur != nullptr && m <= 65536 && !getenv("PRB_TRACE_NO_LONG")) {
      const char *ce = getenv("PRB_TRACE_LONG_CAP");
      const int32_t cap = ce ? std::max(1, atoi(ce)) : 1024;
      if (lt_used == 0) {
        if ((rc = w.slowSlot.ensure((size_t)nmax * 4))) return rc;
        PRB_HIP(hipMemsetAsync(w.slowSlot.p, 0xFF, (size_t)nmax * 4, ctx->stream));
      }
      const size_t have = (size_t)lt_used, need = have + (size_t)m;
      if ((rc = grow_keep(ctx, w.slowCnt, have * 8, need * 8)) || (rc = grow_keep(ctx, w.slowTrace, have * 8 * cap, need * 8 * cap))) return rc;
      PRB_HIP(hipMemsetAsync(w.slowCnt.as<int32_t>() + have * 2, 0xFF, (size_t)m * 8, ctx->stream));
      PRB_HIP(launch_assign_slots(cur, m, lt_used, w.slowSlot.as<int32_t>(), ctx->stream));
      lt_used += (int32_t)m;
      lt = LongTrace{w.slowTrace.as<uint32_t>(), w.slowCnt.as<int32_t>(), w.slowSlot.as<int32_t>(), cap};
    }
    int cap_diag = 512, cap_rec = 2048;
    if (mode != 0) { // caps known to suffice for every hit seen so far
      cap_diag = std::max(512, ctx->max_gap_caps);
      cap_rec = cap_diag * 4;
    }
    uint32_t *other = spare;
    while (m > 0) {
      GapScratch gs;
      if ((rc = scratch_for(m, cap_diag, cap_rec, gs))) return rc;
      PRB_HIP(launch_gapped_wave(U, G, m, cur, qb->view, pd, sc, eo, gs, mode, mode == 0 ? w.overflow.as<uint8_t>() : nullptr,
                                 mode == 0 ? w.tierOf.as<uint8_t>() : nullptr, firstc,
                                 mode == 0 ? w.ntrace.as<int32_t>() : nullptr, off_dev, w.bpOut.as<int32_t>(), ctx->stream, handover,
                                 mode == 0 ? lt : LongTrace{}));
      if (mode != 0) break;
      int64_t again = 0;
      if ((rc = select_overflow(cur, m, other, &again))) return rc;
      uint32_t *done_list = const_cast<uint32_t *>(cur);
      cur = other;
      other = done_list ? done_list : (other == w.listA.as<uint32_t>() ? w.listB.as<uint32_t>() : w.listA.as<uint32_t>());
      m = again;
      if (m == 0) break;
      cap_diag *= 4;
      cap_rec *= 4;
      if (cap_diag > 32768) {
        set_error("gapped extension exceeds the supported extension length (32768)");
        return PRB_ERR_STATE;
      }
      ctx->max_gap_caps = std::max(ctx->max_gap_caps, cap_diag);
    }
    return PRB_OK;
  };
  static const char *const kTierTimer[5] = {"gapped", "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow"};
  bool front_called = false;
  const GapResume no_resume{nullptr, nullptr, nullptr, 0};
  // the cascade for the chunk (U, G, nch, firstc)
  auto extend_chunk = [&]() -> int {
  const int64_t nung = nch; // (everything below is per chunk)
  if ((rc = w.accScratch.ensure(std::max<size_t>(gapped_acc_scratch_bytes(), 8)))) return rc;
  PRB_HIP(hipMemsetAsync(w.tierOf.p, 0, (size_t)nung, ctx->stream)); // no hit carries a resume mark yet
  // state dumps for the hits that outgrow tier 0 (~15 %: room for one hit in four, at most 4 M), tier 1
  // (~4 %: one in eight, at most 2 M) and tier 2 (~0.7 %: one in 32, at most 1 M); rs[t] = dumps of tier t
  GapResume rs[kLdsTiers - 1] = {no_resume, no_resume, no_resume};
  static_assert(kLdsTiers == 4, "one pool of state dumps per LDS tier but the last");
  if (!getenv("PRB_GAPPED_NO_RESUME")) {
    rs[0].cap = (int32_t)std::min<int64_t>(nung / 4 + 1024, 4 << 20);
    rs[1].cap = (int32_t)std::min<int64_t>(nung / 8 + 1024, 2 << 20);
    rs[2].cap = (int32_t)std::min<int64_t>(nung / 32 + 1024, 1 << 20);
    if (const char *e = getenv("PRB_GAPPED_RESUME_CAP")) // testing: pools that run out (those hits are redone instead)
      rs[0].cap = rs[1].cap = rs[2].cap = std::max(1, atoi(e));
    DevBuf *pools[3] = {&w.resumePool, &w.resumePool2, &w.resumePool3};
    if ((rc = w.resumeSlot.ensure((size_t)nung * 4 * 3)) || (rc = w.resumeCount.ensure(16))) return rc;
    for (int t = 0; t < 3; t++) {
      if ((rc = pools[t]->ensure((size_t)rs[t].cap * gapped_resume_bytes(t)))) return rc;
      rs[t].slot = w.resumeSlot.as<int32_t>() + (size_t)t * nung;
      rs[t].pool = pools[t]->as<uint8_t>();
      rs[t].count = w.resumeCount.as<uint32_t>() + t;
    }
    PRB_HIP(hipMemsetAsync(w.resumeCount.p, 0, 16, ctx->stream));
    PRB_HIP(hipMemsetAsync(w.resumeSlot.p, 0xFF, (size_t)nung * 4 * 3, ctx->stream));
  }
  {
    const uint32_t *cur = nullptr; // all of U
    int64_t m = nung;
    uint32_t *bufs[2] = {w.listA.as<uint32_t>(), w.listB.as<uint32_t>()};
    int nb = 0;
    // the front kernel on the list (cur, m): what it completes leaves the list
    auto run_front = [&](bool second_only = false) -> int {
      if ((rc = w.frontScratch.ensure(gapped_front_scratch_bytes()))) return rc;
      if ((rc = ctx->time_begin())) return rc;
      PRB_HIP(launch_gapped_front(U, G, m, cur, qb->view, pd, sc, eo, 0, w.overflow.as<uint8_t>(), w.tierOf.as<uint8_t>(),
                                  w.ntrace.as<int32_t>(), w.count.as<unsigned long long>() + 1, w.frontScratch.p, ctx->stream,
                                  second_only && !getenv("PRB_GAPPED_FRONT_PAIRED")));
      int64_t rest = 0;
      if ((rc = select_overflow(cur, m, bufs[nb], &rest))) return rc;
      if (getenv("PRB_DEBUG_ROWS")) fprintf(stderr, "[front] hits %lld, go on %lld\n", (long long)m, (long long)rest);
      cur = bufs[nb];
      nb ^= 1;
      ctx->timers["gapped_front_hits"].launches += m - rest; // (a counter, not a time: hits completed by the front kernel)
      m = rest;
      return ctx->time_end("gapped_front", 1);
    };
    // the LDS tiers and the wavefront-per-hit kernel on the list (cur, m), each taking what the one before it could not hold
    const char *skip_env = getenv("PRB_GAPPED_SKIP_TIERS"); // experiment: bit t set = LDS tier t is left out behind the front kernel
    int skip_mask = 0;
    // (hold_wave: the list is handed back before the wavefront-per-hit kernel instead - see run_slow)
    auto run_cascade = [&](int handover, bool hold_wave) -> int {
      for (size_t c = 0; c < cascade.size() && m > 0; c++) {
        const int tier = cascade[c];
        if (tier < kLdsTiers - 1 && ((skip_mask >> tier) & 1)) continue;
        if (tier == kWaveTier && hold_wave) break;
        if (tier == kWaveTier && front_free && !front_called && !handover) { // (see below: the longest extensions run nearly alone)
          front_free();
          front_called = true;
        }
        if ((rc = ctx->time_begin())) return rc;
        if (tier == kWaveTier) {
          hs->slow_hits += m;
          ctx->slow_hits += m;
          if ((rc = run_wave(0, cur, m, bufs[nb], nullptr, handover))) return rc;
          m = 0;
        } else {
          if (tier == 0) ctx->timers["gapped_tier0_hits"].launches += m; // (a counter, not a time: hits that entered tier 0)
          PRB_HIP(launch_gapped_lds(U, G, m, cur, qb->view, pd, sc, eo, 0, tier, w.overflow.as<uint8_t>(), w.tierOf.as<uint8_t>(),
                                    firstc, w.ntrace.as<int32_t>(), w.trace.as<uint16_t>(), nullptr, nullptr,
                                    w.count.as<unsigned long long>() + 1, tier >= 1 ? rs[tier - 1] : no_resume,
                                    tier < kLdsTiers - 1 ? rs[tier] : no_resume, ctx->stream, handover, w.accScratch.as<double>()));
          int64_t rest = 0;
          if ((rc = select_overflow(cur, m, bufs[nb], &rest))) return rc;
          if (getenv("PRB_DEBUG_ROWS")) fprintf(stderr, "[tier %d%s] hits %lld, go on %lld\n", tier, handover ? ", first direction" : "", (long long)m, (long long)rest);
          cur = bufs[nb];
          nb ^= 1;
          m = rest;
        }
        if ((rc = ctx->time_end(kTierTimer[tier], 1))) return rc;
      }
      return PRB_OK;
    };
    // In front of the cascade (gapped_front.hip; PRB_GAPPED_FRONT=0 leaves it out): the hits neither direction of which
    // finds anything - four in five - are completed by a kernel that only has to prove that.
    const char *fe = getenv("PRB_GAPPED_FRONT");
    const bool front_on = cascade[0] == 0 && !(fe && atoi(fe) == 0) && gapped_front_supported(sc, eo);
    const char *he = getenv("PRB_GAPPED_HANDOVER");
    const bool handover = front_on && !(he && atoi(he) == 0);
    if (front_on) {
      if ((rc = run_front())) return rc;
      skip_mask = skip_env ? atoi(skip_env) : 0;
    }
    if (!handover) {
      if ((rc = run_cascade(0, false))) return rc;
    } else if (m > 0) {
      // The hits that are left have a direction that finds something - or had too many cells for the front kernel.  Nine
      // SECOND directions in ten still find nothing, and a tier pays for proving that what it pays for 16 anti-diagonals of
      // any extension (8 - 40 ns per hit, against the front kernel's 0.6 per direction).  So the cascade first runs FIRST
      // directions only (GapArgs::handover; a hit whose first direction the front kernel completed runs its second one, as
      // ever), the front kernel then looks at the second directions of what the tiers stopped behind, and only the hits
      // whose second direction finds something too come back to the cascade, for that direction.
      const int64_t m1 = m;
      if ((rc = w.listC.ensure((size_t)m1 * 4))) return rc;
      PRB_HIP(hipMemcpyAsync(w.listC.p, cur, (size_t)m1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
      if ((rc = run_cascade(1, true))) return rc;
      // What outgrew the last LDS tier - ~150 extensions per configs[2] query, a wavefront each for 2 ms on a GPU that is
      // otherwise idle - waits for the second pass's: ONE launch of the wavefront-per-hit kernel for both (a launch lasts
      // as long as its longest extension; these hits run both their directions there).
      int64_t n_slow = m;
      if (n_slow > 0) {
        if ((rc = w.slowList.ensure((size_t)n_slow * 4))) return rc;
        PRB_HIP(hipMemcpyAsync(w.slowList.p, cur, (size_t)n_slow * 4, hipMemcpyDeviceToDevice, ctx->stream));
      }
      // the hits an LDS tier stopped behind their first direction
      PRB_HIP(launch_flag_marked(w.tierOf.as<uint8_t>(), w.listC.as<uint32_t>(), m1, kHandoverMark, w.overflow.as<uint8_t>(), ctx->stream));
      cur = w.listC.as<uint32_t>();
      m = m1;
      int64_t mh = 0;
      if ((rc = select_overflow(cur, m, bufs[nb], &mh))) return rc;
      cur = bufs[nb];
      nb ^= 1;
      m = mh;
      if (m > 0) {
        if (rs[0].slot) { // (the hand-over slots of the first pass are not to be taken for this pass's)
          PRB_HIP(hipMemsetAsync(w.resumeCount.p, 0, 16, ctx->stream));
          PRB_HIP(hipMemsetAsync(w.resumeSlot.p, 0xFF, (size_t)nung * 4 * 3, ctx->stream));
        }
        if ((rc = run_front(true))) return rc; // (all of them stopped behind their first direction: a lane per hit)
        if ((rc = run_cascade(0, true))) return rc;
      }
      if (n_slow + m > 0) {
        if (m > 0) { // both passes' lists as one
          if ((rc = grow_keep(ctx, w.slowList, (size_t)n_slow * 4, (size_t)(n_slow + m) * 4))) return rc;
          PRB_HIP(hipMemcpyAsync(w.slowList.as<uint32_t>() + n_slow, cur, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        cur = w.slowList.as<uint32_t>();
        m += n_slow;
        if (front_free && !front_called) {
          front_free();
          front_called = true;
        }
        if ((rc = ctx->time_begin())) return rc;
        hs->slow_hits += m;
        ctx->slow_hits += m;
        if ((rc = run_wave(0, cur, m, bufs[nb], nullptr, 0))) return rc;
        m = 0;
        if ((rc = ctx->time_end(kTierTimer[kWaveTier], 1))) return rc;
      }
    }
  }
  return PRB_OK;
  }; // extend_chunk

  int64_t m2 = 0;
  if (!chunked) {
    if ((rc = extend_chunk())) return rc;
  } else {
    // chunk by chunk; of each chunk, what is not above the -g threshold is appended to the kept lists: the extended hits as
    // records in hitsB (what the final sort takes), the hits they came from, their first-of-query flags, tiers, chain lengths
    // and trace slots - everything the traceback of the final hits reads, indexed by position in the kept lists from now on
    for (int64_t c0 = 0; c0 < nung; c0 += gchunk) {
      nch = std::min<int64_t>(gchunk, nung - c0);
      U = offset_hits(Uall, c0);
      G = carve_hits(w.hitsC, nch);
      firstc = w.first.as<uint8_t>() + c0;
      if ((rc = extend_chunk())) return rc;
      int64_t mc = 0;
      if ((rc = ctx->time_begin())) return rc;
      if ((rc = compact_below(ctx, w, G, nch, opts.final_threshold, w.cidx, w.hitsB, &mc, m2))) return rc;
      if (mc > 0) {
        if ((rc = grow_keep(ctx, w.keptU, (size_t)m2 * sizeof(HitRec), (size_t)(m2 + mc) * sizeof(HitRec))) ||
            (rc = grow_keep(ctx, w.keptFirst, (size_t)m2, (size_t)(m2 + mc))) || (rc = grow_keep(ctx, w.keptTier, (size_t)m2, (size_t)(m2 + mc))) ||
            (rc = grow_keep(ctx, w.keptNtrace, (size_t)m2 * 4, (size_t)(m2 + mc) * 4)) ||
            (rc = grow_keep(ctx, w.keptTrace, (size_t)m2 * 2 * kTraceCap * 2, (size_t)(m2 + mc) * 2 * kTraceCap * 2)))
          return rc;
        const uint32_t *idx = w.cidx.as<uint32_t>();
        PRB_HIP(launch_gather_hits_to_recs(U, idx, w.keptU.as<HitRec>() + m2, mc, ctx->stream));
        PRB_HIP(launch_gather_u8(firstc, idx, w.keptFirst.as<uint8_t>() + m2, mc, ctx->stream));
        PRB_HIP(launch_gather_u8(w.tierOf.as<uint8_t>(), idx, w.keptTier.as<uint8_t>() + m2, mc, ctx->stream));
        PRB_HIP(launch_gather_u32(w.ntrace.as<uint32_t>(), idx, w.keptNtrace.as<uint32_t>() + m2, mc, ctx->stream));
        PRB_HIP(launch_gather_rows(w.trace.p, idx, static_cast<uint8_t *>(w.keptTrace.p) + (size_t)m2 * 2 * kTraceCap * 2, mc,
                                   2 * kTraceCap * 2, ctx->stream));
      }
      if ((rc = ctx->time_end("filter", 2))) return rc;
      if (getenv("PRB_DEBUG_ROWS")) fprintf(stderr, "[gapped chunk] hits %lld at %lld of %lld, kept %lld\n", (long long)nch, (long long)c0, (long long)nung, (long long)mc);
      m2 += mc;
      if (m2 > (int64_t)UINT32_MAX - 16) {
        set_error("more than 4e9 hits under the -g threshold in one sub-batch: build the database in smaller pages (db -c)");
        return PRB_ERR_NOMEM;
      }
    }
    // from here on "the hits before the gapped stage" are the kept ones, in the order they were kept
    if (m2 > 0) {
      if ((rc = w.hitsA.ensure(hits_bytes(m2))) || (rc = w.cidx.ensure((size_t)m2 * 4))) return rc;
      U = carve_hits(w.hitsA, m2);
      PRB_HIP(launch_iota_u32(w.cidx.as<uint32_t>(), m2, ctx->stream));
      PRB_HIP(launch_gather_recs_to_hits(w.keptU.as<HitRec>(), w.cidx.as<uint32_t>(), U, m2, ctx->stream));
    }
  }
  const uint8_t *first_all = chunked ? w.keptFirst.as<uint8_t>() : w.first.as<uint8_t>();
  const uint8_t *tier_all = chunked ? w.keptTier.as<uint8_t>() : w.tierOf.as<uint8_t>();
  const int32_t *ntrace_all = chunked ? w.keptNtrace.as<int32_t>() : w.ntrace.as<int32_t>();
  const uint16_t *trace_all = chunked ? w.keptTrace.as<uint16_t>() : w.trace.as<uint16_t>();
  firstc = first_all; // (the re-extension of final hits for their pairs indexes the kept lists)

  // The front of the NEXT sub-batch's seed path (bandwidth-bound, ~7 ms per configs[2] query) goes out here at the latest:
  // beside what is left of this sub-batch - the ~150 longest extensions on a wavefront each (2 ms, and 2 ms again for
  // their base pairs), the final sort and filter of a few hundred thousand hits, the copies to the host - the GPU is
  // nearly idle.  (Issued right behind k_seed_extend, even on a stream of the lowest priority, it cost the gapped
  // tiers 290 ms per step: its workgroups take LDS and wave slots that tier 0 fills completely.)
  if (front_free && !front_called) {
    front_free();
    front_called = true;
  }
  // ---- final sort + filter (hits above the -g threshold dropped first) ----
  if (!chunked) {
    if ((rc = ctx->time_begin())) return rc;
    if ((rc = compact_below(ctx, w, G, nung, opts.final_threshold, w.cidx, w.hitsB, &m2))) return rc;
    if ((rc = ctx->time_end("filter", 2))) return rc;
  }
  if (m2 == 0) return PRB_OK;
  if ((rc = w.hitsC.ensure(hits_bytes(m2)))) return rc; // (more than a chunk's worth, possibly, when the stage ran in chunks)
  HitSoA S = carve_hits(w.hitsC, m2); // G is dead after the compaction
  if ((rc = ctx->time_begin())) return rc;
  if ((rc = sort_hits(ctx, w, w.hitsB.as<HitRec>(), S, m2, qb->nq, sb, &perm))) return rc;
  if ((rc = ctx->time_end("sort", 9))) return rc;
  int64_t nfin = 0;
  if ((rc = ctx->time_begin())) return rc;
  if ((rc = filter_hits(ctx, w, S, m2, opts.final_threshold, &nfin))) return rc;
  if ((rc = ctx->time_end("filter", 3))) return rc;
  hs->counts[2] += nfin;
  if (nfin == 0) return PRB_OK;
  // final hits, and for each the index of its pre-gapped state in U (for the traceback):
  // final -> sorted position -> position in the compacted list -> index in U
  if ((rc = w.hitsB.ensure(hits_bytes(nfin)))) return rc; // (the records of the compacted list are dead after the sort)
  HitSoA F = carve_hits(w.hitsB, nfin);
  PRB_HIP(launch_gather_hits(S, w.surv.as<uint32_t>(), F, nfin, ctx->stream));
  if ((rc = w.subset2.ensure((size_t)nfin * 4)) || (rc = w.subset.ensure((size_t)nfin * 4))) return rc;
  PRB_HIP(launch_gather_u32(perm, w.surv.as<uint32_t>(), w.subset2.as<uint32_t>(), nfin, ctx->stream));
  PRB_HIP(launch_gather_u32(w.cidx.as<uint32_t>(), w.subset2.as<uint32_t>(), w.subset.as<uint32_t>(), nfin, ctx->stream));

  // ---- base pairs of the survivors: from the trace slots of the extension pass; the few hits
  // the slots cannot describe (wave-kernel hits, chains longer than a slot) are extended again ----
  if ((rc = w.copy_init())) return rc;
  if (w.copy_pending) { // the last results' copy still reads the buffers that are written next
    PRB_HIP(hipStreamWaitEvent(ctx->stream, w.copy_done, 0));
    w.copy_pending = false;
  }
  if ((rc = ctx->time_begin())) return rc;
  if ((rc = w.bpCount.ensure((size_t)(nfin + 1) * 4)) || (rc = w.bpOff.ensure((size_t)(nfin + 1) * 8)) ||
      (rc = w.tierFin.ensure((size_t)nfin)) || (rc = w.ntraceFin.ensure((size_t)nfin * 4)))
    return rc;
  {
    // host copies of four per-hit arrays + the offsets, in one page-locked block that lives with the workspace (fresh
    // vectors of 3 MB each were an mmap, ~750 page faults and a munmap apiece, per sub-batch)
    const size_t NF = (size_t)nfin;
    const size_t tb_bytes = NF * 4 * 3 + NF + 16 + (NF + 1) * 8;
    if ((rc = w.tb_pinned.ensure(w.tb_pinned.cap >= tb_bytes ? tb_bytes : 2 * tb_bytes))) return rc;
    int64_t *off = static_cast<int64_t *>(w.tb_pinned.p);
    uint32_t *pre = reinterpret_cast<uint32_t *>(off + NF + 1); // index of each final hit's pre-gapped state in U
    int32_t *cnt = reinterpret_cast<int32_t *>(pre + NF);
    uint32_t *ntr = reinterpret_cast<uint32_t *>(cnt + NF);
    uint8_t *tier_fin = reinterpret_cast<uint8_t *>(ntr + NF);
    PRB_HIP(hipMemcpyAsync(pre, w.subset.p, NF * 4, hipMemcpyDeviceToHost, ctx->stream));
    PRB_HIP(launch_bp_count(U, nfin, w.subset.as<uint32_t>(), qb->view, pd, sc, ntrace_all, w.bpCount.as<int32_t>(), ctx->stream));
    PRB_HIP(hipMemcpyAsync(cnt, w.bpCount.p, NF * 4, hipMemcpyDeviceToHost, ctx->stream));
    // which kernel of the cascade completed each final hit, and its chain lengths
    PRB_HIP(launch_gather_u8(tier_all, w.subset.as<uint32_t>(), w.tierFin.as<uint8_t>(), nfin, ctx->stream));
    PRB_HIP(launch_gather_u32(reinterpret_cast<const uint32_t *>(ntrace_all), w.subset.as<uint32_t>(), w.ntraceFin.as<uint32_t>(), nfin,
                              ctx->stream));
    PRB_HIP(hipMemcpyAsync(tier_fin, w.tierFin.p, NF, hipMemcpyDeviceToHost, ctx->stream));
    PRB_HIP(hipMemcpyAsync(ntr, w.ntraceFin.p, NF * 4, hipMemcpyDeviceToHost, ctx->stream));
    PRB_HIP(hipStreamSynchronize(ctx->stream));
    off[0] = 0;
    for (int64_t i = 0; i < nfin; i++) off[i + 1] = off[i] + cnt[i];
    const int64_t total = off[nfin];
    if ((rc = w.bpOut.ensure((size_t)std::max<int64_t>(total, 1) * 8))) return rc;
    PRB_HIP(hipMemcpyAsync(w.bpOff.p, off, (size_t)(nfin + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    PRB_HIP(launch_bp_expand(U, nfin, w.subset.as<uint32_t>(), qb->view, pd, sc, first_all, ntrace_all, tier_all, trace_all, lt,
                             w.bpOff.as<int64_t>(), w.bpOut.as<int32_t>(), ctx->stream));
    std::vector<uint32_t> tlist[kWaveTier + 1];
    std::vector<int64_t> toff[kWaveTier + 1];
    const bool no_slots = getenv("PRB_TRACE_NO_SLOTS") != nullptr; // testing: re-extend every final hit as well
    const char *cap_env = getenv("PRB_TRACE_SLOT_CAP");               // testing: pretend the slots are shorter
    const int slot_cap = cap_env ? std::min(kTraceCap, atoi(cap_env)) : kTraceCap;
    for (int64_t i = 0; i < nfin; i++) {
      const int t = std::min<int>(tier_fin[i] & 7, kWaveTier);
      if (!no_slots && (tier_fin[i] & 7) == kLongTraceTier && lt.slot) continue; // (its chains are on record: launch_bp_expand wrote its pairs)
      if (no_slots || t == kWaveTier || (int)(ntr[i] & 0xFFFF) > slot_cap || (int)(ntr[i] >> 16) > slot_cap) {
        tlist[t].push_back(pre[i]);
        toff[t].push_back(off[i]);
      }
    }
    bool any_rerun = false;
    for (int t = 0; t <= kWaveTier; t++) {
      if (tlist[t].empty()) continue;
      const int64_t m = (int64_t)tlist[t].size();
      if (!any_rerun) {
        if ((rc = ctx->time_end("traceback", 1))) return rc;
        if ((rc = ctx->time_begin())) return rc;
        any_rerun = true;
      }
      if ((rc = w.bpOff2.ensure((size_t)m * 8)) || (rc = w.subset2.ensure((size_t)m * 4))) return rc;
      PRB_HIP(hipMemcpyAsync(w.bpOff2.p, toff[t].data(), (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
      PRB_HIP(hipMemcpyAsync(w.subset2.p, tlist[t].data(), (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream));
      if (t == kWaveTier) {
        if ((rc = run_wave(2, w.subset2.as<uint32_t>(), m, nullptr, w.bpOff2.as<int64_t>()))) return rc;
      } else {
        PRB_HIP(launch_gapped_lds(U, G, m, w.subset2.as<uint32_t>(), qb->view, pd, sc, eo, 2, t, nullptr, nullptr,
                                  first_all, nullptr, nullptr, w.bpOff2.as<int64_t>(), w.bpOut.as<int32_t>(),
                                  w.count.as<unsigned long long>() + 1, GapResume{nullptr, nullptr, nullptr, 0},
                                  GapResume{nullptr, nullptr, nullptr, 0}, ctx->stream, 0, w.accScratch.as<double>()));
      }
      PRB_HIP(hipStreamSynchronize(ctx->stream)); // the staging buffers are reused by the next tier
    }
    if (any_rerun) {
      if ((rc = ctx->time_end("traceback_slow", 0))) return rc;
      if ((rc = ctx->time_begin())) return rc;
    }
    // results: records packed on the device (with their base-pair ranges), one asynchronous copy
    // each for hits and pairs into a pinned slot; the background thread appends them to the hit set
    const int slot = hs->next_slot;
    hs->next_slot ^= 1;
    hs->drain->acquire(slot);
    const int64_t bp_base_pairs = hs->bp_ints_total / 2;
    const int64_t nbp_ints = opts.output_style == 0 ? nfin * 4 : total * 2;
    // (page-locking a fresh 50 MB block takes ~35 ms with the GPU idle: when a slot has to grow, to twice the need, so
    // that the larger sub-batches to come still fit)
    const size_t hit_bytes = (size_t)nfin * sizeof(prb_hit), bp_bytes = (size_t)std::max<int64_t>(nbp_ints, 1) * 4;
    if ((rc = w.packed.ensure(hit_bytes)) || (rc = w.pin_hits[slot].ensure(w.pin_hits[slot].cap >= hit_bytes ? hit_bytes : 2 * hit_bytes)) ||
        (rc = w.pin_bp[slot].ensure(w.pin_bp[slot].cap >= bp_bytes ? bp_bytes : 2 * bp_bytes)))
      return rc;
    const int32_t *bp_src;
    if (opts.output_style == 0) {
      // simplified output: only the first and the last pair of a hit are ever printed
      if ((rc = w.bpEnds.ensure((size_t)nfin * 16))) return rc;
      PRB_HIP(launch_bp_ends(w.bpOff.as<int64_t>(), nfin, w.bpOut.as<int32_t>(), w.bpEnds.as<int32_t>(), ctx->stream));
      PRB_HIP(launch_pack_hits(F, nfin, nullptr, nullptr, bp_base_pairs, w.packed.p, ctx->stream));
      bp_src = w.bpEnds.as<int32_t>();
    } else {
      PRB_HIP(launch_pack_hits(F, nfin, w.bpCount.as<int32_t>(), w.bpOff.as<int64_t>(), bp_base_pairs, w.packed.p, ctx->stream));
      bp_src = w.bpOut.as<int32_t>();
    }
    PRB_HIP(hipEventRecord(w.packed_ready, ctx->stream));
    PRB_HIP(hipStreamWaitEvent(w.copy_stream, w.packed_ready, 0));
    PRB_HIP(hipMemcpyAsync(w.pin_hits[slot].p, w.packed.p, (size_t)nfin * sizeof(prb_hit), hipMemcpyDeviceToHost, w.copy_stream));
    if (nbp_ints)
      PRB_HIP(hipMemcpyAsync(w.pin_bp[slot].p, bp_src, (size_t)nbp_ints * 4, hipMemcpyDeviceToHost, w.copy_stream));
    PRB_HIP(hipEventRecord(w.copy_done, w.copy_stream));
    w.copy_pending = true;
    if (hs->on_device) { // device copies for the final hit gather (prb_gather_hits): no re-upload later
      if ((rc = hs->d_hits.append(w.packed.p, (size_t)nfin * sizeof(prb_hit), ctx->stream)) ||
          (rc = hs->d_bp.append(bp_src, (size_t)nbp_ints * 4, ctx->stream)))
        return rc;
    }
    PRB_HIP(hipEventRecord(hs->drain->ev[slot], w.copy_stream));
    hs->drain->submit(Drainer::Job{slot, nfin, nbp_ints});
    hs->hits_total += nfin;
    hs->bp_ints_total += nbp_ints;
  }
  return ctx->time_end("traceback", 2);
}

} // namespace prb

namespace prb {
static std::unique_ptr<SeedPlan> start_seed_plan(prb_ctx *ctx, const prb_qbatch *qb, const prb_db *db, int32_t page, int32_t max_seed_length,
                                                 double hybrid_threshold) {
  std::unique_ptr<SeedPlan> pl(new SeedPlan());
  SeedPlan *P = pl.get();
  const int32_t nq = qb->nq;
  P->db = db;
  P->page = page;
  P->nq = nq;
  P->max_seed_length = max_seed_length;
  P->hybrid_threshold = hybrid_threshold;
  P->per_q.resize((size_t)nq);
  P->qpairs.assign((size_t)nq, 0);
  P->qrows.assign((size_t)nq, 0);
  P->qents.assign((size_t)nq, 0);
  P->done.reset(new std::atomic<int>[(size_t)nq]);
  for (int32_t q = 0; q < nq; q++) P->done[q].store(0, std::memory_order_relaxed);
  const EnergyParams *params = &ctx->params;
  const DbPage *pg = &db->pages[(size_t)page];
  const int hash_size = db->hdr.hash_size, delta = db->hdr.min_accessible_length;
  P->producer = std::thread([P, qb, params, pg, hash_size, delta, nq] {
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel num_threads(host_threads(nq))
    for (;;) {
      const int32_t q = P->next_query.fetch_add(1, std::memory_order_relaxed);
      if (q >= nq) break;
      seed_dfs(*params, qb->enc.data() + qb->off[q], qb->len[q] + 1, qb->sa.data() + qb->off[q], *pg, hash_size, P->max_seed_length, delta,
               P->hybrid_threshold, P->per_q[q]);
      double pairs = 0;
      int64_t rows = 0, ents = 0;
      for (auto &c : P->per_q[q]) {
        c.query = q;
        pairs += (double)(c.ep_q - c.sp_q + 1) * (double)(c.ep_db - c.sp_db + 1);
        rows += (int64_t)c.ep_db - c.sp_db + 1;
        ents += (int64_t)c.ep_q - c.sp_q + 1;
      }
      P->qpairs[q] = pairs;
      P->qrows[q] = rows;
      P->qents[q] = ents;
      P->done[q].store(1, std::memory_order_release);
    }
    P->dfs_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  });
  return pl;
}
} // namespace prb

extern "C" {

int prb_qbatch_seed_search_begin(prb_ctx *ctx, prb_qbatch *qb, const prb_db *db, int32_t page, const prb_ris_opts *opts) {
  if (!ctx || !qb || !db || !opts || page < 0 || page >= (int32_t)db->pages.size() || opts->max_seed_length < 1 ||
      opts->max_seed_length > 63 || qb->repeat_flag != db->hdr.repeat_flag) {
    set_error("prb_qbatch_seed_search_begin: bad argument");
    return PRB_ERR_ARG;
  }
  try {
    qb->plan = start_seed_plan(ctx, qb, db, page, opts->max_seed_length, opts->hybrid_threshold); // (an unused earlier one is joined and dropped)
  } catch (const std::exception &e) {
    set_error(std::string("prb_qbatch_seed_search_begin: ") + e.what());
    return PRB_ERR_NOMEM;
  }
  return PRB_OK;
}

int prb_search_page(prb_ctx *ctx, prb_qbatch *qb, prb_db *db, int32_t page, const prb_ris_opts *opts, int32_t last_stage,
                    prb_hitset **out) {
  if (!ctx || !qb || !db || !opts || !out || qb->ctx->device != ctx->device || db->ctx->device != ctx->device || page < 0 ||
      page >= (int32_t)db->pages.size() || last_stage < 1 || last_stage > 3) {
    set_error("prb_search_page: bad argument");
    return PRB_ERR_ARG;
  }
  *out = nullptr;
  if (!qb->have_acc || qb->W != db->hdr.maximal_span || qb->delta != db->hdr.min_accessible_length) {
    set_error("prb_search_page: query accessibilities must be computed with the database's span / window parameters");
    return PRB_ERR_STATE;
  }
  if (qb->repeat_flag != db->hdr.repeat_flag) {
    set_error("prb_search_page: query batch was encoded with a different repeat flag than the database");
    return PRB_ERR_STATE;
  }
  if (opts->drop_out_w_gap < 0 || opts->drop_out_w_gap > 30 || opts->drop_out_wo_gap < 1 || opts->drop_out_wo_gap > 15 ||
      opts->min_helix_length < 1 || opts->min_helix_length > 16 || opts->max_seed_length < 1 || opts->max_seed_length > 63) {
    set_error("unsupported option: need 0 <= -x <= 30, 1 <= -y <= 15 (beyond that the reference reads outside its "
              "31-entry loop tables), 1 <= -m <= 16, 1 <= -l <= 63 (the seed search keeps a path of 64 characters)");
    return PRB_ERR_ARG;
  }
  PRB_HIP(hipSetDevice(ctx->device));
  {
    // the page on the device (uploaded now unless it is resident or was prefetched), then - while it is searched -
    // the next page on the copy stream
    int slot = -1, rcp;
    if ((rcp = page_slot(ctx, db, page, -1, ctx->stream, &slot))) return rcp;
    PRB_HIP(hipStreamWaitEvent(ctx->stream, db->slot_ready[(size_t)slot], 0));
    const int next = page + 1 < (int32_t)db->pages.size() ? page + 1 : 0;
    if (db->copy_stream && db->mem.size() >= 2 && next != page && db->slot_of_page[(size_t)next] < 0) {
      int ns = -1;
      if ((rcp = page_slot(ctx, db, next, page, db->copy_stream, &ns))) return rcp;
      db->slot_used[(size_t)slot] = ++db->clock; // (the page being searched is the most recently used one)
    }
  }
  // Seed search proper: DFS over the two suffix arrays, per query, on host threads.  It runs
  // in the background while the GPU already works on the first sub-batches: the consumer below
  // only waits for the queries it is about to submit.  If the caller started it ahead
  // (prb_qbatch_seed_search_begin with the same page and options), that one is used.
  const int32_t nq = qb->nq;
  std::unique_ptr<SeedPlan> plan_owner;
  if (qb->plan && qb->plan->db == db && qb->plan->page == page && qb->plan->max_seed_length == opts->max_seed_length &&
      qb->plan->hybrid_threshold == opts->hybrid_threshold) {
    plan_owner = std::move(qb->plan);
  } else {
    plan_owner = start_seed_plan(ctx, qb, db, page, opts->max_seed_length, opts->hybrid_threshold);
  }
  SeedPlan &plan = *plan_owner;
  std::vector<std::vector<SeedCandidate>> &per_q = plan.per_q;
  std::vector<double> &qpairs = plan.qpairs;
  std::vector<int64_t> &qrows = plan.qrows, &qents = plan.qents;
  auto wait_for = [&](int32_t q) { plan.wait_for(q); };
  std::thread &producer = plan.producer;
  auto *hs = new prb_hitset();
  hs->device = ctx->device;
  hs->on_device = ctx->keep_device_records && last_stage == 3;
  hs->d_hits.hint = ctx->keep_hint_hits;
  hs->d_bp.hint = ctx->keep_hint_bp;
  SearchWs &wsp = ws_of(ctx);
  Drainer drain(&hs->hits, &hs->bp, wsp.pin_hits, wsp.pin_bp);
  hs->drain = &drain;
  const char *env = getenv("PRB_SEARCH_PAIRS");
  const double budget = env ? atof(env) : 4.0e8;
  if (last_stage == 3) { // a stream of similar batches: the last hit set's size, with a twentieth to spare, up front
    drain.hint_hits = ctx->host_hint_hits + ctx->host_hint_hits / 20;
    drain.hint_bp = ctx->host_hint_bp + ctx->host_hint_bp / 20;
  }
  int rc = drain.start();
  double wait_ms = 0;
  std::vector<int64_t> cbase, rbase, ebase;
  // the candidates of queries [a, b) converted straight into page-locked memory (queries in parallel), rows and query
  // entries numbered from 0
  struct SubBatch {
    int32_t q0 = 0, q1 = 0;
    CandDev *cd = nullptr;
    int64_t ncand = 0, nrows = 0, nqent = 0;
  };
  auto convert = [&](int32_t a, int32_t b, PinnedBuf &pin, SubBatch &out) -> int {
    const int32_t nb = b - a;
    cbase.assign((size_t)nb + 1, 0);
    rbase.assign((size_t)nb + 1, 0);
    ebase.assign((size_t)nb + 1, 0);
    for (int32_t k = 0; k < nb; k++) {
      cbase[k + 1] = cbase[k] + (int64_t)per_q[a + k].size();
      rbase[k + 1] = rbase[k] + qrows[a + k];
      ebase[k + 1] = ebase[k] + qents[a + k];
    }
    out.q0 = a;
    out.q1 = b;
    out.ncand = cbase[nb];
    out.nrows = rbase[nb];
    out.nqent = ebase[nb];
    if (int r = pin.ensure((size_t)std::max<int64_t>(out.ncand, 1) * sizeof(CandDev))) return r;
    CandDev *cd = static_cast<CandDev *>(pin.p);
    out.cd = cd;
    HostTimer ht(ctx, "host_cands");
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(8, host_threads(nb)))
    for (int32_t k = 0; k < nb; k++) {
      std::vector<SeedCandidate> &v = per_q[a + k];
      CandDev *o = cd + cbase[k];
      int64_t row = rbase[k], ent = ebase[k];
      for (size_t i = 0; i < v.size(); i++) {
        const SeedCandidate &c = v[i];
        o[i] = CandDev{c.sp_q, c.ep_q, c.sp_db, c.ep_db, c.length, c.query, c.score, row, ent};
        row += (int64_t)c.ep_db - c.sp_db + 1;
        ent += (int64_t)c.ep_q - c.sp_q + 1;
      }
      std::vector<SeedCandidate>().swap(v);
    }
    return PRB_OK;
  };
  const SeedKnobs knobs = seed_knobs();
  const PageDev &pdv = db->mem[(size_t)db->slot_of_page[(size_t)page]].view;
  const bool front_ahead = !getenv("PRB_NO_FRONT_AHEAD");
  SubBatch next; // the sub-batch behind the current one, when it has been prepared ahead (next.q1 > next.q0)
  int parity = 0;
  for (int32_t q0 = 0; q0 < nq && rc == PRB_OK;) {
    SubBatch cur;
    if (next.q1 > next.q0 && next.q0 == q0) {
      cur = next;
    } else {
      const auto tw0 = std::chrono::steady_clock::now();
      int32_t q1 = q0;
      double acc = 0;
      for (;;) { // queries [q0, q1) of this sub-batch: as many as fit the pair budget
        if (q1 >= nq) break;
        wait_for(q1);
        if (q1 > q0 && acc + qpairs[q1] > budget) break;
        acc += qpairs[q1];
        q1++;
      }
      wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
      if ((rc = convert(q0, q1, wsp.cand_pinned[parity], cur))) break;
    }
    next = SubBatch{};
    const int32_t q1 = cur.q1;
    const CandDev *cd = cur.cd;
    const int64_t ncand = cur.ncand, nrows = cur.nrows, nqent = cur.nqent;
    // While this sub-batch is sorted, filtered and extended: the candidates of the next one into the other page-locked
    // buffer and the front of its seed path onto the low-priority stream - if its queries' DFS is done (no waiting here).
    std::function<void()> front_free = [&]() {
      if (!front_ahead || q1 >= nq || last_stage == 1 || !knobs.fused || knobs.row_shift < 0) return;
      int32_t q2 = q1;
      double acc = 0;
      for (;;) {
        if (q2 >= nq) break;
        if (!plan.done[q2].load(std::memory_order_acquire)) return; // (the main loop will wait for it, and take it from there)
        if (q2 > q1 && acc + qpairs[q2] > budget) break;
        acc += qpairs[q2];
        q2++;
      }
      SubBatch nb;
      if (convert(q1, q2, wsp.cand_pinned[parity ^ 1], nb) != PRB_OK) return; // (candidates are converted once: `next` must be set)
      next = nb;
      if (nb.ncand == 0 || nb.ncand > INT32_MAX) return;
      double pairs = 0;
      const int32_t c1 = chunk_end(nb.cd, (int32_t)nb.ncand, 0, knobs.chunk_pairs, &pairs);
      if (pairs >= 4.0e9) return;
      const int64_t cents = (c1 < nb.ncand ? nb.cd[c1].qoff : nb.nqent) - nb.cd[0].qoff;
      SearchWs::FrontStage &F = wsp.front;
      int64_t np = -1;
      if (issue_front(wsp, qb, pdv, db->hdr.min_accessible_length, knobs.row_shift, nb.cd, c1, cents, F.stream, &np) != PRB_OK || np < 0) {
        (void)hipStreamSynchronize(F.stream); // (whatever part of it was issued is not used)
        return;
      }
      if (hipEventRecord(F.done, F.stream) != hipSuccess) {
        (void)hipStreamSynchronize(F.stream);
        return;
      }
      F.ahead = true;
      F.cd = nb.cd;
      F.nc = c1;
      F.np = np;
    };
    {
      HostTimer ht(ctx, "host_search_range");
      rc = search_range(ctx, qb, db, page, *opts, last_stage, cd, ncand, nrows, nqent, hs, front_free);
      if (rc == PRB_ERR_NOMEM) // (the seed pools and the gapped stage's state are bounded by their chunk budgets; what grows with a
                               //  query is the list behind -f itself, ~200 B per hit with its sort keys)
        set_error(std::string(prb_last_error()) + " - the hits of queries " + std::to_string(q0) + ".." + std::to_string(q1 - 1) +
                  " that pass -f against this page do not fit the device (lower PRB_GAPPED_CHUNK_HITS / PRB_SEARCH_CHUNK_PAIRS if it is "
                  "their working state; else build the database in smaller pages, db -c, which bounds the list per page)");
    }
    // the pinned candidates are reused by the next sub-batch: their upload must be over
    if (rc == PRB_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = PRB_ERR_HIP;
    q0 = q1;
    parity ^= 1;
    if (q0 < nq) { // extrapolated from the queries done so far, with a tenth to spare (never below the hint from the last set)
      const double scale = 1.1 * (double)nq / (double)q0;
      drain.hint_hits = std::max(drain.hint_hits.load(), (size_t)((double)hs->counts[2] * scale));
      drain.hint_bp = std::max(drain.hint_bp.load(), (size_t)((double)hs->bp_ints_total * scale));
    }
  }
  if (wsp.front.stream && wsp.front.ahead) { // (an error on the way: nothing of a front issued ahead stays in flight)
    (void)hipStreamSynchronize(wsp.front.stream);
    wsp.front.ahead = false;
  }
  producer.join();
  {
    HostTimer ht(ctx, "host_drain_tail");
    const int drc = drain.finish();
    if (rc == PRB_OK) rc = drc;
    hs->drain = nullptr;
  }
  ctx->timers["host_dfs"].ms += plan.dfs_ms;  // wall time of the background DFS
  ctx->timers["host_dfs"].launches++;
  ctx->timers["host_dfs_wait"].ms += wait_ms; // what the GPU pipeline actually waited for it
  ctx->timers["host_dfs_wait"].launches++;
  if (rc != PRB_OK) {
    delete hs;
    return rc;
  }
  if (hs->on_device) {
    ctx->keep_hint_hits = hs->d_hits.used;
    ctx->keep_hint_bp = hs->d_bp.used;
  }
  if (last_stage == 3) {
    ctx->host_hint_hits = hs->hits.size();
    ctx->host_hint_bp = hs->bp.size();
  }
  *out = hs;
  return PRB_OK;
}

int64_t prb_hitset_size(const prb_hitset *hs) { return !hs ? -1 : hs->ext_hits ? hs->ext_nhits : (int64_t)hs->hits.size(); }
const prb_hit *prb_hitset_hits(const prb_hitset *hs) { return !hs ? nullptr : hs->ext_hits ? hs->ext_hits : hs->hits.data(); }
const int32_t *prb_hitset_basepairs(const prb_hitset *hs, int64_t *count) {
  if (!hs) return nullptr;
  if (count) *count = (hs->ext_hits ? hs->ext_bp_ints : (int64_t)hs->bp.size()) / 2;
  return hs->ext_hits ? hs->ext_bp : hs->bp.data();
}
void prb_hitset_counts(const prb_hitset *hs, int64_t counts[3]) {
  for (int i = 0; i < 3; i++) counts[i] = hs ? hs->counts[i] : 0;
}
void prb_hitset_free(prb_hitset *hs) { delete hs; }

int prb_write_lines(const prb_db *db, int32_t nq, const char *const *qnames, const int32_t *qlen_unmasked,
                    const prb_page_hits *pages, int32_t npages, int32_t output_style, int64_t id0, int fd, int64_t *lines,
                    int64_t *bytes) {
  if (!db || nq < 0 || (nq && (!qnames || !qlen_unmasked)) || !pages || npages != (int32_t)db->pages.size() ||
      output_style < 0 || output_style > 1) {
    set_error("prb_write_lines: bad argument");
    return PRB_ERR_ARG;
  }
  try {
    BatchView v;
    std::vector<std::string> names((size_t)nq);
    for (int32_t q = 0; q < nq; q++) names[q] = qnames[q];
    v.nq = (size_t)nq;
    v.names = names.data();
    v.qlen_unmasked = qlen_unmasked;
    for (int32_t p = 0; p < npages; p++) {
      const prb_page_hits &ph = pages[p];
      if (ph.nhits < 0 || ph.npairs < 0 || (ph.nhits && !ph.hits)) {
        set_error("prb_write_lines: bad page");
        return PRB_ERR_ARG;
      }
      const int32_t nseq = db->pages[p].nseq;
      for (int64_t i = 0; i < ph.nhits; i++) { // the records may come from another process: check before indexing
        const prb_hit &x = ph.hits[i];
        if (x.query < 0 || x.query >= nq || x.db_id < 0 || x.db_id >= nseq || x.bp_count < 0 || x.bp_offset < 0 ||
            x.bp_offset + x.bp_count > ph.npairs || (i && x.query < ph.hits[i - 1].query)) {
          set_error("prb_write_lines: hit record " + std::to_string(i) + " of page " + std::to_string(p) + " is inconsistent");
          return PRB_ERR_ARG;
        }
      }
      v.pages.push_back(PageHits{ph.hits, ph.nhits, ph.basepairs, ph.npairs});
    }
    LineSink sink;
    sink.fd = fd;
    const int64_t next = format_batch(v, db->tabs, output_style, id0, sink, format_threads());
    if (lines) *lines = sink.lines;
    if (bytes) *bytes = sink.bytes;
    if (next < 0) {
      set_error("prb_write_lines: write failed");
      return PRB_ERR_IO;
    }
  } catch (const std::exception &e) {
    set_error(std::string("prb_write_lines: ") + e.what());
    return PRB_ERR_NOMEM;
  }
  return PRB_OK;
}

} // extern "C"
