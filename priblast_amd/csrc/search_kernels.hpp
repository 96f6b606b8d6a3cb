// Host-visible interface of the interaction-search kernels (search_kernels.hip):
// seed expansion, ungapped extension, sort + redundancy filter, gapped extension, traceback.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace prb {

// Integer Turner tables (0.01 kcal/mol) as used by the extension stages
// (ungapped_extension.cpp:157-186, gapped_extension.cpp:366-399, 426-473).
// layout of the concatenated integer table `SearchConst::tab`
struct SearchTab {
  static constexpr int kStack = 0;                 // [7][7]
  static constexpr int kInternal = kStack + 49;    // [31]
  static constexpr int kMismatchI = kInternal + 31; // [7][5][5]
  static constexpr int kInt11 = kMismatchI + 175;  // [8][8][5][5]
  static constexpr int kInt21 = kInt11 + 1600;     // [8][8][5][5][5]
  static constexpr int kInt22 = kInt21 + 8000;     // [8][8][5][5][5][5]
  static constexpr int kDangle5 = kInt22 + 40000;  // [8][5]
  static constexpr int kDangle3 = kDangle5 + 40;   // [8][5]
  static constexpr int kBulge = kDangle3 + 40;     // [31] bulge37
  static constexpr int kTau = kBulge + 31;         // [8] TerminalAU if pair type > 2 else 0
  static constexpr int kZero = kTau + 8;           // one 0
  static constexpr int kCount = kZero + 1;
};

struct SearchConst {
  const int32_t *tab;       // SearchTab
  const int32_t *stack37;   // [7][7]
  const int32_t *internal37; // [31]
  const int32_t *mismatchI37; // [7][5][5]
  const int32_t *int11;     // [8][8][5][5]
  const int32_t *int21;     // [8][8][5][5][5]
  const int32_t *int22;     // [8][8][5][5][5][5]
  const int32_t *dangle5;   // [8][5]
  const int32_t *dangle3;   // [8][5]
  const double *bulge;      // [64]: bulge37[u] for u <= 30, logarithmic extrapolation beyond
  int32_t terminal_au;
  // BP_pair (energy_par.hpp:17-23) packed 3 bits per entry, row a (1..4) at bit 15*(a-1);
  // rtype (energy_par.hpp:26) is the involution ((t-1)^1)+1, checked on the host
  uint64_t bp_rows;
  uint32_t pair_mask, wobble_mask; // bit 5 * a + b: bases a, b pair / form a wobble pair (types 3, 4)
  unsigned char bp_pair[25];
};

struct PageDev {
  const uint8_t *seqs; // page text: reversed sequences, 0 after each
  const int32_t *sa;
  const int32_t *sa_seq;  // sequence that SA entry k lies in (GetSeqIdAndStart's search, done once per page)
  const int32_t *blk_seq; // sequence that character 32 b of the page text lies in (b = 0 .. (nchars - 1) / 32)
  const int32_t *start_pos;
  const int32_t *seq_length;
  const float *acc;  // padded to L per sequence; sequence id at start_pos[id] - id
  const float *cond;
  int32_t nchars, nseq;
};

struct QBatchDev {
  const uint8_t *enc;   // query q at off[q], L+1 codes (trailing 0)
  const int32_t *sa;    // same offsets
  const float *acc;     // same offsets (L values + one 0)
  const float *cond;
  const int64_t *off;
  const int32_t *len;   // L
  int32_t nq;
};

struct CandDev { // SeedCandidate + first row of the candidate (one row per db SA entry)
  int32_t sp_q, ep_q, sp_db, ep_db, length, query;
  double score;
  int64_t row0;
  int64_t qoff; // first of the candidate's ep_q - sp_q + 1 entries in the query-side window sums
};

struct HitSoA {
  int32_t *q_sp, *db_sp, *q_len, *db_len, *db_id, *db_id_start, *query;
  double *e_acc, *e_hyb, *e_tot;
};
constexpr int kHitInts = 7, kHitDoubles = 3;
constexpr size_t kHitBytes = kHitInts * 4 + kHitDoubles * 8;

struct ExtOpts {
  int32_t delta;       // min accessible length
  int32_t drop_wo_gap; // -y
  int32_t drop_w_gap;  // -x
  int32_t min_helix;   // -m
};

// ---- seeds ----
// blk_seq for a page (PageDev::blk_seq, blk_seq_entries(nchars) values)
inline int64_t blk_seq_entries(int64_t nchars) { return (nchars >> 5) + 1; }
hipError_t launch_blk_seq(const PageDev &pg, int32_t *blk_seq, hipStream_t s);
// sa_seq[k] for every SA entry of a page
hipError_t launch_sa_seq(const PageDev &pg, int32_t *sa_seq, hipStream_t s);
// qacc[c.qoff + t] = accessibility energy of the query window of candidate c at its SA entry sp_q + t
hipError_t launch_seed_qacc(const CandDev *cands, int32_t ncand, int64_t nq_entries, const QBatchDev &qb, int delta, double *qacc,
                            hipStream_t s);
// row_perm (optional, both passes alike): the order in which the rows are taken - launch_row_keys gives every row its
// candidate (row_cand) and the key ((query - qmin) << dbits | position in the page text >> shift; 64-bit keys if `wide`,
// else 32-bit) to sort the rows by, val = the row
hipError_t launch_row_keys(const CandDev *cands, int32_t ncand, int64_t nrows, const PageDev &pg, int qmin, int shift, int dbits,
                           bool wide, int32_t *row_cand, void *key, uint32_t *val, hipStream_t s);
hipError_t launch_seed_count(const CandDev *cands, int32_t ncand, int64_t nrows, const QBatchDev &qb, const PageDev &pg,
                             int delta, const double *qacc, int32_t *row_count, int32_t *row_cand, const uint32_t *row_perm,
                             hipStream_t s);
hipError_t launch_seed_emit(const CandDev *cands, int32_t ncand, int64_t nrows, const QBatchDev &qb, const PageDev &pg,
                            int delta, const double *qacc, const int32_t *row_cand, const int64_t *row_off, HitSoA hits,
                            const uint32_t *row_perm, hipStream_t s);
// ---- ungapped ----
hipError_t launch_ungapped(HitSoA hits, int64_t n, const QBatchDev &qb, const PageDev &pg, const SearchConst &sc,
                           ExtOpts o, int max_query_len, hipStream_t s);
// ---- sort keys / gather ----
// One-key form of the sort (when the fields fit 64 bits): field widths and the offsets
struct PackedKeyInfo {
  int32_t qmin, lmax; // first query of the sub-batch; upper bound of q_len / db_len
  int32_t bl, bq, bd; // bits of a length, of q_sp, of db_sp
  int32_t one_len = 0; // every hit has q_len = db_len (hits extended without gaps): the key holds the length once
};
hipError_t launch_make_packed_keys(const HitSoA &hits, int64_t n, const PackedKeyInfo &f, uint64_t *key, uint64_t *k_energy,
                                   uint32_t *idx, hipStream_t s);
// after the stable sort by the packed key: runs of equal keys -> (energy, input index) order; *too_long is
// set if a run is longer than the kernel handles (the caller then sorts by the four keys instead)
hipError_t launch_order_keys(const double *v, int64_t n, uint64_t *key, hipStream_t s); // monotone u64 image of doubles
hipError_t launch_make_keys(const HitSoA &hits, int64_t n, uint64_t *k_energy, uint32_t *k_len, uint32_t *k_qsp,
                            uint64_t *k_pos, uint32_t *idx, hipStream_t s);
hipError_t launch_gather_u64(const uint64_t *src, const uint32_t *idx, uint64_t *dst, int64_t n, hipStream_t s);
hipError_t launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t *dst, int64_t n, hipStream_t s);
// out: n records of prb_hit (include/priblast_hip.h) in device memory; bp_base >= 0 also fills their
// base-pair ranges (bp_count / bp_off per hit, or 2 pairs per hit when those are null)
hipError_t launch_pack_hits(const HitSoA &src, int64_t n, const int32_t *bp_count, const int64_t *bp_off, int64_t bp_base,
                            void *out, hipStream_t s);
hipError_t launch_gather_u8(const uint8_t *src, const uint32_t *idx, uint8_t *dst, int64_t n, hipStream_t s);
hipError_t launch_iota_u32(uint32_t *dst, int64_t n, hipStream_t s); // dst[i] = i
// flags[i] = marks[list[i]] & mask != 0
hipError_t launch_flag_marked(const uint8_t *marks, const uint32_t *list, int64_t n, uint8_t mask, uint8_t *flags, hipStream_t s);
// dst row r = src row idx[r], rows of row_bytes (a multiple of 16) bytes
hipError_t launch_gather_rows(const void *src, const uint32_t *idx, void *dst, int64_t n, int row_bytes, hipStream_t s);
hipError_t launch_gather_hits(const HitSoA &src, const uint32_t *idx, HitSoA dst, int64_t n, hipStream_t s);
// The hits between a threshold compaction and the sort behind it, one 64-byte record each: the sort
// ends in a gather in random order, which then costs one cache line per hit instead of one per field.
struct alignas(16) HitRec {
  int32_t q_sp, db_sp, q_len, db_len, db_id, db_id_start, query, pad0;
  double e_acc, e_hyb, e_tot;
  int64_t pad1;
};
static_assert(sizeof(HitRec) == 64, "one cache line half, two per 128-byte line");
// Seeds to extended hits in one pass over the (query SA entry, database SA entry) pairs of a chunk of candidates
// (search_kernels.hip, "seeds -> extended hits in one pass").  pair0[c] = first pair of candidate c; at most
// kMaxFusedCands candidates of at most kMaxFusedEntries query entries each.
constexpr int64_t kMaxFusedCands = 1 << 20, kMaxFusedEntries = 1 << 12;
hipError_t launch_pair_keys(const CandDev *cands, const int64_t *pair0, int32_t ncand, int64_t npairs, const PageDev &pg, int qmin,
                            int shift, int dbits, bool wide, void *key, uint64_t *val, hipStream_t s);
// A workgroup takes kFusePairs pairs and keeps what survives in its slice of `slices` (kFusePairs records of
// kSliceRecBytes each), slice_count[b] of them; nseed[0] += seeds, nseed[1] = max(nseed[1], length of the longest hit kept).  launch_collect_slices packs the slices into `out`
// (slice b at slice_off[b] = the exclusive scan of the counts).
constexpr int kFusePairs = 2048, kSliceRecBytes = 48;
inline int64_t fused_slices(int64_t npairs) { return (npairs + kFusePairs - 1) / kFusePairs; }
hipError_t launch_seed_extend(const CandDev *cands, const uint64_t *vals, int64_t npairs, const QBatchDev &qb, const PageDev &pg,
                              const SearchConst &sc, ExtOpts o, const double *qacc, double thr, int max_query_len, void *slices,
                              int32_t *slice_count, uint64_t *nseed, hipStream_t s);
hipError_t launch_collect_slices(const void *slices, const int32_t *slice_count, const int64_t *slice_off, int64_t nslices, HitRec *out,
                                 hipStream_t s);
// perm_out = perm with every run of equal keys put in (energy, its parts, input index) order
hipError_t launch_fix_ties(const uint64_t *key_sorted, const uint64_t *e_sorted, const uint32_t *perm, int64_t n, const HitRec *recs,
                           uint32_t *perm_out, int32_t *too_long, hipStream_t s);
hipError_t launch_gather_hits_to_recs(const HitSoA &src, const uint32_t *idx, HitRec *dst, int64_t n, hipStream_t s);
// idx == nullptr: in order
hipError_t launch_gather_recs_to_hits(const HitRec *src, const uint32_t *idx, HitSoA dst, int64_t n, hipStream_t s);
hipError_t launch_make_packed_keys_recs(const HitRec *hits, int64_t n, const PackedKeyInfo &f, uint64_t *key, uint64_t *k_energy,
                                        uint32_t *idx, hipStream_t s);
hipError_t launch_flag_not_above(const double *e_tot, int64_t n, double thr, uint8_t *keep, hipStream_t s);
hipError_t launch_mark_first(const int32_t *query, int64_t n, uint8_t *first, hipStream_t s);
// ---- redundancy filter on a sorted list ----
// state: 0 unknown, 1 active, 2 inactive; keep[i] = 1 for survivors
hipError_t launch_filter_init(const HitSoA &h, int64_t n, double thr, int64_t *db_end_key, uint8_t *state, hipStream_t s);
hipError_t launch_filter_round(const HitSoA &h, int64_t n, const int64_t *pmax, uint8_t *state, int32_t *pending,
                               hipStream_t s);
hipError_t launch_filter_final(const HitSoA &h, int64_t n, const int64_t *pmax, const uint8_t *state, uint8_t *keep,
                               hipStream_t s);
// ---- gapped ----
constexpr size_t kGapWaveLdsBytes = 64 * 1024; // a state block up to this size can live in the workgroup's LDS
struct GapScratch {
  uint8_t *base;           // one block per wavefront; nullptr = in the wavefront's (dynamic) LDS
  size_t bytes_per_thread; // bytes per block
  int32_t cap_rec, cap_diag;
  int32_t nthreads;        // number of wavefronts (= grid size)
};
size_t gapped_wave_scratch_bytes(int cap_diag, int cap_rec);
// State dumps of the hits that outgrow LDS tier 0 / tier 1 (mode 0): the next tier continues from
// them instead of starting over.  slot[x] = -1 or the hit's dump; pool = cap dumps of
// gapped_resume_bytes(tier that writes them) each; *count = dumps taken so far (zero it together
// with the slots).  All null / 0: no dumps.  A launch gets the pool it may continue (rin: tier 1
// <- tier 0's, tier 2 <- tier 1's) and the pool it fills (rout: tiers 0 and 1).
struct GapResume {
  int32_t *slot;
  uint8_t *pool;
  uint32_t *count;
  int32_t cap;
};
size_t gapped_resume_bytes(int tier);
// Gapped extension (gapped_lds.hip).  mode 0: extend hits (coords + energies) into `out`,
// overflow[i] = 1 if the state capacity was too small, bp_count[x] = pairs traced back by the
// two extensions of hit x; mode 2: write the base pairs of list entry i at bp_off[i] (hits
// beyond the capacity are skipped).  launch_bp_count: total pairs per list entry.
// launch_gapped_lds: a group of lanes per hit, state in LDS with fixed capacities: tier 0 = 8 lanes,
// 32 anti-diagonals; tier 1 = 16 lanes, 64 anti-diagonals; tier 2 = 64 lanes, 128 anti-diagonals.
// launch_gapped_wave: one wavefront per hit, state in the HBM scratch (`scratch.nthreads`
// wavefronts, `bytes_per_thread` bytes each = gapped_wave_scratch_bytes(cap_diag, cap_rec)).
hipError_t launch_gapped_lds(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb,
                             const PageDev &pg, const SearchConst &sc, ExtOpts o, int mode, int tier, uint8_t *overflow,
                             uint8_t *tier_out, const uint8_t *first_flag, int32_t *bp_count, uint16_t *trace,
                             const int64_t *bp_off, int32_t *bp_out, unsigned long long *next_work /* 8 bytes of scratch */,
                             const GapResume &rin, const GapResume &rout, hipStream_t s, int handover = 0 /* GapArgs::handover */,
                             double *acc_scratch = nullptr /* gapped_acc_scratch_bytes() of device memory */);
size_t gapped_acc_scratch_bytes();
// Trace slots: the extension pass (mode 0, LDS tiers) leaves the first kTraceCap cells (i | j << 8)
// of each direction's traceback chain of hit x at trace[(2x + direction) * kTraceCap ...];
// launch_bp_expand writes the base pairs of the final hits from them (hits of the wave kernel or
// with longer chains are skipped: they are traced by a mode-2 pass).
constexpr int kTraceCap = 32;
// ids of the gapped kernels a hit can be completed by (tier_out): LDS tiers 0..3, then the
// wavefront-per-hit kernel with its state in HBM scratch
constexpr int kLdsTiers = 4, kWaveTier = 4;
// Long traces: the wavefront-per-hit kernel (mode 0) leaves the whole traceback chain of a direction it ran - cells as
// i | j << 16 - in a pool of its own, trace[(2 * slot[x] + direction) * cap ...], and count[2 * slot[x] + direction] = its
// length (-1: the direction was not run by this kernel - an LDS tier's trace slot has it -, -2: longer than cap).  A hit
// whose two chains are all there is reported as kLongTraceTier instead of kWaveTier, and launch_bp_expand writes its pairs
// from them: no second extension of the ~150 longest hits of a query (2 - 3 ms on an otherwise idle GPU).
constexpr int kLongTraceTier = 5;
struct LongTrace {
  uint32_t *trace = nullptr;
  int32_t *count = nullptr;
  const int32_t *slot = nullptr; // per hit x; -1: none
  int32_t cap = 0;
};
// tier_out[x] of a hit whose first direction is done and whose second one is somebody else's business: out.*[x] and
// bp_count[x] hold its state after direction 0.  kResumeMark: the next kernel of the cascade extends the other
// direction (a hit that outgrew a kernel in direction 1, or that the front kernel handed on with a first direction that
// finds nothing).  kHandoverMark: an LDS tier stopped behind direction 0 on purpose (GapArgs::handover): the second
// direction goes to the front kernel first - it finds nothing nine times in ten, which that kernel proves at a fraction of
// a tier's cost.  The low three bits are the LDS tier that has room for the first direction: a hit is reported as the
// larger of the tiers its two directions needed (the tier that would have to extend it again for its pairs).
constexpr uint8_t kResumeMark = 0x40, kHandoverMark = 0x80, kMarkTier = 0x07;
__host__ __device__ inline bool is_resumed(uint8_t t) { return (t & (kResumeMark | kHandoverMark)) != 0; }

// launch_gapped_front (gapped_front.hip): the kernel in front of the cascade.  It proves that a direction finds nothing
// within its `-x` anti-diagonals (phases that are dense over directions / filled cells / (cell, candidate) pairs of 64
// directions at a time) and completes the hits whose two directions both find nothing; the others are flagged in
// overflow[] (with direction 0 handed over when it found nothing) and go on to the LDS tiers.  Completed hits are
// reported as `tier_id`.  Takes -x <= kFrontMaxDrop.
constexpr int kFrontMaxDrop = 16;
bool gapped_front_supported(const SearchConst &sc, const ExtOpts &o);
size_t gapped_front_scratch_bytes(); // HBM scratch of a launch (accessibility sums of the resident wavefronts; stays in L2)
hipError_t launch_gapped_front(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                               const SearchConst &sc, ExtOpts o, int tier_id, uint8_t *overflow, uint8_t *tier_out, int32_t *bp_count,
                               unsigned long long *next_work, void *scratch, hipStream_t s,
                               bool second_only = false /* every hit of the list has its first direction done (a resume / hand-over mark) */);
hipError_t launch_bp_expand(const HitSoA &in, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                            const SearchConst &sc, const uint8_t *first_flag, const int32_t *ntrace, const uint8_t *tier_of,
                            const uint16_t *trace, const LongTrace &lt, const int64_t *bp_off, int32_t *bp_out, hipStream_t s);
hipError_t launch_bp_count(const HitSoA &in, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                           const SearchConst &sc, const int32_t *ntrace, int32_t *bp_count, hipStream_t s);
hipError_t launch_bp_ends(const int64_t *bp_off, int64_t n, const int32_t *bp, int32_t *ends, hipStream_t s);
hipError_t launch_gapped_wave(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb,
                              const PageDev &pg, const SearchConst &sc, ExtOpts o, GapScratch scratch, int mode,
                              uint8_t *overflow, uint8_t *tier_out, const uint8_t *first_flag, int32_t *bp_count,
                              const int64_t *bp_off, int32_t *bp_out, hipStream_t s, int handover = 0,
                              const LongTrace &lt = LongTrace{});
hipError_t launch_assign_slots(const uint32_t *list, int64_t n, int32_t base, int32_t *slot, hipStream_t s); // slot[list[p]] = base + p

} // namespace prb
