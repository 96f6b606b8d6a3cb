// Interaction search on gfx950: seed expansion, ungapped extension, sort keys, the
// order-exact redundancy filter, gapped extension and traceback.
//
//   k_seed_count/emit <-> SeedSearch::CalcInteractionEnergy, GetSeqIdAndStart  seed_search.cpp:47-151
//   k_ungapped        <-> UngappedExtension::Run / LoopEnergy                   ungapped_extension.cpp:30-186
//   k_filter_*        <-> CheckRedundancy                                       rna_interaction_search.cpp:387-424
//   (gapped extension: gapped_lds.hip)
//
// All of this is HBM/L2-bound gather work on small integer tables; there is no dense
// contraction in it.  Energies are doubles built from exact multiples of 0.01 and
// float-derived accessibilities, summed in the reference's order (-ffp-contract=off).
#include "search_kernels.hpp"

#include <cstdlib>

#include "../../include/priblast_hip.h"

#include "search_device.hpp"

namespace prb {

namespace {

constexpr int kBlock = 256;

// id with start_pos[id] <= sp < start_pos[id+1]  (GetSeqIdAndStart, seed_search.cpp:101-141)
__device__ __forceinline__ int seq_of(const PageDev &pg, int sp) {
  int lo = 0, hi = pg.nseq - 1;
  while (lo < hi) {
    int m = (lo + hi + 1) >> 1;
    if (pg.start_pos[m] <= sp) lo = m;
    else hi = m - 1;
  }
  return lo;
}

// SeedSearch::CalcAccessibility, seed_search.cpp:143-151
__device__ __forceinline__ double window_acc(const float *acc, const float *cond, int64_t sp, int length, int delta) {
  double t = acc[sp];
  for (int i = delta; i < length; i++) t += cond[sp + i];
  return t;
}

__device__ __forceinline__ int find_cand(const CandDev *c, int n, int64_t row) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int m = (lo + hi + 1) >> 1;
    if (c[m].row0 <= row) lo = m;
    else hi = m - 1;
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void k_sa_seq(PageDev pg, int32_t *sa_seq) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < pg.nchars) sa_seq[k] = seq_of(pg, pg.sa[k]);
}

// Query-side window sums (SeedSearch::CalcAccessibility): one per (candidate, query SA entry),
// shared by all the database entries of the candidate.
__global__ __launch_bounds__(kBlock) void k_seed_qacc(const CandDev *__restrict__ cands, int ncand, int64_t n, QBatchDev qb, int delta,
                                                      double *__restrict__ qacc) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  int lo = 0, hi = ncand - 1; // candidate with qoff <= e
  while (lo < hi) {
    const int m = (lo + hi + 1) >> 1;
    if (cands[m].qoff <= e) lo = m;
    else hi = m - 1;
  }
  const CandDev c = cands[lo];
  const int64_t qo = qb.off[c.query];
  const int q_sp = qb.sa[qo + c.sp_q + (int)(e - c.qoff)];
  qacc[e] = window_acc(qb.acc + qo, qb.cond + qo, q_sp, c.length, delta);
}

// One row = one (candidate, db SA entry); the query interval is walked inside the row.  The
// count pass finds the row's candidate - a workgroup's 256 rows span at most 256 candidates,
// whose first rows are put in LDS - and leaves it for the emit pass.
// With `row_perm` the threads take the rows in that order (rows sorted by query and database position, see
// k_row_key): thread t works on row row_perm[t], whose candidate k_row_key has left in row_cand; counts and offsets
// are indexed by t.
template <bool kEmit>
__global__ __launch_bounds__(kBlock) void k_seed(const CandDev *__restrict__ cands, int ncand, int64_t nrows, QBatchDev qb,
                                                 PageDev pg, int delta, const double *__restrict__ qacc,
                                                 int32_t *__restrict__ row_count, int32_t *__restrict__ row_cand,
                                                 const int64_t *__restrict__ row_off, HitSoA hits,
                                                 const uint32_t *__restrict__ row_perm) {
  __shared__ int64_t s_row0[kBlock + 1];
  __shared__ int s_c0;
  const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x; // index of the row's count / offset
  int64_t row = slot;
  int ci;
  if (kEmit || row_perm) {
    if (slot >= nrows) return;
    if (row_perm) row = row_perm[slot];
    ci = row_cand[row];
  } else {
    if (threadIdx.x == 0) s_c0 = find_cand(cands, ncand, (int64_t)blockIdx.x * kBlock);
    __syncthreads();
    const int c0 = s_c0;
    for (int t = threadIdx.x; t <= kBlock; t += kBlock) s_row0[t] = c0 + t < ncand ? cands[c0 + t].row0 : INT64_MAX;
    __syncthreads();
    if (row >= nrows) return;
    int lo = 0, hi = kBlock; // last t with s_row0[t] <= row
    while (lo < hi) {
      const int m = (lo + hi + 1) >> 1;
      if (s_row0[m] <= row) lo = m;
      else hi = m - 1;
    }
    ci = c0 + lo;
    row_cand[row] = ci;
  }
  const CandDev c = cands[ci];
  const int k = c.sp_db + (int)(row - c.row0);
  const int db_sp = pg.sa[k];
  const int id = pg.sa_seq[k];
  const int st = pg.seq_length[id] - (db_sp - pg.start_pos[id]) - c.length;
  const int64_t base = (int64_t)pg.start_pos[id] - id;
  const double dba = window_acc(pg.acc + base, pg.cond + base, st, c.length, delta);
  const int32_t *qsa = qb.sa + qb.off[c.query];
  const double *qa_c = qacc + c.qoff - c.sp_q;
  int cnt = 0;
  int64_t w = kEmit ? row_off[slot] : 0;
  for (int j = c.sp_q; j <= c.ep_q; j++) {
    const double qa = qa_c[j];
    const double ie = qa + dba + c.score;
    if (ie < 0) {
      if (kEmit) {
        hits.q_sp[w] = qsa[j];
        hits.db_sp[w] = db_sp;
        hits.q_len[w] = c.length;
        hits.db_len[w] = c.length;
        hits.db_id[w] = id;
        hits.db_id_start[w] = st;
        hits.query[w] = c.query;
        const double ea = qa + dba;
        hits.e_acc[w] = ea;
        hits.e_hyb[w] = c.score;
        hits.e_tot[w] = ea + c.score;
        w++;
      }
      cnt++;
    }
  }
  if (!kEmit) row_count[slot] = cnt;
}

// Sort key of a row = (query, position in the page text): the rows of a candidate are consecutive entries of the
// suffix array, i.e. RANDOM positions of the database, and a seed's extension reads ~5 cache lines around its
// position - at the configs[2] database (0.9 GB of text + accessibilities) every one of them from HBM (measured:
// 377 B fetched per seed in k_ungapped, 105 B in each k_seed pass).  Emitting the seeds of a query in the order of
// their database positions makes neighbouring threads read neighbouring lines.  The low `shift` bits of the position
// are left out of the key (fewer radix passes): rows of one 2^shift window stay in suffix-array order, which is all
// the same to the caches.  The order of the seeds is free: the list is sorted by coordinates afterwards and the ties of
// that sort are broken by the hits' own fields (k_fix_ties).
template <class Key>
__global__ __launch_bounds__(kBlock) void k_row_key(const CandDev *__restrict__ cands, int ncand, int64_t nrows, PageDev pg, int qmin,
                                                    int shift, int dbits, int32_t *__restrict__ row_cand, Key *__restrict__ key,
                                                    uint32_t *__restrict__ val) {
  __shared__ int64_t s_row0[kBlock + 1];
  __shared__ int s_c0;
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (threadIdx.x == 0) s_c0 = find_cand(cands, ncand, (int64_t)blockIdx.x * kBlock);
  __syncthreads();
  const int c0 = s_c0;
  for (int t = threadIdx.x; t <= kBlock; t += kBlock) s_row0[t] = c0 + t < ncand ? cands[c0 + t].row0 : INT64_MAX;
  __syncthreads();
  if (row >= nrows) return;
  int lo = 0, hi = kBlock; // last t with s_row0[t] <= row
  while (lo < hi) {
    const int m = (lo + hi + 1) >> 1;
    if (s_row0[m] <= row) lo = m;
    else hi = m - 1;
  }
  const int ci = c0 + lo;
  row_cand[row] = ci;
  const CandDev c = cands[ci];
  const int k = c.sp_db + (int)(row - c.row0);
  key[row] = ((Key)(uint32_t)(c.query - qmin) << dbits) | (Key)((uint32_t)pg.sa[k] >> shift);
  val[row] = (uint32_t)row;
}

// ---------------------------------------------------------------------------- ungapped
// symmetric interior loops only: stack / 1x1 / 2x2 / generic without asymmetry term
// (UngappedExtension::LoopEnergy, ungapped_extension.cpp:157-186)
__device__ __forceinline__ double loop_energy_ungapped(const SearchConst &sc, int type, int type2, int i, int j, int p,
                                                       int q, const uint8_t *qs, const uint8_t *ds) {
  const int u1 = p - i - 1, u2 = q - j - 1;
  int z;
  if (u1 == 0 && u2 == 0) {
    z = sc.stack37[type * 7 + type2];
  } else {
    const int a = base_of(qs[i + 1]), b = base_of(ds[j + 1]), c = base_of(qs[p - 1]), d = base_of(ds[q - 1]);
    if (u1 + u2 == 2) z = sc.int11[((type * 8 + type2) * 5 + a) * 5 + b];
    else if (u1 == 1 && u2 == 2) z = sc.int21[(((type * 8 + type2) * 5 + a) * 5 + d) * 5 + b];
    else if (u1 == 2 && u2 == 1) z = sc.int21[(((type2 * 8 + type) * 5 + d) * 5 + a) * 5 + c];
    else if (u1 == 2 && u2 == 2) z = sc.int22[((((type * 8 + type2) * 5 + a) * 5 + c) * 5 + d) * 5 + b];
    else z = sc.internal37[u1 + u2] + sc.mismatchI37[(type * 5 + a) * 5 + b] + sc.mismatchI37[(type2 * 5 + d) * 5 + c];
  }
  return div100(sc, z);
}

// The tables UngappedExtension::LoopEnergy reads: all but the 2x2 table (160 KB, one pair in ten)
// are staged in LDS by the workgroup.
struct UngappedTabs {
  const int32_t *stack37, *internal37, *mismatchI37, *int11; // LDS
  const int32_t *int22;                                        // HBM / L2
};
constexpr int kUtStack = 0, kUtInternal = 49, kUtMismatch = 49 + 31, kUtInt11 = 49 + 31 + 175, kUtTotal = 49 + 31 + 175 + 1600;

// UngappedExtension::LoopEnergy (ungapped_extension.cpp:157-186) on values the walk already holds:
// the loop between the pairs (type, type2) is symmetric, u unpaired bases per strand; a, b = the
// bases next to the first pair (query, db), c, d = next to the second one.
__device__ __forceinline__ double loop_energy_ungapped_abcd(const SearchConst &sc, const UngappedTabs &t, int type, int type2,
                                                           int u, int a, int b, int c, int d) {
  int z;
  if (u == 0) z = t.stack37[type * 7 + type2];
  else if (u == 1) z = t.int11[((type * 8 + type2) * 5 + a) * 5 + b];
  else if (u == 2) z = t.int22[((((type * 8 + type2) * 5 + a) * 5 + c) * 5 + d) * 5 + b];
  else z = t.internal37[2 * u] + t.mismatchI37[(type * 5 + a) * 5 + b] + t.mismatchI37[(type2 * 5 + d) * 5 + c];
  return div100(sc, z);
}

// The query side of a walk: in HBM, or staged in LDS by the workgroup.
struct QueryGlobal {
  const uint8_t *qs;
  const float *qacc, *qcond;
  __device__ __forceinline__ unsigned enc(int i) const { return qs[i]; }
  __device__ __forceinline__ float acc(int i) const { return qacc[i]; }
  __device__ __forceinline__ float cond(int i) const { return qcond[i]; }
};
struct QueryLds {
  const uint8_t *qs; // derived from the kernel's __shared__ block
  const float *qacc, *qcond;
  __device__ __forceinline__ unsigned enc(int i) const { return qs[i]; }
  __device__ __forceinline__ float acc(int i) const { return qacc[i]; }
  __device__ __forceinline__ float cond(int i) const { return qcond[i]; }
};

// One seed hit (UngappedExtension::Run, ungapped_extension.cpp:30-155).  A walk is a chain of
// data-dependent steps, and what it costs is memory latency, not arithmetic (PMC: VALU 16 % busy,
// ~400 ns per load, all exposed).  So: the query side and the small energy tables come from LDS;
// the database side of the next kUngappedAhead positions is fetched at once, speculatively
// (clamped to the arrays; a walk that stops earlier just drops the values); and everything a step
// needs again later is carried in registers - the accessibility of the previous position, the
// bases of the previous position and of the position next to the last pair, the type of the
// last pair (the reference re-reads all of them).
// a hit in registers: the seed on the way in, the extended hit on the way out
struct WalkHit {
  int q_sp, db_sp, len, id, id_start;
  double e_acc, e_hyb, e_tot;
};
template <int kUngappedAhead, class Q>
__device__ __forceinline__ void ungapped_walk_reg(const Q &qv, const UngappedTabs &tabs, WalkHit &w, const PageDev &pg,
                                              const SearchConst &sc, const ExtOpts &o) {
  const uint8_t *ds = pg.seqs;
  const int id = w.id;
  const int64_t base = (int64_t)pg.start_pos[id] - id;
  const int64_t nacc = (int64_t)pg.nchars - pg.nseq; // floats in pg.acc / pg.cond
  const int delta = o.delta, drop = o.drop_wo_gap;
  const int q_sp0 = w.q_sp, db_sp0 = w.db_sp, len0 = w.len;

  double min_e = w.e_tot, e = min_e, min_a = w.e_acc, a = min_a, min_h = w.e_hyb, hy = min_h;
  int i = q_sp0, p = q_sp0, j = db_sp0, min_p = p, min_q = db_sp0;
  int id_start = w.id_start, id_end = id_start + len0 - 1, min_id_start = id_start;
  {
    // walk left (:55-94).  (bq, bd) = bases at (i+1, j+1); (cq, cd) = bases at (p-1, q-1), the
    // position next to the last pair; tp = rtype of the pair at (p, q)
    int bq = base_of(qv.enc(i)), bd = base_of(ds[j]);
    int tp = rtype_of(bp_type(sc, bq, bd));
    int cq = 0, cd = 0;
    float acc_next = qv.acc(i); // qacc[i + 1] of the step to come
    bool done = false;
    while (!done) {
      unsigned dcs[kUngappedAhead];
      float dcn[kUngappedAhead];
#pragma unroll
      for (int s = 0; s < kUngappedAhead; s++) {
        const int jj = j - 1 - s;
        dcs[s] = jj >= 0 ? ds[jj] : 0u;
        int64_t ci = base + id_end + 1 + s;
        ci = ci < nacc ? ci : nacc - 1;
        dcn[s] = pg.cond[ci];
      }
#pragma unroll
      for (int s = 0; s < kUngappedAhead; s++) {
        if (done) continue;
        i--;
        j--;
        id_end++;
        if (i < 0 || j < 0) {
          done = true;
          continue;
        }
        const unsigned qc = qv.enc(i), dc = dcs[s];
        if (qc < 2 || dc < 2) {
          done = true;
          continue;
        }
        const float acc_i = qv.acc(i);
        const double ta = acc_i - acc_next + qv.cond(i + delta) + dcn[s]; // float arithmetic, as the reference
        acc_next = acc_i;
        e += ta;
        a += ta;
        const int nq = base_of(qc), nd = base_of(dc);
        if (i == p - 1) {
          cq = nq;
          cd = nd;
        }
        const int type = bp_type(sc, nq, nd);
        if (type != 0) {
          const double le = loop_energy_ungapped_abcd(sc, tabs, type, tp, p - i - 1, bq, bd, cq, cd);
          e += le;
          hy += le;
          if (e < min_e) {
            min_e = e;
            min_a = a;
            min_h = hy;
            min_p = i;
            min_q = j;
          }
          p = i;
          tp = rtype_of(type);
        }
        bq = nq;
        bd = nd;
        if (min_p - i >= drop) done = true;
      }
    }
  }
  e = min_e;
  a = min_a;
  hy = min_h;
  int k = q_sp0 + len0 - 1, r = k, l = db_sp0 + len0 - 1, min_r = r;
  {
    // walk right (:96-145).  (bq, bd) = bases at (k-1, l-1); (cq, cd) = bases at (r+1, s+1); tr = type
    // of the pair at (r, s)
    int bq = base_of(qv.enc(k)), bd = base_of(ds[l]);
    int tr = bp_type(sc, bq, bd);
    int cq = 0, cd = 0;
    float acc_prev = pg.acc[base + id_start]; // dacc[id_start + 1] of the step to come
    bool done = false;
    while (!done) {
      unsigned dcs[kUngappedAhead];
      float dan[kUngappedAhead], dcn[kUngappedAhead];
#pragma unroll
      for (int s = 0; s < kUngappedAhead; s++) {
        const int ll = l + 1 + s;
        dcs[s] = ll < pg.nchars ? ds[ll] : 0u;
        int64_t ai = base + id_start - 1 - s;
        ai = ai > 0 ? ai : 0;
        int64_t ci = ai + delta;
        ci = ci < nacc ? ci : nacc - 1;
        dan[s] = pg.acc[ai];
        dcn[s] = pg.cond[ci];
      }
#pragma unroll
      for (int s = 0; s < kUngappedAhead; s++) {
        if (done) continue;
        k++;
        l++;
        id_start--;
        const unsigned qc = qv.enc(k), dc = dcs[s];
        if (qc < 2 || dc < 2) {
          done = true;
          continue;
        }
        const float acc_i = dan[s];
        const double ta = qv.cond(k) + acc_i - acc_prev + dcn[s];
        acc_prev = acc_i;
        e += ta;
        a += ta;
        const int nq = base_of(qc), nd = base_of(dc);
        if (k == r + 1) {
          cq = nq;
          cd = nd;
        }
        const int type2 = rtype_of(bp_type(sc, nq, nd));
        if (type2 != 0) {
          // loop between (r, s) and (k, l): a, b next to the first pair, c, d next to the second
          const double le = loop_energy_ungapped_abcd(sc, tabs, tr, type2, k - r - 1, cq, cd, bq, bd);
          e += le;
          hy += le;
          if (e < min_e) {
            min_e = e;
            min_a = a;
            min_h = hy;
            min_r = k;
            min_id_start = id_start;
          }
          r = k;
          tr = rtype_of(type2);
        }
        bq = nq;
        bd = nd;
        if (k - min_r >= drop) done = true;
      }
    }
  }
  w.id_start = min_id_start;
  w.q_sp = min_p;
  w.db_sp = min_q;
  w.len = min_r - min_p + 1;
  w.e_tot = min_e;
  w.e_acc = min_a;
  w.e_hyb = min_h;
}
// the same on a hit of a list
template <class Q>
__device__ __forceinline__ void ungapped_walk(const Q &qv, const UngappedTabs &tabs, HitSoA &h, int64_t x, const PageDev &pg,
                                              const SearchConst &sc, const ExtOpts &o) {
  WalkHit w{h.q_sp[x], h.db_sp[x], h.q_len[x], h.db_id[x], h.db_id_start[x], h.e_acc[x], h.e_hyb[x], h.e_tot[x]};
  ungapped_walk_reg<8>(qv, tabs, w, pg, sc, o);
  h.db_id_start[x] = w.id_start;
  h.q_sp[x] = w.q_sp;
  h.db_sp[x] = w.db_sp;
  h.q_len[x] = w.len;
  h.db_len[x] = w.len;
  h.e_tot[x] = w.e_tot;
  h.e_acc[x] = w.e_acc;
  h.e_hyb[x] = w.e_hyb;
}

// A workgroup takes kUngappedPer x kBlock consecutive seed hits.  Seed hits are emitted query
// by query, and the 64 lanes of a wave sit at 64 unrelated positions of that query (its suffix-array
// interval) but at one position of the database.  When all hits of the workgroup belong to one
// query (all but the few workgroups at query boundaries), the query's codes and accessibilities
// (9 B per nucleotide) are staged in LDS; queries longer than the launch's LDS capacity take the
// HBM path.
constexpr int kUngappedPer = 8;
__global__ __launch_bounds__(kBlock) void k_ungapped(HitSoA h, int64_t n, QBatchDev qb, PageDev pg, SearchConst sc, ExtOpts o,
                                                     int qcap) {
  extern __shared__ __align__(16) uint8_t ungapped_smem[];
  __shared__ int32_t s_tab[kUtTotal];
  const int64_t b0 = (int64_t)blockIdx.x * (kBlock * kUngappedPer);
  if (b0 >= n) return;
  const int64_t b1 = (b0 + kBlock * kUngappedPer < n ? b0 + kBlock * kUngappedPer : n) - 1;
  for (int t = threadIdx.x; t < kUtTotal; t += kBlock)
    s_tab[t] = t < kUtInternal   ? sc.stack37[t]
               : t < kUtMismatch ? sc.internal37[t - kUtInternal]
               : t < kUtInt11    ? sc.mismatchI37[t - kUtMismatch]
                                 : sc.int11[t - kUtInt11];
  const UngappedTabs tabs{s_tab + kUtStack, s_tab + kUtInternal, s_tab + kUtMismatch, s_tab + kUtInt11, sc.int22};
  const int q0 = h.query[b0];
  const int nslots = qb.len[q0] + 1;
  const bool staged = q0 == h.query[b1] && nslots <= qcap; // uniform over the workgroup
  if (staged) {
    float *s_acc = reinterpret_cast<float *>(ungapped_smem), *s_cond = s_acc + qcap;
    uint8_t *s_enc = reinterpret_cast<uint8_t *>(s_cond + qcap);
    const int64_t qo = qb.off[q0];
    for (int t = threadIdx.x; t < nslots; t += kBlock) {
      s_acc[t] = qb.acc[qo + t];
      s_cond[t] = qb.cond[qo + t];
      s_enc[t] = qb.enc[qo + t];
    }
    __syncthreads();
    const QueryLds qv{s_enc, s_acc, s_cond};
    for (int64_t x = b0 + threadIdx.x; x <= b1; x += kBlock) ungapped_walk(qv, tabs, h, x, pg, sc, o);
  } else {
    __syncthreads();
    for (int64_t x = b0 + threadIdx.x; x <= b1; x += kBlock) {
      const int64_t qo = qb.off[h.query[x]];
      const QueryGlobal qv{qb.enc + qo, qb.acc + qo, qb.cond + qo};
      ungapped_walk(qv, tabs, h, x, pg, sc, o);
    }
  }
}

// ---------------------------------------------------------------------------- seeds -> extended hits in one pass
// Every (query SA entry, database SA entry) PAIR of a chunk of candidates is one unit of work:
//   k_pair_key    pair p -> sort key (query, database position >> shift) and a 64-bit value
//                 {database position : 32, candidate : 20, query entry within the candidate : 12}
//   (radix sort of the pairs by key)
//   k_seed_extend pair -> SeedSearch::CalcInteractionEnergy's test (seed_search.cpp:47-99); a seed is walked at once
//                 (UngappedExtension::Run) and, if it is not above the -f threshold, kept (see SliceRec)
//   k_collect_slices  what the workgroups kept -> the list of 64-byte records
// Against k_seed (count) / scan / k_seed (emit) / k_ungapped / threshold compaction this never writes the seeds
// (3.3e9 x 48 B per configs[2] step, written once, read and written by the walk, read by the compaction), and
// neighbouring threads work at neighbouring database positions.  The list comes out in no particular order; the sort
// behind it is total on the hits' own fields (k_fix_ties), so the result is the same.
constexpr int kPairCandBits = 20, kPairEntBits = 12;
static_assert(kMaxFusedCands == (1 << kPairCandBits) && kMaxFusedEntries == (1 << kPairEntBits) && kPairCandBits + kPairEntBits == 32,
              "the value of a pair");
constexpr int kSeqBlkShift = 5; // PageDev::blk_seq: one entry per 32 characters of the page text

__global__ __launch_bounds__(kBlock) void k_blk_seq(PageDev pg, int32_t *blk_seq, int64_t nblk) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b < nblk) blk_seq[b] = seq_of(pg, (int)(b << kSeqBlkShift));
}

template <class Key>
__global__ __launch_bounds__(kBlock) void k_pair_key(const CandDev *__restrict__ cands, const int64_t *__restrict__ pair0, int ncand,
                                                     int64_t npairs, PageDev pg, int qmin, int shift, int dbits,
                                                     Key *__restrict__ key, uint64_t *__restrict__ val) {
  __shared__ int64_t s_p0[kBlock + 1];
  __shared__ int s_c0;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (threadIdx.x == 0) { // candidate of the workgroup's first pair
    const int64_t first = (int64_t)blockIdx.x * kBlock;
    int lo = 0, hi = ncand - 1;
    while (lo < hi) {
      const int m = (lo + hi + 1) >> 1;
      if (pair0[m] <= first) lo = m;
      else hi = m - 1;
    }
    s_c0 = lo;
  }
  __syncthreads();
  const int c0 = s_c0;
  for (int t = threadIdx.x; t <= kBlock; t += kBlock) s_p0[t] = c0 + t < ncand ? pair0[c0 + t] : INT64_MAX;
  __syncthreads();
  if (p >= npairs) return;
  int lo = 0, hi = kBlock; // last t with s_p0[t] <= p
  while (lo < hi) {
    const int m = (lo + hi + 1) >> 1;
    if (s_p0[m] <= p) lo = m;
    else hi = m - 1;
  }
  const int ci = c0 + lo;
  const CandDev c = cands[ci];
  const uint32_t pl = (uint32_t)(p - s_p0[lo]), qw = (uint32_t)(c.ep_q - c.sp_q + 1);
  const uint32_t rowl = pl / qw, jrel = pl - rowl * qw;
  const uint32_t db_sp = (uint32_t)pg.sa[c.sp_db + (int)rowl];
  key[p] = ((Key)(uint32_t)(c.query - qmin) << dbits) | (Key)(db_sp >> shift);
  val[p] = (uint64_t)db_sp | ((uint64_t)(uint32_t)ci << 32) | ((uint64_t)jrel << (32 + kPairCandBits));
}

// What a workgroup keeps goes to ITS slice of a sparse list (kFusePairs records, one per pair it takes: no bound to
// check), the position from a counter in LDS; k_collect_slices then packs the used part of every slice into the list
// proper.  (One global counter for all wavefronts - an atomic add with return per 64 pairs - serialised the whole
// kernel: 844 ms per configs[2] step instead of 300, ~14 ns per atomic on one address.)
struct alignas(16) SliceRec {
  int32_t q_sp, db_sp, len, db_id, db_id_start, query;
  double e_acc, e_hyb, e_tot;
};
static_assert(sizeof(SliceRec) == kSliceRecBytes, "three 16-byte stores");

struct FuseArgs {
  const CandDev *cands;
  const uint64_t *vals;
  int64_t npairs;
  const double *qacc;
  double thr;
  SliceRec *slices;        // kFusePairs per workgroup
  int32_t *slice_count;    // records in each slice
  unsigned long long *nseed; // [0] += seeds, [1] = max(., length of the longest hit kept)
  int qcap;
};

__device__ __forceinline__ int pair_cand(uint64_t v) { return (int)((v >> 32) & ((1u << kPairCandBits) - 1)); }

constexpr int kFusePer = kFusePairs / kBlock;
static_assert(kFusePer * kBlock == kFusePairs, "pairs per workgroup");
template <int kAhead>
__global__ __launch_bounds__(kBlock) void k_seed_extend(FuseArgs f, QBatchDev qb, PageDev pg, SearchConst sc, ExtOpts o) {
  extern __shared__ __align__(16) uint8_t ungapped_smem[];
  __shared__ int32_t s_tab[kUtTotal];
  __shared__ unsigned s_kept, s_seeds, s_maxlen;
  const int64_t b0 = (int64_t)blockIdx.x * kFusePairs;
  if (b0 >= f.npairs) return;
  const int64_t b1 = (b0 + kFusePairs < f.npairs ? b0 + kFusePairs : f.npairs) - 1;
  if (threadIdx.x == 0) {
    s_kept = 0;
    s_seeds = 0;
    s_maxlen = 0;
  }
  for (int t = threadIdx.x; t < kUtTotal; t += kBlock)
    s_tab[t] = t < kUtInternal   ? sc.stack37[t]
               : t < kUtMismatch ? sc.internal37[t - kUtInternal]
               : t < kUtInt11    ? sc.mismatchI37[t - kUtMismatch]
                                 : sc.int11[t - kUtInt11];
  const UngappedTabs tabs{s_tab + kUtStack, s_tab + kUtInternal, s_tab + kUtMismatch, s_tab + kUtInt11, sc.int22};
  // the pairs are sorted by query first: one query for the whole workgroup unless it sits on a boundary
  const int q0 = f.cands[pair_cand(f.vals[b0])].query;
  const int nslots = qb.len[q0] + 1;
  const bool staged = q0 == f.cands[pair_cand(f.vals[b1])].query && nslots <= f.qcap; // uniform over the workgroup
  float *s_acc = reinterpret_cast<float *>(ungapped_smem), *s_cond = s_acc + f.qcap;
  uint8_t *s_enc = reinterpret_cast<uint8_t *>(s_cond + f.qcap);
  if (staged) {
    const int64_t qo = qb.off[q0];
    for (int t = threadIdx.x; t < nslots; t += kBlock) {
      s_acc[t] = qb.acc[qo + t];
      s_cond[t] = qb.cond[qo + t];
      s_enc[t] = qb.enc[qo + t];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  SliceRec *slice = f.slices + b0;
  unsigned nseed = 0, maxlen = 0; // (the longest hit kept: the sort behind this packs the lengths into as many bits)
  for (int it = 0; it < kFusePer; it++) {
    const int64_t x = b0 + (int64_t)it * kBlock + threadIdx.x;
    bool keep = false;
    WalkHit w{};
    int query = 0;
    if (x <= b1) {
      const uint64_t v = f.vals[x];
      const int db_sp = (int)(uint32_t)v, jrel = (int)(v >> (32 + kPairCandBits));
      const CandDev c = f.cands[pair_cand(v)];
      int id = pg.blk_seq[db_sp >> kSeqBlkShift];
      while (id + 1 < pg.nseq && pg.start_pos[id + 1] <= db_sp) id++;
      const int sp0 = pg.start_pos[id];
      const int st = pg.seq_length[id] - (db_sp - sp0) - c.length;
      const int64_t base = (int64_t)sp0 - id;
      const double dba = window_acc(pg.acc + base, pg.cond + base, st, c.length, o.delta);
      const double qa = f.qacc[c.qoff + jrel];
      const double ie = qa + dba + c.score;
      if (ie < 0) {
        nseed++;
        const int64_t qo = qb.off[c.query];
        const double ea = qa + dba;
        w = WalkHit{qb.sa[qo + c.sp_q + jrel], db_sp, c.length, id, st, ea, c.score, ea + c.score};
        query = c.query;
        if (staged) {
          const QueryLds qv{s_enc, s_acc, s_cond};
          ungapped_walk_reg<kAhead>(qv, tabs, w, pg, sc, o);
        } else {
          const QueryGlobal qv{qb.enc + qo, qb.acc + qo, qb.cond + qo};
          ungapped_walk_reg<kAhead>(qv, tabs, w, pg, sc, o);
        }
        keep = !(w.e_tot > f.thr);
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      const int leader = __ffsll((long long)mask) - 1;
      unsigned basepos = 0;
      if (lane == leader) basepos = atomicAdd(&s_kept, (unsigned)__popcll(mask));
      basepos = (unsigned)__shfl((int)basepos, leader);
      if (keep) {
        SliceRec r;
        r.q_sp = w.q_sp;
        r.db_sp = w.db_sp;
        r.len = w.len;
        r.db_id = w.id;
        r.db_id_start = w.id_start;
        r.query = query;
        r.e_acc = w.e_acc;
        r.e_hyb = w.e_hyb;
        r.e_tot = w.e_tot;
        slice[basepos + (unsigned)__popcll(mask & ((1ull << lane) - 1))] = r;
        maxlen = (unsigned)w.len > maxlen ? (unsigned)w.len : maxlen;
      }
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    nseed += (unsigned)__shfl_down((int)nseed, d);
    const unsigned o = (unsigned)__shfl_down((int)maxlen, d);
    maxlen = o > maxlen ? o : maxlen;
  }
  if (lane == 0 && nseed) atomicAdd(&s_seeds, nseed);
  if (lane == 0 && maxlen) atomicMax(&s_maxlen, maxlen);
  __syncthreads();
  if (threadIdx.x == 0) {
    f.slice_count[blockIdx.x] = (int32_t)s_kept;
    if (s_seeds) atomicAdd(f.nseed, (unsigned long long)s_seeds);
    if (s_maxlen) atomicMax(f.nseed + 1, (unsigned long long)s_maxlen);
  }
}

// slice b's records to out[off[b] ...]
__global__ __launch_bounds__(kBlock) void k_collect_slices(const SliceRec *__restrict__ slices, const int32_t *__restrict__ count,
                                                           const int64_t *__restrict__ off, HitRec *__restrict__ out) {
  const int n = count[blockIdx.x];
  const SliceRec *src = slices + (int64_t)blockIdx.x * kFusePairs;
  HitRec *dst = out + off[blockIdx.x];
  for (int t = threadIdx.x; t < n; t += kBlock) {
    const SliceRec a = src[t];
    HitRec r;
    r.q_sp = a.q_sp;
    r.db_sp = a.db_sp;
    r.q_len = a.len;
    r.db_len = a.len;
    r.db_id = a.db_id;
    r.db_id_start = a.db_id_start;
    r.query = a.query;
    r.pad0 = 0;
    r.e_acc = a.e_acc;
    r.e_hyb = a.e_hyb;
    r.e_tot = a.e_tot;
    r.pad1 = 0;
    dst[t] = r;
  }
}

// ------------------------------------------------------------------- sort keys / gather

__global__ __launch_bounds__(kBlock) void k_make_keys(HitSoA h, int64_t n, uint64_t *k_energy, uint32_t *k_len,
                                                      uint32_t *k_qsp, uint64_t *k_pos, uint32_t *idx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // compare(): db_sp asc, q_sp asc, db_len desc, q_len desc (rna_interaction_search.cpp:45-55);
  // ties are then broken by energy asc, its two parts and input order (DESIGN.md "total order")
  const uint64_t eb = (uint64_t)__double_as_longlong(h.e_tot[i]);
  k_energy[i] = (eb >> 63) ? ~eb : (eb | 0x8000000000000000ull);
  k_len[i] = ((uint32_t)(0xFFFF - US(h.db_len[i])) << 16) | (uint32_t)(0xFFFF - US(h.q_len[i]));
  k_qsp[i] = (uint32_t)h.q_sp[i];
  k_pos[i] = ((uint64_t)(uint32_t)h.query[i] << 32) | (uint32_t)h.db_sp[i];
  idx[i] = (uint32_t)i;
}

// The same order from ONE 64-bit key when the fields are narrow enough (they are for every page
// of up to 2^27 characters and sequences of up to a few thousand nucleotides): query (relative to
// the sub-batch) | db_sp | q_sp | lmax - db_len | lmax - q_len.  A single stable radix sort over
// the used bits then leaves only the hits with identical coordinates to be put in (energy, input
// order) order, which k_fix_ties does run by run.
__global__ __launch_bounds__(kBlock) void k_make_packed_keys(HitSoA h, int64_t n, PackedKeyInfo f, uint64_t *key,
                                                             uint64_t *k_energy, uint32_t *idx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t eb = (uint64_t)__double_as_longlong(h.e_tot[i]);
  k_energy[i] = (eb >> 63) ? ~eb : (eb | 0x8000000000000000ull);
  uint64_t k = (uint64_t)(uint32_t)(h.query[i] - f.qmin);
  k = (k << f.bd) | (uint32_t)h.db_sp[i];
  k = (k << f.bq) | (uint32_t)h.q_sp[i];
  k = (k << f.bl) | (uint32_t)(f.lmax - US(h.db_len[i]));
  k = (k << f.bl) | (uint32_t)(f.lmax - US(h.q_len[i]));
  key[i] = k;
  idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void k_make_packed_keys_recs(const HitRec *__restrict__ h, int64_t n, PackedKeyInfo f,
                                                                  uint64_t *key, uint64_t *k_energy, uint32_t *idx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const HitRec r = h[i];
  const uint64_t eb = (uint64_t)__double_as_longlong(r.e_tot);
  k_energy[i] = (eb >> 63) ? ~eb : (eb | 0x8000000000000000ull);
  uint64_t k = (uint64_t)(uint32_t)(r.query - f.qmin);
  k = (k << f.bd) | (uint32_t)r.db_sp;
  k = (k << f.bq) | (uint32_t)r.q_sp;
  k = (k << f.bl) | (uint32_t)(f.lmax - US(r.db_len));
  if (!f.one_len) k = (k << f.bl) | (uint32_t)(f.lmax - US(r.q_len)); // (one_len: q_len = db_len in every hit of the list)
  key[i] = k;
  idx[i] = (uint32_t)i;
}

constexpr int kMaxTieRun = 4096;
// monotone map of a double onto unsigned integers
__device__ __forceinline__ uint64_t order_bits(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// Runs of identical coordinates are put in (energy, hybridization part, accessibility part, input index) order: two
// hits still tied after the three energies are identical records, so the result does not depend on the order in which
// the seeds were produced (chunks of candidates, pairs sorted by database position).  Every element of a run finds its
// own rank among the others (a run is several seeds of one duplex extended to the same hit: a few elements, the same
// total energy more often than not, so most comparisons go on to the records) - independent loads, where one thread
// per run sorting by insertion was a chain of dependent ones (2.1 ms per 2.5e7 hits; this: 1.6 ms).
__global__ __launch_bounds__(kBlock) void k_fix_ties(const uint64_t *__restrict__ key, const uint64_t *__restrict__ e,
                                                     const uint32_t *__restrict__ perm, int64_t n, const HitRec *__restrict__ recs,
                                                     uint32_t *__restrict__ perm_out, int32_t *too_long) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = key[i];
  const uint32_t pi = perm[i];
  const bool left = i > 0 && key[i - 1] == k, right = i + 1 < n && key[i + 1] == k;
  if (!left && !right) {
    perm_out[i] = pi;
    return;
  }
  int64_t s = i, t = i + 1;
  while (s > 0 && key[s - 1] == k && i - s <= kMaxTieRun) s--;
  while (t < n && key[t] == k && t - s <= kMaxTieRun) t++;
  if (t - s > kMaxTieRun) {
    *too_long = 1;
    perm_out[i] = pi;
    return;
  }
  const uint64_t ei = e[i];
  const uint64_t hi = order_bits(recs[pi].e_hyb), ai = order_bits(recs[pi].e_acc);
  int rank = 0;
  for (int64_t b = s; b < t; b++) {
    if (b == i) continue;
    const uint64_t eb = e[b];
    bool before = eb < ei; // element b sorts before this one
    if (eb == ei) {
      const uint32_t pb = perm[b];
      const uint64_t hb = order_bits(recs[pb].e_hyb);
      if (hb != hi) {
        before = hb < hi;
      } else {
        const uint64_t ab = order_bits(recs[pb].e_acc);
        before = ab != ai ? ab < ai : pb < pi;
      }
    }
    rank += before ? 1 : 0;
  }
  perm_out[s + rank] = pi;
}

__global__ __launch_bounds__(kBlock) void k_order_keys(const double *__restrict__ v, int64_t n, uint64_t *__restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) key[i] = order_bits(v[i]);
}

template <class T> __global__ __launch_bounds__(kBlock) void k_gather(const T *src, const uint32_t *idx, T *dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

__global__ __launch_bounds__(kBlock) void k_gather_hits(HitSoA s, const uint32_t *idx, HitSoA d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = idx[i];
  d.q_sp[i] = s.q_sp[j];
  d.db_sp[i] = s.db_sp[j];
  d.q_len[i] = s.q_len[j];
  d.db_len[i] = s.db_len[j];
  d.db_id[i] = s.db_id[j];
  d.db_id_start[i] = s.db_id_start[j];
  d.query[i] = s.query[j];
  d.e_acc[i] = s.e_acc[j];
  d.e_hyb[i] = s.e_hyb[j];
  d.e_tot[i] = s.e_tot[j];
}

__global__ __launch_bounds__(kBlock) void k_gather_hits_to_recs(HitSoA s, const uint32_t *idx, HitRec *d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = idx[i];
  HitRec r;
  r.q_sp = s.q_sp[j];
  r.db_sp = s.db_sp[j];
  r.q_len = s.q_len[j];
  r.db_len = s.db_len[j];
  r.db_id = s.db_id[j];
  r.db_id_start = s.db_id_start[j];
  r.query = s.query[j];
  r.pad0 = 0;
  r.e_acc = s.e_acc[j];
  r.e_hyb = s.e_hyb[j];
  r.e_tot = s.e_tot[j];
  r.pad1 = 0;
  d[i] = r;
}
__global__ __launch_bounds__(kBlock) void k_gather_recs_to_hits(const HitRec *__restrict__ s, const uint32_t *idx, HitSoA d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const HitRec r = s[idx ? (int64_t)idx[i] : i];
  d.q_sp[i] = r.q_sp;
  d.db_sp[i] = r.db_sp;
  d.q_len[i] = r.q_len;
  d.db_len[i] = r.db_len;
  d.db_id[i] = r.db_id;
  d.db_id_start[i] = r.db_id_start;
  d.query[i] = r.query;
  d.e_acc[i] = r.e_acc;
  d.e_hyb[i] = r.e_hyb;
  d.e_tot[i] = r.e_tot;
}

// SoA hits -> the C ABI's records (include/priblast_hip.h), so that one copy brings them to the host.
// bp_base >= 0: the records also get their range in the hit set's base-pair array: hit i has
// bp_count[i] pairs from pair index bp_base + bp_off[i] on (no arrays: the two end pairs, 2 per hit).
__global__ __launch_bounds__(kBlock) void k_pack_hits(HitSoA s, int64_t n, const int32_t *__restrict__ bp_count,
                                                      const int64_t *__restrict__ bp_off, int64_t bp_base, prb_hit *out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  prb_hit h;
  h.q_sp = s.q_sp[i];
  h.db_sp = s.db_sp[i];
  h.q_len = s.q_len[i];
  h.db_len = s.db_len[i];
  h.db_id = s.db_id[i];
  h.db_id_start = s.db_id_start[i];
  h.e_acc = s.e_acc[i];
  h.e_hyb = s.e_hyb[i];
  h.e_tot = s.e_tot[i];
  h.query = s.query[i];
  h.bp_count = bp_base < 0 ? 0 : bp_count ? bp_count[i] : 2;
  h.bp_offset = bp_base < 0 ? 0 : bp_base + (bp_off ? bp_off[i] : 2 * i);
  out[i] = h;
}

// keep[i] = 1 unless E_i > threshold.  A hit above the threshold is flagged by CheckRedundancy
// the moment the sweep reaches it and never flags anything else (as the contained hit of an
// earlier scan it loses: E_a <= threshold < E_b), so it can be dropped BEFORE the sort.
__global__ __launch_bounds__(kBlock) void k_flag_not_above(const double *e_tot, int64_t n, double thr, uint8_t *keep) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) keep[i] = !(e_tot[i] > thr);
}

// first[i] = 1 for the first hit of every query in a (query-sorted) list
__global__ __launch_bounds__(kBlock) void k_mark_first(const int32_t *query, int64_t n, uint8_t *first) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) first[i] = i == 0 || query[i] != query[i - 1];
}

// ------------------------------------------------------------------- redundancy filter
// CheckRedundancy (rna_interaction_search.cpp:387-424) is a sequential sweep over the sorted
// list.  Its result has a closed form (DESIGN.md "redundancy filter"): with
//   R(a,b)  = a before b, a.q range contains b.q range, a.db_end >= b.db_end
//   over(i) = E_i > threshold
//   active(i) = !over(i) and no active a<i with R(a,i) and E_a <= E_i       (i scans)
//   flagA(i)  = active(i) and exists b>i: R(i,b), E_i > E_b, and no active a<i with R(a,b), E_a <= E_b
// the survivors are exactly active(i) && !flagA(i).  `active` is resolved by rounds that
// only ever turn "unknown" into a final state, so any interleaving of threads is safe.
constexpr uint8_t kUnknown = 0, kActive = 1, kInactive = 2;

struct Box {
  int qs, qe, ds, de, query;
  double e;
};
__device__ __forceinline__ Box box_of(const HitSoA &h, int64_t i) {
  Box b;
  b.qs = h.q_sp[i];
  b.ds = h.db_sp[i];
  b.qe = b.qs + US(h.q_len[i]) - 1;
  b.de = b.ds + US(h.db_len[i]) - 1;
  b.query = h.query[i];
  b.e = h.e_tot[i];
  return b;
}
__device__ __forceinline__ bool contains(const Box &a, const Box &b) { return a.qe >= b.qe && a.qs <= b.qs && a.de >= b.de; }
// running maximum of (query, db_end) packed so that a later query always dominates
__device__ __forceinline__ int64_t pack_end(int query, int de) { return ((int64_t)query << 32) | (uint32_t)de; }

__global__ __launch_bounds__(kBlock) void k_filter_init(HitSoA h, int64_t n, double thr, int64_t *end_key, uint8_t *state) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const Box b = box_of(h, i);
  end_key[i] = pack_end(b.query, b.de);
  state[i] = b.e > thr ? kInactive : kUnknown;
  // (state of an over-threshold hit is "inactive": it never scans)
}

__global__ __launch_bounds__(kBlock) void k_filter_round(HitSoA h, int64_t n, const int64_t *__restrict__ pmax,
                                                         uint8_t *state, int32_t *pending) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (state[i] != kUnknown) return;
  const Box b = box_of(h, i);
  const int64_t need = pack_end(b.query, b.de);
  bool wait = false;
  uint8_t res = kActive;
  for (int64_t a = i - 1; a >= 0; a--) {
    if (pmax[a] < need) break; // nothing at or before a (in this query) reaches b's db end
    const Box c = box_of(h, a);
    if (c.query != b.query) break;
    if (contains(c, b) && c.e <= b.e) {
      const uint8_t sa = state[a];
      if (sa == kActive) {
        res = kInactive;
        wait = false;
        break;
      }
      if (sa == kUnknown) wait = true;
    }
  }
  if (wait) {
    *pending = 1;
  } else {
    state[i] = res;
  }
}

__global__ __launch_bounds__(kBlock) void k_filter_final(HitSoA h, int64_t n, const int64_t *__restrict__ pmax,
                                                         const uint8_t *__restrict__ state, uint8_t *keep) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint8_t k = 0;
  if (state[i] == kActive) {
    k = 1;
    const Box a = box_of(h, i);
    for (int64_t j = i + 1; j < n && k; j++) {
      const Box b = box_of(h, j);
      if (b.query != a.query || b.ds > a.de) break;
      if (contains(a, b) && a.e > b.e) {
        // was b already flagged when a's scan reached it?  (by an active hit before a)
        const int64_t need = pack_end(b.query, b.de);
        bool flagged = false;
        for (int64_t c = i - 1; c >= 0; c--) {
          if (pmax[c] < need) break;
          const Box x = box_of(h, c);
          if (x.query != b.query) break;
          if (state[c] == kActive && contains(x, b) && x.e <= b.e) {
            flagged = true;
            break;
          }
        }
        if (!flagged) k = 0;
      }
    }
  }
  keep[i] = k;
}

// The same two scans on a WINDOW of the list held in LDS.  A scan is a chain of dependent loads - the running maximum,
// the box, the state of the hit before, and so on, ~4 hits back and ~4 forward - and as loads from L2 that chain is what
// the kernels above cost (0.96 / 1.5 ms per 23 M hits where the list itself is 0.1 ms of HBM traffic).  A workgroup
// loads the boxes of its 256 hits and of the neighbours on both sides once, coalesced, and scans in LDS; a scan that
// leaves the window goes on in memory.  k_filter_round_tile also repeats its round within the window until nothing
// changes (a state only ever goes from unknown to final, so more rounds, in any interleaving, give the same result):
// what is left for the next launch are the chains that cross a window.
constexpr int kFilterBack = 128, kFilterFwd = 128, kFilterWin = kFilterBack + kBlock + kFilterFwd;
struct FilterWindow {
  int qs[kFilterWin], qe[kFilterWin], ds[kFilterWin], de[kFilterWin], query[kFilterWin];
  double e[kFilterWin];
  int64_t pmax[kFilterWin];
  uint8_t state[kFilterWin];
  int64_t w0, w1; // the hits [w0, w1) of the list
};
__device__ __forceinline__ void window_load(FilterWindow &w, const HitSoA &h, int64_t n, const int64_t *__restrict__ pmax,
                                            const uint8_t *state, int64_t t0, int fwd) {
  const int64_t w0 = t0 > kFilterBack ? t0 - kFilterBack : 0;
  const int64_t w1 = t0 + kBlock + fwd < n ? t0 + kBlock + fwd : n;
  if (threadIdx.x == 0) {
    w.w0 = w0;
    w.w1 = w1;
  }
  for (int64_t g = w0 + threadIdx.x; g < w1; g += kBlock) {
    const int k = (int)(g - w0);
    const int qs = h.q_sp[g], ds = h.db_sp[g];
    w.qs[k] = qs;
    w.ds[k] = ds;
    w.qe[k] = qs + US(h.q_len[g]) - 1;
    w.de[k] = ds + US(h.db_len[g]) - 1;
    w.query[k] = h.query[g];
    w.e[k] = h.e_tot[g];
    w.pmax[k] = pmax[g];
    w.state[k] = state[g];
  }
  __syncthreads();
}
__device__ __forceinline__ Box window_box(const FilterWindow &w, const HitSoA &h, int64_t g) {
  if (g >= w.w0 && g < w.w1) {
    const int k = (int)(g - w.w0);
    return Box{w.qs[k], w.qe[k], w.ds[k], w.de[k], w.query[k], w.e[k]};
  }
  return box_of(h, g);
}
__device__ __forceinline__ int64_t window_pmax(const FilterWindow &w, const int64_t *__restrict__ pmax, int64_t g) {
  return g >= w.w0 && g < w.w1 ? w.pmax[g - w.w0] : pmax[g];
}
__device__ __forceinline__ uint8_t window_state(const FilterWindow &w, const uint8_t *state, int64_t g) {
  return g >= w.w0 && g < w.w1 ? ((const volatile uint8_t *)w.state)[g - w.w0] : ((const volatile uint8_t *)state)[g];
}

__global__ __launch_bounds__(kBlock) void k_filter_round_tile(HitSoA h, int64_t n, const int64_t *__restrict__ pmax, uint8_t *state,
                                                              int32_t *pending) {
  __shared__ FilterWindow w;
  __shared__ int s_progress;
  const int64_t t0 = (int64_t)blockIdx.x * kBlock, i = t0 + threadIdx.x;
  window_load(w, h, n, pmax, state, t0, 0);
  bool mine = i < n && window_state(w, state, i) == kUnknown;
  const Box b = i < n ? window_box(w, h, i) : Box{0, 0, 0, 0, 0, 0.0};
  const int64_t need = pack_end(b.query, b.de);
  for (int iter = 0; iter < 64; iter++) {
    if (threadIdx.x == 0) s_progress = 0;
    __syncthreads();
    if (mine) {
      bool wait = false;
      uint8_t res = kActive;
      for (int64_t a = i - 1; a >= 0; a--) {
        if (window_pmax(w, pmax, a) < need) break; // nothing at or before a (in this query) reaches b's db end
        const Box c = window_box(w, h, a);
        if (c.query != b.query) break;
        if (contains(c, b) && c.e <= b.e) {
          const uint8_t sa = window_state(w, state, a);
          if (sa == kActive) {
            res = kInactive;
            wait = false;
            break;
          }
          if (sa == kUnknown) wait = true;
        }
      }
      if (!wait) {
        ((volatile uint8_t *)w.state)[i - w.w0] = res;
        state[i] = res;
        mine = false;
        s_progress = 1;
      }
    }
    __syncthreads();
    const int p = s_progress;
    __syncthreads();
    if (!p) break;
  }
  if (mine) *pending = 1;
}

__global__ __launch_bounds__(kBlock) void k_filter_final_tile(HitSoA h, int64_t n, const int64_t *__restrict__ pmax,
                                                              const uint8_t *__restrict__ state, uint8_t *keep) {
  __shared__ FilterWindow w;
  const int64_t t0 = (int64_t)blockIdx.x * kBlock, i = t0 + threadIdx.x;
  window_load(w, h, n, pmax, state, t0, kFilterFwd);
  if (i >= n) return;
  uint8_t k = 0;
  if (window_state(w, state, i) == kActive) {
    k = 1;
    const Box a = window_box(w, h, i);
    for (int64_t j = i + 1; j < n && k; j++) {
      const Box b = window_box(w, h, j);
      if (b.query != a.query || b.ds > a.de) break;
      if (contains(a, b) && a.e > b.e) {
        // was b already flagged when a's scan reached it?  (by an active hit before a)
        const int64_t need = pack_end(b.query, b.de);
        bool flagged = false;
        for (int64_t c = i - 1; c >= 0; c--) {
          if (window_pmax(w, pmax, c) < need) break;
          const Box x = window_box(w, h, c);
          if (x.query != b.query) break;
          if (window_state(w, state, c) == kActive && contains(x, b) && x.e <= b.e) {
            flagged = true;
            break;
          }
        }
        if (!flagged) k = 0;
      }
    }
  }
  keep[i] = k;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

} // namespace

hipError_t launch_sa_seq(const PageDev &pg, int32_t *sa_seq, hipStream_t s) {
  if (pg.nchars <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_sa_seq, grid_for(pg.nchars), dim3(kBlock), 0, s, pg, sa_seq);
  return hipGetLastError();
}
hipError_t launch_seed_qacc(const CandDev *cands, int32_t ncand, int64_t nq_entries, const QBatchDev &qb, int delta, double *qacc,
                            hipStream_t s) {
  if (nq_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_seed_qacc, grid_for(nq_entries), dim3(kBlock), 0, s, cands, ncand, nq_entries, qb, delta, qacc);
  return hipGetLastError();
}
hipError_t launch_row_keys(const CandDev *cands, int32_t ncand, int64_t nrows, const PageDev &pg, int qmin, int shift, int dbits,
                           bool wide, int32_t *row_cand, void *key, uint32_t *val, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  if (wide)
    hipLaunchKernelGGL(k_row_key<uint64_t>, grid_for(nrows), dim3(kBlock), 0, s, cands, ncand, nrows, pg, qmin, shift, dbits, row_cand,
                       (uint64_t *)key, val);
  else
    hipLaunchKernelGGL(k_row_key<uint32_t>, grid_for(nrows), dim3(kBlock), 0, s, cands, ncand, nrows, pg, qmin, shift, dbits, row_cand,
                       (uint32_t *)key, val);
  return hipGetLastError();
}
hipError_t launch_seed_count(const CandDev *cands, int32_t ncand, int64_t nrows, const QBatchDev &qb, const PageDev &pg,
                             int delta, const double *qacc, int32_t *row_count, int32_t *row_cand, const uint32_t *row_perm,
                             hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_seed<false>, grid_for(nrows), dim3(kBlock), 0, s, cands, ncand, nrows, qb, pg, delta, qacc, row_count,
                     row_cand, (const int64_t *)nullptr, HitSoA{}, row_perm);
  return hipGetLastError();
}
hipError_t launch_seed_emit(const CandDev *cands, int32_t ncand, int64_t nrows, const QBatchDev &qb, const PageDev &pg,
                            int delta, const double *qacc, const int32_t *row_cand, const int64_t *row_off, HitSoA hits,
                            const uint32_t *row_perm, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_seed<true>, grid_for(nrows), dim3(kBlock), 0, s, cands, ncand, nrows, qb, pg, delta, qacc,
                     (int32_t *)nullptr, const_cast<int32_t *>(row_cand), row_off, hits, row_perm);
  return hipGetLastError();
}
hipError_t launch_blk_seq(const PageDev &pg, int32_t *blk_seq, hipStream_t s) {
  const int64_t nblk = blk_seq_entries(pg.nchars);
  hipLaunchKernelGGL(k_blk_seq, grid_for(nblk), dim3(kBlock), 0, s, pg, blk_seq, nblk);
  return hipGetLastError();
}
hipError_t launch_pair_keys(const CandDev *cands, const int64_t *pair0, int32_t ncand, int64_t npairs, const PageDev &pg, int qmin,
                            int shift, int dbits, bool wide, void *key, uint64_t *val, hipStream_t s) {
  if (npairs <= 0) return hipSuccess;
  if (wide)
    hipLaunchKernelGGL(k_pair_key<uint64_t>, grid_for(npairs), dim3(kBlock), 0, s, cands, pair0, ncand, npairs, pg, qmin, shift, dbits,
                       (uint64_t *)key, val);
  else
    hipLaunchKernelGGL(k_pair_key<uint32_t>, grid_for(npairs), dim3(kBlock), 0, s, cands, pair0, ncand, npairs, pg, qmin, shift, dbits,
                       (uint32_t *)key, val);
  return hipGetLastError();
}
hipError_t launch_seed_extend(const CandDev *cands, const uint64_t *vals, int64_t npairs, const QBatchDev &qb, const PageDev &pg,
                              const SearchConst &sc, ExtOpts o, const double *qacc, double thr, int max_query_len, void *slices,
                              int32_t *slice_count, uint64_t *nseed, hipStream_t s) {
  if (npairs <= 0) return hipSuccess;
  int qcap = (max_query_len + 1 + 3) & ~3; // as launch_ungapped
  if (qcap > 7168) qcap = 7168;
  FuseArgs f{cands, vals, npairs, qacc, thr, static_cast<SliceRec *>(slices), slice_count, reinterpret_cast<unsigned long long *>(nseed),
             qcap};
  // (4, 5 or 6 positions fetched ahead instead of 8: the same 273-290 ms per configs[2] step)
  hipLaunchKernelGGL(k_seed_extend<8>, dim3((unsigned)fused_slices(npairs)), dim3(kBlock), (size_t)qcap * 9, s, f, qb, pg, sc, o);
  return hipGetLastError();
}
hipError_t launch_collect_slices(const void *slices, const int32_t *slice_count, const int64_t *slice_off, int64_t nslices, HitRec *out,
                                 hipStream_t s) {
  if (nslices <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_collect_slices, dim3((unsigned)nslices), dim3(kBlock), 0, s, static_cast<const SliceRec *>(slices), slice_count,
                     slice_off, out);
  return hipGetLastError();
}
hipError_t launch_ungapped(HitSoA hits, int64_t n, const QBatchDev &qb, const PageDev &pg, const SearchConst &sc, ExtOpts o,
                           int max_query_len, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  // LDS slots per query array: the longest query of the batch, at most 7168 (63 KB per workgroup)
  int qcap = (max_query_len + 1 + 3) & ~3;
  if (qcap > 7168) qcap = 7168;
  const int64_t per_block = (int64_t)kBlock * kUngappedPer;
  hipLaunchKernelGGL(k_ungapped, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), (size_t)qcap * 9, s, hits, n,
                     qb, pg, sc, o, qcap);
  return hipGetLastError();
}
hipError_t launch_make_keys(const HitSoA &hits, int64_t n, uint64_t *k_energy, uint32_t *k_len, uint32_t *k_qsp,
                            uint64_t *k_pos, uint32_t *idx, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_make_keys, grid_for(n), dim3(kBlock), 0, s, hits, n, k_energy, k_len, k_qsp, k_pos, idx);
  return hipGetLastError();
}
hipError_t launch_make_packed_keys(const HitSoA &hits, int64_t n, const PackedKeyInfo &f, uint64_t *key, uint64_t *k_energy,
                                   uint32_t *idx, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_make_packed_keys, grid_for(n), dim3(kBlock), 0, s, hits, n, f, key, k_energy, idx);
  return hipGetLastError();
}
hipError_t launch_fix_ties(const uint64_t *key_sorted, const uint64_t *e_sorted, const uint32_t *perm, int64_t n, const HitRec *recs,
                           uint32_t *perm_out, int32_t *too_long, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fix_ties, grid_for(n), dim3(kBlock), 0, s, key_sorted, e_sorted, perm, n, recs, perm_out, too_long);
  return hipGetLastError();
}
hipError_t launch_order_keys(const double *v, int64_t n, uint64_t *key, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_order_keys, grid_for(n), dim3(kBlock), 0, s, v, n, key);
  return hipGetLastError();
}
hipError_t launch_gather_u64(const uint64_t *src, const uint32_t *idx, uint64_t *dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather<uint64_t>, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
  return hipGetLastError();
}
hipError_t launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t *dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather<uint32_t>, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
  return hipGetLastError();
}
namespace {
__global__ __launch_bounds__(256) void k_iota_u32(uint32_t *dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (uint32_t)i;
}
// dst row r = src row idx[r]; rows of `words` 16-byte words, one thread per word
__global__ __launch_bounds__(256) void k_gather_rows(const uint4 *__restrict__ src, const uint32_t *__restrict__ idx, uint4 *dst, int64_t n,
                                                     int words) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = t / words;
  if (r >= n) return;
  const int wi = (int)(t - r * words);
  dst[r * words + wi] = src[(int64_t)idx[r] * words + wi];
}
} // namespace
namespace {
__global__ __launch_bounds__(256) void k_flag_marked(const uint8_t *__restrict__ marks, const uint32_t *__restrict__ list, int64_t n, uint8_t mask,
                                                     uint8_t *flags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[i] = (marks[list[i]] & mask) ? 1 : 0;
}
} // namespace
hipError_t launch_flag_marked(const uint8_t *marks, const uint32_t *list, int64_t n, uint8_t mask, uint8_t *flags, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_flag_marked, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, marks, list, n, mask, flags);
  return hipGetLastError();
}
hipError_t launch_iota_u32(uint32_t *dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, n);
  return hipGetLastError();
}
hipError_t launch_gather_rows(const void *src, const uint32_t *idx, void *dst, int64_t n, int row_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (row_bytes % 16 != 0) return hipErrorInvalidValue;
  const int words = row_bytes / 16;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * words + 255) / 256)), dim3(256), 0, s, static_cast<const uint4 *>(src), idx,
                     static_cast<uint4 *>(dst), n, words);
  return hipGetLastError();
}
hipError_t launch_gather_u8(const uint8_t *src, const uint32_t *idx, uint8_t *dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather<uint8_t>, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
  return hipGetLastError();
}
hipError_t launch_gather_hits(const HitSoA &src, const uint32_t *idx, HitSoA dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_hits, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
  return hipGetLastError();
}
hipError_t launch_gather_hits_to_recs(const HitSoA &src, const uint32_t *idx, HitRec *dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_hits_to_recs, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
  return hipGetLastError();
}
hipError_t launch_gather_recs_to_hits(const HitRec *src, const uint32_t *idx, HitSoA dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_recs_to_hits, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
  return hipGetLastError();
}
hipError_t launch_make_packed_keys_recs(const HitRec *hits, int64_t n, const PackedKeyInfo &f, uint64_t *key, uint64_t *k_energy,
                                        uint32_t *idx, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_make_packed_keys_recs, grid_for(n), dim3(kBlock), 0, s, hits, n, f, key, k_energy, idx);
  return hipGetLastError();
}
hipError_t launch_pack_hits(const HitSoA &src, int64_t n, const int32_t *bp_count, const int64_t *bp_off, int64_t bp_base,
                            void *out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_hits, grid_for(n), dim3(kBlock), 0, s, src, n, bp_count, bp_off, bp_base, static_cast<prb_hit *>(out));
  return hipGetLastError();
}
hipError_t launch_flag_not_above(const double *e_tot, int64_t n, double thr, uint8_t *keep, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_flag_not_above, grid_for(n), dim3(kBlock), 0, s, e_tot, n, thr, keep);
  return hipGetLastError();
}
hipError_t launch_mark_first(const int32_t *query, int64_t n, uint8_t *first, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_mark_first, grid_for(n), dim3(kBlock), 0, s, query, n, first);
  return hipGetLastError();
}
hipError_t launch_filter_init(const HitSoA &h, int64_t n, double thr, int64_t *end_key, uint8_t *state, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_filter_init, grid_for(n), dim3(kBlock), 0, s, h, n, thr, end_key, state);
  return hipGetLastError();
}
hipError_t launch_filter_round(const HitSoA &h, int64_t n, const int64_t *pmax, uint8_t *state, int32_t *pending,
                               hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const bool tiles = !(getenv("PRB_FILTER_TILES") && atoi(getenv("PRB_FILTER_TILES")) == 0);
  if (tiles) hipLaunchKernelGGL(k_filter_round_tile, grid_for(n), dim3(kBlock), 0, s, h, n, pmax, state, pending);
  else hipLaunchKernelGGL(k_filter_round, grid_for(n), dim3(kBlock), 0, s, h, n, pmax, state, pending);
  return hipGetLastError();
}
hipError_t launch_filter_final(const HitSoA &h, int64_t n, const int64_t *pmax, const uint8_t *state, uint8_t *keep,
                               hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const bool tiles = !(getenv("PRB_FILTER_TILES") && atoi(getenv("PRB_FILTER_TILES")) == 0);
  if (tiles) hipLaunchKernelGGL(k_filter_final_tile, grid_for(n), dim3(kBlock), 0, s, h, n, pmax, state, keep);
  else hipLaunchKernelGGL(k_filter_final, grid_for(n), dim3(kBlock), 0, s, h, n, pmax, state, keep);
  return hipGetLastError();
}
} // namespace prb
