// Interaction search on gfx950: seed expansion, ungapped extension, sort keys, the
// order-exact redundancy filter, gapped extension and traceback.
//
//   k_seed_count/emit <-> SeedSearch::CalcInteractionEnergy, GetSeqIdAndStart  seed_search.cpp:47-151
//   k_ungapped        <-> UngappedExtension::Run / LoopEnergy                   ungapped_extension.cpp:30-186
//   k_filter_*        <-> CheckRedundancy                                       rna_interaction_search.cpp:387-424
//   k_gapped          <-> GappedExtension::Run / extension / CheckHelixLength / CalcDangleEnergy /
//                         traceback / LoopEnergy                                gapped_extension.cpp:33-473
//
// All of this is HBM/L2-bound gather work on small integer tables; there is no dense
// contraction in it.  Energies are doubles built from exact multiples of 0.01 and
// float-derived accessibilities, summed in the reference's order (-ffp-contract=off).
#include "search_kernels.hpp"

namespace prb {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int base_of(unsigned c) { return c <= 5 ? (int)c - 1 : (int)c - 5; } // codes 6..9 = soft-masked
__device__ __forceinline__ int get_char(const uint8_t *s, int64_t i) {                          // GetChar, gapped_extension.cpp:401-407
  if (i < 0) return 0;
  unsigned c = s[i];
  return c < 2 ? 0 : base_of(c);
}
__device__ __forceinline__ int bp_type(const SearchConst &sc, int a, int b) { return sc.bp_pair[a * 5 + b]; }

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

// One row = one (candidate, db SA entry); the query interval is walked inside the row.
template <bool kEmit>
__global__ __launch_bounds__(kBlock) void k_seed(const CandDev *__restrict__ cands, int ncand, int64_t nrows, QBatchDev qb,
                                                 PageDev pg, int delta, int32_t *__restrict__ row_count,
                                                 const int64_t *__restrict__ row_off, HitSoA hits) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= nrows) return;
  const CandDev c = cands[find_cand(cands, ncand, row)];
  const int k = c.sp_db + (int)(row - c.row0);
  const int db_sp = pg.sa[k];
  const int id = seq_of(pg, db_sp);
  const int st = pg.seq_length[id] - (db_sp - pg.start_pos[id]) - c.length;
  const int64_t base = (int64_t)pg.start_pos[id] - id;
  const double dba = window_acc(pg.acc + base, pg.cond + base, st, c.length, delta);
  const int64_t qo = qb.off[c.query];
  const int32_t *qsa = qb.sa + qo;
  int cnt = 0;
  int64_t w = kEmit ? row_off[row] : 0;
  for (int j = c.sp_q; j <= c.ep_q; j++) {
    const int q_sp = qsa[j];
    const double qa = window_acc(qb.acc + qo, qb.cond + qo, q_sp, c.length, delta);
    const double ie = qa + dba + c.score;
    if (ie < 0) {
      if (kEmit) {
        hits.q_sp[w] = q_sp;
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
  if (!kEmit) row_count[row] = cnt;
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
  return (double)z / 100.0;
}

__global__ __launch_bounds__(kBlock) void k_ungapped(HitSoA h, int64_t n, QBatchDev qb, PageDev pg, SearchConst sc,
                                                     ExtOpts o) {
  const int64_t x = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (x >= n) return;
  const int query = h.query[x];
  const int64_t qo = qb.off[query];
  const uint8_t *qs = qb.enc + qo;
  const float *qacc = qb.acc + qo, *qcond = qb.cond + qo;
  const uint8_t *ds = pg.seqs;
  const int id = h.db_id[x];
  const int64_t base = (int64_t)pg.start_pos[id] - id;
  const float *dacc = pg.acc + base, *dcond = pg.cond + base;
  const int delta = o.delta, drop = o.drop_wo_gap;
  const int q_sp0 = h.q_sp[x], db_sp0 = h.db_sp[x], len0 = h.q_len[x];

  double min_e = h.e_tot[x], e = min_e, min_a = h.e_acc[x], a = min_a, min_h = h.e_hyb[x], hy = min_h;
  int i = q_sp0, p = q_sp0, j = db_sp0, q = db_sp0, min_p = p, min_q = q;
  int id_start = h.db_id_start[x], id_end = id_start + len0 - 1, min_id_start = id_start;
  for (;;) { // walk left (ungapped_extension.cpp:55-94)
    i--;
    j--;
    id_end++;
    if (i < 0 || j < 0 || qs[i] < 2 || ds[j] < 2) break;
    const double ta = qacc[i] - qacc[i + 1] + qcond[i + delta] + dcond[id_end]; // float arithmetic, as the reference
    e += ta;
    a += ta;
    const int type = bp_type(sc, base_of(qs[i]), base_of(ds[j]));
    if (type != 0) {
      const int type2 = sc.rtype[bp_type(sc, base_of(qs[p]), base_of(ds[q]))];
      const double le = loop_energy_ungapped(sc, type, type2, i, j, p, q, qs, ds);
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
      q = j;
    }
    if (min_p - i >= drop) break;
  }
  e = min_e;
  a = min_a;
  hy = min_h;
  int k = q_sp0 + len0 - 1, r = k, l = db_sp0 + len0 - 1, s = l, min_r = r;
  for (;;) { // walk right (:96-145)
    k++;
    l++;
    id_start--;
    if (qs[k] < 2 || ds[l] < 2) break;
    const double ta = qcond[k] + dacc[id_start] - dacc[id_start + 1] + dcond[id_start + delta];
    e += ta;
    a += ta;
    const int type2 = sc.rtype[bp_type(sc, base_of(qs[k]), base_of(ds[l]))];
    if (type2 != 0) {
      const int type = bp_type(sc, base_of(qs[r]), base_of(ds[s]));
      const double le = loop_energy_ungapped(sc, type, type2, r, s, k, l, qs, ds);
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
      s = l;
    }
    if (k - min_r >= drop) break;
  }
  h.db_id_start[x] = min_id_start;
  h.q_sp[x] = min_p;
  h.db_sp[x] = min_q;
  h.q_len[x] = min_r - min_p + 1;
  h.db_len[x] = min_r - min_p + 1;
  h.e_tot[x] = min_e;
  h.e_acc[x] = min_a;
  h.e_hyb[x] = min_h;
}

// ------------------------------------------------------------------- sort keys / gather
#define US(x) ((int)(uint16_t)(x)) // Hit::GetQLength/GetDbLength are unsigned short (hit.hpp:62-64)

__global__ __launch_bounds__(kBlock) void k_make_keys(HitSoA h, int64_t n, uint64_t *k_energy, uint32_t *k_len,
                                                      uint32_t *k_qsp, uint64_t *k_pos, uint32_t *idx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // compare(): db_sp asc, q_sp asc, db_len desc, q_len desc (rna_interaction_search.cpp:45-55);
  // ties are then broken by energy asc and input order (DESIGN.md "total order")
  const uint64_t eb = (uint64_t)__double_as_longlong(h.e_tot[i]);
  k_energy[i] = (eb >> 63) ? ~eb : (eb | 0x8000000000000000ull);
  k_len[i] = ((uint32_t)(0xFFFF - US(h.db_len[i])) << 16) | (uint32_t)(0xFFFF - US(h.q_len[i]));
  k_qsp[i] = (uint32_t)h.q_sp[i];
  k_pos[i] = ((uint64_t)(uint32_t)h.query[i] << 32) | (uint32_t)h.db_sp[i];
  idx[i] = (uint32_t)i;
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

// ------------------------------------------------------------------------------ gapped
struct Rec { // one filled DP cell == one stem candidate (gapped_extension.hpp:34-60)
  double hyb;
  uint16_t i, j;
  uint16_t pred;  // record index of the predecessor cell
  uint8_t type;   // Stem::type  = rtype[type1]
  uint8_t ptype;  // Cell::type  = the PREDECESSOR's Stem::type (gapped_extension.cpp:256-258)
};
static_assert(sizeof(Rec) == 16, "Rec layout");

// GappedExtension::LoopEnergy, gapped_extension.cpp:426-473
__device__ __forceinline__ double loop_energy_gapped(const SearchConst &sc, int type, int type2, int i, int j, int p, int q,
                                                     const uint8_t *qs, const uint8_t *ds) {
  const int u1 = p - i - 1, u2 = q - j - 1;
  if (u1 == 0 && u2 == 0) return (double)sc.stack37[type * 7 + type2] / 100.0;
  if (u1 == 0 || u2 == 0) {
    const int u = u1 == 0 ? u2 : u1;
    double z = sc.bulge[u];
    if (u == 1) {
      z += sc.stack37[type * 7 + type2];
    } else {
      if (type > 2) z += sc.terminal_au;
      if (type2 > 2) z += sc.terminal_au;
    }
    return z / 100.0;
  }
  const int a = base_of(qs[i + 1]), b = base_of(ds[j + 1]), c = base_of(qs[p - 1]), d = base_of(ds[q - 1]);
  int z;
  if (u1 + u2 == 2) z = sc.int11[((type * 8 + type2) * 5 + a) * 5 + b];
  else if (u1 == 1 && u2 == 2) z = sc.int21[(((type * 8 + type2) * 5 + a) * 5 + d) * 5 + b];
  else if (u1 == 2 && u2 == 1) z = sc.int21[(((type2 * 8 + type) * 5 + d) * 5 + a) * 5 + c];
  else if (u1 == 2 && u2 == 2) z = sc.int22[((((type * 8 + type2) * 5 + a) * 5 + c) * 5 + d) * 5 + b];
  else z = sc.internal37[u1 + u2] + sc.mismatchI37[(type * 5 + a) * 5 + b] + sc.mismatchI37[(type2 * 5 + d) * 5 + c];
  return (double)z / 100.0;
}

// GetBPType, gapped_extension.cpp:321-338
__device__ __forceinline__ int ext_bp_type(const SearchConst &sc, int flag, const uint8_t *qs, const uint8_t *ds, int q_start,
                                           int64_t db_start, int i, int j, int x) {
  int qc, dc;
  if (flag == 0) {
    qc = get_char(qs, (int64_t)q_start - i - x);
    dc = get_char(ds, db_start - j - x);
  } else {
    qc = get_char(qs, (int64_t)q_start + i + x);
    dc = get_char(ds, db_start + j + x);
  }
  int t = bp_type(sc, qc, dc);
  if (flag == 1) t = sc.rtype[t];
  return t;
}
__device__ __forceinline__ bool wobble(int t) { return t == 3 || t == 4; }

// CalcDangleEnergy, gapped_extension.cpp:366-399
__device__ __forceinline__ double dangle_energy_gapped(const SearchConst &sc, int q_pos, int64_t db_pos, int flag,
                                                       const uint8_t *qs, int qn, const uint8_t *ds, int64_t dn) {
  const int qc = get_char(qs, q_pos), dc = get_char(ds, db_pos);
  const int type = flag == 0 ? bp_type(sc, qc, dc) : bp_type(sc, dc, qc);
  const int q_length = qn - 1;
  int x = 0;
  if (type != 0) {
    if (flag == 0) {
      if (q_pos > 0) x += sc.dangle5[type * 5 + get_char(qs, q_pos - 1)];
      if (db_pos > 0 && ds[db_pos - 1] != 0) x += sc.dangle3[type * 5 + get_char(ds, db_pos - 1)];
      if ((db_pos == 0 || ds[db_pos - 1] == 0) && type > 2) x += sc.terminal_au;
    } else {
      if (db_pos < dn - 1 && ds[db_pos + 1] != 0) x += sc.dangle5[type * 5 + get_char(ds, db_pos + 1)];
      if (q_pos < q_length - 1) x += sc.dangle3[type * 5 + get_char(qs, q_pos + 1)];
      if ((db_pos == dn - 1 || ds[db_pos + 1] == 0) && type > 2) x += sc.terminal_au;
    }
  }
  return (double)x / 100.0;
}

struct HitState {
  int q_sp, db_sp, q_len, db_len, id_start;
  double e_tot, e_acc, e_hyb;
};

struct ExtResult {
  bool overflow;
  int best;  // record index of the arg-min cell, 0 if the extension found nothing
  int nrec;
};

// GappedExtension::extension (gapped_extension.cpp:71-319) for one direction.  The
// reference's growing 100x100 Cell matrix is replaced by the append-only list of filled
// cells (`rec`), which is at the same time the stem-candidate list: candidates are pushed in
// (diagonal, i) order and pruned from the front (gapped_extension.cpp:213-217), so the live
// candidates are always the contiguous range [lo, diagonal start).
__device__ ExtResult extend_dir(const SearchConst &sc, const ExtOpts &o, HitState &h, int flag, const uint8_t *qs,
                                const float *qacc, const float *qcond, const uint8_t *ds, const float *dacc,
                                const float *dcond, double *eq, double *ed, Rec *rec, int cap_rec, int cap_diag) {
  const int MAXE = 100000;
  const int delta = o.delta, drop = o.drop_w_gap, min_helix = o.min_helix;
  double min_e = h.e_tot;
  const double first_a = h.e_acc;
  double min_a = first_a;
  int q_start;
  int64_t db_start;
  if (flag == 0) {
    q_start = h.q_sp;
    db_start = h.db_sp;
  } else {
    q_start = h.q_sp + h.q_len - 1;
    db_start = (int64_t)h.db_sp + h.db_len - 1;
  }
  int max_q = MAXE, max_d = MAXE;
  const int id_start = h.id_start, id_end = id_start + h.db_len - 1;
  int min_q_start = q_start;
  int64_t min_db_start = db_start;
  const int q_length = h.q_len, db_length = h.db_len;
  int min_q_len = q_length, min_db_len = db_length, min_id_start = id_start;
  int length = 0, min_length = 0;
  ExtResult res{false, 0, 0};

  int type0 = bp_type(sc, get_char(qs, q_start), get_char(ds, db_start));
  if (flag == 0) type0 = sc.rtype[type0];
  rec[0].hyb = min_e;
  rec[0].i = 0;
  rec[0].j = 0;
  rec[0].pred = 0;
  rec[0].type = (uint8_t)type0;
  rec[0].ptype = (uint8_t)type0;
  int nrec = 1;
  int lo = 0;            // first live candidate
  int d2 = 0, d1 = 0;    // [d2, d1) = records of diagonal length-2 at the top of each iteration
  int best = 0;

  for (;;) {
    length++;
    if (length > cap_diag) {
      res.overflow = true;
      break;
    }
    if (flag == 0) {
      if (max_q == MAXE && (q_start - length < 0 || qs[q_start - length] < 2)) max_q = length - 1;
      if (max_d == MAXE && (db_start - length < 0 || ds[db_start - length] < 2)) max_d = length - 1;
    } else {
      if (max_q == MAXE && qs[q_start + length] < 2) max_q = length - 1;
      if (max_d == MAXE && ds[db_start + length] < 2) max_d = length - 1;
    }
    // cumulative accessibility change of the extension (:156-212)
    if (max_q == MAXE) {
      double v;
      if (flag == 0) {
        const int t = q_start - length;
        if (length == 1) v = qacc[t] - qacc[t + 1] + qcond[t + delta];
        else v = eq[length - 2] + qacc[t] - qacc[t + 1] + qcond[t + delta];
      } else {
        if (length == 1) v = qcond[q_start + length];
        else v = eq[length - 2] + qcond[q_start + length];
      }
      eq[length - 1] = v;
    }
    if (max_d == MAXE) {
      double v;
      if (flag == 0) {
        if (length == 1) v = dcond[id_end + length];
        else v = ed[length - 2] + dcond[id_end + length];
      } else {
        const int t = id_start - length;
        if (length == 1) v = dacc[t] - dacc[t + 1] + dcond[t + delta];
        else v = ed[length - 2] + dacc[t] - dacc[t + 1] + dcond[t + delta];
      }
      ed[length - 1] = v;
    }
    // prune candidates with length - first - second - 2 > drop (:213-217)
    if (length - 2 > drop)
      while (lo < nrec && length - (int)rec[lo].i - (int)rec[lo].j - 2 > drop) lo++;
    const int dstart = nrec; // this diagonal's cells are appended from here
    int pp = d2;             // cursor into diagonal length-2 for the (i-1, j-1) lookups
    const int i_lo = length - max_d > 1 ? length - max_d : 1;
    const int i_hi = max_q < length - 1 ? max_q : length - 1;
    for (int i = i_lo; i <= i_hi; i++) {
      const int j = length - i;
      // CheckHelixLength (:342-364)
      int type1 = ext_bp_type(sc, flag, qs, ds, q_start, db_start, i, j, 0);
      if (type1 != 0) {
        while (pp < d1 && (int)rec[pp].i < i - 1) pp++;
        int pt = 0;
        if (pp < d1 && (int)rec[pp].i == i - 1) pt = rec[pp].ptype; // j-1 matches: same diagonal
        if (pt == 0 || (wobble(type1) && wobble(pt))) {
          for (int x = 1; x <= min_helix - 1; x++) {
            const int t = ext_bp_type(sc, flag, qs, ds, q_start, db_start, i, j, x);
            if (t == 0 || (x == 1 && wobble(type1) && wobble(t))) {
              type1 = 0;
              break;
            }
          }
        }
      }
      if (type1 == 0) continue;
      int min_k = lo;
      double hyb = 1000000.0; // INF
      for (int k = lo; k < dstart; k++) {
        const Rec c = rec[k];
        if ((int)c.i < i && (int)c.j < j) {
          double te;
          if (flag == 0)
            te = loop_energy_gapped(sc, type1, c.type, q_start - i, (int)(db_start - j), q_start - c.i,
                                    (int)(db_start - c.j), qs, ds);
          else
            te = loop_energy_gapped(sc, c.type, type1, q_start + c.i, (int)(db_start + c.j), q_start + i,
                                    (int)(db_start + j), qs, ds);
          te += c.hyb;
          if (te < hyb) {
            hyb = te;
            min_k = k;
          }
        }
      }
      if (nrec >= cap_rec) {
        res.overflow = true;
        break;
      }
      Rec r;
      r.hyb = hyb;
      r.i = (uint16_t)i;
      r.j = (uint16_t)j;
      if (lo >= dstart) min_k = 0; // empty window: the reference reads stem_candidate[0] of an empty list here
      r.pred = (uint16_t)min_k;
      r.type = sc.rtype[type1];
      r.ptype = rec[min_k].type;
      rec[nrec] = r;
      const double ie = eq[i - 1] + ed[j - 1] + hyb;
      if (ie < min_e) {
        min_e = ie;
        min_a = first_a + eq[i - 1] + ed[j - 1];
        min_length = length;
        best = nrec;
        if (flag == 0) {
          min_q_start = q_start - i;
          min_db_start = db_start - j;
        } else {
          min_id_start = id_start - j;
        }
        min_q_len = q_length + i;
        min_db_len = db_length + j;
      }
      nrec++;
    }
    if (res.overflow) break;
    d2 = d1;
    d1 = dstart;
    if (length - min_length >= drop) break;
    if (max_q != MAXE && max_d != MAXE) break;
  }
  res.best = (q_length - min_q_len != 0 && db_length - min_db_len != 0) ? best : 0;
  res.nrec = nrec;
  h.id_start = min_id_start;
  if (flag == 0) {
    h.q_sp = min_q_start;
    h.db_sp = (int)min_db_start;
  }
  h.q_len = min_q_len;
  h.db_len = min_db_len;
  h.e_tot = min_e;
  h.e_acc = min_a;
  h.e_hyb = min_e - min_a;
  return res;
}

__global__ __launch_bounds__(kBlock) void k_gapped(HitSoA in, HitSoA out, int64_t n, const uint32_t *__restrict__ subset,
                                                   QBatchDev qb, PageDev pg, SearchConst sc, ExtOpts o, GapScratch scratch,
                                                   int mode, uint8_t *overflow, const uint8_t *__restrict__ first_flag,
                                                   int32_t *bp_count, const int64_t *__restrict__ bp_off, int32_t *bp_out) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint8_t *mine = scratch.base + (size_t)tid * scratch.bytes_per_thread;
  double *eq = reinterpret_cast<double *>(mine);
  double *ed = eq + scratch.cap_diag;
  Rec *rec = reinterpret_cast<Rec *>(ed + scratch.cap_diag);
  for (int64_t w = tid; w < n; w += scratch.nthreads) {
    const int64_t x = subset ? (int64_t)subset[w] : w;
    const int query = in.query[x];
    const int64_t qo = qb.off[query];
    const uint8_t *qs = qb.enc + qo;
    const int qn = qb.len[query] + 1;
    const float *qacc = qb.acc + qo, *qcond = qb.cond + qo;
    const uint8_t *ds = pg.seqs;
    const int id = in.db_id[x];
    const int64_t base = (int64_t)pg.start_pos[id] - id;
    const float *dacc = pg.acc + base, *dcond = pg.cond + base;
    HitState h;
    h.q_sp = in.q_sp[x];
    h.db_sp = in.db_sp[x];
    h.q_len = in.q_len[x];
    h.db_len = in.db_len[x];
    h.id_start = in.db_id_start[x];
    h.e_tot = in.e_tot[x];
    h.e_acc = in.e_acc[x];
    h.e_hyb = in.e_hyb[x];
    const int diag_q = h.q_sp, diag_d = h.db_sp, diag_len = US(h.q_len); // ungapped region

    int ndiag = 0;
    if (mode != 0) // GetBasePair, rna_interaction_search.cpp:371-385
      for (int t = 0; t < diag_len; t++)
        ndiag += sc.bp_pair[(qs[diag_q + t] - 1) * 5 + (ds[diag_d + t] - 1)] != 0;
    const bool unsorted = mode != 0 && first_flag && first_flag[x]; // hit 0 keeps raw order (:314-317)
    const int64_t out0 = mode == 2 ? bp_off[w] : 0;

    bool ovf = false;
    int nleft = 0, nright = 0;
    for (int flag = 0; flag < 2 && !ovf; flag++) {
      const int q_start = flag == 0 ? h.q_sp : h.q_sp + h.q_len - 1;
      const int64_t db_start = flag == 0 ? (int64_t)h.db_sp : (int64_t)h.db_sp + h.db_len - 1;
      const ExtResult r = extend_dir(sc, o, h, flag, qs, qacc, qcond, ds, dacc, dcond, eq, ed, rec, scratch.cap_rec,
                                     scratch.cap_diag);
      ovf = r.overflow;
      if (mode != 0 && !ovf) {
        // traceback (:300-308, :409-424): from the arg-min cell through the predecessors
        int cnt = 0;
        for (int k = r.best; k != 0 && rec[k].i != 0 && rec[k].j != 0; k = rec[k].pred) cnt++;
        if (flag == 0) nleft = cnt;
        else nright = cnt;
        if (mode == 2) {
          int t = 0;
          for (int k = r.best; k != 0 && rec[k].i != 0 && rec[k].j != 0; k = rec[k].pred, t++) {
            int64_t pos;
            int qv, dv;
            if (flag == 0) { // emitted outermost first = ascending positions
              qv = q_start - rec[k].i;
              dv = (int)(db_start - rec[k].j);
              pos = unsorted ? out0 + ndiag + t : out0 + t;
            } else { // emitted outermost first = descending positions
              qv = q_start + rec[k].i;
              dv = (int)(db_start + rec[k].j);
              pos = unsorted ? out0 + ndiag + nleft + t : out0 + nleft + ndiag + (cnt - 1 - t);
            }
            bp_out[2 * pos] = qv;
            bp_out[2 * pos + 1] = dv;
          }
        }
      }
    }
    if (mode == 0) {
      if (overflow) overflow[w] = ovf ? 1 : 0;
      if (!ovf) {
        // GappedExtension::Run tail (gapped_extension.cpp:49-67): dangling ends on both sides
        const double d0 = dangle_energy_gapped(sc, h.q_sp, h.db_sp, 0, qs, qn, ds, pg.nchars);
        const double d1 = dangle_energy_gapped(sc, h.q_sp + US(h.q_len) - 1, (int64_t)h.db_sp + US(h.db_len) - 1, 1, qs,
                                               qn, ds, pg.nchars);
        double e = h.e_tot, hy = h.e_hyb;
        e += d0;
        e += d1;
        hy += d0;
        hy += d1;
        out.q_sp[x] = h.q_sp;
        out.db_sp[x] = h.db_sp;
        out.q_len[x] = h.q_len;
        out.db_len[x] = h.db_len;
        out.db_id[x] = id;
        out.db_id_start[x] = h.id_start;
        out.query[x] = query;
        out.e_acc[x] = h.e_acc;
        out.e_hyb[x] = hy;
        out.e_tot[x] = e;
      }
    } else if (mode == 1) {
      bp_count[w] = ovf ? -1 : ndiag + nleft + nright;
    } else if (!ovf) {
      int t = 0;
      const int64_t d0 = unsorted ? out0 : out0 + nleft;
      for (int u = 0; u < diag_len; u++)
        if (sc.bp_pair[(qs[diag_q + u] - 1) * 5 + (ds[diag_d + u] - 1)] != 0) {
          bp_out[2 * (d0 + t)] = diag_q + u;
          bp_out[2 * (d0 + t) + 1] = diag_d + u;
          t++;
        }
    }
  }
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

} // namespace

hipError_t launch_seed_count(const CandDev *cands, int32_t ncand, int64_t nrows, const QBatchDev &qb, const PageDev &pg,
                             int delta, int32_t *row_count, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_seed<false>, grid_for(nrows), dim3(kBlock), 0, s, cands, ncand, nrows, qb, pg, delta, row_count,
                     (const int64_t *)nullptr, HitSoA{});
  return hipGetLastError();
}
hipError_t launch_seed_emit(const CandDev *cands, int32_t ncand, int64_t nrows, const QBatchDev &qb, const PageDev &pg,
                            int delta, const int64_t *row_off, HitSoA hits, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_seed<true>, grid_for(nrows), dim3(kBlock), 0, s, cands, ncand, nrows, qb, pg, delta,
                     (int32_t *)nullptr, row_off, hits);
  return hipGetLastError();
}
hipError_t launch_ungapped(HitSoA hits, int64_t n, const QBatchDev &qb, const PageDev &pg, const SearchConst &sc, ExtOpts o,
                           hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_ungapped, grid_for(n), dim3(kBlock), 0, s, hits, n, qb, pg, sc, o);
  return hipGetLastError();
}
hipError_t launch_make_keys(const HitSoA &hits, int64_t n, uint64_t *k_energy, uint32_t *k_len, uint32_t *k_qsp,
                            uint64_t *k_pos, uint32_t *idx, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_make_keys, grid_for(n), dim3(kBlock), 0, s, hits, n, k_energy, k_len, k_qsp, k_pos, idx);
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
hipError_t launch_gather_hits(const HitSoA &src, const uint32_t *idx, HitSoA dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_hits, grid_for(n), dim3(kBlock), 0, s, src, idx, dst, n);
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
  hipLaunchKernelGGL(k_filter_round, grid_for(n), dim3(kBlock), 0, s, h, n, pmax, state, pending);
  return hipGetLastError();
}
hipError_t launch_filter_final(const HitSoA &h, int64_t n, const int64_t *pmax, const uint8_t *state, uint8_t *keep,
                               hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_filter_final, grid_for(n), dim3(kBlock), 0, s, h, n, pmax, state, keep);
  return hipGetLastError();
}
hipError_t launch_gapped(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb,
                         const PageDev &pg, const SearchConst &sc, ExtOpts o, GapScratch scratch, int mode,
                         uint8_t *overflow, const uint8_t *first_flag, int32_t *bp_count, const int64_t *bp_off,
                         int32_t *bp_out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int blocks = scratch.nthreads / kBlock;
  hipLaunchKernelGGL(k_gapped, dim3(blocks), dim3(kBlock), 0, s, in, out, n, subset, qb, pg, sc, o, scratch, mode, overflow,
                     first_flag, bp_count, bp_off, bp_out);
  return hipGetLastError();
}

} // namespace prb
