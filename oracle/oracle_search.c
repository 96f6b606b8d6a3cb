/* oracle/oracle_search.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar restatement of the interaction-search stages of the reference's `ris`:
 *   SeedSearch::Run / SeedSearchCore / SeedSearchNextCharacter  seed_search.cpp:30-45,153-295
 *   SeedSearch::CalcInteractionEnergy / GetSeqIdAndStart        seed_search.cpp:47-151
 *   UngappedExtension::Run / LoopEnergy                          ungapped_extension.cpp:30-186
 *   compare / CheckRedundancy / GetBasePair                      rna_interaction_search.cpp:45-55,371-424
 *   GappedExtension::Run / extension / CheckHelixLength /
 *     CalcDangleEnergy / traceback / LoopEnergy                  gapped_extension.cpp:33-473
 *   ExtendWithGap post-processing (SortBasePair from hit 1!)     rna_interaction_search.cpp:302-320
 *   SaveMyResults / MergeOutput formatting                       rna_interaction_search.cpp:322-369,445-476
 *
 * Known deliberate deviations (documented in DESIGN.md): the hit sort is made a total
 * order (ties of the reference's 4-key comparator are broken by energy, its two parts, then
 * input order) - the reference's std::sort leaves ties unspecified; output lines are written
 * in (query, page, hit) order with a running Id (the reference's order is
 * thread-finish order).  Tests compare sorted bodies with the Id column stripped.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define P (orc_params_get())

void orc_ris_opts_default(orc_ris_opts *o) {
  /* rna_interaction_search_parameters.hpp:54-62 */
  o->max_seed_length = 20;
  o->hybrid_thr = -6.0;
  o->interaction_thr = -4;
  o->final_thr = -8.0;
  o->drop_wo_gap = 5;
  o->drop_w_gap = 16;
  o->min_helix = 3;
  o->output_style = 0;
}

void orc_hits_init(orc_hits *v) { memset(v, 0, sizeof *v); }
void orc_hits_free(orc_hits *v) {
  for (size_t i = 0; i < v->n; i++) free(v->h[i].bp);
  free(v->h);
  memset(v, 0, sizeof *v);
}
static orc_hit *hits_push(orc_hits *v) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 1024;
    v->h = realloc(v->h, v->cap * sizeof(orc_hit));
  }
  orc_hit *h = &v->h[v->n++];
  memset(h, 0, sizeof *h);
  return h;
}
static void add_bp(orc_hit *h, int q, int d) {
  if (h->nbp == h->bp_cap) {
    h->bp_cap = h->bp_cap ? h->bp_cap * 2 : 16;
    h->bp = realloc(h->bp, (size_t)h->bp_cap * 2 * sizeof(int32_t));
  }
  h->bp[2 * h->nbp] = q;
  h->bp[2 * h->nbp + 1] = d;
  h->nbp++;
}

/* code (2..5 plain, 6..9 soft-masked) -> base 1..4 ; ungapped_extension.cpp:69-70 */
static inline int base_of(uint8_t c) { return c <= 5 ? c - 1 : c - 5; }
/* GappedExtension::GetChar, gapped_extension.cpp:401-407 */
static inline int get_char(const uint8_t *s, int i) {
  if (i < 0 || s[i] < 2) return 0;
  return base_of(s[i]);
}

/* ------------------------------------------------------------------ seed search */

typedef struct {
  int spq, epq, spdb, epdb, len;
  double e;
} cand_t;

typedef struct {
  const orc_db *db;
  const orc_page *pg;
  const orc_ris_opts *o;
  const uint8_t *q;
  int qn;
  const int32_t *qsa;
  cand_t *c;
  size_t nc, capc;
  int qseed[64], dseed[64];
} seed_ctx;

static const int STEM_PAIR[6][2] = {{3, 4}, {4, 3}, {4, 5}, {5, 4}, {2, 5}, {5, 2}}; /* seed_search.hpp:38-50 */

/* SeedSearchNextCharacter, seed_search.cpp:232-295: narrow [*s,*e] to the suffixes whose
 * character at `off` equals c (empty result is (1,0)). */
static void narrow(const uint8_t *T, const int32_t *SA, int *s, int *e, uint8_t c, int off) {
  int lo = *s, hi = *e;
  if (lo > hi) { *s = 1; *e = 0; return; }
  /* first position with char >= c */
  int a = lo, b = hi + 1;
  while (a < b) {
    int m = (a + b) / 2;
    if (T[SA[m] + off] < c) a = m + 1; else b = m;
  }
  int first = a;
  a = first; b = hi + 1;
  while (a < b) {
    int m = (a + b) / 2;
    if (T[SA[m] + off] <= c) a = m + 1; else b = m;
  }
  int last = a - 1;
  if (first > last) { *s = 1; *e = 0; return; }
  *s = first;
  *e = last;
}

/* SeedSearchCore, seed_search.cpp:153-230 */
static void seed_dfs(seed_ctx *x, int spq, int epq, int spdb, int epdb, double score, int length) {
  const orc_params *p = P;
  if (length >= x->o->max_seed_length) return;
  int rs[6][4];
  for (int i = 0; i < 6; i++) {
    int s = spq, e = epq;
    narrow(x->q, x->qsa, &s, &e, (uint8_t)STEM_PAIR[i][0], length);
    rs[i][0] = s; rs[i][1] = e;
    s = spdb; e = epdb;
    if (length + 1 > x->db->hash_size) {
      narrow(x->pg->seqs, x->pg->sa, &s, &e, (uint8_t)STEM_PAIR[i][1], length);
    } else {
      int t = STEM_PAIR[i][1] - 2;
      for (int j = 0; j < length; j++) t += (1 << (2 * (length - j))) * (x->dseed[j] - 2);
      s = x->pg->start_hash[length][t];
      e = x->pg->end_hash[length][t];
    }
    rs[i][2] = s; rs[i][3] = e;
  }
  for (int i = 0; i < 6; i++) {
    if (rs[i][0] <= rs[i][1] && rs[i][2] <= rs[i][3]) {
      double ts = 0.0;
      if (length > 0) {
        int type = p->bp_pair[x->qseed[length - 1] - 1][x->dseed[length - 1] - 1];
        int type2 = p->rtype[p->bp_pair[STEM_PAIR[i][0] - 1][STEM_PAIR[i][1] - 1]];
        ts = score + ((double)p->stack37[type][type2]) / 100;
      }
      if (ts < x->o->hybrid_thr && length + 1 >= x->db->min_accessible_length) {
        if (x->nc == x->capc) {
          x->capc = x->capc ? x->capc * 2 : 256;
          x->c = realloc(x->c, x->capc * sizeof(cand_t));
        }
        cand_t *c = &x->c[x->nc++];
        c->spq = rs[i][0]; c->epq = rs[i][1]; c->spdb = rs[i][2]; c->epdb = rs[i][3];
        c->len = length + 1; c->e = ts;
      } else {
        x->qseed[length] = STEM_PAIR[i][0];
        x->dseed[length] = STEM_PAIR[i][1];
        seed_dfs(x, rs[i][0], rs[i][1], rs[i][2], rs[i][3], ts, length + 1);
      }
    }
  }
}

/* SeedSearch::CalcAccessibility, seed_search.cpp:143-151 */
static double window_acc(const float *acc, const float *cond, int sp, int length, int delta) {
  double t = acc[sp];
  for (int i = delta; i < length; i++) t += cond[sp + i];
  return t;
}

/* GetSeqIdAndStart, seed_search.cpp:101-141: id with start_pos[id] <= sp < start_pos[id+1] */
static void seq_id_and_start(const orc_page *pg, int sp, int length, int *id, int *start) {
  int lo = 0, hi = pg->nseq - 1;
  while (lo < hi) {
    int m = (lo + hi + 1) / 2;
    if (pg->start_pos[m] <= sp) lo = m; else hi = m - 1;
  }
  *id = lo;
  *start = pg->seq_length[lo] - (sp - pg->start_pos[lo]) - length;
}

void orc_seed_search(const orc_db *db, int page, const orc_ris_opts *o, const uint8_t *qenc, int qn,
                     const int32_t *qsa, const float *qacc, const float *qcond, orc_hits *out) {
  const orc_page *pg = &db->pages[page];
  seed_ctx x;
  memset(&x, 0, sizeof x);
  x.db = db; x.pg = pg; x.o = o; x.q = qenc; x.qn = qn; x.qsa = qsa;
  seed_dfs(&x, 0, qn - 1, 0, pg->nchars - 1, 0.0, 0); /* seed_search.cpp:41-44 */
  const int delta = db->min_accessible_length;
  /* CalcInteractionEnergy, seed_search.cpp:47-99 */
  for (size_t ci = 0; ci < x.nc; ci++) {
    const cand_t *c = &x.c[ci];
    int nq = c->epq - c->spq + 1;
    double *qa = malloc(sizeof(double) * nq);
    for (int j = 0; j < nq; j++) qa[j] = window_acc(qacc, qcond, qsa[c->spq + j], c->len, delta);
    for (int k = c->spdb; k <= c->epdb; k++) {
      int db_sp = pg->sa[k], id, st;
      seq_id_and_start(pg, db_sp, c->len, &id, &st);
      double dba = window_acc(pg->acc[id], pg->cond[id], st, c->len, delta);
      for (int j = 0; j < nq; j++) {
        double ie = qa[j] + dba + c->e;
        if (ie < 0) {
          orc_hit *h = hits_push(out);
          h->q_sp = qsa[c->spq + j];
          h->db_sp = db_sp;
          h->q_len = h->db_len = c->len;
          h->e_acc = qa[j] + dba;
          h->e_hyb = c->e;
          h->e_tot = h->e_acc + h->e_hyb;
          h->db_id = id;
          h->db_id_start = st;
        }
      }
    }
    free(qa);
  }
  free(x.c);
}

/* ------------------------------------------------------------- ungapped extension */

/* UngappedExtension::LoopEnergy, ungapped_extension.cpp:157-186 (symmetric loops only) */
static double loop_energy_ungapped(int type, int type2, int i, int j, int p, int q, const uint8_t *qs,
                                   const uint8_t *ds) {
  const orc_params *pr = P;
  double z = 0;
  int u1 = p - i - 1, u2 = q - j - 1;
  if (u1 == 0 && u2 == 0) {
    z = pr->stack37[type][type2];
  } else {
    int a = base_of(qs[i + 1]), b = base_of(ds[j + 1]), c = base_of(qs[p - 1]), d = base_of(ds[q - 1]);
    if (u1 + u2 == 2) z = pr->int11_37[type][type2][a][b];
    else if (u1 == 1 && u2 == 2) z = pr->int21_37[type][type2][a][d][b];
    else if (u1 == 2 && u2 == 1) z = pr->int21_37[type2][type][d][a][c];
    else if (u1 == 2 && u2 == 2) z = pr->int22_37[type][type2][a][c][d][b];
    else z = pr->internal37[u1 + u2] + pr->mismatchI37[type][a][b] + pr->mismatchI37[type2][d][c];
  }
  return z / 100.0;
}

/* UngappedExtension::Run body for one hit, ungapped_extension.cpp:38-154 */
static void ungapped_one(orc_hit *h, const orc_page *pg, const uint8_t *qs, const float *qacc,
                         const float *qcond, int delta, int drop) {
  const orc_params *pr = P;
  const uint8_t *ds = pg->seqs;
  const float *dacc = pg->acc[h->db_id], *dcond = pg->cond[h->db_id];
  double min_e = h->e_tot, e = h->e_tot, min_a = h->e_acc, a = h->e_acc, min_h = h->e_hyb, hy = h->e_hyb;
  int i = h->q_sp, p = h->q_sp, j = h->db_sp, q = h->db_sp;
  int min_p = p, min_q = q;
  int id_start = h->db_id_start;
  int id_end = id_start + h->db_len - 1;
  int min_id_start = id_start;
  for (;;) {
    i--; j--; id_end++;
    if (i < 0 || j < 0 || qs[i] < 2 || ds[j] < 2) break;
    double ta = qacc[i] - qacc[i + 1] + qcond[i + delta] + dcond[id_end];
    e += ta;
    a += ta;
    int type = pr->bp_pair[base_of(qs[i])][base_of(ds[j])];
    if (type != 0) {
      int type2 = pr->rtype[pr->bp_pair[base_of(qs[p])][base_of(ds[q])]];
      double le = loop_energy_ungapped(type, type2, i, j, p, q, qs, ds);
      e += le;
      hy += le;
      if (e < min_e) { min_e = e; min_a = a; min_h = hy; min_p = i; min_q = j; }
      p = i; q = j;
    }
    if (min_p - i >= drop) break;
  }
  e = min_e; a = min_a; hy = min_h;
  int k = h->q_sp + h->q_len - 1, r = k, l = h->db_sp + h->q_len - 1, s = l, min_r = r;
  for (;;) {
    k++; l++; id_start--;
    if (qs[k] < 2 || ds[l] < 2) break;
    double ta = qcond[k] + dacc[id_start] - dacc[id_start + 1] + dcond[id_start + delta];
    e += ta;
    a += ta;
    int type2 = pr->rtype[pr->bp_pair[base_of(qs[k])][base_of(ds[l])]];
    if (type2 != 0) {
      int type = pr->bp_pair[base_of(qs[r])][base_of(ds[s])];
      double le = loop_energy_ungapped(type, type2, r, s, k, l, qs, ds);
      e += le;
      hy += le;
      if (e < min_e) { min_e = e; min_a = a; min_h = hy; min_r = k; min_id_start = id_start; }
      r = k; s = l;
    }
    if (k - min_r >= drop) break;
  }
  h->db_id_start = min_id_start;
  h->q_sp = min_p;
  h->db_sp = min_q;
  h->q_len = h->db_len = min_r - min_p + 1;
  h->e_tot = min_e;
  h->e_acc = min_a;
  h->e_hyb = min_h;
}

#define US(x) ((int)(uint16_t)(x)) /* Hit::GetQLength/GetDbLength return unsigned short, hit.hpp:62-64 */

typedef struct { orc_hit h; size_t ord; } sort_rec;

/* compare, rna_interaction_search.cpp:45-55 + total-order tie-break (energy, its hybridization part, its accessibility
 * part, input order): hits that are still tied are identical in every field, so the order of the input is immaterial */
static int hit_cmp(const void *pa, const void *pb) {
  const sort_rec *a = pa, *b = pb;
  if (a->h.db_sp != b->h.db_sp) return a->h.db_sp < b->h.db_sp ? -1 : 1;
  if (a->h.q_sp != b->h.q_sp) return a->h.q_sp < b->h.q_sp ? -1 : 1;
  if (US(a->h.db_len) != US(b->h.db_len)) return US(a->h.db_len) > US(b->h.db_len) ? -1 : 1;
  if (US(a->h.q_len) != US(b->h.q_len)) return US(a->h.q_len) > US(b->h.q_len) ? -1 : 1;
  if (a->h.e_tot != b->h.e_tot) return a->h.e_tot < b->h.e_tot ? -1 : 1;
  if (a->h.e_hyb != b->h.e_hyb) return a->h.e_hyb < b->h.e_hyb ? -1 : 1;
  if (a->h.e_acc != b->h.e_acc) return a->h.e_acc < b->h.e_acc ? -1 : 1;
  return a->ord < b->ord ? -1 : (a->ord > b->ord ? 1 : 0);
}

static void sort_hits(orc_hits *v) {
  sort_rec *r = malloc(sizeof(sort_rec) * (v->n ? v->n : 1));
  for (size_t i = 0; i < v->n; i++) { r[i].h = v->h[i]; r[i].ord = i; }
  qsort(r, v->n, sizeof(sort_rec), hit_cmp);
  for (size_t i = 0; i < v->n; i++) v->h[i] = r[i].h;
  free(r);
}

/* CheckRedundancy, rna_interaction_search.cpp:387-424 */
static void check_redundancy(orc_hits *v, double thr) {
  for (size_t i = 0; i < v->n; i++) {
    orc_hit *a = &v->h[i];
    if (a->e_tot > thr) a->flag = 1;
    if (!a->flag) {
      int aQS = a->q_sp, aDS = a->db_sp;
      int aQE = aQS + US(a->q_len) - 1, aDE = aDS + US(a->db_len) - 1;
      for (size_t j = i + 1; j < v->n; j++) {
        orc_hit *b = &v->h[j];
        if (!b->flag) {
          int bDS = b->db_sp;
          if (aDE < bDS) break;
          int bQS = b->q_sp, bQE = bQS + US(b->q_len) - 1, bDE = bDS + US(b->db_len) - 1;
          if (aQE >= bQE && aQS <= bQS && aDE >= bDE) {
            if (a->e_tot > b->e_tot) a->flag = 1; else b->flag = 1;
          }
        }
      }
    }
  }
  size_t w = 0;
  for (size_t i = 0; i < v->n; i++) {
    if (v->h[i].flag) free(v->h[i].bp);
    else v->h[w++] = v->h[i];
  }
  v->n = w;
}

void orc_extend_ungapped(const orc_db *db, int page, const orc_ris_opts *o, const uint8_t *qenc, int qn,
                         const float *qacc, const float *qcond, orc_hits *hits) {
  const orc_page *pg = &db->pages[page];
  const orc_params *pr = P;
  (void)qn;
  for (size_t x = 0; x < hits->n; x++)
    ungapped_one(&hits->h[x], pg, qenc, qacc, qcond, db->min_accessible_length, o->drop_wo_gap);
  sort_hits(hits);
  check_redundancy(hits, o->interaction_thr);
  /* GetBasePair, rna_interaction_search.cpp:371-385.  The reference indexes BP_pair[5][5] with the
   * raw codes minus 1, i.e. outside the table for soft-masked codes 6..9 (`db -r 1`): undefined
   * behaviour, not restated - masked codes are mapped to their bases as in
   * ungapped_extension.cpp:68-69. */
  for (size_t x = 0; x < hits->n; x++) {
    orc_hit *h = &hits->h[x];
    int len = US(h->q_len);
    for (int j = 0; j < len; j++)
      if (pr->bp_pair[base_of(qenc[h->q_sp + j])][base_of(pg->seqs[h->db_sp + j])] != 0) add_bp(h, h->q_sp + j, h->db_sp + j);
  }
}

/* --------------------------------------------------------------- gapped extension */

typedef struct { int first, second, type; double hyb; } cell_t;
typedef struct { int first, second, type; } stem_t;

typedef struct {
  cell_t *m;
  int dim;
} matrix_t;

static const cell_t CELL0 = {-1, -1, 0, ORC_INF};

static cell_t *cell(matrix_t *M, int i, int j) {
  int need = (i > j ? i : j) + 1;
  if (need > M->dim) {
    int nd = M->dim ? M->dim : 128;
    while (nd < need) nd *= 2;
    cell_t *n = malloc(sizeof(cell_t) * nd * nd);
    for (int a = 0; a < nd * nd; a++) n[a] = CELL0;
    for (int a = 0; a < M->dim; a++) memcpy(n + (size_t)a * nd, M->m + (size_t)a * M->dim, sizeof(cell_t) * M->dim);
    free(M->m);
    M->m = n;
    M->dim = nd;
  }
  return &M->m[(size_t)i * M->dim + j];
}

/* GappedExtension::LoopEnergy, gapped_extension.cpp:426-473 */
static double loop_energy_gapped(int type, int type2, int i, int j, int p, int q, const uint8_t *qs,
                                 const uint8_t *ds) {
  const orc_params *pr = P;
  double z = 0;
  int u1 = p - i - 1, u2 = q - j - 1;
  if (u1 == 0 && u2 == 0) {
    z = pr->stack37[type][type2];
  } else if (u1 == 0 || u2 == 0) {
    int u = u1 == 0 ? u2 : u1;
    z = u <= 30 ? pr->bulge37[u] : pr->bulge37[30] + pr->lxc37 * log(u / 30.);
    if (u == 1) {
      z += pr->stack37[type][type2];
    } else {
      if (type > 2) z += pr->terminal_au;
      if (type2 > 2) z += pr->terminal_au;
    }
  } else {
    int a = base_of(qs[i + 1]), b = base_of(ds[j + 1]), c = base_of(qs[p - 1]), d = base_of(ds[q - 1]);
    if (u1 + u2 == 2) z = pr->int11_37[type][type2][a][b];
    else if (u1 == 1 && u2 == 2) z = pr->int21_37[type][type2][a][d][b];
    else if (u1 == 2 && u2 == 1) z = pr->int21_37[type2][type][d][a][c];
    else if (u1 == 2 && u2 == 2) z = pr->int22_37[type][type2][a][c][d][b];
    else z = pr->internal37[u1 + u2] + pr->mismatchI37[type][a][b] + pr->mismatchI37[type2][d][c];
  }
  return z / 100.0;
}

/* GetBPType, gapped_extension.cpp:321-338 */
static int bp_type(int flag, const uint8_t *qs, const uint8_t *ds, int q_start, int db_start, int i, int j, int x) {
  const orc_params *pr = P;
  int qc, dc;
  if (flag == 0) { qc = get_char(qs, q_start - i - x); dc = get_char(ds, db_start - j - x); }
  else { qc = get_char(qs, q_start + i + x); dc = get_char(ds, db_start + j + x); }
  int t = pr->bp_pair[qc][dc];
  if (flag == 1) t = pr->rtype[t];
  return t;
}
static inline int wobble(int t) { return t == 3 || t == 4; }

/* CheckHelixLength, gapped_extension.cpp:342-364 */
static int check_helix(int flag, const uint8_t *qs, const uint8_t *ds, int q_start, int db_start, int i, int j,
                       matrix_t *M, int min_helix) {
  int t0 = bp_type(flag, qs, ds, q_start, db_start, i, j, 0);
  if (t0 != 0) {
    int pt = cell(M, i - 1, j - 1)->type;
    if (pt == 0 || (wobble(t0) && wobble(pt))) {
      for (int x = 1; x <= min_helix - 1; x++) {
        int t = bp_type(flag, qs, ds, q_start, db_start, i, j, x);
        if (t == 0 || (x == 1 && wobble(t0) && wobble(t))) { t0 = 0; break; }
      }
    }
  }
  return t0;
}

/* GappedExtension::extension, gapped_extension.cpp:71-319 */
static void gapped_dir(orc_hit *h, const orc_page *pg, const uint8_t *qs, const float *qacc, const float *qcond,
                       int delta, int drop, int min_helix, int flag) {
  const orc_params *pr = P;
  const uint8_t *ds = pg->seqs;
  const float *dacc = pg->acc[h->db_id], *dcond = pg->cond[h->db_id];
  const int MAXE = 100000;
  double min_e = h->e_tot, first_a = h->e_acc, min_a = first_a;
  int q_start, db_start;
  if (flag == 0) { q_start = h->q_sp; db_start = h->db_sp; }
  else { q_start = h->q_sp + h->q_len - 1; db_start = h->db_sp + h->db_len - 1; }
  int max_q = MAXE, max_d = MAXE;
  int id_start = h->db_id_start, id_end = id_start + h->db_len - 1;
  int min_q_start = q_start, min_db_start = db_start;
  int q_length = h->q_len, db_length = h->db_len, min_q_len = q_length, min_db_len = db_length;
  int min_id_start = id_start;
  int length = 0, min_length = 0;
  matrix_t M = {0, 0};
  size_t eq_n = 0, ed_n = 0, eq_cap = 128, ed_cap = 128;
  double *eq = malloc(sizeof(double) * eq_cap), *ed = malloc(sizeof(double) * ed_cap);
  int type = pr->bp_pair[get_char(qs, q_start)][get_char(ds, db_start)];
  if (flag == 0) type = pr->rtype[type];
  *cell(&M, 0, 0) = (cell_t){-1, -1, type, min_e};
  size_t sc_n = 0, sc_cap = 128;
  stem_t *sc = malloc(sizeof(stem_t) * sc_cap);
  sc[sc_n++] = (stem_t){0, 0, type};

  for (;;) {
    length++;
    if (flag == 0) {
      if (max_q == MAXE && (q_start - length < 0 || qs[q_start - length] < 2)) max_q = length - 1;
      if (max_d == MAXE && (db_start - length < 0 || ds[db_start - length] < 2)) max_d = length - 1;
    } else {
      if (max_q == MAXE && qs[q_start + length] < 2) max_q = length - 1;
      if (max_d == MAXE && ds[db_start + length] < 2) max_d = length - 1;
    }
    /* cumulative accessibility of the extension, :156-212 */
    if (max_q == MAXE) {
      double v;
      if (flag == 0) {
        if (length == 1) v = qacc[q_start - length] - qacc[q_start - length + 1] + qcond[q_start - length + delta];
        else v = eq[length - 2] + qacc[q_start - length] - qacc[q_start - length + 1] + qcond[q_start - length + delta];
      } else {
        if (length == 1) v = qcond[q_start + length];
        else v = eq[length - 2] + qcond[q_start + length];
      }
      if (eq_n == eq_cap) eq = realloc(eq, sizeof(double) * (eq_cap *= 2));
      eq[eq_n++] = v;
    }
    if (max_d == MAXE) {
      double v;
      if (flag == 0) {
        if (length == 1) v = dcond[id_end + length];
        else v = ed[length - 2] + dcond[id_end + length];
      } else {
        if (length == 1) v = dacc[id_start - length] - dacc[id_start - length + 1] + dcond[id_start - length + delta];
        else v = ed[length - 2] + dacc[id_start - length] - dacc[id_start - length + 1] + dcond[id_start - length + delta];
      }
      if (ed_n == ed_cap) ed = realloc(ed, sizeof(double) * (ed_cap *= 2));
      ed[ed_n++] = v;
    }
    /* prune, :213-217 (stable) */
    if (length - 2 > drop) {
      size_t w = 0;
      for (size_t k = 0; k < sc_n; k++)
        if (!(length - sc[k].first - sc[k].second - 2 > drop)) sc[w++] = sc[k];
      sc_n = w;
    }
    for (int i = 1; i <= length - 1; i++) {
      int j = length - i;
      if (i <= max_q && j <= max_d) {
        int type1 = check_helix(flag, qs, ds, q_start, db_start, i, j, &M, min_helix);
        if (type1 != 0) {
          int min_k = 0;
          int scs = (int)sc_n;
          double hyb = ORC_INF;
          for (int k = 0; k < scs; k++) {
            stem_t c = sc[k];
            if (c.first < i && c.second < j) {
              double te;
              if (flag == 0)
                te = loop_energy_gapped(type1, c.type, q_start - i, db_start - j, q_start - c.first, db_start - c.second, qs, ds);
              else
                te = loop_energy_gapped(c.type, type1, q_start + c.first, db_start + c.second, q_start + i, db_start + j, qs, ds);
              te += cell(&M, c.first, c.second)->hyb;
              if (te < hyb) { hyb = te; min_k = k; }
            }
          }
          *cell(&M, i, j) = (cell_t){sc[min_k].first, sc[min_k].second, sc[min_k].type, hyb};
          double ie = eq[i - 1] + ed[j - 1] + hyb;
          if (sc_n == sc_cap) sc = realloc(sc, sizeof(stem_t) * (sc_cap *= 2));
          sc[sc_n++] = (stem_t){i, j, pr->rtype[type1]};
          if (ie < min_e) {
            min_e = ie;
            min_a = first_a + eq[i - 1] + ed[j - 1];
            min_length = length;
            if (flag == 0) { min_q_start = q_start - i; min_db_start = db_start - j; }
            else min_id_start = id_start - j;
            min_q_len = q_length + i;
            min_db_len = db_length + j;
          }
        }
      }
    }
    if (length - min_length >= drop) break;
    if (max_q != MAXE && max_d != MAXE) break;
  }

  /* traceback, :300-308 / :409-424 */
  if (q_length - min_q_len != 0 && db_length - min_db_len != 0) {
    int i, j;
    if (flag == 0) { i = q_start - min_q_start; j = db_start - min_db_start; }
    else { i = min_q_len - q_length; j = min_db_len - db_length; }
    while (i != 0 && j != 0) {
      if (flag == 0) add_bp(h, q_start - i, db_start - j); else add_bp(h, q_start + i, db_start + j);
      cell_t *c = cell(&M, i, j);
      i = c->first;
      j = c->second;
    }
  }
  h->db_id_start = min_id_start;
  if (flag == 0) { h->q_sp = min_q_start; h->db_sp = min_db_start; }
  h->q_len = min_q_len;
  h->db_len = min_db_len;
  h->e_tot = min_e;
  h->e_acc = min_a;
  h->e_hyb = min_e - min_a;
  free(M.m); free(eq); free(ed); free(sc);
}

/* CalcDangleEnergy, gapped_extension.cpp:366-399 */
static double dangle_energy_gapped(int q_pos, int db_pos, int flag, const uint8_t *qs, int qn, const uint8_t *ds, int dn) {
  const orc_params *pr = P;
  double x = 0;
  int qc = get_char(qs, q_pos), dc = get_char(ds, db_pos);
  int type = flag == 0 ? pr->bp_pair[qc][dc] : pr->bp_pair[dc][qc];
  int q_length = qn - 1;
  if (type != 0) {
    if (flag == 0) {
      if (q_pos > 0) x += pr->dangle5_37[type][get_char(qs, q_pos - 1)];
      if (db_pos > 0 && ds[db_pos - 1] != 0) x += pr->dangle3_37[type][get_char(ds, db_pos - 1)];
      if ((db_pos == 0 || ds[db_pos - 1] == 0) && type > 2) x += pr->terminal_au;
    } else {
      if (db_pos < dn - 1 && ds[db_pos + 1] != 0) x += pr->dangle5_37[type][get_char(ds, db_pos + 1)];
      if (q_pos < q_length - 1) x += pr->dangle3_37[type][get_char(qs, q_pos + 1)];
      if ((db_pos == dn - 1 || ds[db_pos + 1] == 0) && type > 2) x += pr->terminal_au;
    }
  }
  return x / 100.0;
}

static int bp_cmp(const void *a, const void *b) {
  int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

void orc_extend_gapped(const orc_db *db, int page, const orc_ris_opts *o, const uint8_t *qenc, int qn,
                       const float *qacc, const float *qcond, orc_hits *hits) {
  const orc_page *pg = &db->pages[page];
  const int delta = db->min_accessible_length;
  /* CheckHelixLength looks up to min_helix - 1 positions past a cell (GetBPType -> GetChar, gapped_extension.cpp:
   * 321-338, 401-407) without an upper bound: next to the query's end the reference reads one or two bytes behind
   * its vector (undefined; whatever the heap holds).  Defined here - and in the HIP path - as "no base": the query is
   * copied into a zero-padded buffer.  (The page text is padded the same way where it is loaded, oracle_io.c.) */
  uint8_t *qpad = calloc((size_t)qn + 64, 1);
  memcpy(qpad, qenc, (size_t)qn);
  qenc = qpad;
  /* GappedExtension::Run, gapped_extension.cpp:33-69 */
  for (size_t x = 0; x < hits->n; x++) {
    orc_hit *h = &hits->h[x];
    gapped_dir(h, pg, qenc, qacc, qcond, delta, o->drop_w_gap, o->min_helix, 0);
    gapped_dir(h, pg, qenc, qacc, qcond, delta, o->drop_w_gap, o->min_helix, 1);
    double e = h->e_tot, hy = h->e_hyb;
    double d0 = dangle_energy_gapped(h->q_sp, h->db_sp, 0, qenc, qn, pg->seqs, pg->nchars);
    double d1 = dangle_energy_gapped(h->q_sp + US(h->q_len) - 1, h->db_sp + US(h->db_len) - 1, 1, qenc, qn, pg->seqs, pg->nchars);
    e += d0;
    e += d1;
    hy += d0;
    hy += d1;
    h->e_tot = e;
    h->e_hyb = hy;
  }
  /* rna_interaction_search.cpp:314-319: SortBasePair for hits 1..n-1 only */
  for (size_t x = 1; x < hits->n; x++) qsort(hits->h[x].bp, hits->h[x].nbp, 2 * sizeof(int32_t), bp_cmp);
  sort_hits(hits);
  check_redundancy(hits, o->final_thr);
  free(qpad);
}

/* ------------------------------------------------------------------- whole `ris` */

typedef struct {
  char *buf;
  size_t n, cap;
  long nhits;
} sink_t;

static void sink_printf(sink_t *s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void sink_printf(sink_t *s, const char *fmt, ...) {
  for (;;) {
    va_list ap;
    va_start(ap, fmt);
    int k = vsnprintf(s->buf + s->n, s->cap - s->n, fmt, ap);
    va_end(ap);
    if ((size_t)k < s->cap - s->n) { s->n += k; return; }
    s->cap = s->cap ? s->cap * 2 : 4096;
    while (s->cap - s->n <= (size_t)k) s->cap *= 2;
    s->buf = realloc(s->buf, s->cap);
  }
}

/* SaveMyResults, rna_interaction_search.cpp:322-369 (one line per hit, without the Id column) */
static void format_hits(sink_t *s, const orc_page *pg, const orc_hits *hits, const char *qname, int qlen, int style) {
  for (size_t i = 0; i < hits->n; i++) {
    const orc_hit *h = &hits->h[i];
    int id = h->db_id;
    int sp = pg->start_pos[id], dl = pg->seq_length[id];
    sink_printf(s, "%s,%d,%s,%d,%g,%g,%g,", qname, qlen, pg->names[id], pg->seq_length_rep[id], h->e_acc, h->e_hyb, h->e_tot);
    if (style == 1) {
      for (int j = 0; j < h->nbp; j++) sink_printf(s, "(%d:%d) ", h->bp[2 * j], (dl - 1) - (h->bp[2 * j + 1] - sp));
    } else {
      int n = h->nbp;
      sink_printf(s, "(%d-%d:%d-%d) ", h->bp[0], h->bp[2 * (n - 1)], (dl - 1) - (h->bp[1] - sp), (dl - 1) - (h->bp[2 * (n - 1) + 1] - sp));
    }
    sink_printf(s, "\n");
    s->nhits++;
  }
}

long orc_ris(const char *fasta, const char *dbprefix, const char *outpath, const orc_ris_opts *o, int nthreads) {
  if (!P) return -1;
  orc_fmath_init();
  orc_db *db = orc_db_open(dbprefix);
  if (!db) return -2;
  orc_fasta *fa = orc_fasta_read(fasta);
  if (!fa) { orc_db_close(db); return -3; }
  sink_t *sinks = calloc(fa->n ? fa->n : 1, sizeof(sink_t));
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (int qi = 0; qi < fa->n; qi++) {
    int L = fa->lens[qi];
    float *acc = calloc((size_t)L + 1, sizeof(float)), *cond = calloc((size_t)L + 1, sizeof(float));
    uint8_t *enc = malloc((size_t)L + 1);
    int32_t *sa = malloc(sizeof(int32_t) * ((size_t)L + 1));
    orc_raccess(fa->seqs[qi], L, db->maximal_span, db->min_accessible_length, acc, cond, NULL);
    orc_encode_query(fa->seqs[qi], L, db->repeat_flag, enc);
    orc_suffix_array(enc, sa, L + 1);
    int qlen = 0;
    for (int j = 0; j <= L; j++) qlen += enc[j] >= 2 && enc[j] <= 5; /* rna_interaction_search.cpp:179-183 */
    for (int pgi = 0; pgi < db->npages; pgi++) {
      orc_hits hits;
      orc_hits_init(&hits);
      orc_seed_search(db, pgi, o, enc, L + 1, sa, acc, cond, &hits);
      orc_extend_ungapped(db, pgi, o, enc, L + 1, acc, cond, &hits);
      orc_extend_gapped(db, pgi, o, enc, L + 1, acc, cond, &hits);
      format_hits(&sinks[qi], &db->pages[pgi], &hits, fa->names[qi], qlen, o->output_style);
      orc_hits_free(&hits);
    }
    free(acc); free(cond); free(enc); free(sa);
  }
  long total = 0;
  FILE *f = outpath ? fopen(outpath, "w") : NULL;
  if (outpath && !f) return -4;
  if (f) {
    /* MergeOutput header, rna_interaction_search.cpp:445-463 */
    fprintf(f, "RIblast ris result\n");
    fprintf(f, "input:%s,database:%s,RepeatFlag:%d,MaximalSpan:%d,MinAccessibleLength:%d,MaxSeedLength:%d,"
               "InteractionEnergyThreshold:%g,HybridEnergyThreshold:%g,FinalThreshold:%g,DropOutLengthWoGap:%d,"
               "DropOutLengthWGap:%d\n",
            fasta, dbprefix, db->repeat_flag, db->maximal_span, db->min_accessible_length, o->max_seed_length,
            o->interaction_thr, o->hybrid_thr, o->final_thr, o->drop_wo_gap, o->drop_w_gap);
    fprintf(f, "Id,Query name, Query Length, Target name, Target Length, Accessibility Energy, Hybridization Energy, "
               "Interaction Energy, BasePair\n");
  }
  for (int qi = 0; qi < fa->n; qi++) {
    sink_t *s = &sinks[qi];
    if (f && s->n) {
      char *p = s->buf, *end = s->buf + s->n;
      while (p < end) {
        char *nl = memchr(p, '\n', end - p);
        fprintf(f, "%ld,", total++);
        fwrite(p, 1, nl - p + 1, f);
        p = nl + 1;
      }
    } else {
      total += s->nhits;
    }
    free(s->buf);
  }
  if (f) fclose(f);
  free(sinks);
  orc_fasta_free(fa);
  orc_db_close(db);
  return total;
}
