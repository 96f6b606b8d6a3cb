/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, strict IEEE: built with -ffp-contract=off) of the
 * pRIblast `ris` hot path.  It exists to CHECK the HIP path; nothing in the product
 * (priblast_amd/, include/) may include, link or execute it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Every function cites the reference file:line it follows.  Pinned against the
 * compiled, unmodified reference (oracle/_ref, built by oracle/Makefile) through the
 * golden vectors in tests/golden/ (see tests/test_oracle_*.py).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INF 1000000          /* energy_par.hpp:8 */
#define ORC_NINF (-1000000.0)    /* exact "-INF" sentinel of raccess.cpp */
#define ORC_TURN 3               /* energy_par.hpp:9 */
#define ORC_MAXLOOP 30           /* energy_par.hpp:10 */

/* ---- fmath restatement (fmath.hpp:148-216 tables, :439-479 expd, :738-752 log) ---- */
void orc_fmath_init(void);
double orc_expd(double x);
float orc_logf(float x);
const uint64_t *orc_expd_table(void); /* 2048 mantissa words */
const float *orc_log_table(void);     /* 2048 x {app, rev} interleaved */
void orc_fmath_consts(double *a, double *ra, float *c_log2);

/* ---- energy parameters (energy_par.hpp, intloops.hpp as data file) ---- */
typedef struct {
  int bp_pair[5][5];
  int rtype[7];
  int hairpin37[31], bulge37[31], internal37[31];
  int stack37[7][7];
  int mismatchH37[7][5][5], mismatchI37[7][5][5];
  int dangle5_37[8][5], dangle3_37[8][5];
  int int11_37[8][8][5][5];
  int int21_37[8][8][5][5][5];
  int int22_37[8][8][5][5][5][5];
  int terminal_au, ml_closing, ml_intern, ml_base, max_ninio, f_ninio;
  double lxc37, kT;
} orc_params;

/* returns 0 on success; path = priblast_amd/params/rna_andronescu2007.par */
int orc_params_load(const char *path);
const orc_params *orc_params_get(void);

/* ---- Raccess (raccess.cpp:42-50 in-memory overload) ---- */
/* optional debug capture of the DP tables, each (L+1)*(W+2) doubles, row-major [i][d] */
typedef struct {
  double *alpha_outer, *beta_outer; /* L+1 each */
  double *alpha[6];                 /* stem, stemend, multi, multibif, multi1, multi2 */
  double *beta[6];                  /* same order */
} orc_raccess_dbg;

/* acc and cond must hold L floats each.  Returns 0, or <0 on error. */
int orc_raccess(const char *seq, int L, int W, int delta, float *acc, float *cond,
                orc_raccess_dbg *dbg);

/* ---- encoder + suffix array (encoder.hpp:36-79, encoder.cpp:27-44, sais.cpp:656) ---- */
void orc_encode_query(const char *seq, int L, int repeat_flag, uint8_t *out /* L+1 */);
int orc_suffix_array(const uint8_t *T, int32_t *SA, int n);

/* ---- database pages (db_reader.cpp:61-177) ---- */
typedef struct {
  int nseq;
  int nchars;
  int32_t *seq_length;     /* nseq */
  int32_t *start_pos;      /* nseq */
  int32_t *seq_length_rep; /* nseq + 1 */
  uint8_t *seqs;           /* nchars: reversed sequences, 0-terminated each */
  int32_t *sa;             /* nchars */
  int32_t **start_hash;    /* hash_size levels, 4^(i+1) ints */
  int32_t **end_hash;
  float **acc;             /* per sequence, acc_len[i] floats */
  float **cond;            /* per sequence, cond_len[i] floats */
  int32_t *acc_len, *cond_len;
  char **names;
} orc_page;

typedef struct {
  int hash_size, repeat_flag, maximal_span, min_accessible_length; /* .bas */
  int npages;
  orc_page *pages;
} orc_db;

orc_db *orc_db_open(const char *prefix);
void orc_db_close(orc_db *db);

/* ---- hits (hit.hpp:31-118) ---- */
typedef struct {
  int32_t q_sp, db_sp, q_len, db_len, db_id, db_id_start;
  double e_acc, e_hyb, e_tot;
  int32_t flag;
  int32_t nbp, bp_cap;
  int32_t *bp; /* pairs (q, db) */
} orc_hit;

typedef struct {
  size_t n, cap;
  orc_hit *h;
} orc_hits;

void orc_hits_init(orc_hits *v);
void orc_hits_free(orc_hits *v);

typedef struct {
  int max_seed_length;  /* -l 20 */
  double hybrid_thr;    /* -e -6 */
  double interaction_thr; /* -f -4 */
  double final_thr;     /* -g -8 */
  int drop_wo_gap;      /* -y 5 */
  int drop_w_gap;       /* -x 16 */
  int min_helix;        /* -m 3 */
  int output_style;     /* -s 0 */
} orc_ris_opts;

void orc_ris_opts_default(orc_ris_opts *o);

/* stage functions (rna_interaction_search.cpp:264-320) */
void orc_seed_search(const orc_db *db, int page, const orc_ris_opts *o, const uint8_t *qenc,
                     int qn /* L+1 */, const int32_t *qsa, const float *qacc,
                     const float *qcond, orc_hits *out);
void orc_extend_ungapped(const orc_db *db, int page, const orc_ris_opts *o, const uint8_t *qenc,
                         int qn, const float *qacc, const float *qcond, orc_hits *hits);
void orc_extend_gapped(const orc_db *db, int page, const orc_ris_opts *o, const uint8_t *qenc,
                       int qn, const float *qacc, const float *qcond, orc_hits *hits);

/* whole `ris` (rna_interaction_search.cpp:61-92), single thread unless nthreads>1
 * (queries are independent; output order = query order, page order, hit order).
 * Returns the number of hits written, <0 on error. */
long orc_ris(const char *fasta, const char *dbprefix, const char *outpath, const orc_ris_opts *o,
             int nthreads);

/* FASTA (fastafile_reader.cpp:373-409) */
typedef struct {
  int n;
  char **names;
  char **seqs;
  int *lens;
} orc_fasta;
orc_fasta *orc_fasta_read(const char *path);
void orc_fasta_free(orc_fasta *f);

#ifdef __cplusplus
}
#endif
#endif
