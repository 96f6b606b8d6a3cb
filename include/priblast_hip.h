/* include/priblast_hip.h -- C ABI of the MI355X-native `ris` hot path (libpriblast_hip.so).
 *
 * The reference (UDC-GAC/pRIblast) has no FFI / plugin interface; its only internal seam is
 * the five private per-query stage wrappers of RnaInteractionSearch
 * (rna_interaction_search.hpp:51-79, called from rna_interaction_search.cpp:171-197):
 *
 *   CalculateAccessibility  rna_interaction_search.cpp:242-250  -> prb_qbatch_accessibility
 *   ConstructSuffixArray    rna_interaction_search.cpp:252-262  -> prb_qbatch_create (host)
 *   SearchSeed              rna_interaction_search.cpp:264-283  -> prb_search_page (stage 1)
 *   ExtendWithoutGap        rna_interaction_search.cpp:285-300  -> prb_search_page (stage 2)
 *   ExtendWithGap           rna_interaction_search.cpp:302-320  -> prb_search_page (stage 3)
 *   DbReader::LoadDatabases db_reader.cpp:29-59                 -> prb_db_open
 *   Raccess::Run(seq, idx)  raccess.cpp:34-40 (db side)         -> prb_accessibility
 *
 * The entry points below are the batched form of that seam: plain pointers and sizes,
 * opaque handles, integer status returns (0 = ok, <0 = error; prb_last_error() gives the
 * text), no exceptions and no C++/torch types across the boundary.  All device work runs
 * on the handle's own HIP stream; the library never falls back to the CPU - if no HIP
 * device can be initialised prb_ctx_create fails.
 *
 * Layouts follow the reference: accessibility arrays are `float[L]` per sequence
 * (raccess.cpp:484-528), encoded queries are `uint8_t[L+1]` (encoder.cpp:38-44), suffix
 * arrays `int32_t[L+1]`, database files are read unchanged (.bas/.seq/.acc/.nam/.ind).
 */
#ifndef PRIBLAST_HIP_H
#define PRIBLAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRB_OK 0
#define PRB_ERR_ARG (-1)
#define PRB_ERR_IO (-2)
#define PRB_ERR_HIP (-3)
#define PRB_ERR_NOMEM (-4)
#define PRB_ERR_STATE (-5)

typedef struct prb_ctx prb_ctx;       /* one GPU: stream, parameter tables, workspaces.  A context is used by one
                                       * host thread at a time; several contexts may share a device (each has its
                                       * own stream), and query batches / databases made under one of them can be
                                       * used under another one of the same device - e.g. the accessibilities of the
                                       * next batch computed under a second context while the first one searches */
typedef struct prb_db prb_db;         /* database pages resident in HBM */
typedef struct prb_qbatch prb_qbatch; /* a batch of queries: enc + SA + accessibilities */
typedef struct prb_hitset prb_hitset; /* result of prb_search_page, host side */

/* `ris` options, defaults of rna_interaction_search_parameters.hpp:54-62 */
typedef struct prb_ris_opts {
  int32_t max_seed_length;    /* -l 20 */
  double hybrid_threshold;    /* -e -6.0 */
  double interaction_threshold; /* -f -4.0 */
  double final_threshold;     /* -g -8.0 */
  int32_t drop_out_wo_gap;    /* -y 5  */
  int32_t drop_out_w_gap;     /* -x 16 */
  int32_t min_helix_length;   /* -m 3  */
  int32_t output_style;       /* -s 0  */
} prb_ris_opts;

/* POD form of hit.hpp:38-118 (`Hit`) */
typedef struct prb_hit {
  int32_t q_sp, db_sp;          /* start in the query / in the page's reversed db text */
  int32_t q_len, db_len;
  int32_t db_id, db_id_start;   /* sequence index in the page, forward start in it */
  double e_acc, e_hyb, e_tot;   /* kcal/mol */
  int32_t query;                /* index in the batch */
  int32_t bp_count;             /* base pairs in prb_hitset_basepairs, at bp_offset */
  int64_t bp_offset;
} prb_hit;

const char *prb_last_error(void);
const char *prb_version(void);
/* CPUs the process may keep busy (hardware threads, affinity mask, cgroup CPU quota) and the size the library gives
 * each of its pools of host threads by default (suffix arrays + seed DFS; result lines): half of that, at most 32;
 * PRB_HOST_THREADS overrides.  The reference sizes its one pool with OMP_NUM_THREADS (main.cpp / `-a`). */
int prb_cpu_budget(void);
int prb_host_threads_default(void);
void prb_ris_opts_default(prb_ris_opts *o);

/* ---- context ---- */
/* param_file: nearest-neighbour parameter data (priblast_amd/params/rna_andronescu2007.par);
 * NULL = the file next to the library. */
int prb_ctx_create(int device, const char *param_file, prb_ctx **out);
void prb_ctx_destroy(prb_ctx *ctx);
int prb_ctx_synchronize(prb_ctx *ctx);
/* device time (ms, HIP events on the library's stream) of the named stage since the last
 * reset, and launch counts: "raccess", "seed" (keys + sort of the candidates' pairs or rows - what of it is not issued
 * ahead, beside the sub-batch before), "ungapped" (one-pass form: k_seed_extend, seeds found and extended), "sort",
 * "filter", "gapped_front" (the kernel in front of the gapped cascade: the hits whose two directions find nothing;
 * "gapped_front_hits": launches = hits it completed), "gapped" (LDS tier 0; "gapped_tier0_hits": launches = hits that entered it),
 * "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow" (wavefront-per-hit kernel), "traceback", "traceback_slow";
 * host wall-clock pseudo stages: "host_dfs" (background seed DFS), "host_dfs_wait",
 * "host_search_range", "host_cands", "host_drain_tail", "host_download" (the synchronous copy of
 * the hits of last_stage 1 / 2). */
int prb_ctx_stage_ms(prb_ctx *ctx, const char *stage, double *ms, int64_t *launches);
void prb_ctx_reset_timers(prb_ctx *ctx);

/* ---- stage 1: accessibility (Raccess), batched over sequences ----
 * seqs: concatenated characters, sequence i = seqs[offsets[i] .. offsets[i+1]);
 * acc/cond: concatenated float outputs with the same offsets (L floats per sequence,
 * entries the reference leaves untouched are 0), host memory. */
int prb_accessibility(prb_ctx *ctx, int32_t nseq, const char *seqs, const int64_t *offsets,
                      int32_t maximal_span, int32_t min_accessible_length, float *acc, float *cond);

/* diagnostics: runs Raccess for ONE sequence and returns the DP tables in the reference's
 * own layout (raccess.hpp:89-103): alpha_outer/beta_outer hold L+1 doubles; tables[0..5] =
 * Alpha_{stem,stemend,multi,multibif,multi1,multi2}, tables[6..11] = Beta_ same order, each
 * (L+1)*(W+2) doubles, row-major [i][j-i]; NULL entries are skipped. */
int prb_accessibility_tables(prb_ctx *ctx, const char *seq, int32_t len, int32_t maximal_span,
                             int32_t min_accessible_length, float *acc, float *cond,
                             double *alpha_outer, double *beta_outer, double *const tables[12]);

/* ---- host helpers that are part of the seam (stage 2: encode + suffix array) ---- */
int prb_encode_query(const char *seq, int32_t len, int32_t repeat_flag, uint8_t *enc /* len+1 */);
int prb_suffix_array(const uint8_t *text, int32_t n, int32_t *sa);

/* ---- database ---- */
int prb_db_open(prb_ctx *ctx, const char *prefix, prb_db **out);
/* The same with at most max_resident_pages of the database's pages in HBM at a time (0 = all, what
 * prb_db_open does unless PRB_DB_RESIDENT_PAGES says otherwise): prb_search_page uploads the page it is
 * asked for if it is not resident (the least recently used page makes room) and, with two pages or
 * more, uploads the next page on a copy stream of its own while this one is searched.  For databases
 * beyond HBM; results do not depend on it.  prb_db_page_uploads: pages uploaded so far. */
int prb_db_open_streaming(prb_ctx *ctx, const char *prefix, int32_t max_resident_pages, prb_db **out);
int64_t prb_db_page_uploads(const prb_db *db);
void prb_db_close(prb_db *db);
int prb_db_info(const prb_db *db, int32_t *hash_size, int32_t *repeat_flag, int32_t *maximal_span,
                int32_t *min_accessible_length, int32_t *npages);
int prb_db_page_info(const prb_db *db, int32_t page, int32_t *nseq, int64_t *nchars);
/* name / reference-style lengths of sequence `id` of `page` (db_reader.cpp:107-131) */
const char *prb_db_seq_name(const prb_db *db, int32_t page, int32_t id);
int prb_db_seq_lengths(const prb_db *db, int32_t page, int32_t id, int32_t *length,
                       int32_t *length_unmasked, int32_t *start_pos);
/* build a database from sequences (same files the reference's `db` step writes;
 * db_construction.cpp:37-83): accessibilities on the GPU, SA + k-mer table on the host. */
int prb_db_build(prb_ctx *ctx, const char *prefix, int32_t nseq, const char *const *names,
                 const char *seqs, const int64_t *offsets, int32_t repeat_flag, int32_t hash_size,
                 int32_t maximal_span, int32_t min_accessible_length, int32_t page_size);

/* ---- query batches ---- */
int prb_qbatch_create(prb_ctx *ctx, int32_t nq, const char *seqs, const int64_t *offsets,
                      int32_t repeat_flag, prb_qbatch **out);
void prb_qbatch_destroy(prb_qbatch *qb);
/* runs Raccess for every query of the batch; results stay in HBM for the search stages */
int prb_qbatch_accessibility(prb_ctx *ctx, prb_qbatch *qb, int32_t maximal_span,
                             int32_t min_accessible_length);
/* copies of the per-query arrays (NULL pointers are skipped) */
int prb_qbatch_get(prb_qbatch *qb, int32_t q, uint8_t *enc, int32_t *sa, float *acc, float *cond);
int32_t prb_qbatch_length_unmasked(const prb_qbatch *qb, int32_t q);
/* Optional: starts the seed search proper (the suffix-array DFS of SeedSearch::Run, seed_search.cpp:153-295; host
 * threads) of the batch against `page` in the background and returns at once.  A later prb_search_page with the
 * same database, page, -l and -e picks it up instead of starting its own - e.g. begun for the NEXT batch while
 * this one is searched.  It needs neither the accessibilities nor the GPU. */
int prb_qbatch_seed_search_begin(prb_ctx *ctx, prb_qbatch *qb, const prb_db *db, int32_t page, const prb_ris_opts *opts);

/* ---- stages 3-5: seed search, ungapped and gapped extension for all queries of a batch
 * against one page.  last_stage: 1 = seeds, 2 = after ungapped extension + filter,
 * 3 = final (after gapped extension + filter). ---- */
int prb_search_page(prb_ctx *ctx, prb_qbatch *qb, prb_db *db, int32_t page, const prb_ris_opts *opts,
                    int32_t last_stage, prb_hitset **out);
int64_t prb_hitset_size(const prb_hitset *hs);
const prb_hit *prb_hitset_hits(const prb_hitset *hs);
/* pairs (q, db) as int32[2], indexed by prb_hit.bp_offset.  With opts->output_style == 0
 * (simplified output, which prints only the two ends of an interaction,
 * rna_interaction_search.cpp:355-363) final hits carry exactly two pairs, the first and the
 * last of the reference's list; with output_style == 1 they carry every pair. */
const int32_t *prb_hitset_basepairs(const prb_hitset *hs, int64_t *count);
/* number of hits per stage for the whole call: seeds, after ungapped+filter, final */
void prb_hitset_counts(const prb_hitset *hs, int64_t counts[3]);
void prb_hitset_free(prb_hitset *hs);

/* ---- output: SaveMyResults (rna_interaction_search.cpp:322-369) ----
 * The result lines of one batch of queries, grouped query by query and page by page and numbered
 * from id0 on (the `Id` column; MergeOutput, rna_interaction_search.cpp:464-476), written to the file
 * descriptor `fd` (fd < 0: formatted and counted only).  pages[p] = the hits of the batch against
 * page p with their pair array, as prb_hitset_hits / prb_hitset_basepairs return them (or as they
 * arrived from another rank); hits ascending by `query`.  Host threads format in parallel. */
typedef struct prb_page_hits {
  const prb_hit *hits;
  int64_t nhits;
  const int32_t *basepairs; /* int32[2 * npairs] */
  int64_t npairs;
} prb_page_hits;
int prb_write_lines(const prb_db *db, int32_t nq, const char *const *qnames, const int32_t *qlen_unmasked,
                    const prb_page_hits *pages, int32_t npages, int32_t output_style, int64_t id0, int fd,
                    int64_t *lines, int64_t *bytes);

/* ---- multi-GPU: one process per GPU, the final hit gather over RCCL (xGMI) ----
 * Replaces MergeOutput's MPI token ring (rna_interaction_search.cpp:426-487) and, with the caller
 * dealing batches to ranks, the area / dynamic schedulers (rna_interaction_search.cpp:143-160,
 * 202-230).  Queries are independent end to end: this gather is the only exchange of the `ris` step.
 *   rank 0: prb_comm_unique_id(id); the 128 bytes reach the other ranks by any side channel (a file,
 *   torch.distributed, MPI ...); every rank: prb_comm_create(ctx, nranks, rank, id, &comm).
 * From then on the final hit sets of `ctx` also keep their packed records in HBM, and
 * prb_gather_hits - a collective, called by every rank once per (batch round, database page) -
 * moves them device to device: on `root`, *out is a new hit set with the hits of all ranks in rank
 * order, `query` shifted by the number of queries of the lower ranks and pair offsets rebased;
 * prb_hitset_gathered_queries gives every rank's batch size and the unmasked lengths of all the
 * queries in the same order.  Elsewhere *out = NULL.  A rank without a batch in this round passes mine = NULL,
 * nq = 0.  The gathered hit set borrows a pinned buffer of the communicator until it is
 * freed with prb_hitset_free (from any thread; the buffer outlives prb_comm_destroy if it has to).  The gather works on a stream of
 * its own, so one host thread may gather (and print) batch k while another one searches batch k + 1 on the
 * context - as long as every rank issues its gathers in the same order. */
typedef struct prb_comm prb_comm;
#define PRB_COMM_ID_BYTES 128
int prb_comm_unique_id(char id[PRB_COMM_ID_BYTES]);
int prb_comm_create(prb_ctx *ctx, int32_t nranks, int32_t rank, const char id[PRB_COMM_ID_BYTES], prb_comm **out);
void prb_comm_destroy(prb_comm *comm);
int prb_gather_hits(prb_comm *comm, const prb_hitset *mine, int32_t nq, const int32_t *qlen_unmasked, int32_t root,
                    prb_hitset **out);
int prb_hitset_gathered_queries(const prb_hitset *hs, int32_t *nranks, const int32_t **nq_of_rank,
                                const int32_t **qlen_unmasked);
/* The placement rule of prb_gather_hits as a pure host function (no GPU, no communicator): counts[3 * k + {0, 1, 2}] =
 * hits, pair-array ints and queries rank k brings; bases[3 * k + j] = where rank k's share of kind j starts in the
 * gathered arrays (k = 0 .. nranks; entry nranks = the totals).  The root receives rank k's records at bases[3k],
 * adds bases[3k + 2] to their `query` and bases[3k + 1] / 2 to their `bp_offset`.  This is what replaces the order
 * in which the reference's ranks append their temporary files (rna_interaction_search.cpp:426-487). */
int prb_gather_plan(int32_t nranks, const int64_t *counts /* 3 * nranks */, int64_t *bases /* 3 * (nranks + 1) */);
/* keep (on != 0) the packed records of later final hit sets in HBM without a communicator (tests) */
void prb_ctx_keep_device_records(prb_ctx *ctx, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
