/* oracle/oracle_io.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * FASTA input (fastafile_reader.cpp:373-409) and the on-disk database pages
 * (.bas rna_interaction_search_parameters.cpp:97-114; .seq/.acc/.nam/.ind
 * db_reader.cpp:61-177, written by db_construction.cpp:371-436 / raccess.cpp:447-480). */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static char *read_line(FILE *f) {
  size_t cap = 256, n = 0;
  char *b = malloc(cap);
  int c;
  int any = 0;
  while ((c = fgetc(f)) != EOF) {
    any = 1;
    if (c == '\n') break;
    if (n + 2 > cap) b = realloc(b, cap *= 2);
    b[n++] = (char)c;
  }
  if (!any) {
    free(b);
    return NULL;
  }
  b[n] = 0;
  return b;
}

orc_fasta *orc_fasta_read(const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) return NULL;
  orc_fasta *fa = calloc(1, sizeof *fa);
  int cap = 0;
  char *cur = NULL;
  size_t curlen = 0, curcap = 0;
  char *line;
  int first = 1;
  while ((line = read_line(f))) {
    size_t n = strlen(line);
    if (first || line[0] == '>') {
      if (!first) {
        fa->seqs[fa->n - 1] = cur ? cur : calloc(1, 1);
        fa->lens[fa->n - 1] = (int)curlen;
      }
      if (fa->n == cap) {
        cap = cap ? cap * 2 : 64;
        fa->names = realloc(fa->names, cap * sizeof(char *));
        fa->seqs = realloc(fa->seqs, cap * sizeof(char *));
        fa->lens = realloc(fa->lens, cap * sizeof(int));
      }
      fa->names[fa->n] = strdup(n ? line + 1 : line);
      fa->n++;
      cur = NULL;
      curlen = curcap = 0;
      first = 0;
    } else {
      while (n && (line[n - 1] == '\r' || line[n - 1] == '\n')) n--;
      if (curlen + n + 1 > curcap) {
        curcap = (curlen + n + 1) * 2;
        cur = realloc(cur, curcap);
      }
      memcpy(cur + curlen, line, n);
      curlen += n;
      cur[curlen] = 0;
    }
    free(line);
  }
  if (fa->n) {
    fa->seqs[fa->n - 1] = cur ? cur : calloc(1, 1);
    fa->lens[fa->n - 1] = (int)curlen;
  }
  fclose(f);
  return fa;
}

void orc_fasta_free(orc_fasta *f) {
  if (!f) return;
  for (int i = 0; i < f->n; i++) {
    free(f->names[i]);
    free(f->seqs[i]);
  }
  free(f->names);
  free(f->seqs);
  free(f->lens);
  free(f);
}

static FILE *open_ext(const char *prefix, const char *ext, const char *mode) {
  size_t n = strlen(prefix) + strlen(ext) + 1;
  char *p = malloc(n);
  snprintf(p, n, "%s%s", prefix, ext);
  FILE *f = fopen(p, mode);
  free(p);
  return f;
}

static int rd_i32(FILE *f, int32_t *v) { return fread(v, 4, 1, f) == 1 ? 0 : -1; }

orc_db *orc_db_open(const char *prefix) {
  FILE *bas = open_ext(prefix, ".bas", "rb");
  FILE *seq = open_ext(prefix, ".seq", "rb");
  FILE *acc = open_ext(prefix, ".acc", "rb");
  FILE *nam = open_ext(prefix, ".nam", "r");
  FILE *ind = open_ext(prefix, ".ind", "rb");
  if (!bas || !seq || !acc || !nam || !ind) return NULL;
  orc_db *db = calloc(1, sizeof *db);
  int32_t v[4];
  if (fread(v, 4, 4, bas) != 4) return NULL;
  fclose(bas);
  db->hash_size = v[0];
  db->repeat_flag = v[1];
  db->maximal_span = v[2];
  db->min_accessible_length = v[3];
  int cap = 0;
  for (;;) {
    int32_t nseq;
    if (rd_i32(seq, &nseq)) break; /* EOF on .seq ends the DB, db_reader.cpp:73-82 */
    if (db->npages == cap) {
      cap = cap ? cap * 2 : 4;
      db->pages = realloc(db->pages, cap * sizeof(orc_page));
    }
    orc_page *pg = &db->pages[db->npages++];
    memset(pg, 0, sizeof *pg);
    pg->nseq = nseq;
    pg->seq_length = malloc(sizeof(int32_t) * nseq);
    pg->start_pos = malloc(sizeof(int32_t) * nseq);
    if (fread(pg->seq_length, 4, nseq, seq) != (size_t)nseq) return NULL;
    int t = 0;
    for (int i = 0; i < nseq; i++) { /* db_reader.cpp:107-112 */
      pg->start_pos[i] = t;
      t += pg->seq_length[i] + 1;
    }
    int32_t nchars;
    if (rd_i32(seq, &nchars)) return NULL;
    pg->nchars = nchars;
    pg->seqs = calloc((size_t)nchars + 64, 1); /* zero padding behind the text: see orc_extend_gapped */
    if (fread(pg->seqs, 1, nchars, seq) != (size_t)nchars) return NULL;
    /* seq_length_rep: count of unmasked codes per sequence, db_reader.cpp:122-131 */
    pg->seq_length_rep = calloc(nseq + 2, sizeof(int32_t));
    {
      int k = 0, c = 0;
      for (int i = 0; i < nchars; i++) {
        if (pg->seqs[i] == 0) {
          if (k <= nseq) pg->seq_length_rep[k] = c;
          k++;
          c = 0;
        } else if (pg->seqs[i] >= 2 && pg->seqs[i] <= 5) {
          c++;
        }
      }
    }
    pg->acc = malloc(sizeof(float *) * nseq);
    pg->cond = malloc(sizeof(float *) * nseq);
    pg->acc_len = malloc(sizeof(int32_t) * nseq);
    pg->cond_len = malloc(sizeof(int32_t) * nseq);
    for (int i = 0; i < nseq; i++) { /* db_reader.cpp:133-150 */
      int32_t n;
      if (rd_i32(acc, &n)) return NULL;
      if (n < 0) n = 0;
      pg->acc_len[i] = n;
      pg->acc[i] = malloc(sizeof(float) * (n ? n : 1));
      if (fread(pg->acc[i], 4, n, acc) != (size_t)n) return NULL;
      if (rd_i32(acc, &n)) return NULL;
      pg->cond_len[i] = n;
      pg->cond[i] = malloc(sizeof(float) * (n ? n : 1));
      if (fread(pg->cond[i], 4, n, acc) != (size_t)n) return NULL;
    }
    pg->names = malloc(sizeof(char *) * nseq);
    for (int i = 0; i < nseq; i++) {
      char *l = read_line(nam);
      pg->names[i] = l ? l : strdup("");
    }
    int32_t nsa;
    if (rd_i32(ind, &nsa)) return NULL;
    pg->sa = malloc(sizeof(int32_t) * (nsa ? nsa : 1));
    if (fread(pg->sa, 4, nsa, ind) != (size_t)nsa) return NULL;
    pg->start_hash = malloc(sizeof(int32_t *) * db->hash_size);
    pg->end_hash = malloc(sizeof(int32_t *) * db->hash_size);
    for (int pass = 0; pass < 2; pass++) { /* db_reader.cpp:163-174 */
      size_t n = 4;
      for (int i = 0; i < db->hash_size; i++, n *= 4) {
        int32_t *h = malloc(sizeof(int32_t) * n);
        if (fread(h, 4, n, ind) != n) return NULL;
        (pass == 0 ? pg->start_hash : pg->end_hash)[i] = h;
      }
    }
  }
  fclose(seq);
  fclose(acc);
  fclose(nam);
  fclose(ind);
  return db;
}

void orc_db_close(orc_db *db) {
  if (!db) return;
  for (int p = 0; p < db->npages; p++) {
    orc_page *pg = &db->pages[p];
    for (int i = 0; i < pg->nseq; i++) {
      free(pg->acc[i]);
      free(pg->cond[i]);
      free(pg->names[i]);
    }
    for (int i = 0; i < db->hash_size; i++) {
      free(pg->start_hash[i]);
      free(pg->end_hash[i]);
    }
    free(pg->acc); free(pg->cond); free(pg->acc_len); free(pg->cond_len); free(pg->names);
    free(pg->start_hash); free(pg->end_hash); free(pg->sa); free(pg->seqs);
    free(pg->seq_length); free(pg->start_pos); free(pg->seq_length_rep);
  }
  free(db->pages);
  free(db);
}
