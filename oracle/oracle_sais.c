/* oracle/oracle_sais.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference builds suffix arrays with Yuta Mori's SA-IS (sais.cpp:656-667, called from
 * rna_interaction_search.cpp:252-262 and db_construction.cpp:331-335).  A suffix array is
 * unique for a given text, so the oracle uses the simplest correct construction
 * (prefix doubling, O(n log^2 n)); tests pin it against the reference's own SA dumps
 * (tests/golden/c1_q.sa, c1db.ind).  Plus the query encoder, encoder.hpp:36-79 /
 * encoder.cpp:38-44.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* qsort has no context argument: per thread, because orc_ris builds the suffix arrays of several queries at once
 * (shared globals here gave a wrong suffix array once in a few runs - a test that failed now and then) */
static _Thread_local const int32_t *g_rank;
static _Thread_local int g_k, g_n;

static int cmp_pair(const void *pa, const void *pb) {
  int32_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
  if (g_rank[a] != g_rank[b]) return g_rank[a] < g_rank[b] ? -1 : 1;
  int32_t ra = a + g_k < g_n ? g_rank[a + g_k] : -1;
  int32_t rb = b + g_k < g_n ? g_rank[b + g_k] : -1;
  return ra < rb ? -1 : (ra > rb ? 1 : 0);
}

int orc_suffix_array(const uint8_t *T, int32_t *SA, int n) {
  if (n <= 0) return 0;
  int32_t *rank = malloc((size_t)n * sizeof(int32_t));
  int32_t *tmp = malloc((size_t)n * sizeof(int32_t));
  for (int i = 0; i < n; i++) {
    SA[i] = i;
    rank[i] = T[i];
  }
  for (int k = 1;; k <<= 1) {
    g_rank = rank;
    g_k = k;
    g_n = n;
    qsort(SA, (size_t)n, sizeof(int32_t), cmp_pair);
    tmp[SA[0]] = 0;
    for (int i = 1; i < n; i++) tmp[SA[i]] = tmp[SA[i - 1]] + (cmp_pair(&SA[i - 1], &SA[i]) < 0);
    memcpy(rank, tmp, (size_t)n * sizeof(int32_t));
    if (rank[SA[n - 1]] == n - 1) break;
    if (k > n) break;
  }
  free(rank);
  free(tmp);
  return 0;
}

/* encoder.hpp:36-79 (code table) + encoder.cpp:38-44 (query: forward, trailing sentinel) */
void orc_encode_query(const char *seq, int L, int repeat_flag, uint8_t *out) {
  uint8_t table[256];
  memset(table, 1, sizeof table);
  table['A'] = 2; table['C'] = 3; table['G'] = 4; table['T'] = 5; table['U'] = 5;
  if (repeat_flag == 1) {
    table['a'] = 6; table['c'] = 7; table['g'] = 8; table['t'] = 9; table['u'] = 9;
  } else if (repeat_flag == 2) {
    table['a'] = 2; table['c'] = 3; table['g'] = 4; table['t'] = 5; table['u'] = 5;
  }
  for (int i = 0; i < L; i++) out[i] = table[(unsigned char)seq[i]];
  out[L] = 0;
}
