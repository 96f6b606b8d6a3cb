/* oracle/oracle_params.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Loads the nearest-neighbour parameter data file (values of energy_par.hpp:6-174 and
 * intloops.hpp:6/309/1788, written by tools/dump_params.cpp). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static orc_params g_p;
static int g_loaded = 0;

const orc_params *orc_params_get(void) { return g_loaded ? &g_p : NULL; }

static int read_ints(FILE *f, int *dst, int n) {
  char tok[64];
  for (int i = 0; i < n; i++) {
    if (fscanf(f, "%63s", tok) != 1) return -1;
    dst[i] = strcmp(tok, "INF") == 0 ? ORC_INF : atoi(tok);
  }
  return 0;
}

int orc_params_load(const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  memset(&g_p, 0, sizeof(g_p));
  double temperature = 0, gasconst = 0, k0 = 0;
  char line[512];
  int ntab = 0;
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#' || line[0] == '\n') continue;
    char name[64];
    if (line[0] == '$') {
      char val[64];
      if (sscanf(line + 1, "%63s %63s", name, val) != 2) { fclose(f); return -2; }
      if (!strcmp(name, "temperature")) temperature = atoi(val);
      else if (!strcmp(name, "gasconst")) gasconst = strtod(val, NULL);
      else if (!strcmp(name, "k0")) k0 = strtod(val, NULL);
      else if (!strcmp(name, "lxc37")) g_p.lxc37 = strtod(val, NULL);
      else if (!strcmp(name, "terminal_au")) g_p.terminal_au = atoi(val);
      else if (!strcmp(name, "ml_closing")) g_p.ml_closing = atoi(val);
      else if (!strcmp(name, "ml_intern")) g_p.ml_intern = atoi(val);
      else if (!strcmp(name, "ml_base")) g_p.ml_base = atoi(val);
      else if (!strcmp(name, "max_ninio")) g_p.max_ninio = atoi(val);
      else if (!strcmp(name, "f_ninio")) g_p.f_ninio = atoi(val);
      continue;
    }
    if (line[0] != '@') continue;
    if (sscanf(line + 1, "%63s", name) != 1) { fclose(f); return -2; }
    int *dst = NULL, n = 0;
#define TAB(nm, field) if (!strcmp(name, nm)) { dst = (int *)g_p.field; n = (int)(sizeof(g_p.field) / sizeof(int)); }
    TAB("bp_pair", bp_pair) TAB("rtype", rtype) TAB("hairpin", hairpin37) TAB("bulge", bulge37)
    TAB("internal", internal37) TAB("stack", stack37) TAB("mismatch_hairpin", mismatchH37)
    TAB("mismatch_interior", mismatchI37) TAB("dangle5", dangle5_37) TAB("dangle3", dangle3_37)
    TAB("int11", int11_37) TAB("int21", int21_37) TAB("int22", int22_37)
#undef TAB
    if (!dst) { fclose(f); return -3; }
    if (read_ints(f, dst, n)) { fclose(f); return -4; }
    ntab++;
  }
  fclose(f);
  if (ntab != 13) return -5;
  /* energy_par.hpp:12-13: kT = (temperature + K0) * GASCONST */
  g_p.kT = (temperature + k0) * gasconst;
  g_loaded = 1;
  return 0;
}
