// tools/dump_params.cpp -- one-off data extraction, run in the container that has the
// reference:  g++ -I/root/reference/src tools/dump_params.cpp -o /tmp/dump_params &&
//             /tmp/dump_params > priblast_amd/params/rna_andronescu2007.par
//
// The nearest-neighbour energy parameters (Andronescu et al. 2007 "RNA-Params", the set
// the reference compiles in as energy_par.hpp:6-174 and intloops.hpp:6/309/1788) are
// numeric DATA that any implementation must share; they are written out as a plain
// parameter file (own layout: "@ name dims" + whitespace-separated integers, units of
// 0.01 kcal/mol, INF spelled out) which both the product and the oracle load at run time.
#include <cstdio>

#include "energy_par.hpp"
#include "intloops.hpp"

static void emit(const char *name, const int *p, int n, const char *dims, int per_line) {
  printf("@ %s %s\n", name, dims);
  for (int i = 0; i < n; i++) {
    if (p[i] == INF)
      printf("INF");
    else
      printf("%d", p[i]);
    putchar(((i + 1) % per_line == 0 || i + 1 == n) ? '\n' : ' ');
  }
}

int main() {
  printf("# pRIblast-hip nearest-neighbour energy parameter file, format 1\n");
  printf("# parameter set: Andronescu et al. 2007 (RNA-Params), as used by RIblast/pRIblast\n");
  printf("# units: 0.01 kcal/mol at 37 C; INF = %d; pair types 1..6 = CG GC GU UG AU UA; bases 0..4 = @ A C G U\n", INF);
  printf("$ temperature %d\n$ gasconst %.5f\n$ k0 %.2f\n$ lxc37 %.3f\n", temperature, GASCONST, K0, lxc37);
  printf("$ turn %d\n$ maxloop %d\n", TURN, MAXLOOP);
  printf("$ terminal_au %d\n$ ml_closing %d\n$ ml_intern %d\n$ ml_base %d\n$ max_ninio %d\n$ f_ninio %d\n",
         TerminalAU, ML_closing37, ML_intern37, ML_BASE37, MAX_NINIO, F_ninio37);
  emit("bp_pair", &BP_pair[0][0], 25, "5 5", 5);
  emit("rtype", rtype, 7, "7", 7);
  emit("hairpin", hairpin37, 31, "31", 31);
  emit("bulge", bulge37, 31, "31", 31);
  emit("internal", internal_loop37, 31, "31", 31);
  emit("stack", &stack37[0][0], 49, "7 7", 7);
  emit("mismatch_hairpin", &mismatchH37[0][0][0], 175, "7 5 5", 25);
  emit("mismatch_interior", &mismatchI37[0][0][0], 175, "7 5 5", 25);
  emit("dangle5", &dangle5_37[0][0], 40, "8 5", 5);
  emit("dangle3", &dangle3_37[0][0], 40, "8 5", 5);
  emit("int11", &int11_37[0][0][0][0], 8 * 8 * 25, "8 8 5 5", 25);
  emit("int21", &int21_37[0][0][0][0][0], 8 * 8 * 125, "8 8 5 5 5", 125);
  emit("int22", &int22_37[0][0][0][0][0][0], 8 * 8 * 625, "8 8 5 5 5 5", 125);
  return 0;
}
