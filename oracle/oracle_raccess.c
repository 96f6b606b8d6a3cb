/* oracle/oracle_raccess.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar restatement of the reference's Raccess (windowed McCaskill inside/outside +
 * accessibility of every length-delta window), in-memory overload:
 *   Raccess::Run                      raccess.cpp:42-50
 *   set_energy_parameters             raccess.hpp:105-158
 *   Initiallize                       raccess.cpp:52-97
 *   CalcInsideVariable                raccess.cpp:99-242
 *   CalcOutsideVariable               raccess.cpp:258-412
 *   logsumexp                         raccess.cpp:414-419
 *   CalcAccessibility(acc, cond)      raccess.cpp:484-528
 *   Exterior/Hairpin/Multi prob.      raccess.cpp:530-612
 *   biloop, linear / log-sum          raccess.cpp:614-771
 *   LoopEnergy / HairpinEnergy        raccess.cpp:773-832
 *
 * Numerics contract: strict IEEE (no contraction), every sum folded in the
 * reference's loop order, -INF = -1e6 is an exact sentinel compared with ==/!=.
 * Tables are flat bands: T[i*(W+2) + d] <-> reference T[i][d].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define NINF ORC_NINF

typedef struct {
  /* scaled parameters, raccess.hpp:70-85 */
  double hairpin[31], bulge[31], internal[31], ninio[ORC_MAXLOOP + 1];
  double mismatchH[7][5][5], mismatchI[7][5][5], stack[7][7];
  double dangle5[8][5], dangle3[8][5];
  double int11[8][8][5][5], int21[8][8][5][5][5], int22[8][8][5][5][5][5];
  double TermAU, MLclosing, MLintern, MLbase;
  double lxc37, kT;
  const orc_params *P;
  int L, W, delta, S; /* S = W + 2 = row stride */
  int *seq;           /* 1-based codes 0..4, seq[0] = 0 */
  double *ao, *bo;
  double *a_stem, *a_stemend, *a_multi, *a_multibif, *a_multi1, *a_multi2;
  double *b_stem, *b_stemend, *b_multi, *b_multibif, *b_multi1, *b_multi2;
} RA;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* raccess.cpp:414-419 */
static inline double lse(double x, double y) {
  return x > y ? x + (double)orc_logf((float)(orc_expd(y - x) + 1.0))
               : y + (double)orc_logf((float)(orc_expd(x - y) + 1.0));
}

/* raccess.hpp:105-158 */
static void scale_params(RA *r) {
  const orc_params *P = r->P;
  double kT = P->kT;
  r->kT = kT;
  r->lxc37 = P->lxc37;
  r->MLclosing = -P->ml_closing * 10 / kT;
  r->MLintern = -P->ml_intern * 10. / kT;
  r->MLbase = -P->ml_base * 10. / kT;
  r->TermAU = -P->terminal_au * 10 / kT;
  for (int i = 0; i <= 30; i++) {
    r->hairpin[i] = -P->hairpin37[i] * 10. / kT;
    r->bulge[i] = -P->bulge37[i] * 10. / kT;
    r->internal[i] = -P->internal37[i] * 10. / kT;
  }
  memset(r->dangle5, 0, sizeof r->dangle5);
  memset(r->dangle3, 0, sizeof r->dangle3);
  for (int i = 0; i < 7; i++) {
    for (int j = 0; j < 5; j++)
      for (int k = 0; k < 5; k++) {
        r->mismatchI[i][j][k] = -P->mismatchI37[i][j][k] * 10.0 / kT;
        r->mismatchH[i][j][k] = -P->mismatchH37[i][j][k] * 10.0 / kT;
      }
    for (int j = 0; j < 7; j++) r->stack[i][j] = -P->stack37[i][j] * 10. / kT;
    for (int j = 0; j <= 4; j++) {
      r->dangle5[i][j] = -P->dangle5_37[i][j] * 10. / kT;
      r->dangle3[i][j] = -P->dangle3_37[i][j] * 10. / kT;
      if (i > 2) r->dangle3[i][j] += r->TermAU;
    }
  }
  for (int i = 0; i <= 7; i++)
    for (int j = 0; j <= 7; j++)
      for (int k = 0; k < 5; k++)
        for (int l = 0; l < 5; l++) {
          r->int11[i][j][k][l] = -P->int11_37[i][j][k][l] * 10. / kT;
          for (int m = 0; m < 5; m++) {
            r->int21[i][j][k][l][m] = -P->int21_37[i][j][k][l][m] * 10. / kT;
            for (int n = 0; n < 5; n++)
              r->int22[i][j][k][l][m][n] = -P->int22_37[i][j][k][l][m][n] * 10. / kT;
          }
        }
  for (int i = 0; i <= ORC_MAXLOOP; i++)
    r->ninio[i] = -imin(P->max_ninio, i * P->f_ninio) * 10 / kT;
}

/* raccess.cpp:773-817 */
static double loop_energy(const RA *r, int type, int type2, int i, int j, int p, int q) {
  const int *s = r->seq;
  double z = 0;
  int u1 = p - i - 1;
  int u2 = j - q - 1;
  if (u1 == 0 && u2 == 0) {
    z = r->stack[type][type2];
  } else if (u1 == 0 || u2 == 0) {
    int u = u1 == 0 ? u2 : u1;
    z = u <= 30 ? r->bulge[u] : r->bulge[30] - r->lxc37 * log(u / 30.) * 10. / r->kT;
    if (u == 1) {
      z += r->stack[type][type2];
    } else {
      if (type > 2) z += r->TermAU;
      if (type2 > 2) z += r->TermAU;
    }
  } else if (u1 + u2 == 2) {
    z = r->int11[type][type2][s[i + 1]][s[j - 1]];
  } else if (u1 == 1 && u2 == 2) {
    z = r->int21[type][type2][s[i + 1]][s[q + 1]][s[j - 1]];
  } else if (u1 == 2 && u2 == 1) {
    z = r->int21[type2][type][s[q + 1]][s[i + 1]][s[p - 1]];
  } else if (u1 == 2 && u2 == 2) {
    z = r->int22[type][type2][s[i + 1]][s[p - 1]][s[q + 1]][s[j - 1]];
  } else {
    z = r->internal[u1 + u2] + r->mismatchI[type][s[i + 1]][s[j - 1]] +
        r->mismatchI[type2][s[q + 1]][s[p - 1]];
    z += r->ninio[abs(u1 - u2)];
  }
  return z;
}

/* raccess.cpp:819-832 */
static double hairpin_energy(const RA *r, int type, int i, int j) {
  int d = j - i - 1;
  double q = d <= 30 ? r->hairpin[d] : r->hairpin[30] - r->lxc37 * log(d / 30.) * 10. / r->kT;
  if (d != 3) {
    q += r->mismatchH[type][r->seq[i + 1]][r->seq[j - 1]];
  } else if (type > 2) {
    q += r->TermAU;
  }
  return q;
}

/* raccess.cpp:244-256 */
static double dangle_energy(const RA *r, int type, int a, int b) {
  double x = 0;
  if (type != 0) {
    if (a > 0) x += r->dangle5[type][r->seq[a]];
    if (b < r->L) x += r->dangle3[type][r->seq[b + 1]];
    if (b == r->L && type > 2) x += r->TermAU;
  }
  return x;
}

#define T(tab, i, d) (r->tab[(size_t)(i) * r->S + (d)])
#define BP(a, b) (r->P->bp_pair[a][b])

/* raccess.cpp:99-242 */
static void inside(RA *r) {
  const int L = r->L, W = r->W;
  const int *s = r->seq;
  const int *rtype = r->P->rtype;
  for (int j = ORC_TURN + 1; j <= L; j++) {
    for (int i = j - ORC_TURN; i >= imax(0, j - W - 1); i--) {
      const int d = j - i;
      /* Alpha_stem, :102-129 */
      int type = BP(s[i + 1], s[j]);
      int type2 = BP(s[i + 2], s[j - 1]);
      double temp = 0;
      int flag = 0;
      if (type != 0) {
        type2 = rtype[type2];
        if (T(a_stem, i + 1, d - 2) != NINF) {
          if (type2 != 0)
            temp = T(a_stem, i + 1, d - 2) + loop_energy(r, type, type2, i + 1, j, i + 2, j - 1);
          flag = 1;
        }
        if (T(a_stemend, i + 1, d - 2) != NINF) {
          temp = flag == 1 ? lse(temp, T(a_stemend, i + 1, d - 2)) : T(a_stemend, i + 1, d - 2);
          flag = 1;
        }
        T(a_stem, i, d) = flag == 0 ? NINF : temp;
      } else {
        T(a_stem, i, d) = NINF;
      }

      /* Alpha_multibif, :131-143 */
      temp = 0;
      flag = 0;
      for (int k = i + 1; k <= j - 1; k++) {
        double m1 = T(a_multi1, i, k - i), m2 = T(a_multi2, k, j - k);
        if (m1 != NINF && m2 != NINF) {
          temp = flag == 0 ? m1 + m2 : lse(temp, m1 + m2);
          flag = 1;
        }
      }
      T(a_multibif, i, d) = flag == 0 ? NINF : temp;

      /* Alpha_multi2, :145-162 */
      temp = 0;
      flag = 0;
      if (type != 0 && T(a_stem, i, d) != NINF) {
        temp = T(a_stem, i, d) + r->MLintern + dangle_energy(r, type, i, j);
        flag = 1;
      }
      if (T(a_multi2, i, d - 1) != NINF) {
        T(a_multi2, i, d) = T(a_multi2, i, d - 1) + r->MLbase;
        if (flag == 1) T(a_multi2, i, d) = lse(temp, T(a_multi2, i, d));
      } else {
        T(a_multi2, i, d) = flag == 0 ? NINF : temp;
      }

      /* Alpha_multi1, :164-175 */
      {
        double m2 = T(a_multi2, i, d), mb = T(a_multibif, i, d);
        if (m2 != NINF && mb != NINF) T(a_multi1, i, d) = lse(m2, mb);
        else if (m2 == NINF) T(a_multi1, i, d) = mb;
        else T(a_multi1, i, d) = m2;
      }

      /* Alpha_multi, :177-191 */
      flag = 0;
      if (T(a_multi, i + 1, d - 1) != NINF) {
        T(a_multi, i, d) = T(a_multi, i + 1, d - 1) + r->MLbase;
        flag = 1;
      }
      if (flag == 1) {
        if (T(a_multibif, i, d) != NINF) T(a_multi, i, d) = lse(T(a_multi, i, d), T(a_multibif, i, d));
      } else {
        T(a_multi, i, d) = T(a_multibif, i, d);
      }

      /* Alpha_stemend, :193-226 */
      if (j != L) {
        type = BP(s[i], s[j + 1]);
        if (type != 0) {
          temp = hairpin_energy(r, type, i, j + 1);
          for (int p = i; p <= imin(i + ORC_MAXLOOP, j - ORC_TURN - 2); p++) {
            int u1 = p - i;
            for (int q = imax(p + ORC_TURN + 2, j - ORC_MAXLOOP + u1); q <= j; q++) {
              type2 = BP(s[p + 1], s[q]);
              if (T(a_stem, p, q - p) != NINF) {
                if (type2 != 0 && !(p == i && q == j)) {
                  type2 = rtype[type2];
                  temp = lse(temp, T(a_stem, p, q - p) + loop_energy(r, type, type2, i, j + 1, p + 1, q));
                }
              }
            }
          }
          int tt = rtype[type];
          temp = lse(temp, T(a_multi, i, d) + r->MLclosing + r->MLintern + r->dangle3[tt][s[i + 1]] +
                               r->dangle5[tt][s[j]]);
          T(a_stemend, i, d) = temp;
        } else {
          T(a_stemend, i, d) = NINF;
        }
      }
    }
  }

  /* Alpha_outer, :230-241 */
  for (int i = 1; i <= L; i++) {
    double temp = r->ao[i - 1];
    for (int p = imax(0, i - W - 1); p < i; p++) {
      if (T(a_stem, p, i - p) != NINF) {
        int type = BP(s[p + 1], s[i]);
        double ao = T(a_stem, p, i - p) + dangle_energy(r, type, p, i);
        temp = lse(temp, ao + r->ao[p]);
      }
    }
    r->ao[i] = temp;
  }
}

/* raccess.cpp:258-412 */
static void outside(RA *r) {
  const int L = r->L, W = r->W;
  const int *s = r->seq;
  const int *rtype = r->P->rtype;
  /* Beta_outer, :260-271 */
  for (int i = L - 1; i >= 0; i--) {
    double temp = r->bo[i + 1];
    for (int p = i + 1; p <= imin(i + W + 1, L); p++) {
      if (T(a_stem, i, p - i) != NINF) {
        int type = BP(s[i + 1], s[p]);
        double bo = T(a_stem, i, p - i) + dangle_energy(r, type, i, p);
        temp = lse(temp, bo + r->bo[p]);
      }
    }
    r->bo[i] = temp;
  }

  for (int q = L; q >= ORC_TURN + 1; q--) {
    for (int p = imax(0, q - W - 1); p <= q - ORC_TURN; p++) {
      const int d = q - p;
      double temp = 0;
      if (p != 0 && q != L) {
        /* Beta_stemend, :278-279 */
        T(b_stemend, p, d) = d >= W ? NINF : T(b_stem, p - 1, d + 2);

        /* Beta_multi, :281-308 */
        int flag = 0;
        if (d + 1 <= W + 1) {
          if (T(b_multi, p - 1, d + 1) != NINF) {
            temp = T(b_multi, p - 1, d + 1) + r->MLbase;
            flag = 1;
          }
        }
        int type = BP(s[p], s[q + 1]);
        int tt = rtype[type];
        if (flag == 1) {
          if (T(b_stemend, p, d) != NINF)
            temp = lse(temp, T(b_stemend, p, d) + r->MLclosing + r->MLintern + r->dangle3[tt][s[p + 1]] +
                                 r->dangle5[tt][s[q]]);
        } else {
          if (T(b_stemend, p, d) != NINF)
            temp = T(b_stemend, p, d) + r->MLclosing + r->MLintern + r->dangle3[tt][s[p + 1]] +
                   r->dangle5[tt][s[q]];
          else
            temp = NINF;
        }
        T(b_multi, p, d) = temp;

        /* Beta_multi1, :310-324 */
        temp = 0;
        flag = 0;
        for (int k = q + 1; k <= imin(L, p + W); k++) {
          double bb = T(b_multibif, p, k - p), m2 = T(a_multi2, q, k - q);
          if (bb != NINF && m2 != NINF) {
            temp = flag == 0 ? bb + m2 : lse(temp, bb + m2);
            flag = 1;
          }
        }
        T(b_multi1, p, d) = flag == 1 ? temp : NINF;

        /* Beta_multi2, :326-352 */
        temp = 0;
        flag = 0;
        if (T(b_multi1, p, d) != NINF) {
          temp = T(b_multi1, p, d);
          flag = 1;
        }
        if (d <= W) {
          if (T(b_multi2, p, d + 1) != NINF) {
            temp = flag == 1 ? lse(temp, T(b_multi2, p, d + 1) + r->MLbase) : T(b_multi2, p, d + 1) + r->MLbase;
            flag = 1;
          }
        }
        for (int k = imax(0, q - W); k < p; k++) {
          double bb = T(b_multibif, k, q - k), m1 = T(a_multi1, k, p - k);
          if (bb != NINF && m1 != NINF) {
            temp = flag == 0 ? bb + m1 : lse(temp, bb + m1);
            flag = 1;
          }
        }
        T(b_multi2, p, d) = flag == 0 ? NINF : temp;

        /* Beta_multibif, :354-364 */
        {
          double m1 = T(b_multi1, p, d), m = T(b_multi, p, d);
          if (m1 != NINF && m != NINF) T(b_multibif, p, d) = lse(m1, m);
          else if (m == NINF) T(b_multibif, p, d) = m1;
          else T(b_multibif, p, d) = m;
        }
      }

      /* Beta_stem, :367-409 */
      int type2 = BP(s[p + 1], s[q]);
      if (type2 != 0) {
        temp = r->ao[p] + r->bo[q] + dangle_energy(r, type2, p, q);
        type2 = rtype[type2];
        for (int i = imax(1, p - ORC_MAXLOOP); i <= p; i++) {
          for (int j = q; j <= imin(q + ORC_MAXLOOP - p + i, L - 1); j++) {
            int type = BP(s[i], s[j + 1]);
            if (type != 0 && !(i == p && j == q)) {
              if (j - i <= W + 1 && T(b_stemend, i, j - i) != NINF)
                temp = lse(temp, T(b_stemend, i, j - i) + loop_energy(r, type, type2, i, j + 1, p + 1, q));
            }
          }
        }
        if (p != 0 && q != L) {
          int type = BP(s[p], s[q + 1]);
          if (type != 0) {
            if (d + 2 <= W + 1 && T(b_stem, p - 1, d + 2) != NINF)
              temp = lse(temp, T(b_stem, p - 1, d + 2) + loop_energy(r, type, type2, p, q + 1, p + 1, q));
          }
        }
        T(b_stem, p, d) = temp;
        if (T(b_multi2, p, d) != NINF) {
          type2 = rtype[type2];
          temp = T(b_multi2, p, d) + r->MLintern + dangle_energy(r, type2, p, q);
          T(b_stem, p, d) = lse(temp, T(b_stem, p, d));
        }
      } else {
        T(b_stem, p, d) = NINF;
      }
    }
  }
}

/* raccess.cpp:530-534 */
static double exterior_prob(const RA *r, int x, int w) {
  return orc_expd(r->ao[x - 1] + r->bo[x + w - 1] - r->ao[r->L]);
}

/* raccess.cpp:581-612 */
static double multi_prob(const RA *r, int x, int w) {
  const int L = r->L, W = r->W;
  double probability = 0.0, temp = 0.0;
  int flag = 0;
  for (int i = x + w - 1; i <= imin(x + W, L); i++) {
    double b = T(b_multi, x - 1, i - x + 1), a = T(a_multi, x + w - 1, i - x - w + 1);
    if (b != NINF && a != NINF) {
      temp = flag == 0 ? b + a : lse(temp, b + a);
      flag = 1;
    }
  }
  for (int i = imax(0, x + w - 1 - W); i < x; i++) {
    double b = T(b_multi2, i, x + w - 1 - i), a = T(a_multi2, i, x - i - 1);
    if (b != NINF && a != NINF) {
      temp = flag == 0 ? b + a : lse(temp, b + a);
      flag = 1;
    }
  }
  if (flag == 1) probability = orc_expd(temp - r->ao[L]);
  return probability;
}

/* raccess.cpp:536-579 */
static void hairpin_prob(const RA *r, double *hp, double *chp) {
  const int L = r->L, W = r->W, w = r->delta;
  const int *s = r->seq;
  for (int x = 1; x + w - 1 <= L; x++) {
    double temp = 0.0, c_temp = 0.0;
    int flag = 0, c_flag = 0;
    for (int i = imax(1, x - W); i < x; i++) {
      for (int j = x + w; j <= imin(i + W, L); j++) {
        int type = BP(s[i], s[j]);
        if (T(b_stemend, i, j - i - 1) != NINF) {
          double h = T(b_stemend, i, j - i - 1) + hairpin_energy(r, type, i, j);
          if (j == x + w) {
            temp = flag == 1 ? lse(temp, h) : h;
            flag = 1;
          } else {
            c_temp = c_flag == 1 ? lse(c_temp, h) : h;
            c_flag = 1;
          }
        }
      }
    }
    if (flag == 1 && c_flag == 1) temp = lse(temp, c_temp);
    if (flag == 0 && c_flag == 1) {
      temp = c_temp;
      flag = 1;
    }
    if (flag == 1) hp[x - 1] = orc_expd(temp - r->ao[L]);
    if (c_flag == 1) chp[x - 1] = orc_expd(c_temp - r->ao[L]);
  }
}

/* raccess.cpp:614-681 (logsum == 0) and :683-771 (logsum == 1) */
static void biloop_prob(const RA *r, double *bp, double *cbp, int logsum) {
  const int L = r->L, W = r->W, w = r->delta;
  const int *s = r->seq;
  const int *rtype = r->P->rtype;
  unsigned char *bf = calloc((size_t)L + 1, 1), *cf = calloc((size_t)L + 1, 1);
  for (int i = 1; i < L - ORC_TURN - 2; i++) {
    for (int j = i + ORC_TURN + 3; j <= imin(i + W, L); j++) {
      int type = BP(s[i], s[j]);
      if (type == 0) continue;
      for (int p = i + 1; p <= imin(i + ORC_MAXLOOP + 1, j - ORC_TURN - 2); p++) {
        int u1 = p - i - 1;
        for (int q = imax(p + ORC_TURN + 1, j - ORC_MAXLOOP + u1 - 1); q < j; q++) {
          int type2 = BP(s[p], s[q]);
          if (type2 != 0 && !(p == i + 1 && q == j - 1)) {
            type2 = rtype[type2];
            if (T(b_stemend, i, j - i - 1) != NINF && T(a_stem, p - 1, q - p + 1) != NINF) {
              double e = T(b_stemend, i, j - i - 1) + loop_energy(r, type, type2, i, j, p, q) +
                         T(a_stem, p - 1, q - p + 1);
              if (!logsum) {
                double t = orc_expd(e);
                for (int k = i + 1; k <= p - w; k++) {
                  if (k == p - w) bp[k - 1] += t; else cbp[k - 1] += t;
                }
                for (int k = q + 1; k <= j - w; k++) {
                  if (k == j - w) bp[k - 1] += t; else cbp[k - 1] += t;
                }
              } else {
                for (int k = i + 1; k <= p - w; k++) {
                  if (k == p - w) { bp[k - 1] = bf[k - 1] ? lse(bp[k - 1], e) : e; bf[k - 1] = 1; }
                  else { cbp[k - 1] = cf[k - 1] ? lse(cbp[k - 1], e) : e; cf[k - 1] = 1; }
                }
                for (int k = q + 1; k <= j - w; k++) {
                  if (k == j - w) { bp[k - 1] = bf[k - 1] ? lse(bp[k - 1], e) : e; bf[k - 1] = 1; }
                  else { cbp[k - 1] = cf[k - 1] ? lse(cbp[k - 1], e) : e; cf[k - 1] = 1; }
                }
              }
            }
          }
        }
      }
    }
  }
  const double Z = r->ao[L];
  if (!logsum) {
    /* :667-680; note the float cast before fmath::log (overflow quirk) */
    for (int i = 0; i < L; i++) {
      if (bp[i] != 0) {
        bp[i] = orc_logf((float)(bp[i] + cbp[i]));
        bp[i] = orc_expd(bp[i] - Z);
      }
      if (cbp[i] != 0) {
        cbp[i] = orc_logf((float)(cbp[i]));
        cbp[i] = orc_expd(cbp[i] - Z);
      }
    }
  } else {
    /* :754-770 */
    for (int i = 0; i < L; i++) {
      if (bf[i] == 1 && cf[i] == 1) bp[i] = lse(bp[i], cbp[i]);
      if (bf[i] == 0 && cf[i] == 1) bp[i] = cbp[i];
      if (bf[i] == 1) bp[i] = orc_expd(bp[i] - Z);
      if (cf[i] == 1) cbp[i] = orc_expd(cbp[i] - Z);
    }
  }
  free(bf);
  free(cf);
}

static double *band(const RA *r) {
  size_t n = (size_t)(r->L + 1) * r->S;
  double *t = malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++) t[i] = NINF;
  return t;
}

int orc_raccess(const char *seq, int L, int W, int delta, float *acc, float *cond, orc_raccess_dbg *dbg) {
  const orc_params *P = orc_params_get();
  if (!P) return -1;
  orc_fmath_init();
  RA *r = calloc(1, sizeof(RA));
  r->P = P;
  r->L = L;
  r->W = W;
  r->delta = delta;
  r->S = W + 2;
  scale_params(r);

  /* Initiallize, raccess.cpp:52-97 */
  r->seq = calloc((size_t)L + 3, sizeof(int));
  for (int i = 0; i < L; i++) {
    char c = seq[i];
    int v = 0;
    if (c == 'A' || c == 'a') v = 1;
    else if (c == 'C' || c == 'c') v = 2;
    else if (c == 'G' || c == 'g') v = 3;
    else if (c == 'T' || c == 't' || c == 'U' || c == 'u') v = 4;
    r->seq[i + 1] = v;
  }
  r->ao = calloc((size_t)L + 1, sizeof(double));
  r->bo = calloc((size_t)L + 1, sizeof(double));
  r->a_stem = band(r); r->a_stemend = band(r); r->a_multi = band(r);
  r->a_multibif = band(r); r->a_multi1 = band(r); r->a_multi2 = band(r);
  r->b_stem = band(r); r->b_stemend = band(r); r->b_multi = band(r);
  r->b_multibif = band(r); r->b_multi1 = band(r); r->b_multi2 = band(r);

  inside(r);
  outside(r);

  if (dbg) {
    size_t n = (size_t)(L + 1) * r->S * sizeof(double);
    memcpy(dbg->alpha_outer, r->ao, (size_t)(L + 1) * sizeof(double));
    memcpy(dbg->beta_outer, r->bo, (size_t)(L + 1) * sizeof(double));
    double *A[6] = {r->a_stem, r->a_stemend, r->a_multi, r->a_multibif, r->a_multi1, r->a_multi2};
    double *B[6] = {r->b_stem, r->b_stemend, r->b_multi, r->b_multibif, r->b_multi1, r->b_multi2};
    for (int k = 0; k < 6; k++) {
      if (dbg->alpha[k]) memcpy(dbg->alpha[k], A[k], n);
      if (dbg->beta[k]) memcpy(dbg->beta[k], B[k], n);
    }
  }

  /* CalcAccessibility(acc, cond), raccess.cpp:484-528 */
  for (int i = 0; i < L; i++) acc[i] = 0.0f, cond[i] = 0.0f;
  double *bp = calloc((size_t)L + 1, sizeof(double)), *cbp = calloc((size_t)L + 1, sizeof(double));
  double *hp = calloc((size_t)L + 1, sizeof(double)), *chp = calloc((size_t)L + 1, sizeof(double));
  double pf = r->ao[L];
  biloop_prob(r, bp, cbp, !(pf >= -690 && pf <= 690));
  hairpin_prob(r, hp, chp);
  const double kT = r->kT;
  double prob = 0.0;
  for (int i = 1; i + delta - 1 <= L; i++) {
    prob += exterior_prob(r, i, delta);
    prob += hp[i - 1];
    prob += bp[i - 1];
    prob += multi_prob(r, i, delta);
    acc[i - 1] = (float)((-orc_logf((float)prob) * kT) / 1000);
    prob = 0.0;
  }
  for (int i = 1; i + delta - 1 < L; i++) {
    prob += exterior_prob(r, i, delta + 1);
    prob += chp[i - 1];
    prob += cbp[i - 1];
    prob += multi_prob(r, i, delta + 1);
    cond[i + delta - 1] = (float)((-orc_logf((float)prob) * kT) / 1000 - acc[i - 1]);
    prob = 0.0;
  }

  free(bp); free(cbp); free(hp); free(chp);
  free(r->a_stem); free(r->a_stemend); free(r->a_multi); free(r->a_multibif); free(r->a_multi1); free(r->a_multi2);
  free(r->b_stem); free(r->b_stemend); free(r->b_multi); free(r->b_multibif); free(r->b_multi1); free(r->b_multi2);
  free(r->ao); free(r->bo); free(r->seq);
  free(r);
  return 0;
}
