#include "energy.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace prb {

std::string load_energy_params(const std::string &path, EnergyParams &p) {
  std::ifstream f(path);
  if (!f) return "cannot open parameter file " + path;
  std::memset(&p, 0, sizeof(EnergyParams));
  struct Tab {
    int *dst;
    int n;
  };
  std::map<std::string, Tab> tabs = {
      {"bp_pair", {&p.bp_pair[0][0], 25}},
      {"rtype", {p.rtype, 7}},
      {"hairpin", {p.hairpin37, 31}},
      {"bulge", {p.bulge37, 31}},
      {"internal", {p.internal37, 31}},
      {"stack", {&p.stack37[0][0], 49}},
      {"mismatch_hairpin", {&p.mismatchH37[0][0][0], 175}},
      {"mismatch_interior", {&p.mismatchI37[0][0][0], 175}},
      {"dangle5", {&p.dangle5_37[0][0], 40}},
      {"dangle3", {&p.dangle3_37[0][0], 40}},
      {"int11", {&p.int11_37[0][0][0][0], 1600}},
      {"int21", {&p.int21_37[0][0][0][0][0], 8000}},
      {"int22", {&p.int22_37[0][0][0][0][0][0], 40000}},
  };
  std::map<std::string, double> scal;
  std::string line;
  size_t seen = 0;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line.substr(1));
    std::string name;
    if (line[0] == '$') {
      std::string val;
      ls >> name >> val;
      scal[name] = std::strtod(val.c_str(), nullptr);
    } else if (line[0] == '@') {
      ls >> name;
      auto it = tabs.find(name);
      if (it == tabs.end()) return "unknown table '" + name + "' in " + path;
      for (int i = 0; i < it->second.n; i++) {
        std::string tok;
        if (!(f >> tok)) return "truncated table '" + name + "' in " + path;
        it->second.dst[i] = tok == "INF" ? kInf : std::atoi(tok.c_str());
      }
      seen++;
    }
  }
  if (seen != tabs.size()) return "parameter file " + path + " is missing tables";
  for (const char *k : {"temperature", "gasconst", "k0", "lxc37", "terminal_au", "ml_closing", "ml_intern",
                        "ml_base", "max_ninio", "f_ninio"})
    if (!scal.count(k)) return std::string("parameter file is missing scalar ") + k;
  p.terminal_au = (int)scal["terminal_au"];
  p.ml_closing = (int)scal["ml_closing"];
  p.ml_intern = (int)scal["ml_intern"];
  p.ml_base = (int)scal["ml_base"];
  p.max_ninio = (int)scal["max_ninio"];
  p.f_ninio = (int)scal["f_ninio"];
  p.lxc37 = scal["lxc37"];
  p.kT = (scal["temperature"] + scal["k0"]) * scal["gasconst"]; // energy_par.hpp:12-13
  return "";
}

// Boltzmann exponent of an energy given in 0.01 kcal/mol: -E*10/kT (raccess.hpp:106-157).
static inline double boltz(int e, double kT) { return (double)(-e * 10) / kT; }

void build_raccess_tables(const EnergyParams &p, RaccessTables &t) {
  using S = RaSmallLayout;
  using B = RaBigLayout;
  const double kT = p.kT;
  t.small.assign(S::kCount, 0.0);
  t.big.assign(B::kCount, 0.0);
  double *s = t.small.data();
  const double term_au = boltz(p.terminal_au, kT);
  s[S::kTermAU] = term_au;
  s[S::kMLclosing] = boltz(p.ml_closing, kT);
  s[S::kMLintern] = boltz(p.ml_intern, kT);
  s[S::kMLbase] = boltz(p.ml_base, kT);
  s[S::kKT] = kT;
  for (int i = 0; i <= 30; i++) {
    s[S::kBulge + i] = boltz(p.bulge37[i], kT);
    s[S::kInternal + i] = boltz(p.internal37[i], kT);
    s[S::kNinio + i] = boltz(std::min(p.max_ninio, i * p.f_ninio), kT);
  }
  // hairpin[d]: table up to 30, logarithmic extrapolation beyond (raccess.cpp:823); the
  // extrapolated entries are evaluated here, on the host libm, so the device never calls log().
  for (int d = 0; d < S::kHairpinN; d++)
    s[S::kHairpin + d] = d <= 30 ? boltz(p.hairpin37[d], kT)
                                 : boltz(p.hairpin37[30], kT) - p.lxc37 * std::log(d / 30.) * 10. / kT;
  for (int i = 0; i < 7; i++) {
    for (int j = 0; j < 7; j++) s[S::kStack + i * 7 + j] = boltz(p.stack37[i][j], kT);
    for (int j = 0; j < 5; j++)
      for (int k = 0; k < 5; k++) {
        s[S::kMismatchI + (i * 5 + j) * 5 + k] = boltz(p.mismatchI37[i][j][k], kT);
        s[S::kMismatchH + (i * 5 + j) * 5 + k] = boltz(p.mismatchH37[i][j][k], kT);
      }
    for (int j = 0; j < 5; j++) {
      s[S::kDangle5 + i * 5 + j] = boltz(p.dangle5_37[i][j], kT);
      double d3 = boltz(p.dangle3_37[i][j], kT);
      if (i > 2) d3 += term_au; // raccess.hpp:132-134
      s[S::kDangle3 + i * 5 + j] = d3;
    }
  }
  double *b = t.big.data();
  const int *i11 = &p.int11_37[0][0][0][0];
  const int *i21 = &p.int21_37[0][0][0][0][0];
  const int *i22 = &p.int22_37[0][0][0][0][0][0];
  for (int i = 0; i < 1600; i++) b[B::kInt11 + i] = boltz(i11[i], kT);
  for (int i = 0; i < 8000; i++) b[B::kInt21 + i] = boltz(i21[i], kT);
  for (int i = 0; i < 40000; i++) b[B::kInt22 + i] = boltz(i22[i], kT);
}

// Tables of the table-driven exp/log the reference computes with (herumi fmath; see
// fmath.hpp:148-216 for the constructors these values must agree with):
//   expd: mantissa bits of 2^(i/2048); log: log(1+i/2048) and the local slope per ulp of
//   the 12 low mantissa bits.
void build_fmath_tables(FmathTables &t) {
  constexpr int N = 2048;
  t.expd_tbl.resize(N);
  t.log_tbl.resize(2 * N);
  t.expd_a = (double)N / std::log(2.0);
  t.expd_ra = 1 / t.expd_a;
  for (int i = 0; i < N; i++) {
    double v = std::pow(2.0, i * (1.0 / N));
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    t.expd_tbl[i] = bits & ((1ULL << 52) - 1);
  }
  t.c_log2 = ::logf(2.0f) / (1 << 23);
  const double eps = 1 / double(1 << 24);
  const double step = 1 / double(N);
  for (int i = 0; i < N; i++) {
    double x = 1 + double(i) / N;
    double lx = std::log(x);
    t.log_tbl[2 * i] = (float)lx;
    t.log_tbl[2 * i + 1] = i < N - 1 ? (float)((std::log(x + step - eps) - lx) / ((step - eps) * (1 << 23)))
                                     : (float)(1 / (x * (1 << 23)));
  }
}

} // namespace prb
