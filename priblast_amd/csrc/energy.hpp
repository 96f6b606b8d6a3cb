// Nearest-neighbour energy parameters: raw integer tables (0.01 kcal/mol) loaded from the
// parameter data file, and the Boltzmann-scaled double tables Raccess works in.
//
// Values correspond to the reference's energy_par.hpp:6-174 / intloops.hpp:6,309,1788; the
// scaling to -E*10/kT restates Raccess::set_energy_parameters (raccess.hpp:105-158).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace prb {

constexpr int kInf = 1000000;  // energy_par.hpp:8
constexpr int kTurn = 3;       // energy_par.hpp:9
constexpr int kMaxLoop = 30;   // energy_par.hpp:10

struct EnergyParams {
  int bp_pair[5][5];
  int rtype[7];
  int hairpin37[31], bulge37[31], internal37[31];
  int stack37[7][7];
  int mismatchH37[7][5][5], mismatchI37[7][5][5];
  int dangle5_37[8][5], dangle3_37[8][5];
  int int11_37[8][8][5][5];
  int int21_37[8][8][5][5][5];
  int int22_37[8][8][5][5][5][5];
  int terminal_au = 0, ml_closing = 0, ml_intern = 0, ml_base = 0, max_ninio = 0, f_ninio = 0;
  double lxc37 = 0, kT = 0;
};

// returns "" on success, else an error message
std::string load_energy_params(const std::string &path, EnergyParams &out);

// ---- layout of the scaled (double) Raccess tables as uploaded to the device ----
// "small" block: staged in LDS by every workgroup.
struct RaSmallLayout {
  static constexpr int kStack = 0;                  // [7][7]
  static constexpr int kBulge = kStack + 49;        // [31]
  static constexpr int kInternal = kBulge + 31;     // [31]
  static constexpr int kNinio = kInternal + 31;     // [31]
  static constexpr int kMismatchI = kNinio + 31;    // [7][5][5]
  static constexpr int kMismatchH = kMismatchI + 175;
  static constexpr int kDangle5 = kMismatchH + 175; // [8][5]
  static constexpr int kDangle3 = kDangle5 + 40;
  static constexpr int kTermAU = kDangle3 + 40;
  static constexpr int kMLclosing = kTermAU + 1;
  static constexpr int kMLintern = kMLclosing + 1;
  static constexpr int kMLbase = kMLintern + 1;
  static constexpr int kKT = kMLbase + 1;
  static constexpr int kHairpin = kKT + 1;          // [kHairpinN]: index = loop size d
  static constexpr int kHairpinN = 264;             // d = 0 .. 263 (maximal span up to 262)
  static constexpr int kCount = kHairpin + kHairpinN;
};
// "big" block: int11 / int21 / int22, read through L1/L2.
struct RaBigLayout {
  static constexpr int kInt11 = 0;                 // [8][8][5][5]
  static constexpr int kInt21 = kInt11 + 1600;     // [8][8][5][5][5]
  static constexpr int kInt22 = kInt21 + 8000;     // [8][8][5][5][5][5]
  static constexpr int kCount = kInt22 + 40000;
};

struct RaccessTables {
  std::vector<double> small; // RaSmallLayout::kCount
  std::vector<double> big;   // RaBigLayout::kCount
};
void build_raccess_tables(const EnergyParams &p, RaccessTables &out);

// ---- fmath tables (2048-entry expd mantissa table, 2048 x {app, rev} log table) ----
struct FmathTables {
  std::vector<uint64_t> expd_tbl; // 2048
  std::vector<float> log_tbl;     // 4096, {app, rev} interleaved
  double expd_a = 0, expd_ra = 0;
  float c_log2 = 0;
};
void build_fmath_tables(FmathTables &out);

} // namespace prb
