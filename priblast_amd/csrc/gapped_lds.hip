// Gapped extension on gfx950 (GappedExtension::Run / extension / CheckHelixLength / traceback /
// CalcDangleEnergy / LoopEnergy, gapped_extension.cpp:33-473).
//
// A GROUP of lanes owns one hit and keeps the whole DP state of one direction on chip:
//   * the list of filled cells - which is also the stem-candidate list: cells are pushed in
//     (anti-diagonal, i) order and pruned from the front (gapped_extension.cpp:213-217), so the
//     live candidates are always a contiguous range [lo, start of this anti-diagonal);
//   * the cumulative accessibility changes eq[], ed[] (gapped_extension.cpp:156-212);
//   * Cell::type (= the PREDECESSOR's stem type, gapped_extension.cpp:256-258) by i for the last
//     three anti-diagonals, for CheckHelixLength's (i-1, j-1) lookup.
// Cells of an anti-diagonal are checked G at a time (one per lane); the candidates of a filled
// cell are scanned G at a time and reduced with an order-preserving (energy, index) minimum,
// i.e. "first candidate in list order wins under strict <"; the scalars of the recurrence are
// kept redundantly in every lane.  The reference's growing 100x100 Cell matrix never exists.
// The groups of a wavefront run all of this in lockstep, so a loop costs what its busiest group
// needs: the kernels are shaped to keep that maximum small (one loop over the filled cells of an
// anti-diagonal - of two anti-diagonals in tier 0, dir_step_pair -, two cells at once on half a group each, one
// instruction stream for both directions) rather than to keep every lane busy.
//
// Two forms, a cascade of five kernels (a hit goes on to the next one when it outgrows the state a
// kernel has room for; the LDS tiers hand their state over, so the next tier continues instead of
// starting again):
//   k_gapped_lds   state in LDS.  Tier 0: 8 lanes per hit, 32 anti-diagonals and 48 cells per
//                  direction (1.25 KB per hit, 128 hits and 4 wavefronts per SIMD on a CU); tier 1:
//                  8 lanes, 40 / 64; tier 2: 16 lanes, 64 / 120; tier 3: a wavefront per hit, 128 / 512
//   k_gapped_wave  G = 64, state in HBM scratch sized at run time: the rest
// Why: an extension is small (median 16 anti-diagonals, ~10 filled cells per direction) but its
// cell list is re-read for every filled cell; per-thread scratch in HBM made every access a
// dependent HBM round trip (r01 baseline profile: 117 ns/hit; LDS form: 19 ns/hit).
#include <algorithm>
#include <type_traits>

#include "gapped_args.hpp"
#include "search_device.hpp"
#include "search_kernels.hpp"

namespace prb {

// Developer-only cycle breakdown of the gapped kernel (make PROF=1 builds libpriblast_hip_prof.so;
// tools/gapped_profile.py reads it).  Not part of the product build.
#ifdef PRB_GAP_PROFILE
constexpr int kProfSlots = 24;
__device__ unsigned long long g_gap_prof[10 * kProfSlots]; // [tier * 2 + (mode != 0)][region]
struct GapProf {
  unsigned long long last, acc[kProfSlots];
  int cells_now = 0, cells_prev = 0; // filled cells of this group's anti-diagonal (this step / the step before)
  __device__ __forceinline__ void start() {
    for (int i = 0; i < kProfSlots; i++) acc[i] = 0;
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int r) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[r] += t - last;
    last = t;
  }
  __device__ __forceinline__ void count(int r) { acc[r] += 1; }
  __device__ __forceinline__ void flush(int kind) {
    if ((threadIdx.x & 63) == 0)
      for (int i = 0; i < kProfSlots; i++) atomicAdd(&g_gap_prof[kind * kProfSlots + i], acc[i]);
  }
};
#define GP_MARK(r) prof.mark(r)
#define GP_COUNT(r) prof.count(r)
#else
struct GapProf {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void flush(int) {}
};
#define GP_MARK(r)
#define GP_COUNT(r)
#endif

namespace {

constexpr int kInitStage = 16; // extension lengths whose accessibility sums dir_init prepares
// LDS tiers: lanes per hit, (anti-diagonals, filled cells) per direction, groups (= hits) per
// workgroup, staged extension lengths (<= lanes per hit)
struct Tier0 { // 1,280 B per hit: 4 workgroups of 256 threads (32 hits) are exactly the 160 KB of a CU = 4 wavefronts per SIMD
               // (32 anti-diagonals instead of 30: 16 % fewer hits go on to tier 1, -40 ms per configs[2] step)
#ifndef PRB_T0_CAPR
#define PRB_T0_CAPR 48
#define PRB_T0_WGCU 4
#endif
#ifndef PRB_T0_ACC_GLOBAL
#define PRB_T0_ACC_GLOBAL false
#endif
  static constexpr int kG = 8, kCapD = 32, kCapR = PRB_T0_CAPR, kGroups = 32, kWavesPerSimd = PRB_T0_WGCU, kWgPerCu = PRB_T0_WGCU;
  static constexpr bool kResumable = true; // a hit that outgrows it is continued, not redone, by the next tier
  static constexpr bool kResumes = false;
  static constexpr bool kPairSteps = true; // two anti-diagonals per step where that is safe (dir_step_pair)
  static constexpr bool kAccGlobal = PRB_T0_ACC_GLOBAL;
};
#ifndef PRB_T12_ACC_GLOBAL
#define PRB_T12_ACC_GLOBAL true
#endif
#ifndef PRB_T12_WPS
#define PRB_T12_WPS (PRB_T12_ACC_GLOBAL ? 4 : 3)
#endif
#ifndef PRB_T1_CAPD // (geometry experiments: tools/tier_geometry.sh)
#define PRB_T1_CAPD 40
#define PRB_T1_CAPR 64
#define PRB_T1_GROUPS 32
#define PRB_T1_WGCU (PRB_T12_ACC_GLOBAL ? 4 : 3)
#endif
struct Tier1 { // 1.65 KB per hit, 3 workgroups of 256 threads (32 hits) per CU: the hits a little too long for tier 0
  static constexpr int kG = 8, kCapD = PRB_T1_CAPD, kCapR = PRB_T1_CAPR, kGroups = PRB_T1_GROUPS, kWavesPerSimd = PRB_T12_WPS, kWgPerCu = PRB_T1_WGCU;
  static constexpr bool kResumable = true, kResumes = true; // continues the state dumps of tier 0, leaves its own
  using From = Tier0;
  using FromRec = Rec32;
  static constexpr bool kPairSteps = false;
  static constexpr bool kAccGlobal = PRB_T12_ACC_GLOBAL; // the cumulative accessibility sums in a block of memory (L2) instead of LDS: see LdsLive
};
struct Tier2 { // 3.2 KB per hit, 3 workgroups of 256 threads (16 hits) per CU
  static constexpr int kG = 16, kCapD = 64, kCapR = 120, kGroups = 16, kWavesPerSimd = PRB_T12_WPS, kWgPerCu = PRB_T12_ACC_GLOBAL ? 4 : 3;
  static constexpr bool kResumable = true, kResumes = true; // continues the state dumps of tier 1, leaves its own
  using From = Tier1;
  using FromRec = Rec32;
  static constexpr bool kPairSteps = false;
  static constexpr bool kAccGlobal = PRB_T12_ACC_GLOBAL;
};
struct Tier3 { // 9.3 KB per hit, one wavefront per hit, 16 single-wave workgroups per CU
  static constexpr int kG = 64, kCapD = 128, kCapR = 512, kGroups = 1, kWavesPerSimd = 4, kWgPerCu = 16;
  static constexpr bool kResumable = false, kResumes = true; // continues the state dumps of tier 2
  using From = Tier2;
  using FromRec = Rec32;
  static constexpr bool kPairSteps = false;
  static constexpr bool kAccGlobal = false;
};

// A filled cell r is (i, j, pred = index of the predecessor cell, type = Stem::type,
// the bases at (i+1, j+1) that a later loop closing on this cell needs); Cell::type lives in the rotating
// rows instead.  The LDS forms pack a record into 32 bits, the HBM form into 64.
struct Rec32W { // tier 3: no room for the neighbour bases, they are read from the staged windows
  using word = uint32_t; // i:8 | j:8 | pred:9 | type:3
  static constexpr bool kBases = false;
  static __device__ __forceinline__ word pack(int i, int j, int pred, int type, int, int) {
    return (word)i | ((word)j << 8) | ((word)pred << 16) | ((word)type << 25);
  }
  static __device__ __forceinline__ int i(word v) { return v & 0xFF; }
  static __device__ __forceinline__ int j(word v) { return (v >> 8) & 0xFF; }
  static __device__ __forceinline__ int pred(word v) { return (v >> 16) & 0x1FF; }
  static __device__ __forceinline__ int type(word v) { return (v >> 25) & 7; }
  static __device__ __forceinline__ int qa(word) { return 0; }
  static __device__ __forceinline__ int da(word) { return 0; }
};
struct Rec64 {
  using word = uint64_t; // i:16 | j:16 | type:4 | qa:4 | da:4 | pred:20
  static constexpr bool kBases = true;
  static __device__ __forceinline__ word pack(int i, int j, int pred, int type, int qa, int da) {
    return (word)(uint32_t)i | ((word)(uint32_t)j << 16) | ((word)type << 32) | ((word)qa << 36) | ((word)da << 40) |
           ((word)(uint32_t)pred << 44);
  }
  static __device__ __forceinline__ int i(word v) { return (int)(v & 0xFFFF); }
  static __device__ __forceinline__ int j(word v) { return (int)((v >> 16) & 0xFFFF); }
  static __device__ __forceinline__ int pred(word v) { return (int)(v >> 44); }
  static __device__ __forceinline__ int type(word v) { return (int)((v >> 32) & 0xF); }
  static __device__ __forceinline__ int qa(word v) { return (int)((v >> 36) & 0xF); }
  static __device__ __forceinline__ int da(word v) { return (int)((v >> 40) & 0xF); }
};
static_assert(Tier3::kCapD + 16 <= 255 && Tier3::kCapR <= 512 && Tier0::kCapR * 8 >= 6 * kInitStage * 4 && Tier2::kCapD + 16 <= 127 && Tier2::kCapR <= 127, "Rec32 field widths");
static_assert(Tier0::kCapD <= Tier1::kCapD && Tier1::kCapD <= Tier2::kCapD && Tier2::kCapD <= Tier3::kCapD && Tier0::kCapR <= Tier1::kCapR &&
                  Tier1::kCapR <= Tier2::kCapR && Tier2::kCapR <= Tier3::kCapR,
              "a tier continues the dumps of the one before it");

template <class T, class Rec> struct LdsState {
  double eq[T::kCapD], ed[T::kCapD];
  double hyb[T::kCapR];
  typename Rec::word info[T::kCapR];
  // Cell::type rows: a cell has i <= kCapD - 1; whole 8-byte words, so a row is reset with one store per lane
  static constexpr int kPtabLen = (T::kCapD + 7) & ~7;
  alignas(8) uint8_t ptab[3][kPtabLen];
  uint8_t qb[T::kCapD + 16], db[T::kCapD + 16]; // bases along the extension, 0 = end of sequence / masked
};
// What of it lives in LDS.  The cumulative accessibility sums - 16 B per anti-diagonal, a third of the state, written once per
// 16 lengths and read twice per filled cell, behind its scan - of tiers 1 and 2 live in a block of memory per resident group
// instead (it stays in L2): 1.0 / 1.8 KB of LDS per hit instead of 1.6 / 2.8, four wavefronts per SIMD instead of three
// (these tiers are bound by latency, not by issue: a third fewer wavefronts cost them a third more time).
template <class T, class Rec, bool kAccGlobal = T::kAccGlobal> struct LdsLive : LdsState<T, Rec> {};
template <class T, class Rec> struct LdsLive<T, Rec, true> {
  double hyb[T::kCapR];
  typename Rec::word info[T::kCapR];
  static constexpr int kPtabLen = (T::kCapD + 7) & ~7;
  alignas(8) uint8_t ptab[3][kPtabLen];
  uint8_t qb[T::kCapD + 16], db[T::kCapD + 16];
};
template <class T, class Rec> struct LdsStore {
  using R = Rec;
  static constexpr bool kResumable = T::kResumable;
  static constexpr bool kAccGlobal = T::kAccGlobal;
  static constexpr int kCapD = T::kCapD;
  LdsLive<T, Rec> &s;
  double *acc; // kAccGlobal: eq[kCapD], ed[kCapD] of this group
  __device__ __forceinline__ int cap_d() const { return T::kCapD; }
  __device__ __forceinline__ int cap_r() const { return T::kCapR; }
  __device__ __forceinline__ double &eq(int i) const {
    if constexpr (T::kAccGlobal) return acc[i];
    else return s.eq[i];
  }
  __device__ __forceinline__ double &ed(int i) const {
    if constexpr (T::kAccGlobal) return acc[T::kCapD + i];
    else return s.ed[i];
  }
  __device__ __forceinline__ double &hyb(int r) const { return s.hyb[r]; }
  __device__ __forceinline__ typename Rec::word &info(int r) const { return s.info[r]; }
  __device__ __forceinline__ uint8_t &ptab(int row, int i) const { return s.ptab[row][i]; }
  __device__ __forceinline__ int ptab_len() const { return LdsState<T, Rec>::kPtabLen; }
  __device__ __forceinline__ int win_len() const { return T::kCapD + 16; }
  __device__ __forceinline__ uint8_t &qb(int t) const { return s.qb[t]; }
  __device__ __forceinline__ uint8_t &db(int t) const { return s.db[t]; }
};
struct HbmStore { // one block of the scratch per group
  using R = Rec64;
  static constexpr bool kResumable = false;
  static constexpr bool kAccGlobal = false;
  double *eq_, *ed_, *hyb_;
  uint64_t *info_;
  uint8_t *ptab_, *qb_, *db_;
  int capd, capr;
  __device__ __forceinline__ int cap_d() const { return capd; }
  __device__ __forceinline__ int cap_r() const { return capr; }
  __device__ __forceinline__ double &eq(int i) const { return eq_[i]; }
  __device__ __forceinline__ double &ed(int i) const { return ed_[i]; }
  __device__ __forceinline__ double &hyb(int r) const { return hyb_[r]; }
  __device__ __forceinline__ uint64_t &info(int r) const { return info_[r]; }
  __device__ __forceinline__ uint8_t &ptab(int row, int i) const { return ptab_[(size_t)row * (capd + 4) + i]; }
  __device__ __forceinline__ int ptab_len() const { return capd + 4; }
  __device__ __forceinline__ int win_len() const { return capd + 16; }
  __device__ __forceinline__ uint8_t &qb(int t) const { return qb_[t]; }
  __device__ __forceinline__ uint8_t &db(int t) const { return db_[t]; }
};

// State traffic inside a group is produced and consumed by lanes of ONE wavefront; its
// memory instructions execute in order, so ordering only needs the compiler held back
// (LDS) plus completion of the stores (HBM scratch: workgroup-scope fence).
template <bool kLds> __device__ __forceinline__ void group_sync() {
  if (kLds) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// behind writes of the accessibility sums when they live in memory (LdsStore::kAccGlobal): the other lanes' reads see them
template <class Store> __device__ __forceinline__ void acc_sync() {
  if constexpr (Store::kAccGlobal) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// (energy, key) minimum over a group of G lanes, every lane ends with the result.  Groups of 8 and
// 16 lanes stay on the VALU (DPP: quad permutes, then half-row / row mirror) - the step is on the
// critical path of every filled cell and an LDS-routed shuffle costs ~100 cycles of latency each.
// Two passes, both on the VALU alone: the minimum energy of the group (v_min_f64 on DPP-permuted
// copies), then the smallest key among the lanes that hold it (v_min_i32).  The same winner as a
// lexicographic (energy, key) comparison, in a third of the instructions.
template <int kCtrl> __device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), kCtrl, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), kCtrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int kCtrl> __device__ __forceinline__ int dpp_i32(int x) { return __builtin_amdgcn_update_dpp(0, x, kCtrl, 0xF, 0xF, false); }
// The same for an eight-lane group that may be working as two halves of four (two cells at once):
// `halves` stops the reduction at the quads.
__device__ __forceinline__ void group_min_halves(double &te, int &key, bool halves) {
  double m = te;
  m = __builtin_fmin(m, dpp_f64<0xB1>(m)); // quad_perm [1,0,3,2]
  m = __builtin_fmin(m, dpp_f64<0x4E>(m)); // quad_perm [2,3,0,1]
  const double m8 = __builtin_fmin(m, dpp_f64<0x141>(m)); // row_half_mirror: lane i <-> 7 - i
  m = halves ? m : m8;
  int k = te == m ? key : 0x7fffffff;
  k = min(k, dpp_i32<0xB1>(k));
  k = min(k, dpp_i32<0x4E>(k));
  const int k8 = min(k, dpp_i32<0x141>(k));
  te = m;
  key = halves ? k : k8;
}
template <int G> __device__ __forceinline__ void group_min(double &te, int &key) {
  if (G == 8 || G == 16) {
    double m = te;
    m = __builtin_fmin(m, dpp_f64<0xB1>(m));  // quad_perm [1,0,3,2]
    m = __builtin_fmin(m, dpp_f64<0x4E>(m));  // quad_perm [2,3,0,1]
    m = __builtin_fmin(m, dpp_f64<0x141>(m)); // row_half_mirror: lane i <-> 7 - i
    if (G == 16) m = __builtin_fmin(m, dpp_f64<0x140>(m)); // row_mirror: lane i <-> 15 - i
    int k = te == m ? key : 0x7fffffff;
    k = min(k, dpp_i32<0xB1>(k));
    k = min(k, dpp_i32<0x4E>(k));
    k = min(k, dpp_i32<0x141>(k));
    if (G == 16) k = min(k, dpp_i32<0x140>(k));
    te = m;
    key = k;
  } else if (G == 64) {
    // rows of 16 lanes on the VALU (DPP), then the four rows through two exchanges: a third of the LDS-routed shuffles
    group_min<16>(te, key);
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
      const double ote = __shfl_xor(te, m);
      const int ok = __shfl_xor(key, m);
      if (ote < te || (ote == te && ok < key)) {
        te = ote;
        key = ok;
      }
    }
  } else {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) {
      const double ote = __shfl_xor(te, m);
      const int ok = __shfl_xor(key, m);
      if (ote < te || (ote == te && ok < key)) {
        te = ote;
        key = ok;
      }
    }
  }
}

struct DirResult {
  bool overflow;
  int best; // cell index of the arg-min, 0 = nothing found
};

// What a group knows about the hit it is extending.
// The arrays a hit's sequences live in (wave-uniform); a hit only carries its offsets into them.
struct SeqBases {
  const uint8_t *qenc;
  const float *qacc, *qcond, *dacc, *dcond;
};
struct HitCtx {
  int64_t x, out0;
  int64_t qo, dbase; // query: offset of its codes / accessibilities; database sequence: offset of its accessibilities
  HitState h;
  int query, id, qn;
  int diag_q, diag_d, diag_len, ndiag, nleft, nright;
  bool unsorted, ovf;
  bool resumed; // direction 0 was completed by an earlier kernel of the cascade: start at direction 1
  int tier0;    // ... and the LDS tier that has room for it
};
// The scalars of one direction's recurrence (kept redundantly in every lane of the group).  Only
// what cannot be had from the hit (HitCtx::h stays as the direction found it until dir_finish) and
// from the cell of the minimum: registers decide how many hits a compute unit extends at a time.
struct DirState {
  double min_e;
  double acc_prev; // lane 0: eq[length-2]; lane 1 (or 0 when G == 1): ed[length-2]
  int min_ci, min_cj; // the cell of the minimum (0, 0: the start)
  int length, best, nrec, lo, tq0, td0;
  int staged; // eq[] / ed[] are computed for the extension lengths 1..staged
  // set when a resumable kernel runs out of cells in the middle of an anti-diagonal: the chunk to
  // go on with, and the end of the candidate window of that anti-diagonal (0: not in the middle)
  int resume_i0, resume_dstart;
  bool overflow;
};

// Cumulative accessibility change of the extension (gapped_extension.cpp:156-212) for the nb
// lengths from L0 on: the terms are fetched in parallel into the float scratch sf(term, t),
// the sums are sequential (lane 0: query side -> eq[], lane 1: db side -> ed[]).
template <int G, bool kLds, class Store>
__device__ __forceinline__ void stage_acc(const SeqBases &sb, const HitCtx &c, int flag, int delta, const Store &S, int gl, DirState &d, int L0,
                                          int nb, float *scratch /* [6][nb], or nullptr: no parallel fetch */) {
  const float *qacc = sb.qacc + c.qo, *qcond = sb.qcond + c.qo, *dacc = sb.dacc + c.dbase, *dcond = sb.dcond + c.dbase;
  const DirOrigin og = dir_origin(c.h, flag);
  // term k of length len: query side k = 0..2, db side k = 3..5
  auto term = [&](int k, int len) -> float {
    if (flag == 0) {
      const int p = og.q_start - len;
      return k == 0 ? qacc[p] : k == 1 ? qacc[p + 1] : k == 2 ? qcond[p + delta] : dcond[og.id_end + len];
    }
    const int p = og.id_start - len;
    return k == 0 ? qcond[og.q_start + len] : k == 3 ? dacc[p] : k == 4 ? dacc[p + 1] : dcond[p + delta];
  };
  if (scratch) {
    for (int t = gl; t < nb; t += G) {
      const int len = L0 + t;
      if (len < d.tq0) {
        scratch[0 * nb + t] = term(0, len);
        if (flag == 0) {
          scratch[1 * nb + t] = term(1, len);
          scratch[2 * nb + t] = term(2, len);
        }
      }
      if (len < d.td0) {
        scratch[3 * nb + t] = term(3, len);
        if (flag == 1) {
          scratch[4 * nb + t] = term(4, len);
          scratch[5 * nb + t] = term(5, len);
        }
      }
    }
    group_sync<kLds>();
  }
  auto get = [&](int k, int t) -> float { return scratch ? scratch[k * nb + t] : term(k, L0 + t); };
  // The two chains run side by side on lanes 0 and 1, in one instruction stream whatever the direction
  // (the groups of a wavefront are in different directions most of the time): the side the extension
  // runs along the accessible region of has three terms per length, the other one.
  if (gl < 2) {
    const bool qside = gl == 0, three = qside == (flag == 0);
    const int kb = qside ? 0 : 3, lim = qside ? d.tq0 : d.td0;
    double *out = qside ? &S.eq(0) : &S.ed(0);
    for (int k = 0; k < nb && L0 + k < lim && L0 + k <= S.cap_d(); k++) {
      const int len = L0 + k;
      const float x = get(kb, k), y = get(kb + 1, k), z = get(kb + 2, k); // (y, z: whatever the scratch holds on the one-term side)
      double v;
      if (len == 1) v = three ? (double)(x - y + z) : (double)x; // float arithmetic, as the reference
      else v = three ? d.acc_prev + x - y + z : d.acc_prev + x;
      d.acc_prev = v;
      out[len - 1] = v;
    }
  }
  d.staged = L0 + nb - 1 < S.cap_d() ? L0 + nb - 1 : S.cap_d();
  group_sync<kLds>();
  acc_sync<Store>();
}

// The same for the lengths beyond what dir_init prepared (one direction in ten gets here), when the
// cell list is in use and cannot serve as scratch: every lane fetches the terms of its lengths into
// registers (all loads in flight together), then the group walks the 16 lengths in order, the
// terms coming from their lanes by shuffle; every lane carries both running sums.
template <int G, bool kLds, class Store>
__device__ __forceinline__ void stage_acc_regs(const SeqBases &sb, const HitCtx &c, int flag, int delta, const Store &S, int gl, int gbase,
                                               DirState &d, int L0) {
  constexpr int nb = kInitStage, per = (nb + G - 1) / G;
  const float *qacc = sb.qacc + c.qo, *qcond = sb.qcond + c.qo, *dacc = sb.dacc + c.dbase, *dcond = sb.dcond + c.dbase;
  const DirOrigin og = dir_origin(c.h, flag);
  float t0[per], t1[per], t2[per], t3[per]; // flag 0: query acc, acc+1, cond | db cond;  flag 1: query cond | db acc, acc+1, cond
#pragma unroll
  for (int r = 0; r < per; r++) {
    t0[r] = t1[r] = t2[r] = t3[r] = 0.0f;
    const int t = gl + r * G, len = L0 + t;
    if (t < nb) {
      if (len < d.tq0) {
        if (flag == 0) {
          const int p = og.q_start - len;
          t0[r] = qacc[p];
          t1[r] = qacc[p + 1];
          t2[r] = qcond[p + delta];
        } else {
          t0[r] = qcond[og.q_start + len];
        }
      }
      if (len < d.td0) {
        if (flag == 0) {
          t3[r] = dcond[og.id_end + len];
        } else {
          const int p = og.id_start - len;
          t1[r] = dacc[p];
          t2[r] = dacc[p + 1];
          t3[r] = dcond[p + delta];
        }
      }
    }
  }
  double veq = __shfl(d.acc_prev, gbase), ved = __shfl(d.acc_prev, gbase + (G > 1 ? 1 : 0));
#pragma unroll
  for (int k = 0; k < nb; k++) {
    const int src = gbase + (k % G), r = k / G, len = L0 + k;
    const float a0 = __shfl(t0[r], src), a1 = __shfl(t1[r], src), a2 = __shfl(t2[r], src), a3 = __shfl(t3[r], src);
    if (len <= S.cap_d()) {
      if (len < d.tq0) {
        double v;
        if (flag == 0) v = len == 1 ? (double)(a0 - a1 + a2) : veq + a0 - a1 + a2; // float arithmetic at len 1, as the reference
        else v = len == 1 ? (double)a0 : veq + a0;
        veq = v;
        if (gl == 0) S.eq(len - 1) = v;
      }
      if (len < d.td0) {
        double v;
        if (flag == 0) v = len == 1 ? (double)a3 : ved + a3;
        else v = len == 1 ? (double)(a1 - a2 + a3) : ved + a1 - a2 + a3;
        ved = v;
        if (gl == 0) S.ed(len - 1) = v;
      }
    }
  }
  d.acc_prev = (G > 1 && gl == 1) ? ved : veq;
  d.staged = L0 + nb - 1 < S.cap_d() ? L0 + nb - 1 : S.cap_d();
  group_sync<kLds>();
  acc_sync<Store>();
}

// Bases along the extension: window[t] = GetChar(seq, start -/+ t) (gapped_extension.cpp:401-407); the
// first 0 at t >= 1 is where the reference sets max_q_extension / max_db_extension (:131-154).
// Also clears the three predecessor-type rows.
template <int G, bool kLds, class Store>
__device__ __forceinline__ void stage_windows(const SeqBases &sb, const HitCtx &c, int flag, const uint8_t *ds, int64_t dn, const Store &S, int gl,
                                              DirState &d, bool clear_rows = true) {
  const uint8_t *qs = sb.qenc + c.qo;
  const int qn = c.qn;
  const int wn = S.win_len();
  const DirOrigin og = dir_origin(c.h, flag);
  d.tq0 = wn;
  d.td0 = wn;
  for (int t = gl; t < wn; t += G) {
    const int64_t qp = flag == 0 ? (int64_t)og.q_start - t : (int64_t)og.q_start + t;
    const int64_t dp = flag == 0 ? og.db_start - t : og.db_start + t;
    const int qc = (qp >= 0 && qp < qn) ? get_char(qs, qp) : 0;
    const int dc = (dp >= 0 && dp < dn) ? get_char(ds, dp) : 0;
    S.qb(t) = (uint8_t)qc;
    S.db(t) = (uint8_t)dc;
    if (t >= 1 && qc == 0 && t < d.tq0) d.tq0 = t;
    if (t >= 1 && dc == 0 && t < d.td0) d.td0 = t;
  }
  if (clear_rows) {
    uint32_t *pz = reinterpret_cast<uint32_t *>(&S.ptab(0, 0)); // 3 rows, each a multiple of 4 bytes
    for (int t = gl; t < 3 * S.ptab_len() / 4; t += G) pz[t] = 0;
  }
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) {
    const int ta = __shfl_xor(d.tq0, m), tb = __shfl_xor(d.td0, m);
    d.tq0 = ta < d.tq0 ? ta : d.tq0;
    d.td0 = tb < d.td0 ? tb : d.td0;
  }
  group_sync<kLds>();
}

// GappedExtension::extension (gapped_extension.cpp:71-319) for one direction, by a group of G lanes,
// in three pieces so that the groups of a wavefront can be at different points of different
// hits: dir_init (state + everything the anti-diagonal loop touches staged on chip: the bases
// along both strands, so GetBPType / CheckHelixLength / LoopEnergy never go back to HBM),
// dir_step (one anti-diagonal; true when the direction is finished) and dir_finish.
template <int G, bool kLds, class Store>
__device__ __forceinline__ void dir_init(const SearchConst &sc, const SeqBases &sb, const HitCtx &c, int flag, const uint8_t *ds, int64_t dn,
                                         const Store &S, int gl, int delta, DirState &d) {
  using R = typename Store::R;
  d.min_e = c.h.e_tot;
  d.min_ci = 0;
  d.min_cj = 0;
  d.length = 0;
  d.best = 0;
  d.overflow = false;
  d.resume_i0 = 0;
  d.resume_dstart = 0;

  stage_windows<G, kLds>(sb, c, flag, ds, dn, S, gl, d);
  // accessibility sums of the first kInitStage lengths (nearly every direction ends within them);
  // the cell list is still empty, so its energies' storage serves as the float scratch
  d.acc_prev = 0;
  {
    stage_acc<G, kLds>(sb, c, flag, delta, S, gl, d, 1, kInitStage, reinterpret_cast<float *>(&S.hyb(0)));
  }
  int type0 = bp_type(sc, S.qb(0), S.db(0));
  if (flag == 0) type0 = rtype_of(type0);
  if (gl == 0) {
    S.hyb(0) = d.min_e;
    S.info(0) = R::pack(0, 0, 0, type0, S.qb(1), S.db(1));
    S.ptab(0, 0) = (uint8_t)type0; // cell (0,0) lies on anti-diagonal 0
  }
  group_sync<kLds>();
  d.nrec = 1;
  d.lo = 0;
}

template <int G, bool kLds, class Store>
__device__ __forceinline__ bool dir_step(const SearchConst &sc, const SeqBases &sb, const ExtOpts &o, const HitCtx &c, int flag, const Store &S,
                                         int gl /* lane in group */, int gbase /* first lane of the group in its wavefront */,
                                         DirState &d, GapProf &prof) {
  using R = typename Store::R;
  const int delta = o.delta, drop = o.drop_w_gap, min_helix = o.min_helix;
  const bool mid = d.resume_i0 != 0; // continuing an anti-diagonal that a smaller kernel could not finish
  if (!mid) {
    if (d.length >= S.cap_d()) { // no room for another anti-diagonal: the state stays that of the last complete one
      d.overflow = true;
      return true;
    }
    d.length++;
  }
  // max_q_extension / max_db_extension as the reference has them after its checks at this d.length:
  // unbounded while the strand goes on, else the last position with a base
  const bool q_open = d.length < d.tq0, d_open = d.length < d.td0;
  // d.length mod 3 (multiply-shift while the lengths are small) and (d.length - 2) mod 3
  const int cur = kLds ? d.length - 3 * ((d.length * 171) >> 9) : d.length % 3, d2 = cur == 2 ? 0 : cur + 1;
  static_assert(!kLds || Tier3::kCapD + 16 < 256, "multiply-shift division by 3");
  if (!mid) {
    // cumulative accessibility change beyond the lengths prepared so far (one direction in ten gets here)
    if (d.length > d.staged) stage_acc_regs<G, kLds>(sb, c, flag, delta, S, gl, gbase, d, d.length);
    GP_MARK(1);
    GP_COUNT(10);
    // recycle the row of anti-diagonal d.length-3 for this one
    if constexpr (kLds) {
      uint64_t *row = reinterpret_cast<uint64_t *>(&S.ptab(cur, 0));
      for (int t = gl; t < S.ptab_len() / 8; t += G) row[t] = 0;
    } else {
      for (int t = gl; t <= d.length; t += G) S.ptab(cur, t) = 0;
    }
    group_sync<kLds>();
    // prune candidates with d.length - first - second - 2 > drop (:213-217): a prefix of the list
    if (d.length - 2 > drop) {
      while (d.lo < d.nrec) {
        const auto v = S.info(d.lo);
        if (d.length - R::i(v) - R::j(v) - 2 > drop) d.lo++;
        else break;
      }
    }
  }
  GP_MARK(2);
  const int dstart = mid ? d.resume_dstart : d.nrec;
  // cells 1 <= i <= d.length - 1 with i <= max_q and d.length - i <= max_d
  const int i_first = d.length - d.td0 + 1 > 1 ? d.length - d.td0 + 1 : 1;
  const int i_lo = mid ? d.resume_i0 : i_first;
  const int i_hi = (d.length < d.tq0 ? d.length : d.tq0) - 1;
  d.resume_i0 = 0;

  // the first G live candidates, for the first scan round of every filled cell of this anti-diagonal
  typename R::word v_first = 0;
  double h_first = 0;
  if (d.lo + gl < dstart) {
    v_first = S.info(d.lo + gl);
    h_first = S.hyb(d.lo + gl);
  }

  // One filled cell (ci, d.length - ci) of pair type ctype: its best predecessor among the live
  // candidates [d.lo, dstart), the new record, the running minimum.  Returns the predecessor's type
  // (what the reference stores as Cell::type), or -1 when the cell list is full.
  // `two` (eight-lane groups only): the group handles TWO cells of the anti-diagonal at once, lanes 0-3
  // the first (ci, ctype as those lanes see them), lanes 4-7 the next one: the eight groups of a
  // wavefront run this in lockstep, so what counts is the number of iterations of the busiest group.
  // Cells of one anti-diagonal never see each other, their records go into the list in cell order, and
  // the running minimum is updated first cell first (strict '<'), as if they had come one by one.
  constexpr bool kTwoCells = G == 8 && kLds;
  auto fill_cell = [&](int ci, int ctype, bool two) -> int {
    GP_COUNT(12);
    const bool pair_mode = kTwoCells && two, hi = pair_mode && gl >= G / 2;
    const int sub = pair_mode ? (gl & (G / 2 - 1)) : gl, stride = pair_mode ? G / 2 : G;
    const int cj = d.length - ci;
    // scan the live candidates [d.lo, dstart), `stride` per round; strict '<' keeps the first
    const int nq = S.qb(ci - 1), nd = S.db(cj - 1); // the bases next to the new pair on the loop side
    const double eq_c = S.eq(ci - 1), ed_c = S.ed(cj - 1); // needed after the scan; fetched behind it
    const int fq = S.qb(ci + 1), fd = S.db(cj + 1);        // likewise: the new record's far-side bases
    double bte = 1000000.0;                                // INF
    int bkp = d.lo << 3; // candidate index << 3 | its type (0: none looked at), so the type comes out of the reduction
    // (the records of the next round are fetched while this round's energies are looked up; those of
    // the first round are the same for every cell of the anti-diagonal and were fetched at its start -
    // with two cells, lanes 4-7 take theirs from lanes 0-3)
    typename R::word vn = v_first;
    double hn = h_first;
    if constexpr (kTwoCells) {
      if (pair_mode) {
        const typename R::word v4 = (typename R::word)dpp_i32<0x114>((int)v_first); // row_shr:4
        const double h4 = dpp_f64<0x114>(h_first);
        vn = hi ? v4 : v_first;
        hn = hi ? h4 : h_first;
        if (sub + d.lo >= dstart) vn = 0; // (what the owner of this slot would hold)
      }
    }
    for (int k0 = d.lo; k0 < dstart; k0 += stride) {
      GP_COUNT(13);
      const int k = k0 + sub;
      const auto v = vn;
      const double hk = hn;
      if (k + stride < dstart) {
        vn = S.info(k + stride);
        hn = S.hyb(k + stride);
      }
      if (k < dstart) {
        const int ri = R::i(v), rj = R::j(v);
        if (ri < ci && rj < cj) {
          // LoopEnergy between the new pair and candidate k (:230-247); the loop lies between
          // offsets (ri, rj) and (ci, cj) from the start
          const int rq = R::kBases ? R::qa(v) : (int)S.qb(ri + 1), rd = R::kBases ? R::da(v) : (int)S.db(rj + 1);
          // (the groups of a wavefront are in different directions most of the time: the outer and the
          // inner pair are put in place with selects, so there is one copy of the look-up, not one per direction)
          const bool f0 = flag == 0;
          const int rt = R::type(v);
          double te = loop_energy_abcd(sc, f0 ? ctype : rt, f0 ? rt : ctype, ci - ri - 1, cj - rj - 1, f0 ? nq : rq, f0 ? nd : rd,
                                       f0 ? rq : nq, f0 ? rd : nd);
          te += hk;
          if (te < bte) {
            bte = te;
            bkp = (k << 3) | R::type(v);
          }
        }
      }
    }
    GP_MARK(4);
    if constexpr (kTwoCells) group_min_halves(bte, bkp, pair_mode);
    else group_min<G>(bte, bkp); // "first candidate in list order wins under strict <"
    GP_MARK(5);
    if (d.nrec >= S.cap_r()) return -1; // (never with two cells: the caller made sure of room for both)
    int bk = bkp >> 3, ptype = bkp & 7;
    if (d.lo >= dstart) bk = 0;                  // empty window: the reference reads stem_candidate[0] of an empty list
    if (ptype == 0) ptype = R::type(S.info(bk)); // no candidate qualified: the type of that default entry
    const int rec = d.nrec + (hi ? 1 : 0);
    if (sub == 0) {
      S.hyb(rec) = bte;
      S.info(rec) = R::pack(ci, cj, bk, rtype_of(ctype), fq, fd);
      S.ptab(cur, ci) = (uint8_t)ptype;
    }
    const double ie = eq_c + ed_c + bte;
    if constexpr (kTwoCells) {
      // the first cell (lanes 0-3), then the second (lanes 4-7): each half gets the other's values
      // from its mirror lane
      const double ie_o = dpp_f64<0x141>(ie);
      const int ci_o = dpp_i32<0x141>(ci);
      const double ie_a = hi ? ie_o : ie, ie_b = hi ? ie : ie_o;
      const int ci_a = hi ? ci_o : ci, ci_b = hi ? ci : ci_o;
      if (ie_a < d.min_e) { // (:260-278; everything else the reference notes down here follows from the cell)
        d.min_e = ie_a;
        d.best = d.nrec;
        d.min_ci = ci_a;
        d.min_cj = d.length - ci_a;
      }
      if (pair_mode && ie_b < d.min_e) {
        d.min_e = ie_b;
        d.best = d.nrec + 1;
        d.min_ci = ci_b;
        d.min_cj = d.length - ci_b;
      }
      d.nrec += pair_mode ? 2 : 1;
    } else {
      if (ie < d.min_e) { // (:260-278; everything else the reference notes down here follows from the cell)
        d.min_e = ie;
        d.best = d.nrec;
        d.min_ci = ci;
        d.min_cj = cj;
      }
      d.nrec++;
    }
    GP_MARK(6);
    return ptype;
  };

  // A wavefront per hit (tier 3, the wavefront-per-hit kernel): up to FOUR filled cells of the chunk at once, 16 lanes each -
  // the cells of an anti-diagonal never see each other (fill_cell) -: each row of 16 scans the live candidates for its cell
  // and reduces on the VALU (DPP); records and the running minimum then take the cells in their order.  A hit this long has
  // ~5 cells per anti-diagonal and ~100 live candidates: the scan rounds are the same in number, the per-cell reduction over 64
  // lanes (LDS-routed shuffles) and the per-cell bookkeeping are shared by four.
  // `cellbits`: the filled cells of the chunk still to do (bit b: cell i0 + b), n = 2..4 of them are taken.
  auto fill_quad = [&](int i0, unsigned long long cellbits, int n, unsigned long long tb0, unsigned long long tb1, unsigned long long tb2) {
    GP_COUNT(12);
    const int sub = gl >> 4, sl = gl & 15;
    // the sub-th set bit of cellbits
    unsigned long long mbits = cellbits;
    for (int t = 0; t < sub; t++) mbits &= mbits - 1;
    const bool act = sub < n;
    const int b = act ? __builtin_ctzll(mbits) : __builtin_ctzll(cellbits);
    const int ci = i0 + b, cj = d.length - ci;
    const int ctype = (int)((tb0 >> (gbase + b)) & 1) | (int)(((tb1 >> (gbase + b)) & 1) << 1) | (int)(((tb2 >> (gbase + b)) & 1) << 2);
    const int nq = S.qb(ci - 1), nd = S.db(cj - 1);
    const double eq_c = S.eq(ci - 1), ed_c = S.ed(cj - 1);
    const int fq = S.qb(ci + 1), fd = S.db(cj + 1);
    double bte = 1000000.0; // INF
    int bkp = d.lo << 3;
    for (int k0 = d.lo; k0 < dstart; k0 += 16) {
      GP_COUNT(13);
      const int k = k0 + sl;
      if (k < dstart) {
        const auto v = S.info(k);
        const double hk = S.hyb(k);
        const int ri = R::i(v), rj = R::j(v);
        if (ri < ci && rj < cj) {
          const int rq = R::kBases ? R::qa(v) : (int)S.qb(ri + 1), rd = R::kBases ? R::da(v) : (int)S.db(rj + 1);
          const bool f0 = flag == 0;
          const int rt = R::type(v);
          double te = loop_energy_abcd(sc, f0 ? ctype : rt, f0 ? rt : ctype, ci - ri - 1, cj - rj - 1, f0 ? nq : rq, f0 ? nd : rd,
                                       f0 ? rq : nq, f0 ? rd : nd);
          te += hk;
          if (te < bte) {
            bte = te;
            bkp = (k << 3) | R::type(v);
          }
        }
      }
    }
    GP_MARK(4);
    group_min<16>(bte, bkp);
    GP_MARK(5);
    int bk = bkp >> 3, ptype = bkp & 7;
    if (d.lo >= dstart) bk = 0;
    if (ptype == 0) ptype = R::type(S.info(bk));
    if (act && sl == 0) {
      const int rec = d.nrec + sub;
      S.hyb(rec) = bte;
      S.info(rec) = R::pack(ci, cj, bk, rtype_of(ctype), fq, fd);
      S.ptab(cur, ci) = (uint8_t)ptype;
    }
    const double ie = eq_c + ed_c + bte;
#pragma unroll
    for (int t = 0; t < 4; t++) { // the running minimum, cell by cell (:260-278)
      if (t < n) {
        const int l16 = 16 * t;
        const double ie_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ie), l16), __builtin_amdgcn_readlane(__double2loint(ie), l16));
        const int ci_t = __builtin_amdgcn_readlane(ci, l16);
        if (ie_t < d.min_e) {
          d.min_e = ie_t;
          d.best = d.nrec + t;
          d.min_ci = ci_t;
          d.min_cj = d.length - ci_t;
        }
      }
    }
    d.nrec += n;
    GP_MARK(6);
  };

  // CheckHelixLength (:342-364) with GetBPType (:321-338) for cell (i, d.length - i) on the staged
  // bases, without branches (every lane of the wavefront walks through it anyway): pairs and wobble
  // pairs are bits of two 25-bit masks indexed by 5 * query base + database base; the pair type is
  // looked up at the end.  The cell needs min_helix - 1 further pairs ahead, the first of them not a
  // second wobble, when its diagonal predecessor holds no type or a wobble one under a wobble pair.
  auto cell_type = [&](int i) -> int {
    if (i > i_hi) return 0;
    const int j = d.length - i;
    const unsigned q0 = S.qb(i), d0 = S.db(j), q1 = S.qb(i + 1), d1 = S.db(j + 1), q2 = S.qb(i + 2), d2b = S.db(j + 2);
    const int pt = S.ptab(d2, i - 1);
    const unsigned x0 = q0 * 5 + d0, x1 = q1 * 5 + d1, x2 = q2 * 5 + d2b;
    const unsigned p0 = (sc.pair_mask >> x0) & 1, w0 = (sc.wobble_mask >> x0) & 1;
    unsigned ahead = 1;
    if (min_helix >= 2) ahead = ((sc.pair_mask >> x1) & 1) & ~(w0 & (sc.wobble_mask >> x1));
    if (min_helix >= 3) ahead &= sc.pair_mask >> x2;
    for (int x = 3; x <= min_helix - 1; x++) ahead &= sc.pair_mask >> ((unsigned)S.qb(i + x) * 5 + S.db(j + x));
    const unsigned need = (unsigned)(pt == 0) | (w0 & (unsigned)wobble(pt));
    int t = (int)((sc.bp_rows >> ((x0 * 3 - 15) & 63)) & 7); // row q0 - 1, column d0 (meaningless without a pair)
    if (flag == 1) t = ((t - 1) ^ 1) + 1;                     // rtype
    return (p0 & (ahead | ~need) & 1) ? t : 0;
  };

  if constexpr (G == 8 && kLds) {
    // Eight-lane groups: all chunks of the anti-diagonal are checked first and its filled cells handled
    // in ONE loop, so a wavefront runs as many fill iterations as its busiest group has cells on the
    // anti-diagonal - not, chunk after chunk, as many as the busiest group of that chunk.  The types of
    // a chunk's cells are packed into one word (3 bits per cell, an OR butterfly over DPP): every lane
    // of the group reads any cell's type without going through LDS.
    constexpr int kChunks = (Store::kCapD + G - 1) / G;
    using cells_t = typename std::conditional<kChunks <= 4, uint32_t, unsigned long long>::type;
    uint32_t tw[kChunks];
    cells_t cells = 0;
#pragma unroll
    for (int c = 0; c < kChunks; c++) {
      tw[c] = 0;
      const int i0 = i_lo + c * G;
      if (i0 <= i_hi) {
        const int type1 = cell_type(i0 + gl);
        uint32_t w = (uint32_t)type1 << (3 * gl);
        w |= (uint32_t)dpp_i32<0xB1>((int)w);
        w |= (uint32_t)dpp_i32<0x4E>((int)w);
        w |= (uint32_t)dpp_i32<0x141>((int)w);
        tw[c] = w;
        cells |= (cells_t)((__ballot(type1 != 0) >> gbase) & 0xFF) << (8 * c);
        GP_COUNT(11);
      }
    }
    GP_MARK(3);
#ifdef PRB_GAP_PROFILE
    prof.cells_now = kChunks <= 4 ? __popc((uint32_t)cells) : __popcll((unsigned long long)cells);
#endif
    while (cells) { // filled cells of the anti-diagonal, ascending i, two at a time where there are two
      const cells_t rest = cells & (cells - 1);
      const bool two = rest != 0 && d.nrec + 2 <= S.cap_r();
      const int b0 = kChunks <= 4 ? __builtin_ctz((uint32_t)cells) : __builtin_ctzll((unsigned long long)cells);
      if (Store::kResumable && d.nrec >= S.cap_r()) { // out of cells: stop in front of this one
        d.overflow = true;
        d.resume_i0 = i_lo + b0;
        d.resume_dstart = dstart;
        break;
      }
      int b = b0;
      if (two) {
        const int b1 = kChunks <= 4 ? __builtin_ctz((uint32_t)rest) : __builtin_ctzll((unsigned long long)rest);
        if (gl >= G / 2) b = b1;
        cells = rest & (rest - 1);
      } else {
        cells = rest;
      }
      uint32_t w = tw[0];
#pragma unroll
      for (int c = 1; c < kChunks; c++) w = (b >> 3) == c ? tw[c] : w;
      const int ptype = fill_cell(i_lo + b, (int)((w >> (3 * (b & 7))) & 7), two);
      if (ptype < 0) {
        d.overflow = true;
        break;
      }
    }
    group_sync<kLds>();
    GP_MARK(3);
  } else {
    for (int i0 = i_lo; i0 <= i_hi && !d.overflow; i0 += G) {
      const int type1 = cell_type(i0 + gl);
      // which cells of the chunk are filled (the group's share of a ballot), and their types as three
      // bit planes: a lane reads any cell's type from them without going through LDS
      using mask_t = typename std::conditional<G <= 32, uint32_t, unsigned long long>::type;
      const unsigned long long tb0 = __ballot(type1 & 1), tb1 = __ballot(type1 & 2), tb2 = __ballot(type1 & 4);
      const unsigned long long anyb = tb0 | tb1 | tb2;
      mask_t vmask = G < 64 ? (mask_t)((anyb >> gbase) & ((1ull << (G & 63)) - 1)) : (mask_t)anyb;
      GP_MARK(3);
      GP_COUNT(11);
      if (Store::kResumable && d.nrec + __popcll((unsigned long long)vmask) > S.cap_r()) { // out of cells: stop in front of this chunk
        d.overflow = true;
        d.resume_i0 = i0;
        d.resume_dstart = dstart;
        break;
      }
      while (vmask) { // filled cells of this chunk, ascending i
        if constexpr (G == 64) {
          const int left = __popcll((unsigned long long)vmask), n = left < 4 ? left : 4;
          if (n >= 2 && d.nrec + n <= S.cap_r()) {
            fill_quad(i0, (unsigned long long)vmask, n, tb0, tb1, tb2);
            for (int t = 0; t < n; t++) vmask &= vmask - 1;
            continue;
          }
        }
        const int b = G <= 32 ? __builtin_ctz((uint32_t)vmask) : __builtin_ctzll((unsigned long long)vmask);
        vmask &= vmask - 1;
        const int ctype = (int)((tb0 >> (gbase + b)) & 1) | (int)(((tb1 >> (gbase + b)) & 1) << 1) |
                          (int)(((tb2 >> (gbase + b)) & 1) << 2);
        const int ptype = fill_cell(i0 + b, ctype, false);
        if (ptype < 0) {
          d.overflow = true;
          break;
        }
      }
      group_sync<kLds>();
      GP_MARK(3);
    }
  }
  if (d.overflow) return true;
  if (d.length - (d.min_ci + d.min_cj) >= drop) return true;
  if (!q_open && !d_open) return true;
  return false;
}

// dir_step for eight-lane groups with at most 32 cells per anti-diagonal, TWO anti-diagonals per call where that
// changes nothing.  The eight groups of a wavefront run their fill loops in lockstep, so a step costs what its busiest
// group needs - 1.31 iterations per anti-diagonal where a group has 0.68 cells (PROF build, configs[2] shape).  Cells
// of anti-diagonal L + 1 never take a cell of L as predecessor (ri < ci and rj < cj with ri + rj = L, ci + cj = L + 1
// is impossible), and their helix check reads the types of L - 1: the two anti-diagonals only meet in the cell list
// and in the running minimum, both of which take A's cells before B's.  One fill loop over both: 2.10 iterations
// instead of 2.62, one pass through the prologue and the loop top instead of two.
// B is taken along only when A cannot be the last anti-diagonal (no improvement for `drop` lengths / both strands at
// their ends), the state has room for both in full, and B's accessibility sums are staged; else the call is dir_step.
// The list as B's cells see it: live candidates [loB, dstartA) - A's cells lie behind and never qualify -, and the
// default predecessor of a cell without candidates is entry loB, which is A's first cell when everything older has
// been pruned (its type is known from the check, the record itself may not be written yet).
template <class Store>
__device__ __forceinline__ bool dir_step_pair(const SearchConst &sc, const SeqBases &sb, const ExtOpts &o, const HitCtx &c, int flag,
                                              const Store &S, int gl, int gbase, DirState &d, GapProf &prof) {
  constexpr int G = 8;
  using R = typename Store::R;
  static_assert(Store::kCapD <= 32, "a 32-bit mask of cells per anti-diagonal");
  constexpr int kChunks = (Store::kCapD + G - 1) / G;
  const int delta = o.delta, drop = o.drop_w_gap, min_helix = o.min_helix;
  const bool mid = d.resume_i0 != 0; // continuing an anti-diagonal that could not be finished
  if (!mid) {
    if (d.length >= S.cap_d()) {
      d.overflow = true;
      return true;
    }
    d.length++;
  }
  const int LA = d.length;
  const bool openA = LA < d.tq0 || LA < d.td0;
  const int curA = LA - 3 * ((LA * 171) >> 9), d2A = curA == 2 ? 0 : curA + 1; // LA mod 3, (LA - 2) mod 3
  if (!mid) {
    if (LA > d.staged) stage_acc_regs<G, true>(sb, c, flag, delta, S, gl, gbase, d, LA);
    GP_MARK(1);
    GP_COUNT(10);
    uint64_t *row = reinterpret_cast<uint64_t *>(&S.ptab(curA, 0));
    for (int t = gl; t < S.ptab_len() / 8; t += G) row[t] = 0;
    group_sync<true>();
    if (LA - 2 > drop) {
      while (d.lo < d.nrec) {
        const auto v = S.info(d.lo);
        if (LA - R::i(v) - R::j(v) - 2 > drop) d.lo++;
        else break;
      }
    }
  }
  GP_MARK(2);
  const int dstartA = mid ? d.resume_dstart : d.nrec, loA = d.lo;
  const int i_firstA = LA - d.td0 + 1 > 1 ? LA - d.td0 + 1 : 1;
  const int i_loA = mid ? d.resume_i0 : i_firstA;
  const int i_hiA = (LA < d.tq0 ? LA : d.tq0) - 1;
  d.resume_i0 = 0;

  // pair type of the cell (i, j) whose bases pair, as dir_step's cell_type has it
  auto pair_type = [&](int i, int j) -> int {
    const unsigned x0 = (unsigned)S.qb(i) * 5 + S.db(j);
    int t = (int)((sc.bp_rows >> ((x0 * 3 - 15) & 63)) & 7);
    if (flag == 1) t = ((t - 1) ^ 1) + 1;
    return t;
  };
  // CheckHelixLength + GetBPType (see dir_step's cell_type) for cell (i, j): x0, x1, x2 = 5 * query base + database base
  // of the cell and of the two positions ahead of it on its diagonal, pt = Cell::type of its diagonal predecessor
  auto helix_ok = [&](unsigned x0, unsigned x1, unsigned x2, int pt, int i, int j) -> bool {
    const unsigned p0 = (sc.pair_mask >> x0) & 1, w0 = (sc.wobble_mask >> x0) & 1;
    unsigned ahead = 1;
    if (min_helix >= 2) ahead = ((sc.pair_mask >> x1) & 1) & ~(w0 & (sc.wobble_mask >> x1));
    if (min_helix >= 3) ahead &= sc.pair_mask >> x2;
    for (int x = 3; x <= min_helix - 1; x++) ahead &= sc.pair_mask >> ((unsigned)S.qb(i + x) * 5 + S.db(j + x));
    const unsigned need = (unsigned)(pt == 0) | (w0 & (unsigned)wobble(pt));
    return (p0 & (ahead | ~need) & 1) != 0;
  };

  // ---- anti-diagonal B = LA + 1 in the same fill loop?  (LB mod 3 = (LA - 2) mod 3 = d2A; (LB - 2) mod 3 = d2B) ----
  const int LB = LA + 1, d2B = curA == 0 ? 2 : curA - 1;
  const bool try_pair = !mid && openA && LA - (d.min_ci + d.min_cj) < drop && LA < S.cap_d() && LB <= d.staged;
  uint32_t cellsA = 0, cellsB = 0; // bit b: cell i = i_loA + b (both)
  int loB = loA;
  bool paired = false;
  if (try_pair) {
    if (LB - 2 > drop) {
      while (loB < dstartA) {
        const auto v = S.info(loB);
        if (LB - R::i(v) - R::j(v) - 2 > drop) loB++;
        else break;
      }
    }
    // both checks in one pass: cell i of A is (i, LA - i), of B (i, LA + 1 - i) - the same three query bases, four
    // database bases instead of twice three
    const int i_loB = LB - d.td0 + 1 > 1 ? LB - d.td0 + 1 : 1, i_hiB = (LB < d.tq0 ? LB : d.tq0) - 1; // (i_hiB >= i_hiA)
#pragma unroll
    for (int ch = 0; ch < kChunks; ch++) {
      const int i0 = i_loA + ch * G;
      if (i0 <= i_hiB) {
        const int i = i0 + gl, j = LA - i, jc = j < 0 ? 0 : j; // (a lane beyond both ranges computes on whatever it reads)
        const unsigned q0 = (unsigned)S.qb(i) * 5, q1 = (unsigned)S.qb(i + 1) * 5, q2 = (unsigned)S.qb(i + 2) * 5;
        const unsigned d0 = S.db(jc), d1 = S.db(jc + 1), d2b = S.db(jc + 2), d3 = S.db(jc + 3);
        const bool okA = i <= i_hiA && helix_ok(q0 + d0, q1 + d1, q2 + d2b, S.ptab(d2A, i - 1), i, j);
        const bool okB = i >= i_loB && i <= i_hiB && helix_ok(q0 + d1, q1 + d2b, q2 + d3, S.ptab(d2B, i - 1), i, j + 1);
        cellsA |= (uint32_t)((__ballot(okA) >> gbase) & 0xFF) << (8 * ch);
        cellsB |= (uint32_t)((__ballot(okB) >> gbase) & 0xFF) << (8 * ch);
        GP_COUNT(11);
      }
    }
    if (d.nrec + __popc(cellsA) + __popc(cellsB) <= S.cap_r()) {
      paired = true;
      // B's types go to the row of LA - 2: the check of A has read it - and will not read it again, which it would if
      // A ran out of cells half way and the next tier took over (hence only now that there is room for A and B in full)
      uint64_t *row = reinterpret_cast<uint64_t *>(&S.ptab(d2A, 0));
      for (int t = gl; t < S.ptab_len() / 8; t += G) row[t] = 0;
      group_sync<true>();
      GP_COUNT(10);
    } else {
      cellsB = 0;
    }
  } else {
#pragma unroll
    for (int ch = 0; ch < kChunks; ch++) {
      const int i0 = i_loA + ch * G;
      if (i0 <= i_hiA) {
        const int i = i0 + gl, j = LA - i, jc = j < 0 ? 0 : j;
        const unsigned x0 = (unsigned)S.qb(i) * 5 + S.db(jc), x1 = (unsigned)S.qb(i + 1) * 5 + S.db(jc + 1),
                       x2 = (unsigned)S.qb(i + 2) * 5 + S.db(jc + 2);
        const bool okA = i <= i_hiA && helix_ok(x0, x1, x2, S.ptab(d2A, i - 1), i, j);
        cellsA |= (uint32_t)((__ballot(okA) >> gbase) & 0xFF) << (8 * ch);
        GP_COUNT(11);
      }
    }
  }
  const int nA = __popc(cellsA);
  int typeA0 = 0; // Stem::type of A's first cell as its record will hold it (0: A has no cell)
  if (paired && nA != 0) {
    const int iA0 = i_loA + __builtin_ctz(cellsA);
    typeA0 = rtype_of(pair_type(iA0, LA - iA0));
  }
#ifdef PRB_GAP_PROFILE
  prof.cells_now = nA + __popc(cellsB);
#endif
  GP_MARK(3);

  // one filled cell, or two at once on four lanes each (dir_step's fill_cell; a cell brings its anti-diagonal along)
  auto fill_cell = [&](int ci, bool isB, bool two) -> int {
    GP_COUNT(12);
    const bool pair_mode = two, hi = pair_mode && gl >= G / 2;
    const int sub = pair_mode ? (gl & (G / 2 - 1)) : gl, stride = pair_mode ? G / 2 : G;
    const int L = LA + (isB ? 1 : 0), cur = isB ? (curA == 2 ? 0 : curA + 1) : curA, lo = isB ? loB : loA;
    const int cj = L - ci;
    const int ctype = pair_type(ci, cj);
    const int nq = S.qb(ci - 1), nd = S.db(cj - 1);
    double bte = 1000000.0; // INF
    int bkp = lo << 3;
    // (no records fetched ahead at the start of the step, nor the cell's own values ahead of the scan, as dir_step does:
    // the registers they occupy cost more here than the LDS latency they hide)
    typename R::word vn = 0;
    double hn = 0;
    if (lo + sub < dstartA) {
      vn = S.info(lo + sub);
      hn = S.hyb(lo + sub);
    }
    for (int k0 = lo; k0 < dstartA; k0 += stride) {
      GP_COUNT(13);
      const int k = k0 + sub;
      const auto v = vn;
      const double hk = hn;
      if (k + stride < dstartA) {
        vn = S.info(k + stride);
        hn = S.hyb(k + stride);
      }
#ifdef PRB_GAP_PROFILE
      { // lanes with a candidate / with a qualifying candidate in this round (wave-level)
        const bool has = k < dstartA, ok = has && R::i(v) < ci && R::j(v) < cj;
        prof.acc[22] += (unsigned long long)__popcll(__ballot(has));
        prof.acc[23] += (unsigned long long)__popcll(__ballot(ok));
      }
#endif
      if (k < dstartA) {
        const int ri = R::i(v), rj = R::j(v);
        if (ri < ci && rj < cj) {
          const int rq = R::qa(v), rd = R::da(v);
          const bool f0 = flag == 0;
          const int rt = R::type(v);
          double te = loop_energy_abcd(sc, f0 ? ctype : rt, f0 ? rt : ctype, ci - ri - 1, cj - rj - 1, f0 ? nq : rq, f0 ? nd : rd,
                                       f0 ? rq : nq, f0 ? rd : nd);
          te += hk;
          if (te < bte) {
            bte = te;
            bkp = (k << 3) | R::type(v);
          }
        }
      }
    }
    GP_MARK(4);
    group_min_halves(bte, bkp, pair_mode);
    GP_MARK(5);
    if (d.nrec >= S.cap_r()) return -1;
    int bk = bkp >> 3, ptype = bkp & 7;
    // empty window - for a cell of B the list ends behind A's cells -: the reference reads stem_candidate[0] of an empty list
    const bool a_first = isB && typeA0 != 0 && lo >= dstartA; // the default entry is A's first cell
    if (lo >= dstartA && !a_first) bk = 0;
    if (ptype == 0) ptype = a_first ? typeA0 : R::type(S.info(bk)); // no candidate qualified: the type of the default entry
    const int rec = d.nrec + (hi ? 1 : 0);
    if (sub == 0) {
      S.hyb(rec) = bte;
      S.info(rec) = R::pack(ci, cj, bk, rtype_of(ctype), S.qb(ci + 1), S.db(cj + 1));
      S.ptab(cur, ci) = (uint8_t)ptype;
    }
    const double ie = S.eq(ci - 1) + S.ed(cj - 1) + bte;
    // the first cell (lanes 0-3), then the second (lanes 4-7): each half gets the other's values from its mirror lane
    const double ie_o = dpp_f64<0x141>(ie);
    const int ci_o = dpp_i32<0x141>(ci), cj_o = dpp_i32<0x141>(cj);
    const double ie_a = hi ? ie_o : ie, ie_b = hi ? ie : ie_o;
    const int ci_a = hi ? ci_o : ci, ci_b = hi ? ci : ci_o, cj_a = hi ? cj_o : cj, cj_b = hi ? cj : cj_o;
    if (ie_a < d.min_e) {
      d.min_e = ie_a;
      d.best = d.nrec;
      d.min_ci = ci_a;
      d.min_cj = cj_a;
    }
    if (pair_mode && ie_b < d.min_e) {
      d.min_e = ie_b;
      d.best = d.nrec + 1;
      d.min_ci = ci_b;
      d.min_cj = cj_b;
    }
    d.nrec += pair_mode ? 2 : 1;
    GP_MARK(6);
    return ptype;
  };

  unsigned long long cells = (unsigned long long)cellsA | ((unsigned long long)cellsB << 32);
  while (cells) { // filled cells of A, then of B, ascending i, two at a time where there are two
    const unsigned long long rest = cells & (cells - 1);
    const bool two = rest != 0 && d.nrec + 2 <= S.cap_r();
    const int b0 = __builtin_ctzll(cells);
    if (Store::kResumable && d.nrec >= S.cap_r()) { // out of cells (never with B along): stop in front of this one
      d.overflow = true;
      d.resume_i0 = i_loA + b0;
      d.resume_dstart = dstartA;
      break;
    }
    int b = b0;
    if (two) {
      if (gl >= G / 2) b = __builtin_ctzll(rest);
      cells = rest & (rest - 1);
    } else {
      cells = rest;
    }
    const bool isB = b >= 32;
    const int ptype = fill_cell(i_loA + (b & 31), isB, two);
    if (ptype < 0) {
      d.overflow = true;
      break;
    }
  }
  group_sync<true>();
  GP_MARK(3);
  if (d.overflow) return true;
  if (paired) {
    d.length = LA + 1;
    d.lo = loB;
    if (LB - (d.min_ci + d.min_cj) >= drop) return true;
    return !(LB < d.tq0 || LB < d.td0);
  }
  if (LA - (d.min_ci + d.min_cj) >= drop) return true;
  return !openA;
}

// The hit after the direction (:300-318): grown by the cell of the minimum, if there is one.
template <class Store> __device__ __forceinline__ DirResult dir_finish(const DirState &d, HitState &h, int flag, const Store &S) {
  DirResult r;
  r.overflow = d.overflow;
  r.best = d.min_ci != 0 ? d.best : 0; // (a filled cell has i >= 1 and j >= 1)
  if (d.min_ci != 0) {
    h.e_acc = h.e_acc + S.eq(d.min_ci - 1) + S.ed(d.min_cj - 1); // first_accessibility + accQ[i-1] + accDb[j-1] (:266)
    if (flag == 0) {
      h.q_sp -= d.min_ci;
      h.db_sp -= d.min_cj;
    } else {
      h.id_start -= d.min_cj;
    }
    h.q_len += d.min_ci;
    h.db_len += d.min_cj;
  }
  h.e_tot = d.min_e; // (hybridization energy = h.e_tot - h.e_acc, taken where the hit is written)
  return r;
}

// A state dump: everything the next tier needs to go on where this one ran out of capacity.
struct ResumeHeader {
  DirState d;       // lane 0's copy, as of the last complete anti-diagonal
  double acc_prev1; // lane 1's accumulator (the db-side chain)
  HitState h;       // the hit as the running direction found it
  int32_t flag, nleft;
};
// slot size for dumps of tier T (read by tier T + 1)
template <class T, class Rec> constexpr size_t resume_bytes() {
  return (sizeof(ResumeHeader) + sizeof(LdsState<T, Rec>) + 63) & ~(size_t)63;
}

template <int G, class T, class Rec>
__device__ __forceinline__ void resume_dump(const GapResume &ro, const HitCtx &c, const DirState &d, int flag,
                                            const LdsStore<T, Rec> &S, int gl, int gbase) {
  int slot = -1;
  if (gl == 0) {
    slot = (int)atomicAdd(ro.count, 1u);
    if (slot >= ro.cap) slot = -1;
  }
  slot = __shfl(slot, gbase);
  if (slot < 0) return;
  uint8_t *dst = ro.pool + (size_t)slot * resume_bytes<T, Rec>();
  ResumeHeader *H = reinterpret_cast<ResumeHeader *>(dst);
  if (gl == 0) {
    H->d = d;
    H->h = c.h;
    H->flag = flag;
    H->nleft = c.nleft;
    ro.slot[c.x] = slot;
  } else if (gl == 1) {
    H->acc_prev1 = d.acc_prev;
  }
  if constexpr (!T::kAccGlobal) { // the state as it lies in LDS, word by word
    uint32_t *w = reinterpret_cast<uint32_t *>(dst + sizeof(ResumeHeader));
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&S.s);
#pragma unroll 2 // (rare path: keep its registers out of the way)
    for (int t = gl; t < (int)(sizeof(LdsState<T, Rec>) / 4); t += G) w[t] = src[t];
  } else { // what resume_load reads of it (the windows of bases are staged afresh there), in the same layout
    LdsState<T, Rec> *o = reinterpret_cast<LdsState<T, Rec> *>(dst + sizeof(ResumeHeader));
    for (int t = gl; t < T::kCapD; t += G) {
      o->eq[t] = S.eq(t);
      o->ed[t] = S.ed(t);
    }
    for (int t = gl; t < d.nrec; t += G) {
      o->hyb[t] = S.hyb(t);
      o->info[t] = S.info(t);
    }
    uint32_t *pw = reinterpret_cast<uint32_t *>(&o->ptab[0][0]);
    const uint32_t *ps = reinterpret_cast<const uint32_t *>(&S.s.ptab[0][0]);
    for (int t = gl; t < 3 * LdsState<T, Rec>::kPtabLen / 4; t += G) pw[t] = ps[t];
  }
}

// Receiving side: the group's LDS state and scalars from a dump of the smaller tier TS (its own,
// longer windows are staged afresh; the cell records are re-packed when the formats differ).
template <int G, class TS, class RecS, class Store>
__device__ __forceinline__ void resume_load(const GapResume &ri, int slot, const GapArgs &a, HitCtx &c, DirState &d, int &flag,
                                            const Store &S, int gl) {
  using R = typename Store::R;
  const uint8_t *src = ri.pool + (size_t)slot * resume_bytes<TS, RecS>();
  const ResumeHeader *H = reinterpret_cast<const ResumeHeader *>(src);
  d = H->d;
  if (G > 1 && gl == 1) d.acc_prev = H->acc_prev1;
  d.overflow = false;
  c.h = H->h;
  c.nleft = H->nleft;
  flag = H->flag;
  const LdsState<TS, RecS> *s0 = reinterpret_cast<const LdsState<TS, RecS> *>(src + sizeof(ResumeHeader));
  stage_windows<G, true>(SeqBases{a.qb.enc, a.qb.acc, a.qb.cond, a.pg.acc, a.pg.cond}, c, flag, a.pg.seqs, a.pg.nchars, S, gl,
                         d); // also clears the rows
  for (int t = gl; t < TS::kCapD; t += G) {
    S.eq(t) = s0->eq[t];
    S.ed(t) = s0->ed[t];
  }
  for (int t = gl; t < d.nrec; t += G) {
    S.hyb(t) = s0->hyb[t];
    const auto v = s0->info[t];
    S.info(t) = R::pack(RecS::i(v), RecS::j(v), RecS::pred(v), RecS::type(v), RecS::qa(v), RecS::da(v));
  }
  constexpr int kLenS = LdsState<TS, RecS>::kPtabLen;
  for (int t = gl; t < 3 * kLenS; t += G) {
    const int row = t / kLenS, i = t - row * kLenS;
    S.ptab(row, i) = s0->ptab[row][i];
  }
  group_sync<true>();
  acc_sync<Store>();
}

// Hit w of the work list.  kMode 0: extend, write the hit to out, the number of traced-back
// pairs to bp_count[x] and the traced cells to the trace slot; 2: write the base pairs of the
// final alignment at bp_off[w] (the extension is recomputed; only for the few final hits the
// trace slots cannot describe).
template <int kMode> __device__ __forceinline__ void hit_load(const GapArgs &a, int64_t w, HitCtx &c) {
  const SearchConst &sc = a.sc;
  const int64_t x = a.subset ? (int64_t)a.subset[w] : w;
  c.x = x;
  c.query = a.in.query[x];
  c.qo = a.qb.off[c.query];
  c.qn = a.qb.len[c.query] + 1;
  c.id = a.in.db_id[x];
  c.dbase = (int64_t)a.pg.start_pos[c.id] - c.id;
  HitState &h = c.h;
  h.q_sp = a.in.q_sp[x];
  h.db_sp = a.in.db_sp[x];
  h.q_len = a.in.q_len[x];
  h.db_len = a.in.db_len[x];
  h.id_start = a.in.db_id_start[x];
  h.e_tot = a.in.e_tot[x];
  h.e_acc = a.in.e_acc[x];
  c.diag_q = h.q_sp; // the ungapped region
  c.diag_d = h.db_sp;
  c.diag_len = US(h.q_len);
  c.ndiag = 0;
  if (kMode != 0) // GetBasePair, rna_interaction_search.cpp:371-385 (every lane counts; cheap)
    for (int t = 0; t < c.diag_len; t++)
      c.ndiag += diag_pairs(sc, a.qb.enc[c.qo + c.diag_q + t], a.pg.seqs[c.diag_d + t]);
  c.unsorted = kMode != 0 && a.first_flag && a.first_flag[x]; // hit 0 keeps raw pair order (:314-317)
  c.out0 = kMode == 2 ? a.bp_off[w] : 0;
  c.ovf = false;
  c.nleft = 0;
  c.nright = 0;
  c.resumed = false;
  c.tier0 = 0;
  if (kMode == 0 && is_resumed(a.tier_out[x])) {
    c.tier0 = a.tier_out[x] & kMarkTier;
    h.q_sp = a.out.q_sp[x];
    h.db_sp = a.out.db_sp[x];
    h.q_len = a.out.q_len[x];
    h.db_len = a.out.db_len[x];
    h.id_start = a.out.db_id_start[x];
    h.e_tot = a.out.e_tot[x];
    h.e_acc = a.out.e_acc[x];
    c.nleft = a.bp_count[x] & 0xFFFF;
    c.resumed = true;
  }
}

// After a direction: traceback (:300-308, :409-424) from the arg-min cell through the
// predecessors.  The extension pass leaves the chain in the hit's trace slot, so that the base
// pairs of the few hits that survive the final filter can be written without extending them again.
template <int kMode, bool kLds, class Store>
__device__ __forceinline__ void hit_dir_done(const GapArgs &a, HitCtx &c, int flag, const DirOrigin &og /* before dir_finish */,
                                             const DirResult &r, const Store &S, int gl) {
  using R = typename Store::R;
  c.ovf = r.overflow;
  if (c.ovf) return;
  int cnt = 0;
  int lslot = -1; // the wavefront-per-hit kernel: the hit's slot of the long traces
  if constexpr (kMode == 0 && !kLds) lslot = a.lt.slot ? a.lt.slot[c.x] : -1;
  for (int k = r.best; k != 0; k = R::pred(S.info(k)), cnt++) {
    if (kMode == 0 && kLds && gl == 0 && cnt < kTraceCap) {
      const auto v = S.info(k);
      a.trace[(c.x * 2 + flag) * kTraceCap + cnt] = (uint16_t)(R::i(v) | (R::j(v) << 8));
    }
    if (kMode == 0 && !kLds && gl == 0 && lslot >= 0 && cnt < a.lt.cap) {
      const auto v = S.info(k);
      a.lt.trace[((int64_t)lslot * 2 + flag) * a.lt.cap + cnt] = (uint32_t)R::i(v) | ((uint32_t)R::j(v) << 16);
    }
  }
  if (kMode == 0 && !kLds && gl == 0 && lslot >= 0) a.lt.count[lslot * 2 + flag] = cnt <= a.lt.cap ? cnt : -2;
  if (flag == 0) c.nleft = cnt;
  else c.nright = cnt;
  if (kMode == 2 && gl == 0) {
    int t = 0;
    for (int k = r.best; k != 0; k = R::pred(S.info(k)), t++) {
      const auto v = S.info(k);
      int64_t pos;
      int qv, dv;
      if (flag == 0) { // emitted outermost first = ascending positions
        qv = og.q_start - R::i(v);
        dv = (int)(og.db_start - R::j(v));
        pos = c.unsorted ? c.out0 + c.ndiag + t : c.out0 + t;
      } else { // emitted outermost first = descending positions
        qv = og.q_start + R::i(v);
        dv = (int)(og.db_start + R::j(v));
        pos = c.unsorted ? c.out0 + c.ndiag + c.nleft + t : c.out0 + c.nleft + c.ndiag + (cnt - 1 - t);
      }
      a.bp_out[2 * pos] = qv;
      a.bp_out[2 * pos + 1] = dv;
    }
  }
}

// After both directions (or an overflow): lane 0 of the group writes the results of hit w.
template <int kMode>
__device__ __forceinline__ void hit_store(const GapArgs &a, int64_t w, const HitCtx &c, int flag /* last direction run */, int gl) {
  if (gl != 0) return;
  const SearchConst &sc = a.sc;
  const HitState &h = c.h;
  const uint8_t *ds = a.pg.seqs, *qs = a.qb.enc + c.qo;
  if (kMode == 0) {
    a.overflow[w] = c.ovf ? 1 : 0;
    if (c.ovf && flag == 1) { // direction 0 is done (c.h is its result): leave it for the next kernel
      const int64_t x = c.x;
      a.out.q_sp[x] = h.q_sp;
      a.out.db_sp[x] = h.db_sp;
      a.out.q_len[x] = h.q_len;
      a.out.db_len[x] = h.db_len;
      a.out.db_id_start[x] = h.id_start;
      a.out.e_acc[x] = h.e_acc;
      a.out.e_hyb[x] = h.e_tot - h.e_acc;
      a.out.e_tot[x] = h.e_tot;
      a.bp_count[x] = c.nleft;
      a.tier_out[x] = (uint8_t)(kResumeMark | ((c.resumed ? c.tier0 : a.tier_id) & kMarkTier));
    }
    if (!c.ovf) {
      const int64_t x = c.x;
      int tcode = a.tier_id > c.tier0 ? a.tier_id : c.tier0;
      if (a.tier_id == kWaveTier && a.lt.slot) { // both chains on record?  (the first one possibly in an LDS tier's trace slot)
        const int ls = a.lt.slot[x];
        if (ls >= 0) {
          const int c0 = a.lt.count[ls * 2], c1 = a.lt.count[ls * 2 + 1];
          if ((c0 >= 0 || (c0 == -1 && c.nleft <= kTraceCap)) && c1 >= 0) tcode = kLongTraceTier;
        }
      }
      a.tier_out[x] = (uint8_t)tcode;
      a.bp_count[x] = c.nleft | (c.nright << 16);
      // GappedExtension::Run tail (gapped_extension.cpp:49-67): dangling ends on both sides
      const double d0 = dangle_energy_gapped(sc, h.q_sp, h.db_sp, 0, qs, c.qn, ds, a.pg.nchars);
      const double d1 = dangle_energy_gapped(sc, h.q_sp + US(h.q_len) - 1, (int64_t)h.db_sp + US(h.db_len) - 1, 1, qs, c.qn,
                                             ds, a.pg.nchars);
      double e = h.e_tot, hy = h.e_tot - h.e_acc; // (the hybridization energy as the last direction left it, :311)
      e += d0;
      e += d1;
      hy += d0;
      hy += d1;
      a.out.q_sp[x] = h.q_sp;
      a.out.db_sp[x] = h.db_sp;
      a.out.q_len[x] = h.q_len;
      a.out.db_len[x] = h.db_len;
      a.out.db_id[x] = c.id;
      a.out.db_id_start[x] = h.id_start;
      a.out.query[x] = c.query;
      a.out.e_acc[x] = h.e_acc;
      a.out.e_hyb[x] = hy;
      a.out.e_tot[x] = e;
    }
  } else if (!c.ovf) {
    int t = 0;
    const int64_t d0 = c.unsorted ? c.out0 : c.out0 + c.nleft;
    for (int u = 0; u < c.diag_len; u++)
      if (diag_pairs(sc, qs[c.diag_q + u], ds[c.diag_d + u])) {
        a.bp_out[2 * (d0 + t)] = c.diag_q + u;
        a.bp_out[2 * (d0 + t) + 1] = c.diag_d + u;
        t++;
      }
  }
}

// GapArgs::handover: the first direction is done, by this kernel; the hit is left for the front kernel's proof that the
// second one finds nothing (the cascade's own business again only if it does find something).
__device__ __forceinline__ void hit_handover(const GapArgs &a, int64_t w, const HitCtx &c, int gl) {
  if (gl != 0) return;
  const int64_t x = c.x;
  const HitState &h = c.h;
  a.overflow[w] = 0;
  a.out.q_sp[x] = h.q_sp;
  a.out.db_sp[x] = h.db_sp;
  a.out.q_len[x] = h.q_len;
  a.out.db_len[x] = h.db_len;
  a.out.db_id_start[x] = h.id_start;
  a.out.e_acc[x] = h.e_acc;
  a.out.e_hyb[x] = h.e_tot - h.e_acc;
  a.out.e_tot[x] = h.e_tot;
  a.bp_count[x] = c.nleft;
  a.tier_out[x] = (uint8_t)(kHandoverMark | (a.tier_id & kMarkTier));
}

// total base pairs of list entry w = complementary positions of the ungapped diagonal
// (GetBasePair, rna_interaction_search.cpp:371-385) + the pairs traced by the two extensions
__global__ __launch_bounds__(256) void k_bp_count(HitSoA in, int64_t n, const uint32_t *__restrict__ subset, QBatchDev qb,
                                                  PageDev pg, SearchConst sc, const int32_t *__restrict__ ntrace,
                                                  int32_t *bp_count) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  const int64_t x = subset ? (int64_t)subset[w] : w;
  const uint8_t *qs = qb.enc + qb.off[in.query[x]] + in.q_sp[x];
  const uint8_t *ds = pg.seqs + in.db_sp[x];
  const int len = US(in.q_len[x]);
  int c = 0;
  for (int t = 0; t < len; t++) c += diag_pairs(sc, qs[t], ds[t]);
  bp_count[w] = c + (ntrace[x] & 0xFFFF) + (int)((uint32_t)ntrace[x] >> 16);
}

// LDS form.  The grid is what is resident at once; every group takes hits from a global work
// counter.  The groups of a wavefront share one loop whose body is "one anti-diagonal of whatever
// hit / direction the group is at", and they change direction / hit only at the iterations that
// are multiples of P = the drop-out length: a direction without any improvement - nine in ten -
// takes exactly P anti-diagonals, so the groups that started together also finish together and
// their (latency-bound) transitions stay one convergent piece of code, while a group with a
// longer extension just misses a boundary or two instead of holding up the whole wavefront
// (PROF build: with hit-by-hit lockstep 36 % of the wave time was spent waiting for the longest
// extension; with free-running groups 40 % on each other's divergent transitions).
template <int kMode, class T, class Rec, bool kPair>
#ifdef PRB_TIERS_VGPR128 // (experiment: what the tiers cost at the registers of four wavefronts per SIMD)
__global__ __launch_bounds__(T::kG *T::kGroups) __attribute__((amdgpu_num_vgpr(128))) void k_gapped_lds(GapArgs a) {
#else
__global__ __launch_bounds__(T::kG *T::kGroups, T::kWavesPerSimd) void k_gapped_lds(GapArgs a) {
#endif
  static_assert(!kPair || (T::kPairSteps && T::kG == 8), "dir_step_pair");
  __shared__ LdsLive<T, Rec> lds[T::kGroups];
  constexpr int G = T::kG;
  const int gl = threadIdx.x & (G - 1);
  const int gbase = (threadIdx.x & 63) & ~(G - 1);
  const int gid = threadIdx.x / G;
  const LdsStore<T, Rec> S{lds[gid], T::kAccGlobal ? a.acc_scratch + ((size_t)blockIdx.x * T::kGroups + gid) * 2 * T::kCapD : nullptr};
  const int64_t ngroups = (int64_t)gridDim.x * T::kGroups;
  GapProf prof;
  prof.start();
  int64_t w = (int64_t)blockIdx.x * T::kGroups + gid;
  enum { kLoad, kInit, kRun, kFinished, kDone };
  int phase = w < a.n ? kLoad : kDone, flag = 0;
  // (with two anti-diagonals per step a direction without improvement takes (drop + 1) / 2 steps)
  const int period = a.period > 0 ? a.period : kPair ? (a.o.drop_w_gap > 1 ? (a.o.drop_w_gap + 1) / 2 : 1) : (a.o.drop_w_gap > 1 ? a.o.drop_w_gap : 1);
  int tick = 0; // iterations since the last boundary
  HitCtx c;
  DirState d;
  const SeqBases sb{a.qb.enc, a.qb.acc, a.qb.cond, a.pg.acc, a.pg.cond};
  while (__ballot(phase != kDone) != 0) {
    GP_MARK(14);
    // (a.early: not only every `period` iterations, but as soon as that many groups wait for a boundary)
    if (a.early > 0 && tick != 0 && __popcll(__ballot(phase == kFinished)) >= a.early * G) tick = 0;
    if (tick == 0) {
      if (phase == kFinished) {
        DirResult r{true, 0};
        const DirOrigin og = dir_origin(c.h, flag);
        if (!d.overflow) {
          r = dir_finish(d, c.h, flag, S); // (an overflowed direction leaves c.h as it was)
        } else {
          if constexpr (kMode == 0 && T::kResumable)
            if (a.rout.slot) resume_dump<G, T, Rec>(a.rout, c, d, flag, S, gl, gbase);
        }
        hit_dir_done<kMode, true>(a, c, flag, og, r, S, gl);
        group_sync<true>();
        GP_MARK(7);
        const bool hand = kMode == 0 && a.handover && !c.ovf && flag == 0 && !c.resumed;
        if (c.ovf || flag == 1 || hand) {
          if (hand) hit_handover(a, w, c, gl);
          else hit_store<kMode>(a, w, c, flag, gl);
          unsigned long long nw = 0;
          if (gl == 0) nw = (unsigned long long)ngroups + atomicAdd(a.next_work, 1ull);
          w = (int64_t)__shfl(nw, gbase);
          phase = w < a.n ? kLoad : kDone;
          GP_MARK(9);
        } else {
          flag = 1;
          phase = kInit;
        }
      }
      if (phase == kLoad) {
        hit_load<kMode>(a, w, c);
        flag = c.resumed ? 1 : 0;
        phase = kInit;
        if constexpr (kMode == 0 && T::kResumes) {
          const int slot = a.rin.slot ? a.rin.slot[c.x] : -1;
          if (slot >= 0) {
            resume_load<G, typename T::From, typename T::FromRec>(a.rin, slot, a, c, d, flag, S, gl);
            phase = kRun;
          }
        }
        GP_MARK(8);
      }
      if (phase == kInit) {
        dir_init<G, true>(a.sc, sb, c, flag, a.pg.seqs, a.pg.nchars, S, gl, a.o.delta, d);
        phase = kRun;
        GP_MARK(0);
      }
    }
    GP_MARK(15);
#ifdef PRB_GAP_PROFILE
    prof.cells_now = 0;
#endif
    if constexpr (kPair) {
      if (phase == kRun && dir_step_pair(a.sc, sb, a.o, c, flag, S, gl, gbase, d, prof)) phase = kFinished;
    } else {
      if (phase == kRun && dir_step<G, true>(a.sc, sb, a.o, c, flag, S, gl, gbase, d, prof)) phase = kFinished;
    }
#ifdef PRB_GAP_PROFILE
    if constexpr (G == 8) { // wave-level fill iterations: as they are (two cells per iteration), and if two anti-diagonals shared a loop
      auto wave_max = [](int v) {
        for (int m = 32; m >= 8; m >>= 1) {
          const int o = __shfl_xor(v, m);
          v = o > v ? o : v;
        }
        return v;
      };
      const int cn = prof.cells_now;
      prof.acc[16] += (unsigned long long)wave_max((cn + 1) / 2);
      prof.acc[17] += (unsigned long long)wave_max(cn);
      auto wave_sum = [](int v) {
        for (int m = 32; m >= 8; m >>= 1) v += __shfl_xor(v, m);
        return v;
      };
      prof.acc[18] += (unsigned long long)wave_sum(cn);
      if (tick & 1) {
        prof.acc[19] += (unsigned long long)wave_max((cn + prof.cells_prev + 1) / 2);
        prof.acc[20] += 1;
      } else {
        prof.cells_prev = cn;
      }
      prof.acc[21] += 1;
    }
#endif
    tick = tick + 1 == period ? 0 : tick + 1;
  }
  prof.flush(a.tier_id * 2 + (kMode != 0));
}

// Run-time sized form: one wavefront per hit, so the plain nesting (hit, direction, anti-diagonal).  The state block
// lies in HBM scratch - or, kInLds, in the workgroup's dynamic LDS when it fits (the first attempt's 512
// anti-diagonals / 2,048 cells are 43 KB): the same code through the same generic pointers, every access an LDS
// instead of an HBM round trip.  These are the ~150 longest extensions of 2e7, one wavefront each with the GPU
// otherwise idle: what counts is the latency of the longest one (2.5 ms from HBM scratch, twice per query).
template <int kMode, bool kInLds> __global__ __launch_bounds__(64) void k_gapped_wave(GapArgs a, GapScratch scratch) {
  extern __shared__ __align__(16) uint8_t wave_smem[];
  const int gl = threadIdx.x;
  uint8_t *mine = kInLds ? wave_smem : scratch.base + (size_t)blockIdx.x * scratch.bytes_per_thread; // one block per wavefront
  HbmStore S;
  S.capd = scratch.cap_diag;
  S.capr = scratch.cap_rec;
  S.eq_ = reinterpret_cast<double *>(mine);
  S.ed_ = S.eq_ + S.capd;
  S.hyb_ = S.ed_ + S.capd;
  S.info_ = reinterpret_cast<uint64_t *>(S.hyb_ + S.capr);
  S.ptab_ = reinterpret_cast<uint8_t *>(S.info_ + S.capr);
  S.qb_ = S.ptab_ + 3 * ((size_t)S.capd + 4);
  S.db_ = S.qb_ + S.capd + 16;
  GapProf prof;
  prof.start();
  HitCtx c;
  DirState d;
  const SeqBases sb{a.qb.enc, a.qb.acc, a.qb.cond, a.pg.acc, a.pg.cond};
  for (int64_t w = blockIdx.x; w < a.n; w += gridDim.x) {
    hit_load<kMode>(a, w, c);
    int last = 0;
    bool hand = false;
    for (int flag = c.resumed ? 1 : 0; flag < 2 && !c.ovf && !hand; flag++) {
      GP_MARK(8);
      dir_init<64, false>(a.sc, sb, c, flag, a.pg.seqs, a.pg.nchars, S, gl, a.o.delta, d);
      GP_MARK(0);
      while (!dir_step<64, false>(a.sc, sb, a.o, c, flag, S, gl, 0, d, prof)) {
      }
      DirResult r{true, 0};
      const DirOrigin og = dir_origin(c.h, flag);
      if (!d.overflow) r = dir_finish(d, c.h, flag, S);
      hit_dir_done<kMode, false>(a, c, flag, og, r, S, gl);
      group_sync<false>();
      last = flag;
      hand = kMode == 0 && a.handover && !c.ovf && flag == 0 && !c.resumed;
      GP_MARK(7);
    }
    if (hand) hit_handover(a, w, c, gl);
    else hit_store<kMode>(a, w, c, last, gl);
    GP_MARK(9);
  }
  prof.flush(kWaveTier * 2 + (kMode != 0));
}

} // namespace

#ifdef PRB_GAP_PROFILE
extern "C" int prb_debug_gap_profile(unsigned long long *out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gap_prof), sizeof(unsigned long long) * 10 * kProfSlots) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[10 * kProfSlots] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gap_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

size_t gapped_resume_bytes(int tier) {
  return tier == 0 ? resume_bytes<Tier0, Rec32>() : tier == 1 ? resume_bytes<Tier1, Rec32>() : resume_bytes<Tier2, Rec32>();
}

// the block of memory launch_gapped_lds wants for the accessibility sums of the resident groups of tiers 1 / 2 (0: none)
size_t gapped_acc_scratch_bytes() {
  if (!Tier1::kAccGlobal && !Tier0::kAccGlobal) return 0;
  const size_t t0 = (size_t)256 * Tier0::kWgPerCu * Tier0::kGroups * 2 * Tier0::kCapD * sizeof(double);
  const size_t t1 = (size_t)256 * Tier1::kWgPerCu * Tier1::kGroups * 2 * Tier1::kCapD * sizeof(double);
  const size_t t2 = (size_t)256 * Tier2::kWgPerCu * Tier2::kGroups * 2 * Tier2::kCapD * sizeof(double);
  return std::max(t0, std::max(t1, t2));
}

size_t gapped_wave_scratch_bytes(int cap_diag, int cap_rec) {
  size_t b = (size_t)cap_diag * 16 + (size_t)cap_rec * 16 + 3 * ((size_t)cap_diag + 4) + 2 * ((size_t)cap_diag + 16);
  return (b + 255) & ~(size_t)255;
}

namespace {
template <class T, class Rec> hipError_t launch_tier(const GapArgs &a, int mode, hipStream_t s) {
  const int64_t want = (a.n + T::kGroups - 1) / T::kGroups;
  const dim3 grid((unsigned)std::min<int64_t>(want, 256 * T::kWgPerCu)), blk(T::kG * T::kGroups);
  if (hipError_t e = hipMemsetAsync(a.next_work, 0, sizeof(unsigned long long), s); e != hipSuccess) return e;
  if constexpr (T::kPairSteps) {
    const char *pe = getenv("PRB_GAPPED_PAIR"); // (read per launch: the tests switch it inside one process)
    const bool pair = !(pe && atoi(pe) == 0);
    if (pair) {
      if (mode == 0) hipLaunchKernelGGL((k_gapped_lds<0, T, Rec, true>), grid, blk, 0, s, a);
      else hipLaunchKernelGGL((k_gapped_lds<2, T, Rec, true>), grid, blk, 0, s, a);
      return hipGetLastError();
    }
  }
  size_t pad = 0; // experiment (PRB_GAPPED_LDS_PAD="b1,b2,b3"): unused dynamic LDS per workgroup of tiers 1 - 3, i.e. fewer wavefronts per compute unit
  if (const char *pe = getenv("PRB_GAPPED_LDS_PAD")) {
    int b[4] = {0, 0, 0, 0};
    sscanf(pe, "%d,%d,%d", &b[1], &b[2], &b[3]);
    pad = (size_t)b[a.tier_id < 4 ? a.tier_id : 0];
  }
  if (mode == 0) hipLaunchKernelGGL((k_gapped_lds<0, T, Rec, false>), grid, blk, pad, s, a);
  else hipLaunchKernelGGL((k_gapped_lds<2, T, Rec, false>), grid, blk, 0, s, a);
  return hipGetLastError();
}
} // namespace

hipError_t launch_gapped_lds(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb,
                             const PageDev &pg, const SearchConst &sc, ExtOpts o, int mode, int tier, uint8_t *overflow,
                             uint8_t *tier_out, const uint8_t *first_flag, int32_t *bp_count, uint16_t *trace,
                             const int64_t *bp_off, int32_t *bp_out, unsigned long long *next_work, const GapResume &rin,
                             const GapResume &rout, hipStream_t s, int handover, double *acc_scratch) {
  if (n <= 0) return hipSuccess;
  GapArgs a{in,      out,   n,      subset, qb,        pg,         sc, o, overflow, tier_out, tier, first_flag, bp_count,
            trace,   bp_off, bp_out, next_work, rin,     rout};
  a.handover = mode == 0 ? handover : 0;
  a.acc_scratch = acc_scratch;
  if (((tier == 1 || tier == 2) && Tier1::kAccGlobal || tier == 0 && Tier0::kAccGlobal) && !acc_scratch) return hipErrorInvalidValue;
  if (const char *pe = getenv("PRB_GAPPED_PERIOD")) { // experiment: "p0,p1,p2,p3" lockstep iterations between boundaries, per tier (0: default)
    int p[4] = {0, 0, 0, 0};
    sscanf(pe, "%d,%d,%d,%d", &p[0], &p[1], &p[2], &p[3]);
    a.period = mode == 0 && tier >= 0 && tier < 4 ? p[tier] : 0;
  }
  // Tier 2's four groups per wavefront hold hits that arrive in the middle of a direction and end anywhere: a boundary as soon as
  // two of them wait costs it 220 ms per configs[2] step instead of 238 (tiers 0 and 1, eight groups: 2 - 4 waiting groups cost
  // more in transitions than they save in waiting; tier 3 has one group per wavefront and waits for nobody).
  a.early = tier == 2 ? 2 : 0;
  if (const char *ee = getenv("PRB_GAPPED_EARLY")) { // experiment: "k0,k1,k2,k3" groups waiting that make a boundary, per tier
    int k[4] = {0, 0, 0, 0};
    sscanf(ee, "%d,%d,%d,%d", &k[0], &k[1], &k[2], &k[3]);
    a.early = tier >= 0 && tier < 4 ? k[tier] : 0;
  }
  if (tier == 0) return launch_tier<Tier0, Rec32>(a, mode, s);
  if (tier == 1) return launch_tier<Tier1, Rec32>(a, mode, s);
  if (tier == 2) return launch_tier<Tier2, Rec32>(a, mode, s);
  return launch_tier<Tier3, Rec32W>(a, mode, s);
}

hipError_t launch_gapped_wave(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb,
                              const PageDev &pg, const SearchConst &sc, ExtOpts o, GapScratch scratch, int mode,
                              uint8_t *overflow, uint8_t *tier_out, const uint8_t *first_flag, int32_t *bp_count,
                              const int64_t *bp_off, int32_t *bp_out, hipStream_t s, int handover, const LongTrace &lt) {
  if (n <= 0) return hipSuccess;
  GapArgs a{in,      out,    n,      subset,  qb,      pg,      sc,      o, overflow, tier_out, kWaveTier, first_flag, bp_count,
            nullptr, bp_off, bp_out, nullptr, GapResume{nullptr, nullptr, nullptr, 0}, GapResume{nullptr, nullptr, nullptr, 0}};
  a.handover = mode == 0 ? handover : 0;
  if (mode == 0) a.lt = lt;
  const int blocks = scratch.nthreads; // here: number of wavefronts that own a state block
  if (scratch.base == nullptr) { // state in LDS
    const size_t lds = scratch.bytes_per_thread;
    if (mode == 0) hipLaunchKernelGGL((k_gapped_wave<0, true>), dim3(blocks), dim3(64), lds, s, a, scratch);
    else hipLaunchKernelGGL((k_gapped_wave<2, true>), dim3(blocks), dim3(64), lds, s, a, scratch);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_gapped_wave<0, false>), dim3(blocks), dim3(64), 0, s, a, scratch);
    else hipLaunchKernelGGL((k_gapped_wave<2, false>), dim3(blocks), dim3(64), 0, s, a, scratch);
  }
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_assign_slots(const uint32_t *__restrict__ list, int64_t n, int32_t base, int32_t *slot) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < n) slot[list[p]] = base + (int32_t)p;
}
hipError_t launch_assign_slots(const uint32_t *list, int64_t n, int32_t base, int32_t *slot, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_assign_slots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, list, n, base, slot);
  return hipGetLastError();
}

// Base pairs of final hit w from the trace slot its extension pass left (same layout as mode 2
// of the gapped kernels writes).  Hits completed by the wave kernel (tier 3) or with a chain
// longer than the slot are left to the mode-2 pass.
__global__ __launch_bounds__(256) void k_bp_expand(HitSoA in, int64_t n, const uint32_t *__restrict__ subset, QBatchDev qb,
                                                   PageDev pg, SearchConst sc, const uint8_t *__restrict__ first_flag,
                                                   const int32_t *__restrict__ ntrace, const uint8_t *__restrict__ tier_of,
                                                   const uint16_t *__restrict__ trace, LongTrace lt, const int64_t *__restrict__ bp_off,
                                                   int32_t *bp_out) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  const int64_t x = subset[w];
  const int nleft = ntrace[x] & 0xFFFF, nright = (int)((uint32_t)ntrace[x] >> 16);
  // a hit of the wavefront-per-hit kernel with its chains on record (kLongTraceTier): the second direction from its long
  // trace, the first one from there too unless an LDS tier ran it (then from that tier's trace slot)
  const int ls = (tier_of[x] == kLongTraceTier && lt.slot) ? lt.slot[x] : -1;
  if (ls < 0 && (tier_of[x] >= kWaveTier || nleft > kTraceCap || nright > kTraceCap)) return;
  const uint32_t *lt_left = ls >= 0 && lt.count[ls * 2] >= 0 ? lt.trace + ((int64_t)ls * 2) * lt.cap : nullptr;
  const uint32_t *lt_right = ls >= 0 ? lt.trace + ((int64_t)ls * 2 + 1) * lt.cap : nullptr;
  const int q_sp = in.q_sp[x], db_sp = in.db_sp[x], len = US(in.q_len[x]);
  const uint8_t *qs = qb.enc + qb.off[in.query[x]] + q_sp;
  const uint8_t *ds = pg.seqs + db_sp;
  const bool unsorted = first_flag[x] != 0; // hit 0 of a query keeps the raw pair order
  const int64_t out0 = bp_off[w];
  int ndiag = 0;
  const int64_t d0 = unsorted ? out0 : out0 + nleft;
  for (int t = 0; t < len; t++)
    if (diag_pairs(sc, qs[t], ds[t])) {
      bp_out[2 * (d0 + ndiag)] = q_sp + t;
      bp_out[2 * (d0 + ndiag) + 1] = db_sp + t;
      ndiag++;
    }
  const uint16_t *sl = trace + x * 2 * kTraceCap;
  for (int t = 0; t < nleft; t++) {
    const int ci = lt_left ? (int)(lt_left[t] & 0xFFFF) : (sl[t] & 0xFF), cj = lt_left ? (int)(lt_left[t] >> 16) : (sl[t] >> 8);
    const int64_t pos = unsorted ? out0 + ndiag + t : out0 + t;
    bp_out[2 * pos] = q_sp - ci;
    bp_out[2 * pos + 1] = db_sp - cj;
  }
  const int q_end = q_sp + in.q_len[x] - 1, db_end = db_sp + in.db_len[x] - 1;
  for (int t = 0; t < nright; t++) {
    const int ci = lt_right ? (int)(lt_right[t] & 0xFFFF) : (sl[kTraceCap + t] & 0xFF),
              cj = lt_right ? (int)(lt_right[t] >> 16) : (sl[kTraceCap + t] >> 8);
    const int64_t pos = unsorted ? out0 + ndiag + nleft + t : out0 + nleft + ndiag + (nright - 1 - t);
    bp_out[2 * pos] = q_end + ci;
    bp_out[2 * pos + 1] = db_end + cj;
  }
}

hipError_t launch_bp_expand(const HitSoA &in, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                            const SearchConst &sc, const uint8_t *first_flag, const int32_t *ntrace, const uint8_t *tier_of,
                            const uint16_t *trace, const LongTrace &lt, const int64_t *bp_off, int32_t *bp_out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bp_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, subset, qb, pg, sc, first_flag,
                     ntrace, tier_of, trace, lt, bp_off, bp_out);
  return hipGetLastError();
}

// first and last pair of every list entry (all the simplified output prints,
// rna_interaction_search.cpp:355-363): ends[4w..4w+3] = (q0, db0, qN, dbN)
__global__ __launch_bounds__(256) void k_bp_ends(const int64_t *__restrict__ bp_off, int64_t n, const int32_t *__restrict__ bp,
                                                 int32_t *ends) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  const int64_t a = bp_off[w], b = bp_off[w + 1] - 1;
  ends[4 * w] = bp[2 * a];
  ends[4 * w + 1] = bp[2 * a + 1];
  ends[4 * w + 2] = bp[2 * b];
  ends[4 * w + 3] = bp[2 * b + 1];
}

hipError_t launch_bp_ends(const int64_t *bp_off, int64_t n, const int32_t *bp, int32_t *ends, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bp_ends, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bp_off, n, bp, ends);
  return hipGetLastError();
}

hipError_t launch_bp_count(const HitSoA &in, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                           const SearchConst &sc, const int32_t *ntrace, int32_t *bp_count, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bp_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, subset, qb, pg, sc, ntrace,
                     bp_count);
  return hipGetLastError();
}

} // namespace prb
