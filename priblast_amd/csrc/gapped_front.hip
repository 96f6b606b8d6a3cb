// Gapped extension, the kernel IN FRONT of the cascade: it PROVES, exactly, that a direction finds nothing
// (GappedExtension::extension, gapped_extension.cpp:71-319, for the nine directions in ten that end after `-x`
// anti-diagonals without an improvement), and completes the hits whose two directions both find nothing - four in
// five.  Everything else (a direction that improves, one with more filled cells than fit) goes on to the LDS tiers of
// gapped_lds.hip untouched, or with its first direction handed over as done.
//
// Why another kernel: the LDS tiers give a hit 8 lanes that walk its anti-diagonals in lockstep with seven other
// hits; an extension is so small and irregular (0.7 filled cells per anti-diagonal, ~5 candidates per cell) that
// 70 % of the lanes idle in every instruction (profiles/r02_gapped_traffic.json: 2,928 wave-level VALU instructions
// per hit at 20 of 64 lanes).  Here the work of 64 directions (32 hits x 2: a direction that finds nothing leaves the
// hit as it was, so the two are independent) is cut into phases that are each dense over their own kind of item:
//   * per DIRECTION (one lane each): the bases along both strands as bit planes, the cells of two whole
//     anti-diagonals checked at once with a few logic operations (CheckHelixLength + GetBPType for all i: pairs and
//     wobble pairs as masks over i, the look-ahead from the masks of anti-diagonals L + 2x shifted by x, the
//     predecessor test from two masks of anti-diagonal L - 2);
//   * per FILLED CELL of those two anti-diagonals, all 64 directions pooled (one lane each, 64 at a time): its record,
//     and which of its direction's earlier cells are candidates (first < i && second < j) as a bit mask;
//   * per (CELL, CANDIDATE) PAIR, pooled again: the loop energy + the candidate's energy - the expensive piece of
//     the recurrence (:230-247), now on full wavefronts;
//   * per cell again: the minimum over its pairs in list order (strict '<': the first candidate wins), the
//     improvement test against the hit's energy (:260-278), the type bits the next anti-diagonals' checks need.
// A direction is given up (its hit goes on to the cascade) as soon as a cell improves on the hit's energy - then it
// would have to run on beyond `-x` - or its list outgrows kFR cells.  No pruning (:213-217) can happen within the
// first `-x` + 2 anti-diagonals, no traceback is needed for a direction that finds nothing, and the start cell is a
// candidate of every cell, so none of the special cases of the general kernels exist here.
// Results are bit-identical to the cascade's (tests/test_gpu_search.py::test_fallback_kernels_match_tier1 runs it
// with and without this kernel; every stage-dump and option test goes through it).
#include <algorithm>
#include <cstdlib>

#include "gapped_args.hpp"
#include "search_device.hpp"
#include "search_kernels.hpp"

namespace prb {

// Developer-only cycle breakdown (make prof; tools/front_profile.py): wave-cycles per phase, summed over wavefronts,
// and counts of tiles / steps / rounds / cells / pairs / directions given up.  Not part of the product build.
#ifdef PRB_GAP_PROFILE
__device__ unsigned long long g_front_prof[32];
#define FP_DECL unsigned long long fp_t0_ = __builtin_amdgcn_s_memtime(), fp_acc_[32] = {}
#define FP_MARK(k)                                              \
  do {                                                          \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    fp_acc_[k] += t_ - fp_t0_;                                  \
    fp_t0_ = t_;                                                \
  } while (0)
#define FP_COUNT(k, v) fp_acc_[k] += (unsigned long long)(v)
#define FP_FLUSH()                                                        \
  do {                                                                    \
    if (lane == 0)                                                        \
      for (int i_ = 0; i_ < 32; i_++) atomicAdd(&g_front_prof[i_], fp_acc_[i_]); \
  } while (0)
#else
#define FP_DECL
#define FP_MARK(k)
#define FP_COUNT(k, v)
#define FP_FLUSH()
#endif

namespace {

constexpr int kFD = kFrontMaxDrop; // anti-diagonals covered: the largest -x this kernel takes
#ifndef PRB_FRONT_RING
#define PRB_FRONT_RING 24
#endif
constexpr int kFR = PRB_FRONT_RING;            // cells per direction, the start included
constexpr int kFWaves = 1;         // wavefronts per workgroup: one
// Directions per wavefront.  The per-direction phases leave the lanes beyond them idle, the pooled phases use all 64; a
// wavefront's time is the latency of its chain of phases (LDS and L2 round trips), so what counts is how many wavefronts a
// compute unit holds: 64 directions = 19 KB of LDS (with the cells' energies pooled, see FrontLds) = 8 wavefronts per
// compute unit; 32 directions = twice as many (then the registers are the limit), which was slower.
constexpr int kFDirs = 64;
constexpr int kFCells = 192;      // filled cells of a wavefront's directions in one step (two anti-diagonals)
#ifndef PRB_FRONT_PAIRS
#define PRB_FRONT_PAIRS 256
#endif
constexpr int kFPairs = PRB_FRONT_PAIRS;      // (cell, candidate) pairs in one round of at most 64 cells

// the part of SearchTab every loop class but the 1x1 / 1x2 / 2x1 / 2x2 interior loops reads from, staged in LDS
struct FrontTab {
  static constexpr int kBulge = SearchTab::kInt11;       // [31]
  static constexpr int kTau = kBulge + 31;               // [8]
  static constexpr int kZero = kTau + 8;
  static constexpr int kCount = kZero + 1;
};
static_assert(SearchTab::kStack == 0 && SearchTab::kInt11 == 255, "stack, internal and mismatch tables lead SearchTab");

// cumulative accessibility arrays of a wavefront's 64 directions, in HBM / L2 (written once per direction, read once
// per filled cell, a phase ahead of their use)
struct FrontAcc {
  double eq[kFD][kFDirs], ed[kFD][kFDirs];
};

// The energies of the cells (8 of a record's 12 bytes) are pooled over the wavefront's 64 directions: a direction may fill
// kFR cells, the average one fills eleven, so kFPool entries - the start cells first, then every step's cells in list order -
// stand in for 64 x kFR: 19 KB of LDS per wavefront instead of 24.7, eight wavefronts per compute unit instead of six (what
// bounds the kernel is the latency of a wavefront's chain of phases).  A record holds its pool index; a step whose cells
// do not all fit gives up the directions of those that do not, as for a direction with more than kFR cells.
#ifndef PRB_FRONT_POOL
#define PRB_FRONT_POOL 832
#endif
constexpr int kFPool = PRB_FRONT_POOL;
static_assert(kFPool >= 2 * kFDirs && kFPool <= 1024, "FRec::pool");
static_assert(kFrontMaxDrop <= 16, "FrontLds::wfp: 16 cells per anti-diagonal");
struct FRec { // i:5 | j:5 | pool:10 | type:3 | qa:3 | da:3 (Rec32 with the pool index of the cell's energy for a predecessor)
  static __device__ __forceinline__ uint32_t pack(int i, int j, int pool, int type, int qa, int da) {
    return (uint32_t)i | ((uint32_t)j << 5) | ((uint32_t)pool << 10) | ((uint32_t)type << 20) | ((uint32_t)qa << 23) | ((uint32_t)da << 26);
  }
  static __device__ __forceinline__ int i(uint32_t v) { return v & 31; }
  static __device__ __forceinline__ int j(uint32_t v) { return (v >> 5) & 31; }
  static __device__ __forceinline__ int pool(uint32_t v) { return (v >> 10) & 1023; }
  static __device__ __forceinline__ int type(uint32_t v) { return (v >> 20) & 7; }
  static __device__ __forceinline__ int qa(uint32_t v) { return (v >> 23) & 7; }
  static __device__ __forceinline__ int da(uint32_t v) { return (v >> 26) & 7; }
};

struct FrontLds { // per wavefront; arrays per direction are [slot][lane]: conflict-free whatever slot a lane is at
  double hyb[kFPool]; // (pooled: FRec::pool)
  double te[kFPairs];
  int32_t tab[FrontTab::kCount];
  uint32_t info[kFR][kFDirs]; // FRec
  uint32_t cells[kFCells]; // direction lane | i << 8 | second anti-diagonal of the step << 13 | record index << 16
  uint32_t wfp[2][kFDirs]; // per direction, per anti-diagonal of the step: cells (bit i - 1) whose stored type is not 0, << 16: is a wobble
  uint8_t improved[kFDirs];
  uint32_t pairs[kFPairs]; // the cell: ci | cj << 5 | ctype << 10 | nq << 13 | nd << 16 | direction lane << 19; candidate << 25
};

// the bases of one strand along the extension: bit t of lo / hi / valid describes position t
// (A = hi, C = hi | lo, G = 0, U = lo; valid = GetChar != 0)
struct Planes {
  uint32_t lo, hi, valid;
};
__device__ __forceinline__ int plane_base(uint32_t lo, uint32_t hi, uint32_t valid, int t) { // 0 = none, 1..4 = A, C, G, U
  const uint32_t l = (lo >> t) & 1, h = (hi >> t) & 1, v = (valid >> t) & 1;
  return v ? (int)(((h * 2 + l + 2) & 3) + 1) : 0;
}

// 32 positions of a strand starting at `start`, step dir (-1 / +1), from the text s[0, n): nine aligned
// words, shifted into place with v_alignbyte, then one bit per position and plane
__device__ __forceinline__ Planes load_planes(const uint8_t *s, int64_t n, int64_t start, int dir) {
  Planes p{0, 0, 0};
  const int64_t first = dir > 0 ? start : start - 31; // positions [first, first + 31]
  const uintptr_t addr = reinterpret_cast<uintptr_t>(s) + (uintptr_t)first;
  const int sh = (int)(addr & 3);
  const uint32_t *sw = reinterpret_cast<const uint32_t *>(addr - (uintptr_t)sh);
  uint32_t w[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int64_t pos_lo = first - sh + 4 * k; // the word holds positions pos_lo .. pos_lo + 3
    w[k] = (pos_lo + 3 >= 0 && pos_lo < n) ? sw[k] : 0u;
  }
  uint32_t x[8];
#pragma unroll
  for (int k = 0; k < 8; k++) x[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], (uint32_t)sh); // bytes of positions first + 4k ..
  // codes 2..5 (and 6..9, soft-masked) = A, C, G, U: bit 0 and bit 1 of the code are the planes; four positions per word
  // are gathered with one multiplication (bits 0, 8, 16, 24 -> bits 28..31)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t c = x[k];
    // valid: the code is >= 2, i.e. any of bits 1..3 of the byte is set (codes are <= 9)
    const uint32_t any = (c | (c >> 1) | (c >> 2)) >> 1; // bit 0 of every byte: bits 1 | 2 | 3 of the code
    const uint32_t lo4 = ((c & 0x01010101u) * 0x10204080u) >> 28;
    const uint32_t hi4 = (((c >> 1) & 0x01010101u) * 0x10204080u) >> 28;
    const uint32_t ok4 = ((any & 0x01010101u) * 0x10204080u) >> 28;
    p.lo |= lo4 << (4 * k);
    p.hi |= hi4 << (4 * k);
    p.valid |= ok4 << (4 * k);
  }
  // positions outside the text
  {
    const int64_t lo_t = first < 0 ? -first : 0;                    // first position inside
    const int64_t hi_t = first + 31 >= n ? n - 1 - first : 31;      // last position inside (may be < lo_t)
    uint32_t inside = 0;
    if (hi_t >= lo_t) inside = (hi_t >= 31 ? ~0u : ((1u << (hi_t + 1)) - 1u)) & ~((1u << lo_t) - 1u);
    p.valid &= inside;
  }
  if (dir < 0) { // position first + u is step 31 - u of the extension
    p.lo = __builtin_bitreverse32(p.lo);
    p.hi = __builtin_bitreverse32(p.hi);
    p.valid = __builtin_bitreverse32(p.valid);
  }
  p.lo &= p.valid;
  p.hi &= p.valid;
  return p;
}

// pairs / wobble pairs of all cells (i, len - i) of anti-diagonal len as masks over i; the database planes are
// bit-reversed (bit 31 - t = position t)
__device__ __forceinline__ void diag_masks(const Planes &q, const Planes &dr, int len, uint32_t &pairs, uint32_t &wob) {
  const int s = 31 - len; // bit i of (dr >> s) = position len - i of the database strand
  const uint32_t dl = dr.lo >> s, dh = dr.hi >> s, dv = dr.valid >> s;
  const uint32_t both = q.valid & dv, ldiff = q.lo ^ dl;
  const uint32_t gu = ~(q.hi | dh); // G or U on both sides
  wob = both & ldiff & gu;
  pairs = both & ldiff & ((q.hi ^ dh) | gu);
}

// exclusive prefix sum over the 64 lanes, on the VALU alone (DPP row shifts, then the row broadcasts of the GFX9 family)
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
  (void)lane;
  total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (global stores of this wavefront before its later loads from other lanes)
__device__ __forceinline__ void wave_mem_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// loop_energy_abcd (search_device.hpp) with the small tables in LDS: only the 1x1 / 1x2 / 2x1 / 2x2 interior loops
// go to the big tables in memory.  The same three integers, the same sum, the same division.
__device__ __forceinline__ double loop_energy_front(const SearchConst &sc, const int32_t *lt, int type, int type2, int u1, int u2, int a,
                                                    int b, int c, int d) {
  using T = SearchTab;
  using F = FrontTab;
  const int tt = type * 8 + type2;
  const int st = T::kStack + type * 7 + type2;
  const bool z1 = u1 == 0, z2 = u2 == 0;
  const int u = z1 ? u2 : u1;
  const bool c11 = u1 == 1 && u2 == 1, c12 = u1 == 1 && u2 == 2, c21 = u1 == 2 && u2 == 1, c22 = u1 == 2 && u2 == 2;
  const bool special = c11 || c12 || c21 || c22;
  int i1 = T::kInternal + u1 + u2;
  int i2 = special ? F::kZero : T::kMismatchI + (type * 5 + a) * 5 + b;
  int i3 = special ? F::kZero : T::kMismatchI + (type2 * 5 + d) * 5 + c;
  if (z1 || z2) {
    const bool stack = z1 && z2;
    i1 = stack ? st : F::kBulge + u;
    i2 = stack ? F::kZero : (u == 1 ? st : F::kTau + type);
    i3 = (stack || u == 1) ? F::kZero : F::kTau + type2;
  }
  int z1v = lt[special ? F::kZero : i1];
  if (special) {
    const int i11 = T::kInt11 + (tt * 5 + a) * 5 + b;
    const int i21a = T::kInt21 + ((tt * 5 + a) * 5 + d) * 5 + b;
    const int i21b = T::kInt21 + (((type2 * 8 + type) * 5 + d) * 5 + a) * 5 + c;
    const int i22 = T::kInt22 + (((tt * 5 + a) * 5 + c) * 5 + d) * 5 + b;
    z1v = sc.tab[c11 ? i11 : c12 ? i21a : c21 ? i21b : i22];
  }
  return div100(sc, z1v + lt[i2] + lt[i3]);
}

// kOneDir: every hit of the list has its first direction behind it (the cascade's second pass): a lane per hit, all of them
// second directions - 64 hits per wavefront instead of 32 with half the lanes out of the proof from the start.
template <bool kOneDir>
__global__ __launch_bounds__(64 * kFWaves) void k_gapped_front(GapArgs a, FrontAcc *acc_blocks) {
  __shared__ FrontLds lds[kFWaves];
  const int lane = threadIdx.x & 63;
  FrontLds &S = lds[threadIdx.x >> 6];
  FrontAcc &A = acc_blocks[(int64_t)blockIdx.x * kFWaves + (threadIdx.x >> 6)];
  const SearchConst &sc = a.sc;
  for (int t = lane; t < FrontTab::kCount; t += 64)
    S.tab[t] = t < SearchTab::kInt11    ? sc.tab[t]
               : t < FrontTab::kTau     ? sc.tab[SearchTab::kBulge + (t - FrontTab::kBulge)]
               : t < FrontTab::kZero    ? sc.tab[SearchTab::kTau + (t - FrontTab::kTau)]
                                        : 0;
  const int flag = kOneDir ? 1 : lane & 1; // even lanes extend to the left, odd lanes to the right (gapped_extension.cpp:33-48)
  const bool f0 = flag == 0;
  const int drop = a.o.drop_w_gap, min_helix = a.o.min_helix, delta = a.o.delta;
  const int64_t nwaves = (int64_t)gridDim.x * kFWaves;
  constexpr int kTileHits = kOneDir ? kFDirs : kFDirs / 2;
  const int64_t ntiles = (a.n + kTileHits - 1) / kTileHits;
  const bool isdir = lane < kFDirs; // (the lanes beyond the directions only take part in the pooled phases)
  const int me = isdir ? lane : 0;
  int64_t tile = (int64_t)blockIdx.x * kFWaves + (threadIdx.x >> 6);
  FP_DECL;
  while (tile < ntiles) {
    FP_MARK(0);
    FP_COUNT(16, 1);
    // the next tile: one atomic per wavefront, asked for now and looked at when this tile is done
    unsigned long long nt = 0;
    if (lane == 0) nt = atomicAdd(a.next_work, 1ull);
    // ---- the hit, the direction's origin (:88-128) ----
    const int64_t w = tile * kTileHits + (kOneDir ? lane : lane >> 1);
    const bool live = isdir && w < a.n;
    int64_t x = 0, qo = 0, dbase = 0;
    int query = 0, id = 0, qn = 1, q_sp = 0, db_sp = 0, q_len = 0, db_len = 0, id_start = 0;
    double e_tot = 0, e_acc = 0;
    bool resumed = false;
    int tier0 = 0, nleft0 = 0;
    if (live) {
      x = a.subset ? (int64_t)a.subset[w] : w;
      query = a.in.query[x];
      qo = a.qb.off[query];
      qn = a.qb.len[query] + 1;
      id = a.in.db_id[x];
      dbase = (int64_t)a.pg.start_pos[id] - id;
      q_sp = a.in.q_sp[x];
      db_sp = a.in.db_sp[x];
      q_len = a.in.q_len[x];
      db_len = a.in.db_len[x];
      id_start = a.in.db_id_start[x];
      e_tot = a.in.e_tot[x];
      e_acc = a.in.e_acc[x];
      const uint8_t mark = a.tier_out[x];
      if (is_resumed(mark)) { // the first direction is done already (an LDS tier stopped behind it): only the second one is looked at
        resumed = true;
        tier0 = mark & kMarkTier;
        nleft0 = a.bp_count[x] & 0xFFFF;
        q_sp = a.out.q_sp[x];
        db_sp = a.out.db_sp[x];
        q_len = a.out.q_len[x];
        db_len = a.out.db_len[x];
        id_start = a.out.db_id_start[x];
        e_tot = a.out.e_tot[x];
        e_acc = a.out.e_acc[x];
      }
    }
    const uint8_t *qs = a.qb.enc + qo;
    const int q_start = f0 ? q_sp : q_sp + US(q_len) - 1;
    const int64_t db_start = f0 ? (int64_t)db_sp : (int64_t)db_sp + US(db_len) - 1;
    const int id_end = id_start + US(db_len) - 1;
    bool dead = !live || (resumed && f0); // the direction is out of the proof: not there, done already, improved, or too many cells
    if (kOneDir && !resumed) dead = true;  // (not what the list was promised to hold: the whole hit is the cascade's)
    FP_MARK(1);

    // ---- bases along both strands (:131-154 and GetChar) ----
    Planes q{0, 0, 0}, dr{0, 0, 0};
    int tq0 = 1, td0 = 1;
    int type0 = 0, q1 = 0, d1 = 0;
    if (live) {
      q = load_planes(qs, qn, q_start, f0 ? -1 : 1);
      const Planes dd = load_planes(a.pg.seqs, a.pg.nchars, db_start, f0 ? -1 : 1);
      tq0 = (q.valid >> 1) == 0x7FFFFFFFu ? 32 : __builtin_ctz(~(q.valid >> 1)) + 1;
      td0 = (dd.valid >> 1) == 0x7FFFFFFFu ? 32 : __builtin_ctz(~(dd.valid >> 1)) + 1;
      const int q0 = plane_base(q.lo, q.hi, q.valid, 0), d0 = plane_base(dd.lo, dd.hi, dd.valid, 0);
      q1 = plane_base(q.lo, q.hi, q.valid, 1);
      d1 = plane_base(dd.lo, dd.hi, dd.valid, 1);
      type0 = bp_type(sc, q0, d0);
      if (f0) type0 = rtype_of(type0);
      dr.lo = __builtin_bitreverse32(dd.lo);
      dr.hi = __builtin_bitreverse32(dd.hi);
      dr.valid = __builtin_bitreverse32(dd.valid);
    }
    FP_MARK(2);
    // the last anti-diagonal a direction without improvement looks at (:292-297)
    const int open_until = tq0 > td0 ? tq0 : td0;
    const int lend = drop < open_until ? drop : open_until;

    // ---- cumulative accessibility change of the extension lengths 1 .. kFD (:156-212) ----
    // one side has three terms per length (the strand the extension runs down along), the other one
    {
      const float *a3 = f0 ? a.qb.acc + qo : a.pg.acc + dbase, *c3 = f0 ? a.qb.cond + qo : a.pg.cond + dbase;
      const float *c1 = f0 ? a.pg.cond + dbase + id_end : a.qb.cond + qo + q_start;
      const int p0 = f0 ? q_start : id_start;
      const int t3 = f0 ? tq0 : td0, t1 = f0 ? td0 : tq0;
      double s3 = 0, s1 = 0;
      // all the terms first (their loads in flight together): a3[p0 - len] of one length is a3[p0 - len + 1] of the next
      float ta[kFD + 1], tz[kFD], tw[kFD];
      ta[0] = (live && 1 < t3 && 1 <= lend) ? a3[p0] : 0.0f;
#pragma unroll
      for (int len = 1; len <= kFD; len++) {
        const bool v3 = live && len < t3 && len <= lend, v1 = live && len < t1 && len <= lend;
        ta[len] = v3 ? a3[p0 - len] : 0.0f;
        tz[len - 1] = v3 ? c3[p0 - len + delta] : 0.0f;
        tw[len - 1] = v1 ? c1[len] : 0.0f;
      }
#pragma unroll
      for (int len = 1; len <= kFD; len++) {
        const float xx = ta[len], yy = ta[len - 1], zz = tz[len - 1], ww = tw[len - 1];
        s3 = len == 1 ? (double)(xx - yy + zz) : s3 + xx - yy + zz; // float arithmetic at length 1, as the reference
        s1 = len == 1 ? (double)ww : s1 + ww;
        if (isdir) {
          A.eq[len - 1][lane] = f0 ? s3 : s1;
          A.ed[len - 1][lane] = f0 ? s1 : s3;
        }
      }
    }
    if (isdir) {
      S.hyb[lane] = e_tot;
      S.info[0][lane] = FRec::pack(0, 0, lane, type0, q1, d1);
      S.improved[lane] = 0;
    }
    int nrec = 1;
    int pool_used = kFDirs; // (wave-uniform)
    // cells of anti-diagonals L - 2 and L - 1 whose stored type (Cell::type = the predecessor's) is not 0 / is a wobble
    uint32_t F2 = type0 != 0 ? 1u : 0u, W2 = wobble(type0) ? 1u : 0u, F1 = 0, W1 = 0;
    wave_mem_sync();
    FP_MARK(3);

    for (int LA = 2; LA <= kFD; LA += 2) {
      if (__ballot(!dead && LA <= lend) == 0) break;
      FP_COUNT(17, 1);
      // ---- per direction: the filled cells of anti-diagonals LA and LA + 1 (CheckHelixLength, :342-364) ----
      uint32_t fA = 0, fB = 0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int L = LA + h;
        uint32_t p0, wb0, ahead = ~0u;
        diag_masks(q, dr, L, p0, wb0);
        for (int xx = 1; xx <= min_helix - 1; xx++) { // (wave-uniform)
          uint32_t px, wbx;
          Planes qx{q.lo >> xx, q.hi >> xx, q.valid >> xx};
          diag_masks(qx, dr, L + xx, px, wbx); // bit i: (i + xx, L - i + xx)
          ahead &= px;
          if (xx == 1) ahead &= ~(wb0 & wbx);
        }
        const uint32_t Fp = h == 0 ? F2 : F1, Wp = h == 0 ? W2 : W1;
        const uint32_t need = ~(Fp << 1) | (wb0 & (Wp << 1));
        // cells 1 <= i <= L - 1 with i <= max_q and L - i <= max_d
        const int i_first = L - td0 + 1 > 1 ? L - td0 + 1 : 1;
        const int i_hi = (L < tq0 ? L : tq0) - 1;
        const uint32_t range = i_hi >= i_first ? (((i_hi >= 31 ? ~0u : ((1u << (i_hi + 1)) - 1))) & ~((1u << i_first) - 1)) : 0u;
        const uint32_t f = (!dead && L <= lend) ? (p0 & (ahead | ~need) & range) : 0u;
        if (h == 0) fA = f;
        else fB = f;
      }
      int cnt = __popc(fA) + __popc(fB);
      FP_MARK(4);
      if (nrec + cnt > kFR) { // more cells than a direction has room for: the cascade's business
        FP_COUNT(21, __popcll(__ballot(nrec + cnt > kFR)));
        dead = true;
        fA = fB = 0;
        cnt = 0;
      }
      int total;
      int base = wave_excl_scan(cnt, lane, total);
      if (total > kFCells || pool_used + total > kFPool) { // (next to never: 64 directions with 4+ cells each on two anti-diagonals, or 13+ each so far)
        if (base + cnt > kFCells || pool_used + base + cnt > kFPool) {
          dead = true;
          fA = fB = 0;
          cnt = 0;
        }
        base = wave_excl_scan(cnt, lane, total);
      }
      {
        int r = 0;
        for (uint32_t m = fA; m; m &= m - 1, r++) S.cells[base + r] = (uint32_t)lane | ((uint32_t)__builtin_ctz(m) << 8) | ((uint32_t)(nrec + r) << 16);
        for (uint32_t m = fB; m; m &= m - 1, r++)
          S.cells[base + r] = (uint32_t)lane | ((uint32_t)__builtin_ctz(m) << 8) | (1u << 13) | ((uint32_t)(nrec + r) << 16);
      }
      if (isdir) {
        S.wfp[0][lane] = 0;
        S.wfp[1][lane] = 0;
      }
      wave_lds_sync();
      FP_MARK(5);
      FP_COUNT(19, total);

      // ---- the cells of the step, 64 (or as many as have at most kFPairs candidates between them) at a time ----
      for (int c0 = 0; c0 < total;) {
        FP_COUNT(18, 1);
        const bool has = c0 + lane < total;
        const uint32_t cell = has ? S.cells[c0 + lane] : 0u;
        const int dl = cell & 63, ci = (cell >> 8) & 31, isB = (cell >> 13) & 1, rec = (int)(cell >> 16);
        const int cj = LA + isB - ci;
        const bool cf0 = !kOneDir && (dl & 1) == 0;
        // the direction's strands and list length, from its lane
        const uint32_t cq_lo = __shfl(q.lo, dl), cq_hi = __shfl(q.hi, dl), cq_v = __shfl(q.valid, dl);
        const uint32_t cd_lo = __shfl(dr.lo, dl), cd_hi = __shfl(dr.hi, dl), cd_v = __shfl(dr.valid, dl);
        const int nrecb = __shfl(nrec, dl);
        const int qc = plane_base(cq_lo, cq_hi, cq_v, ci), nq = plane_base(cq_lo, cq_hi, cq_v, ci > 0 ? ci - 1 : 0),
                  fq = plane_base(cq_lo, cq_hi, cq_v, ci + 1);
        const int sh = 31 - cj - 1; // positions cj + 1, cj, cj - 1 of the database strand -> bits 0, 1, 2
        const uint32_t dlo = cd_lo >> (sh & 31), dhi = cd_hi >> (sh & 31), dvv = cd_v >> (sh & 31);
        const int fd = plane_base(dlo, dhi, dvv, 0), dc = plane_base(dlo, dhi, dvv, 1), nd = plane_base(dlo, dhi, dvv, 2);
        int ctype = bp_type(sc, qc, dc);
        if (!cf0) ctype = rtype_of(ctype);
        uint32_t qmask = 0; // candidates: the direction's cells before this step with first < i && second < j (:232)
        // (the accessibility sums of the cell, for the improvement test two phases on: on their way meanwhile)
        double eqc = 0, edc = 0;
        if (has) {
          eqc = A.eq[ci - 1][dl];
          edc = A.ed[cj - 1][dl];
        }
        const int nscan = has ? nrecb : 0;
        for (int k = 0; __ballot(k < nscan) != 0; k += 2) { // (two records per turn: their reads are in flight together)
          const uint32_t v0 = S.info[k][dl], v1 = S.info[k + 1 < kFR ? k + 1 : k][dl];
          if (k < nscan && FRec::i(v0) < ci && FRec::j(v0) < cj) qmask |= 1u << k;
          if (k + 1 < nscan && FRec::i(v1) < ci && FRec::j(v1) < cj) qmask |= 2u << k;
        }
        const uint32_t cdw = (uint32_t)ci | ((uint32_t)cj << 5) | ((uint32_t)ctype << 10) | ((uint32_t)nq << 13) | ((uint32_t)nd << 16) |
                             ((uint32_t)dl << 19);
        const int pool = pool_used + c0 + lane; // the cell's place in the pool: its place in the step's list
        if (has) S.info[rec][dl] = FRec::pack(ci, cj, pool, rtype_of(ctype), fq, fd);
        FP_MARK(6);
        int npc = __popc(qmask), ptotal;
        int poff = wave_excl_scan(npc, lane, ptotal);
        // this round: the leading cells whose pairs fit (a cell has at most kFR of them, so at least 25 cells)
        const unsigned long long fits = __ballot(has && poff + npc <= kFPairs);
        const int ncr = fits == ~0ull ? 64 : __builtin_ctzll(~fits); // (the predicate is monotone in the lane; lane 0 always fits)
        const bool mine = lane < ncr;
        if (ncr < 64) ptotal = __shfl(poff, ncr); // pairs of the cells that take part
        if (mine) {
          int r = 0;
          for (uint32_t m = qmask; m; m &= m - 1, r++) S.pairs[poff + r] = cdw | ((uint32_t)__builtin_ctz(m) << 25);
        }
        wave_lds_sync();
        FP_MARK(7);
        FP_COUNT(20, ptotal);
        // ---- per pair: LoopEnergy + the candidate's energy (:233-247) ----
        for (int p0 = 0; p0 < ptotal; p0 += 64) {
          const int p = p0 + lane;
          if (p < ptotal) {
            const uint32_t cdv = S.pairs[p];
            const int k = cdv >> 25;
            const int pci = cdv & 31, pcj = (cdv >> 5) & 31, pct = (cdv >> 10) & 7, pnq = (cdv >> 13) & 7, pnd = (cdv >> 16) & 7,
                      pdl = (cdv >> 19) & 63;
            const bool pf0 = !kOneDir && (pdl & 1) == 0;
            const uint32_t v = S.info[k][pdl];
            const double hk = S.hyb[FRec::pool(v)]; // (needed last: the read is under way while the loop energy is looked up)
            const int ri = FRec::i(v), rj = FRec::j(v), rq = FRec::qa(v), rd = FRec::da(v), rt = FRec::type(v);
            double te = loop_energy_front(sc, S.tab, pf0 ? pct : rt, pf0 ? rt : pct, pci - ri - 1, pcj - rj - 1, pf0 ? pnq : rq,
                                          pf0 ? pnd : rd, pf0 ? rq : pnq, pf0 ? rd : pnd);
            te += hk;
            S.te[p] = te;
          }
        }
        wave_lds_sync();
        FP_MARK(8);
        // ---- per cell: the best candidate (strict '<': the first in list order wins), the new cell (:249-278) ----
        if (mine) {
          double bte = 1000000.0; // INF
          int bk = 0;
          int r = 0;
          for (uint32_t m = qmask; m; m &= m - 1, r++) {
            const double te = S.te[poff + r];
            if (te < bte) {
              bte = te;
              bk = __builtin_ctz(m);
            }
          }
          // (the start cell is a candidate of every cell, so there is a best one; its type is what the cell stores)
          const int ptype = FRec::type(S.info[bk][dl]);
          S.hyb[pool] = bte;
          if (ptype != 0) atomicOr(&S.wfp[isB][dl], (wobble(ptype) ? 0x10001u : 1u) << (ci - 1)); // (a cell has 1 <= i <= 16)
          const double ie = eqc + edc + bte;
          const double min_e = S.hyb[dl]; // the hit's energy (the start cell's entry): nothing has improved on it so far
          if (ie < min_e) S.improved[dl] = 1;
        }
        wave_lds_sync();
        FP_MARK(9);
        c0 += ncr;
      }
      // ---- per direction again: what the next anti-diagonals' checks need ----
      nrec += cnt;
      pool_used += total;
      {
        const uint32_t m0 = S.wfp[0][me], m1 = S.wfp[1][me];
        F2 = (m0 & 0xFFFFu) << 1;
        W2 = (m0 >> 16) << 1;
        F1 = (m1 & 0xFFFFu) << 1;
        W1 = (m1 >> 16) << 1;
      }
      FP_COUNT(22, __popcll(__ballot(!dead && S.improved[me] != 0)));
      if (S.improved[me]) dead = true;
      FP_MARK(10);
      // (the next step's LA - 2 and LA - 1 are this step's two anti-diagonals: F2 / W2 and F1 / W1 as just set)
    }

    FP_COUNT(23, pool_used > 640);
    FP_COUNT(24, pool_used > 704);
    FP_COUNT(25, pool_used > 768);
    FP_COUNT(26, pool_used);
    FP_MARK(11);
    // ---- the hit: done when neither direction finds anything (GappedExtension::Run tail, :49-67) ----
    bool ok = live && (!dead || (resumed && f0));
    bool ok_other;
    double dng = 0, dng_other = 0; // of the hit's first / second end, as the lane that writes the hit sees them
    if constexpr (kOneDir) {
      ok_other = ok;  // the second direction: this lane's
      ok = live && resumed; // the first one: done before
      if (live) {
        dng = dangle_energy_gapped(sc, q_sp, (int64_t)db_sp, 0, qs, qn, a.pg.seqs, a.pg.nchars);
        dng_other = dangle_energy_gapped(sc, q_sp + US(q_len) - 1, (int64_t)db_sp + US(db_len) - 1, 1, qs, qn, a.pg.seqs, a.pg.nchars);
      }
    } else {
      ok_other = __shfl_xor(ok ? 1 : 0, 1) != 0;
      if (live) {
        const int qp = f0 ? q_sp : q_sp + US(q_len) - 1;
        const int64_t dp = f0 ? (int64_t)db_sp : (int64_t)db_sp + US(db_len) - 1;
        dng = dangle_energy_gapped(sc, qp, dp, flag, qs, qn, a.pg.seqs, a.pg.nchars);
      }
      dng_other = __shfl_xor(dng, 1);
    }
    if (live && (kOneDir || f0)) {
      if (ok && ok_other) {
        const double d0 = dng, d1 = dng_other;
        double e = e_tot, hy = e_tot - e_acc;
        e += d0;
        e += d1;
        hy += d0;
        hy += d1;
        a.overflow[w] = 0;
        a.tier_out[x] = (uint8_t)(resumed ? tier0 : a.tier_id);
        a.bp_count[x] = nleft0;
        a.out.q_sp[x] = q_sp;
        a.out.db_sp[x] = db_sp;
        a.out.q_len[x] = q_len;
        a.out.db_len[x] = db_len;
        a.out.db_id[x] = id;
        a.out.db_id_start[x] = id_start;
        a.out.query[x] = query;
        a.out.e_acc[x] = e_acc;
        a.out.e_hyb[x] = hy;
        a.out.e_tot[x] = e;
      } else {
        a.overflow[w] = 1;
        if (resumed) {
          a.tier_out[x] = (uint8_t)(kResumeMark | tier0); // (the second direction finds something: the cascade's, from where it stands)
        } else if (ok) { // the first direction is done (it changes nothing): leave it for the next kernel, as the tiers do among themselves
          a.out.q_sp[x] = q_sp;
          a.out.db_sp[x] = db_sp;
          a.out.q_len[x] = q_len;
          a.out.db_len[x] = db_len;
          a.out.db_id_start[x] = id_start;
          a.out.e_acc[x] = e_acc;
          a.out.e_hyb[x] = e_tot - e_acc;
          a.out.e_tot[x] = e_tot;
          a.bp_count[x] = 0;
          a.tier_out[x] = kResumeMark;
        }
      }
    }
    wave_mem_sync(); // (the accessibility block is written again for the next tile)
    tile = nwaves + (int64_t)__shfl(nt, 0);
    FP_MARK(12);
  }
  FP_FLUSH();
}

} // namespace

#ifdef PRB_GAP_PROFILE
extern "C" int prb_debug_front_profile(unsigned long long *out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_front_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_front_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

bool gapped_front_supported(const SearchConst &sc, const ExtOpts &o) {
  // the bit-parallel cell check is written for the standard pairing rule (A-U, C-G, G-U; wobble = G-U) with bases
  // 1..4 = A, C, G, U; 32 positions of a strand are held: kFD anti-diagonals + the look-ahead of the helix check
  uint32_t pm = 0, wm = 0;
  auto set = [&](uint32_t &m, int q, int d) { m |= 1u << (q * 5 + d); };
  set(pm, 1, 4), set(pm, 4, 1), set(pm, 2, 3), set(pm, 3, 2), set(pm, 3, 4), set(pm, 4, 3);
  set(wm, 3, 4), set(wm, 4, 3);
  return sc.pair_mask == pm && sc.wobble_mask == wm && o.min_helix >= 1 && o.min_helix <= 7 && o.drop_w_gap >= 1 &&
         o.drop_w_gap <= kFD;
}

static int front_blocks_per_cu() {
  const size_t granule = 2048, lds = (sizeof(FrontLds) * kFWaves + granule - 1) / granule * granule;
  return std::max<int>(1, std::min<int>(12, (int)((160 * 1024) / lds))); // (133 registers: three wavefronts per SIMD)
}
size_t gapped_front_scratch_bytes() { return (size_t)256 * front_blocks_per_cu() * kFWaves * sizeof(FrontAcc); }

hipError_t launch_gapped_front(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                               const SearchConst &sc, ExtOpts o, int tier_id, uint8_t *overflow, uint8_t *tier_out, int32_t *bp_count,
                               unsigned long long *next_work, void *scratch, hipStream_t s, bool second_only) {
  if (n <= 0) return hipSuccess;
  GapArgs a{in,      out,     n,       subset,    qb, pg, sc, o, overflow, tier_out, tier_id, nullptr, bp_count,
            nullptr, nullptr, nullptr, next_work, GapResume{nullptr, nullptr, nullptr, 0}, GapResume{nullptr, nullptr, nullptr, 0}};
  if (hipError_t e = hipMemsetAsync(next_work, 0, sizeof(unsigned long long), s); e != hipSuccess) return e;
  const int per_tile = second_only ? kFDirs : kFDirs / 2;
  const int64_t tiles = (n + per_tile - 1) / per_tile, want_blocks = (tiles + kFWaves - 1) / kFWaves;
  const dim3 grid((unsigned)std::min<int64_t>(want_blocks, 256 * front_blocks_per_cu())), blk(64 * kFWaves);
  if (second_only) hipLaunchKernelGGL(k_gapped_front<true>, grid, blk, 0, s, a, static_cast<FrontAcc *>(scratch));
  else hipLaunchKernelGGL(k_gapped_front<false>, grid, blk, 0, s, a, static_cast<FrontAcc *>(scratch));
  return hipGetLastError();
}

} // namespace prb
