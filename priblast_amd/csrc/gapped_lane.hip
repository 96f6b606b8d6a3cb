// Gapped extension, first kernel of the cascade: ONE LANE PER HIT (GappedExtension::Run / extension /
// CheckHelixLength / traceback, gapped_extension.cpp:33-473).
//
// Why: an extension is tiny - nine directions in ten find nothing and end after exactly `-x` (16)
// anti-diagonals with ~11 filled cells, each looking at ~5 candidates - so a group of 8 or more lanes
// per hit (gapped_lds.hip) spends most of its instructions on idle lanes, on reductions across the
// group and on waiting for the busiest group of its wavefront (round 1: 3,074 wave-level VALU
// instructions per hit, ~30 % useful fill work).  Here a lane owns a hit and nothing crosses lanes:
//   * the cells of a whole anti-diagonal are checked at once with bit-parallel masks: the two strands
//     are three bit planes each (low / high bit of the base, valid), the database side bit-reversed, so
//     that "q[i] pairs d[len - i]" for all i is a shift and a few logic operations; CheckHelixLength's
//     look-ahead uses the masks of anti-diagonals len + 2 and len + 4 shifted, its predecessor test two
//     masks of typed cells of anti-diagonal len - 2;
//   * the lanes run free: a lane's unit of work is "one candidate of one filled cell" (plus, when it has
//     no cell pending, the step to its next anti-diagonal), so no lane waits for another lane's cells;
//     only the changes of direction / hit happen at common iterations (every kPeriod), because their code
//     is long and would otherwise be walked through by the whole wavefront at every iteration;
//   * state per lane: the records of the filled-cell list (4 bytes per cell) in LDS, transposed ([slot][lane]:
//     conflict-free whatever slot a lane is at) - 12 KB per wavefront; the cells' energies and the two cumulative
//     accessibility arrays in a per-wavefront block of HBM that stays in L2 (accessibility sums read one cell ahead
//     of their use); everything else in registers - 160 of them, i.e. three wavefronts per SIMD.
// STATUS (round 2): correct (bit-identical, see below) but NOT faster than the LDS tiers, so the cascade uses it
// only when PRB_GAPPED_LANE is set.  Measured on MI355X, 37.7 M hits (profiles/r02_gapped_lane.md): this form
// (27 anti-diagonals / 48 cells per direction) completes 86 % of the hits in 186 ms and leaves tier 0 50 ms of
// work: 236 ms against tier 0's 190 ms on everything.  It needs ~1,300 wave-level VALU instructions per hit where
// tier 0 needs 3,074, but at ~25 % of the lanes active (advance / candidate / close are three different pieces of
// code and every iteration walks through all three) and 60 % VALU busy.  Forms tried on the way: state in LDS and
// 256 registers (one wavefront per SIMD: 233 ms for 20 / 24 capacities - latency bound, time inversely
// proportional to the resident wavefronts); a vote for ONE piece per iteration (half the instructions, but every
// lane waits for its piece to come up: 261 ms); cell lists pooled per wavefront in 4-cell chunks (the pool that
// fits is too small for 64 live lists: 53 % complete).  What took it from one to three wavefronts per SIMD was a
// dynamically indexed register array in the window set-up (the extension's direction differs per lane): memory
// order + v_bfrev instead, 256 -> 160 registers.  To beat tier 0 clearly it would need half its instructions
// again (cheaper changes of direction: ~300 of the 1,300; the anti-diagonal masks kept incrementally).
// A direction that outgrows the capacities (kD anti-diagonals, kR cells) is left to the LDS tiers of
// gapped_lds.hip: the hit is flagged, and if its first direction was completed here that result is
// handed over (kResumeMark) exactly as between those tiers.  Results are bit-identical to them
// (tests/test_gpu_search.py::test_fallback_kernels_match_tier1 runs the cascade with and without).
#include <algorithm>
#include <cstdlib>

#include "gapped_args.hpp"
#include "search_device.hpp"
#include "search_kernels.hpp"

namespace prb {
namespace {

constexpr int kPeriod = 16; // iterations between two points where lanes change direction / hit

template <int D, int R> struct LaneLds { // the records of the cell lists; per wavefront
  uint32_t info[R][64];
};
// the cumulative accessibility arrays of a wavefront's 64 directions, in HBM / L2 (written once per direction,
// read twice per filled cell, one cell ahead of their use): eq[slot][lane], ed[slot][lane]
template <int D, int R = kLaneCapR> struct LaneAcc {
  double eq[D][64], ed[D][64];
  double hyb[R][64]; // the cells' energies: written once, read once per later cell that has it as a candidate
};

// the bases of one strand along the extension: bit t of lo / hi / valid describes position t
// (A = hi, C = hi | lo, G = 0, U = lo; valid = GetChar != 0)
struct Planes {
  uint32_t lo, hi, valid;
};
__device__ __forceinline__ int plane_base(const Planes &p, int t) { // 0 = none, 1..4 = A, C, G, U
  const uint32_t l = (p.lo >> t) & 1, h = (p.hi >> t) & 1, v = (p.valid >> t) & 1;
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
#pragma unroll
  for (int u = 0; u < 32; u++) { // in memory order (static register indices); a leftward extension is the mirror image
    const uint32_t c = (x[u >> 2] >> ((u & 3) * 8)) & 0xFF;
    const int64_t pos = first + u;
    const uint32_t ok = (pos >= 0 && pos < n && c >= 2) ? 1u : 0u;
    p.lo |= (c & 1) << u;
    p.hi |= ((c >> 1) & 1) << u;
    p.valid |= ok << u;
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

struct LaneState {
  // the hit
  int64_t w, x, qo, dbase;
  HitState h;
  int query, id, qn;
  int nleft, nright;
  // the running direction
  Planes q, d;       // d: bit-reversed (bit 31 - t = position t)
  uint32_t t0, t1, t2, w0m, w1m, w2m; // typed cells / wobble-typed cells of anti-diagonals length, length - 1, length - 2
  uint32_t pend;     // filled cells of the current anti-diagonal still to be done
  double min_e, acc_q, acc_d, bte, eqc, edc; // eqc / edc: eq[ci - 1], ed[cj - 1] of the current cell
  int flag, length, lo, nrec, dstart, k, best, min_ci, min_cj, tq0, td0, staged, bkp;
  bool overflow;
};

// accessibility sums of extension length len (1-based), appended to eq / ed
template <int D>
__device__ __forceinline__ void lane_stage(const GapArgs &a, LaneAcc<D> &A, int lane, LaneState &st, int len) {
  const float *qacc = a.qb.acc + st.qo, *qcond = a.qb.cond + st.qo, *dacc = a.pg.acc + st.dbase, *dcond = a.pg.cond + st.dbase;
  const DirOrigin og = dir_origin(st.h, st.flag);
  const int delta = a.o.delta;
  if (len < st.tq0) {
    double v;
    if (st.flag == 0) {
      const int p = og.q_start - len;
      const float x = qacc[p], y = qacc[p + 1], z = qcond[p + delta];
      v = len == 1 ? (double)(x - y + z) : st.acc_q + x - y + z; // float arithmetic at length 1, as the reference
    } else {
      const float x = qcond[og.q_start + len];
      v = len == 1 ? (double)x : st.acc_q + x;
    }
    st.acc_q = v;
    A.eq[len - 1][lane] = v;
  }
  if (len < st.td0) {
    double v;
    if (st.flag == 0) {
      const float x = dcond[og.id_end + len];
      v = len == 1 ? (double)x : st.acc_d + x;
    } else {
      const int p = og.id_start - len;
      const float x = dacc[p], y = dacc[p + 1], z = dcond[p + delta];
      v = len == 1 ? (double)(x - y + z) : st.acc_d + x - y + z;
    }
    st.acc_d = v;
    A.ed[len - 1][lane] = v;
  }
  st.staged = len;
}

template <int D, int R>
__device__ __forceinline__ void lane_dir_init(const GapArgs &a, LaneLds<D, R> &S, LaneAcc<D> &A, int lane, LaneState &st) {
  const SearchConst &sc = a.sc;
  const uint8_t *qs = a.qb.enc + st.qo;
  const DirOrigin og = dir_origin(st.h, st.flag);
  const int dir = st.flag == 0 ? -1 : 1;
  st.q = load_planes(qs, st.qn, og.q_start, dir);
  Planes dd = load_planes(a.pg.seqs, a.pg.nchars, og.db_start, dir);
  st.tq0 = (st.q.valid >> 1) == 0x7FFFFFFFu ? 32 : __builtin_ctz(~(st.q.valid >> 1)) + 1;
  st.td0 = (dd.valid >> 1) == 0x7FFFFFFFu ? 32 : __builtin_ctz(~(dd.valid >> 1)) + 1;
  const int q0 = plane_base(st.q, 0), d0 = plane_base(dd, 0), q1 = plane_base(st.q, 1), d1 = plane_base(dd, 1);
  st.d.lo = __builtin_bitreverse32(dd.lo);
  st.d.hi = __builtin_bitreverse32(dd.hi);
  st.d.valid = __builtin_bitreverse32(dd.valid);
  st.min_e = st.h.e_tot;
  st.min_ci = 0;
  st.min_cj = 0;
  st.length = 0;
  st.best = 0;
  st.overflow = false;
  st.pend = 0;
  // cumulative accessibility change (gapped_extension.cpp:156-212) of the first 16 lengths - nearly every
  // direction ends within them; further ones as they are reached
  st.acc_q = 0;
  st.acc_d = 0;
  st.staged = 0;
#pragma unroll 2
  for (int len = 1; len <= (D < 16 ? D : 16); len++) lane_stage<D>(a, A, lane, st, len); // (registers: the loads of two lengths in flight together)
  int type0 = bp_type(sc, q0, d0);
  if (st.flag == 0) type0 = rtype_of(type0);
  A.hyb[0][lane] = st.min_e;
  S.info[0][lane] = Rec32::pack(0, 0, 0, type0, q1, d1);
  st.t0 = type0 != 0 ? 1u : 0u; // cell (0, 0) lies on anti-diagonal 0
  st.w0m = wobble(type0) ? 1u : 0u;
  st.t1 = st.t2 = st.w1m = st.w2m = 0;
  st.nrec = 1;
  st.lo = 0;
}

// pairs / wobble pairs of all cells (i, len - i) of anti-diagonal len as masks over i
__device__ __forceinline__ void diag_masks(const LaneState &st, int len, uint32_t &pairs, uint32_t &wob) {
  const int s = 31 - len; // bit i of (d >> s) = position len - i of the database strand
  const uint32_t dl = st.d.lo >> s, dh = st.d.hi >> s, dv = st.d.valid >> s;
  const uint32_t both = st.q.valid & dv, ldiff = st.q.lo ^ dl;
  const uint32_t gu = ~(st.q.hi | dh); // G or U on both sides
  wob = both & ldiff & gu;
  pairs = both & ldiff & ((st.q.hi ^ dh) | gu);
}

constexpr int kLaneWaves = 4; // wavefronts per workgroup (independent of each other)
template <int D, int R>
__global__ __launch_bounds__(64 * kLaneWaves, 3) void k_gapped_lane(GapArgs a) {
  __shared__ LaneLds<D, R> lds[kLaneWaves];
  const int lane = threadIdx.x & 63;
  LaneLds<D, R> &S = lds[threadIdx.x >> 6];
  const int64_t wave_id = (int64_t)blockIdx.x * kLaneWaves + (threadIdx.x >> 6);
  LaneAcc<D> &A = reinterpret_cast<LaneAcc<D> *>(a.lane_scratch)[wave_id];
  const SearchConst &sc = a.sc;
  const int drop = a.o.drop_w_gap, min_helix = a.o.min_helix;
  enum { kLoad, kInit, kRun, kFinished, kDone };
  LaneState st;
  st.w = wave_id * 64 + lane;
  const int64_t nlanes = (int64_t)gridDim.x * kLaneWaves * 64;
  int phase = st.w < a.n ? kLoad : kDone;
  int tick = 0;
  while (__ballot(phase != kDone) != 0) {
    if (tick == 0) {
      if (phase == kFinished) {
        // ---- the direction is over (gapped_extension.cpp:300-318) ----
        const DirOrigin og = dir_origin(st.h, st.flag);
        (void)og;
        int cnt = 0;
        if (!st.overflow) {
          const int best = st.min_ci != 0 ? st.best : 0;
          if (st.min_ci != 0) {
            st.h.e_acc = st.h.e_acc + A.eq[st.min_ci - 1][lane] + A.ed[st.min_cj - 1][lane];
            if (st.flag == 0) {
              st.h.q_sp -= st.min_ci;
              st.h.db_sp -= st.min_cj;
            } else {
              st.h.id_start -= st.min_cj;
            }
            st.h.q_len += st.min_ci;
            st.h.db_len += st.min_cj;
          }
          st.h.e_tot = st.min_e;
          // traceback through the predecessors: the chain goes to the hit's trace slot
          for (int k = best; k != 0; cnt++) {
            const uint32_t v = S.info[k][lane];
            if (cnt < kTraceCap) a.trace[(st.x * 2 + st.flag) * kTraceCap + cnt] = (uint16_t)(Rec32::i(v) | (Rec32::j(v) << 8));
            k = Rec32::pred(v);
          }
          if (st.flag == 0) st.nleft = cnt;
          else st.nright = cnt;
        }
        if (st.overflow || st.flag == 1) {
          // ---- the hit is over (or handed on) ----
          const int64_t x = st.x;
          const HitState &h = st.h;
          a.overflow[st.w] = st.overflow ? 1 : 0;
          if (st.overflow && st.flag == 1) { // direction 0 is done: leave it for the next kernel
            a.out.q_sp[x] = h.q_sp;
            a.out.db_sp[x] = h.db_sp;
            a.out.q_len[x] = h.q_len;
            a.out.db_len[x] = h.db_len;
            a.out.db_id_start[x] = h.id_start;
            a.out.e_acc[x] = h.e_acc;
            a.out.e_hyb[x] = h.e_tot - h.e_acc;
            a.out.e_tot[x] = h.e_tot;
            a.bp_count[x] = st.nleft;
            a.tier_out[x] = kResumeMark;
          }
          if (!st.overflow) {
            const uint8_t *ds = a.pg.seqs, *qs = a.qb.enc + st.qo;
            a.tier_out[x] = (uint8_t)a.tier_id;
            a.bp_count[x] = st.nleft | (st.nright << 16);
            // GappedExtension::Run tail (gapped_extension.cpp:49-67): dangling ends on both sides
            const double d0 = dangle_energy_gapped(sc, h.q_sp, h.db_sp, 0, qs, st.qn, ds, a.pg.nchars);
            const double d1 = dangle_energy_gapped(sc, h.q_sp + US(h.q_len) - 1, (int64_t)h.db_sp + US(h.db_len) - 1, 1, qs, st.qn, ds,
                                                   a.pg.nchars);
            double e = h.e_tot, hy = h.e_tot - h.e_acc;
            e += d0;
            e += d1;
            hy += d0;
            hy += d1;
            a.out.q_sp[x] = h.q_sp;
            a.out.db_sp[x] = h.db_sp;
            a.out.q_len[x] = h.q_len;
            a.out.db_len[x] = h.db_len;
            a.out.db_id[x] = st.id;
            a.out.db_id_start[x] = h.id_start;
            a.out.query[x] = st.query;
            a.out.e_acc[x] = h.e_acc;
            a.out.e_hyb[x] = hy;
            a.out.e_tot[x] = e;
          }
          // next hit: one atomic per wavefront, the lanes that need one take consecutive numbers
          const unsigned long long want = __ballot(true);
          const int rank = __popcll(want & ((1ull << lane) - 1));
          unsigned long long base = 0;
          if (rank == 0) base = atomicAdd(a.next_work, (unsigned long long)__popcll(want));
          base = __shfl(base, __builtin_ctzll(want));
          st.w = nlanes + (int64_t)base + rank;
          phase = st.w < a.n ? kLoad : kDone;
        } else {
          st.flag = 1;
          phase = kInit;
        }
      }
      if (phase == kLoad) {
        const int64_t x = a.subset ? (int64_t)a.subset[st.w] : st.w;
        st.x = x;
        st.query = a.in.query[x];
        st.qo = a.qb.off[st.query];
        st.qn = a.qb.len[st.query] + 1;
        st.id = a.in.db_id[x];
        st.dbase = (int64_t)a.pg.start_pos[st.id] - st.id;
        st.h.q_sp = a.in.q_sp[x];
        st.h.db_sp = a.in.db_sp[x];
        st.h.q_len = a.in.q_len[x];
        st.h.db_len = a.in.db_len[x];
        st.h.id_start = a.in.db_id_start[x];
        st.h.e_tot = a.in.e_tot[x];
        st.h.e_acc = a.in.e_acc[x];
        st.nleft = 0;
        st.nright = 0;
        st.flag = 0;
        phase = kInit;
      }
      if (phase == kInit) {
        lane_dir_init<D, R>(a, S, A, lane, st);
        phase = kRun;
      }
    }
    tick = tick + 1 == kPeriod ? 0 : tick + 1;

    // Every iteration walks through all three pieces (a vote for ONE piece per iteration was tried: fewer instructions,
    // but every lane then waits for its piece to come up - slower).
    const bool want_a = phase == kRun && st.pend == 0;
    {
      // ---- A: on to the next anti-diagonal (gapped_extension.cpp:129-219, 292-297) ----
      if (want_a) {
        bool stop = false;
        if (st.length >= 1) {
          if (st.length - (st.min_ci + st.min_cj) >= drop) stop = true;
          if (!(st.length < st.tq0) && !(st.length < st.td0)) stop = true;
        }
        if (!stop && st.length >= D) { // no room for another anti-diagonal
          st.overflow = true;
          stop = true;
        }
        if (stop) {
          phase = kFinished;
        } else {
          st.length++;
          const int len = st.length;
          if (len > st.staged) lane_stage<D>(a, A, lane, st, len);
          st.t2 = st.t1;
          st.t1 = st.t0;
          st.t0 = 0;
          st.w2m = st.w1m;
          st.w1m = st.w0m;
          st.w0m = 0;
          // prune candidates with len - first - second - 2 > drop (:213-217): a prefix of the list
          if (len - 2 > drop) {
            while (st.lo < st.nrec) {
              const uint32_t v = S.info[st.lo][lane];
              if (len - Rec32::i(v) - Rec32::j(v) - 2 > drop) st.lo++;
              else break;
            }
          }
          // CheckHelixLength (:342-364) for every cell of the anti-diagonal at once
          uint32_t p0, wb0, p1, wb1, p2, wb2;
          diag_masks(st, len, p0, wb0);
          uint32_t ahead = ~0u;
          if (min_helix >= 2) {
            diag_masks(st, len + 2, p1, wb1);
            ahead = (p1 >> 1) & ~(wb0 & (wb1 >> 1));
          }
          if (min_helix >= 3) {
            diag_masks(st, len + 4, p2, wb2);
            ahead &= p2 >> 2;
          }
          if (min_helix >= 4) { // (the host launches this kernel for -m <= 4 only)
            uint32_t p3, wb3;
            diag_masks(st, len + 6, p3, wb3);
            ahead &= p3 >> 3;
          }
          const uint32_t need = ~(st.t2 << 1) | (wb0 & (st.w2m << 1));
          // cells 1 <= i <= len - 1 with i <= max_q and len - i <= max_d
          const int i_first = len - st.td0 + 1 > 1 ? len - st.td0 + 1 : 1;
          const int i_hi = (len < st.tq0 ? len : st.tq0) - 1;
          const uint32_t range = i_hi >= i_first ? (((i_hi >= 31 ? ~0u : ((1u << (i_hi + 1)) - 1))) & ~((1u << i_first) - 1)) : 0u;
          st.pend = p0 & (ahead | ~need) & range;
          st.dstart = st.nrec;
          st.k = st.lo;
          st.bte = 1000000.0; // INF
          st.bkp = st.lo << 3;
          if (st.pend != 0) { // the first cell's accessibility sums, on their way while its candidates are looked at
            const int ci = __builtin_ctz(st.pend);
            st.eqc = A.eq[ci - 1][lane];
            st.edc = A.ed[len - ci - 1][lane];
          }
        }
      }
    }
    {
      // ---- E: one candidate of the first pending cell (:220-254) ----
      if (phase == kRun && st.pend != 0 && st.k < st.dstart) {
        const int ci = __builtin_ctz(st.pend), cj = st.length - ci;
        const bool f0 = st.flag == 0;
        const uint32_t v = S.info[st.k][lane];
        const double hk = A.hyb[st.k][lane];
        const int ri = Rec32::i(v), rj = Rec32::j(v);
        if (ri < ci && rj < cj) {
          // the bases of the new pair and next to it on the loop side (bit 31 - t of the database planes = position t)
          const int q0 = plane_base(st.q, ci), nq = plane_base(st.q, ci - 1);
          Planes dfw;
          const int sh = 31 - cj - 1; // positions cj + 1, cj, cj - 1 -> bits 0, 1, 2
          dfw.lo = st.d.lo >> sh;
          dfw.hi = st.d.hi >> sh;
          dfw.valid = st.d.valid >> sh;
          const int nd = plane_base(dfw, 2), d0 = plane_base(dfw, 1);
          int ctype = bp_type(sc, q0, d0);
          if (!f0) ctype = rtype_of(ctype);
          const int rq = Rec32::qa(v), rd = Rec32::da(v), rt = Rec32::type(v);
          double te = loop_energy_abcd(sc, f0 ? ctype : rt, f0 ? rt : ctype, ci - ri - 1, cj - rj - 1, f0 ? nq : rq, f0 ? nd : rd,
                                       f0 ? rq : nq, f0 ? rd : nd);
          te += hk;
          if (te < st.bte) { // strict '<': the first candidate in list order wins
            st.bte = te;
            st.bkp = (st.k << 3) | rt;
          }
        }
        st.k++;
      }
    }
    {
      // ---- F: the cell is done: its record, the running minimum (:256-278) ----
      if (phase == kRun && st.pend != 0 && st.k >= st.dstart) {
        if (st.nrec >= R) {
          st.overflow = true;
          phase = kFinished;
        } else {
          const int ci = __builtin_ctz(st.pend), cj = st.length - ci;
          const int q0 = plane_base(st.q, ci), fq = plane_base(st.q, ci + 1);
          Planes dfw;
          const int sh = 31 - cj - 1;
          dfw.lo = st.d.lo >> sh;
          dfw.hi = st.d.hi >> sh;
          dfw.valid = st.d.valid >> sh;
          const int d0 = plane_base(dfw, 1), fd = plane_base(dfw, 0);
          int ctype = bp_type(sc, q0, d0);
          if (st.flag != 0) ctype = rtype_of(ctype);
          int bk = st.bkp >> 3, ptype = st.bkp & 7;
          if (st.lo >= st.dstart) bk = 0;                        // empty window: the reference reads stem_candidate[0] of an empty list
          if (ptype == 0) ptype = Rec32::type(S.info[bk][lane]); // no candidate qualified: the type of that default entry
          A.hyb[st.nrec][lane] = st.bte;
          S.info[st.nrec][lane] = Rec32::pack(ci, cj, bk, rtype_of(ctype), fq, fd);
          st.t0 |= (ptype != 0 ? 1u : 0u) << ci;
          st.w0m |= (wobble(ptype) ? 1u : 0u) << ci;
          const double ie = st.eqc + st.edc + st.bte;
          if (ie < st.min_e) {
            st.min_e = ie;
            st.best = st.nrec;
            st.min_ci = ci;
            st.min_cj = cj;
          }
          st.nrec++;
          st.pend &= st.pend - 1;
          st.k = st.lo;
          st.bte = 1000000.0;
          st.bkp = st.lo << 3;
          if (st.pend != 0) {
            const int ni = __builtin_ctz(st.pend);
            st.eqc = A.eq[ni - 1][lane];
            st.edc = A.ed[st.length - ni - 1][lane];
          }
        }
      }
    }
  }
}

} // namespace

bool gapped_lane_supported(const SearchConst &sc, const ExtOpts &o) {
  // the bit-parallel cell check is written for the standard pairing rule (A-U, C-G, G-U; wobble = G-U) with bases
  // 1..4 = A, C, G, U, looks 3 pairs ahead at most, and holds 32 positions of a strand
  uint32_t pm = 0, wm = 0;
  auto set = [&](uint32_t &m, int q, int d) { m |= 1u << (q * 5 + d); };
  set(pm, 1, 4), set(pm, 4, 1), set(pm, 2, 3), set(pm, 3, 2), set(pm, 3, 4), set(pm, 4, 3);
  set(wm, 3, 4), set(wm, 4, 3);
  return sc.pair_mask == pm && sc.wobble_mask == wm && o.min_helix >= 1 && o.min_helix <= 4 && o.drop_w_gap >= 1 &&
         o.drop_w_gap + 1 <= 18;
}

size_t gapped_lane_scratch_bytes() { return (size_t)256 * 16 * sizeof(LaneAcc<kLaneCapD>); } // a block per resident wavefront

hipError_t launch_gapped_lane(const HitSoA &in, HitSoA out, int64_t n, const uint32_t *subset, const QBatchDev &qb, const PageDev &pg,
                              const SearchConst &sc, ExtOpts o, int tier_id, uint8_t *overflow, uint8_t *tier_out, int32_t *bp_count,
                              uint16_t *trace, unsigned long long *next_work, void *scratch, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  GapArgs a{in,       out,   n,       subset,  qb,        pg,      sc, o, overflow, tier_out, tier_id, nullptr, bp_count,
            trace,    nullptr, nullptr, next_work, GapResume{nullptr, nullptr, nullptr, 0}, GapResume{nullptr, nullptr, nullptr, 0}};
  a.lane_scratch = scratch;
  const int64_t want = (n + 63) / 64;
  if (hipError_t e = hipMemsetAsync(next_work, 0, sizeof(unsigned long long), s); e != hipSuccess) return e;
  {
    constexpr size_t lds = sizeof(LaneLds<kLaneCapD, kLaneCapR>) * kLaneWaves;
    const int blocks_per_cu = std::min<int>(12 / kLaneWaves, (int)((160 * 1024) / lds)); // three wavefronts per SIMD (160 VGPRs)
    const int64_t want_blocks = (want + kLaneWaves - 1) / kLaneWaves;
    const dim3 grid((unsigned)std::min<int64_t>(want_blocks, 256 * blocks_per_cu)), blk(64 * kLaneWaves);
    hipLaunchKernelGGL((k_gapped_lane<kLaneCapD, kLaneCapR>), grid, blk, 0, s, a);
  }
  return hipGetLastError();
}

} // namespace prb
