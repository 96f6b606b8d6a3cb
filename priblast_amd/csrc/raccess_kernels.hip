// Raccess on gfx950: windowed McCaskill inside/outside and the accessibility of every
// length-delta window (what Raccess::Run(seq, acc, cond) computes, raccess.cpp:42-50).
//
// Mapping (DESIGN.md "Raccess kernels"): one 64-lane wavefront owns one sequence and walks
// the DP column by column (column = end position); the <= W-1 cells of a column sit on the
// lanes (64 per pass).  A column is processed as phases that are parallel over cells, in
// the dependency order derived in SURVEY.md 8(a'); every sum is folded in the reference's
// order by the lane that owns the cell (bit-exact), and the (u1,u2) interior-loop
// enumeration is wave-uniform so the loop-energy branch never diverges.  Band tables live in
// HBM in END-MAJOR layout T[end][span], so the lanes of a column read and write contiguous
// doubles; the exp/log tables and the small Turner tables are staged in LDS once per
// workgroup (4 waves = 4 sequences share one image).
//
//   k_inside   <-> Raccess::CalcInsideVariable   raccess.cpp:99-242
//   k_outside  <-> Raccess::CalcOutsideVariable  raccess.cpp:258-412
//   k_biloop   <-> CalcBulgeAndInternalProbability / CalcLogSum...  raccess.cpp:614-771
//   k_access   <-> CalcAccessibility + Exterior/Hairpin/Multi probabilities  raccess.cpp:484-612
#include "raccess_kernels.hpp"

#include "raccess_device.hpp"

namespace prb {

namespace {

// Developer-only cycle breakdown per phase (make prof; tools/raccess_profile.py): wave-cycles of the first
// wavefront of a launch, accumulated by lane 0 at every phase boundary.  Not part of the product build.
#ifdef PRB_GAP_PROFILE
__device__ unsigned long long g_ra_prof[32];
#define RA_COUNT(k, v)            \
  do {                            \
    if (ra_p_) g_ra_prof[k] += (v); \
  } while (0)
#define RA_PROF_DECL unsigned long long ra_t0_ = __builtin_amdgcn_s_memtime(); const bool ra_p_ = blockIdx.x == 0 && threadIdx.x == 0
#define RA_PROF(k)                                                  \
  do {                                                              \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
    if (ra_p_) g_ra_prof[k] += t_ - ra_t0_;                         \
    ra_t0_ = t_;                                                    \
  } while (0)
#else
#define RA_PROF_DECL
#define RA_PROF(k)
#define RA_COUNT(k, v)
#endif

constexpr int kWave = 64;
#ifndef PRB_RA_AHEAD // (throughput-mode experiments: terms fetched ahead, wavefronts per workgroup, wavefronts per SIMD asked of the compiler)
#define PRB_RA_AHEAD 8
#define PRB_RA_WPB 4
#define PRB_RA_WPS 3
#endif
constexpr int kRaAhead = PRB_RA_AHEAD; // interior-loop terms fetched ahead of the fold in k_inside / k_outside
constexpr int kHB = 8;      // ... per batch when helper wavefronts prepare them (a barrier per batch: fewer, longer rounds)
constexpr int kWavesPerBlock = PRB_RA_WPB;
constexpr int kBlock = kWave * kWavesPerBlock;

// band table ids inside one sequence's workspace block
enum : int {
  A_STEM = 0, A_STEMEND, A_MULTI, A_MULTIBIF, A_MULTI1, A_MULTI2, A_MULTI1_T,
  B_STEM, B_STEMEND, B_MULTI, B_MULTIBIF, B_MULTI1, B_MULTI2
};
static_assert(B_MULTI2 + 1 == kRaBands, "band count");
enum : int { V_AO = 0, V_BO, V_BP, V_CBP, V_BFLAG, V_CFLAG };

struct SeqView {
  int L, W, S;  // S = W + 2 = row length of a band
  int64_t rows; // L + 2
  double *band;
  double *vec;
  const unsigned char *s; // codes 0..4: s[0] = 0, s[1..L], s[L+1] = 0
  int64_t out_off;
  __device__ __forceinline__ double *tab(int t) const { return band + (int64_t)t * rows * S; }
  __device__ __forceinline__ double *v(int t) const { return vec + (int64_t)t * rows; }
};

__device__ __forceinline__ SeqView make_view(const RaBatch &b, int idx) {
  const RaSeqDesc d = b.desc[idx];
  SeqView v;
  v.L = d.L;
  v.W = b.W;
  v.S = b.W + 2;
  v.rows = (int64_t)d.L + 2;
  v.band = b.band + d.band_off;
  v.vec = b.vec + d.vec_off;
  v.s = b.codes + d.code_off;
  v.out_off = d.out_off;
  return v;
}

// element (start, end) of an end-major band: row `end`, column `end - start`
#define EM(tab, start, end) ((tab)[(int64_t)(end) * S + ((end) - (start))])
// start-major copy (alpha_multi1 only): row `start`, column `end - start`
#define SM(tab, start, end) ((tab)[(int64_t)(start) * S + ((end) - (start))])

// Orders this wave's global stores before its later loads issued by other lanes.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Raccess::CalcDangleEnergy, raccess.cpp:244-256
__device__ __forceinline__ double dangle_energy(const RaLds &lds, const SeqView &v, int type, int a, int b) {
  using SL = RaSmallLayout;
  double x = 0;
  if (type != 0) {
    if (a > 0) x += lds.small[SL::kDangle5 + type * 5 + v.s[a]];
    if (b < v.L) x += lds.small[SL::kDangle3 + type * 5 + v.s[b + 1]];
    if (b == v.L && type > 2) x += lds.small[SL::kTermAU];
  }
  return x;
}

// Which cells of a band table hold a value: one 128-bit mask per START position (bit = span), kept in LDS for
// the last 128 starts.  The two big folds (Alpha_stemend over the enclosed stems, Beta_stem over the enclosing
// pairs) walk their <= 496 (u1, u2) terms in the reference's order; only the terms whose table entry is not
// -INF take part (about one in three for the cells that fold at all), and a lane finds them as the set bits of a
// window of the row masks instead of testing every term.
struct RowMasks {
  uint32_t m[128][5]; // 128 bits + a zero word, so that a 32-bit window may start in the last word
};
__device__ __forceinline__ void rowmask_clear(RowMasks &r, int start) {
#pragma unroll
  for (int k = 0; k < 5; k++) r.m[start & 127][k] = 0;
}
__device__ __forceinline__ void rowmask_set(RowMasks &r, int start, int span) { r.m[start & 127][span >> 5] |= 1u << (span & 31); }
// bits [lo, hi] of row `start` as a 32-bit word, bit 0 = span lo (0 <= lo <= 127, hi - lo <= 31; 0 if hi < lo)
__device__ __forceinline__ uint32_t rowmask_window(const RowMasks &r, int start, int lo, int hi) {
  if (hi < lo) return 0;
  const uint32_t *row = r.m[start & 127];
  const int k = lo >> 5;
  const uint32_t a = row[k], b = row[k + 1];
  const uint32_t w = __builtin_amdgcn_alignbit(b, a, (uint32_t)(lo & 31)); // (b:a) >> (lo & 31)
  const int width = hi - lo + 1;
  return width >= 32 ? w : (w & ((1u << width) - 1));
}

// lane `l` (wave-uniform) of a double, whatever the execution mask: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// Per-cell values kept in registers across the phases of a column: one double per pass over the cells (NP passes of
// 64 lanes; NP = 2 covers maximal spans up to 127, NP = 4 up to 255).  `pass` / the index are wave-uniform.
template <int NP>
__device__ __forceinline__ void reg_set(double (&reg)[NP], int pass, double v) {
#pragma unroll
  for (int k = 0; k < NP; k++)
    if (pass == k) reg[k] = v;
}
// element idx (0 .. 64 * NP - 1) of values laid out one per lane, pass after pass
template <int NP>
__device__ __forceinline__ double lane_value_asc(const double (&reg)[NP], int idx) {
  double r = readlane_f64(reg[0], idx & 63);
#pragma unroll
  for (int k = 1; k < NP; k++)
    if ((idx >> 6) == k) r = readlane_f64(reg[k], idx & 63);
  return r;
}
// the value the lane that owns cell `d` holds, for cells laid out d = dtop - lane (dtop = dmax, dmax - 64, ...)
template <int NP>
__device__ __forceinline__ double cell_value_desc(const double (&reg)[NP], int dmax, int d) {
  return lane_value_asc<NP>(reg, dmax - d);
}

// ---- kH = 3 (see k_inside): the wavefronts of a sequence's workgroup hand batches of terms to each other through LDS
// counters instead of workgroup barriers, so that the sequence's own wavefront is free to run its chains meanwhile.
// All wavefronts of a workgroup are resident together, so waiting on one another cannot deadlock; a wait that does not
// end (a bug) trips a watchdog: the wavefront raises the workgroup's abort flag and g_ra_watchdog and ENDS - a barrier
// does not wait for ended wavefronts -, the others follow at their next wait, the host reports PRB_ERR_STATE.
__device__ int g_ra_watchdog;
struct FoldSync {
  int prod[2]; // batches helper 1 / helper 2 have written (running count over the whole sequence)
  int cons;    // batches the folding wavefront has taken
  int abort;
};
// false: give up (the caller returns from the kernel)
__device__ __forceinline__ bool fold_wait(FoldSync &fs, int &counter, int target) {
  for (int n = 0;; n++) {
    if (__hip_atomic_load(&counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return true;
    if (__hip_atomic_load(&fs.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || n > (1 << 21)) {
      __hip_atomic_store(&fs.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      g_ra_watchdog = 1;
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void fold_post(int &counter, int value, int lane) {
  // (the LDS operations of a wavefront complete in order: the batch written by all its lanes is there before the count)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&counter, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

__global__ void k_fill(double *p, int64_t n, double value) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = value;
}

// ------------------------------------------------------------------------------ inside
// kH = 0: a wavefront per sequence, four sequences per workgroup (throughput: thousands of sequences side by side).
// kH = 1, 2: a WORKGROUP per sequence - its first wavefront owns the sequence as before, the others are helpers for the
// one phase that is 35 % of a pass, the big fold: they walk the same row masks (the iterator is integer work, every
// wavefront runs it and so knows without a word how many rounds there are), evaluate the terms of the NEXT batch of eight -
// fetch, loop energies - into LDS while the owner folds the batch before; a workgroup barrier per batch.  The fold order
// is untouched.  For the few sequences of a query batch, whose time is the latency of one wavefront's chain.
// kH = 3: two helpers and a FOLDING wavefront; the sequence's own wavefront keeps phases 1 - 3 and the last step of phase 4.
// The big fold of column j needs nothing of phases 2 and 3 until its very end (Alpha_multi, one term), and they need nothing
// of it: phase 1, a barrier, then the fold (helpers -> folding wavefront, LDS counters) BESIDE phases 2 + 3, a barrier, the
// last term.  A column costs phase 1 + the longer of the two instead of their sum.
template <int NP, int kH>
__global__ __launch_bounds__(kH ? kWave * (1 + kH) : kBlock, kH ? 1 : PRB_RA_WPS) void k_inside(RaBatch b, RaConst c) {
  constexpr bool kF = kH == 3;
  __shared__ RaLds lds;
  __shared__ RowMasks rowmasks[kH ? 1 : kWavesPerBlock];
  __shared__ double xbuf[kH ? 2 : 1][kHB][kH ? kWave : 1]; // kH > 0: the terms of two batches, [batch & 1][term][lane]
  __shared__ int xflag[2];                                       // ... and whether there is a batch after them
  __shared__ FoldSync fsync;
  __shared__ double fold_out[kF ? NP : 1][kF ? kWave : 1]; // kF: the folds of a column's cells, [pass][lane]
  if (kF && threadIdx.x == 0) fsync.prod[0] = fsync.prod[1] = fsync.cons = fsync.abort = 0;
  int seq = 0; // kF: batches so far (the same count in the helpers and the folding wavefront)
  ra_load_lds(lds, c);
  const int lane = threadIdx.x & 63;
  const int role = kH ? (int)(threadIdx.x >> 6) : 0; // 0: the sequence's own wavefront
  const int idx = kH ? (int)blockIdx.x : (int)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (idx >= b.nseq) return; // (kH > 0: the whole workgroup)
  RowMasks &rm = rowmasks[kH ? 0 : (threadIdx.x >> 6)]; // rows of Alpha_stem
  RA_PROF_DECL;
  const bool use_masks = b.W + 2 + kMaxLoop <= 128; // spans <= W + 1 fit a mask, live rows <= W + 2 + MAXLOOP fit the ring (kH > 0: the host made sure)
  if (role == 0)
    for (int t = lane; t < 128; t += kWave) rowmask_clear(rm, t);
  using SL = RaSmallLayout;
  const SeqView v = make_view(b, idx);
  const int L = v.L, W = v.W, S = v.S;
  const unsigned char *s = v.s;
  double *a_stem = v.tab(A_STEM), *a_stemend = v.tab(A_STEMEND), *a_multi = v.tab(A_MULTI);
  double *a_multibif = v.tab(A_MULTIBIF), *a_multi1 = v.tab(A_MULTI1), *a_multi2 = v.tab(A_MULTI2);
  double *a_multi1t = v.tab(A_MULTI1_T);
  double *ao = v.v(V_AO);
  const double MLbase = lds.small[SL::kMLbase], MLintern = lds.small[SL::kMLintern],
               MLclosing = lds.small[SL::kMLclosing];

  for (int j = kTurn + 1; j <= L; j++) {
    // cells of column j: spans d = 3..dmax, start i = j - d >= max(0, j-W-1).
    // lane mapping: d = dtop - lane, dtop = dmax, dmax-64, ...  (the last pass holds the
    // small spans, which are the cheap ones in every inside phase)
    const int dmax = imin(j, W + 1);
    double to_reg[NP], mb_reg[NP];
    if (role == 0) { // ---- phases 1 - 3: the sequence's own wavefront ----
    if (lane == 0) rowmask_clear(rm, j - 2); // the row that gets its first cell in column j + 1

    // phase 1: Alpha_stem (raccess.cpp:102-129) and Alpha_multi2 (:145-162): both need only
    // column j-1 and the cell itself.  Also, per cell, the term it contributes to Alpha_outer[j] (:230-241),
    // kept in a register for the chain of phase 3.
#pragma unroll
    for (int k = 0; k < NP; k++) to_reg[k] = mb_reg[k] = kNegInf;
    int pass = 0;
    for (int dtop = dmax; dtop >= kTurn; dtop -= kWave, pass++) {
      const int d = dtop - lane;
      if (d < kTurn) continue;
      const int i = j - d;
      const int type = ra_bp(lds, s[i + 1], s[j]);
      double stem = kNegInf;
      if (type != 0) {
        const int type2 = ra_rtype(ra_bp(lds, s[i + 2], s[j - 1]));
        double temp = 0;
        bool flag = false;
        const double ps = EM(a_stem, i + 1, j - 1), pe = EM(a_stemend, i + 1, j - 1);
        if (ps != kNegInf) {
          if (type2 != 0) temp = ps + lds.small[SL::kStack + type * 7 + type2];
          flag = true;
        }
        if (pe != kNegInf) {
          temp = flag ? ra_lse(lds, temp, pe) : pe;
          flag = true;
        }
        stem = flag ? temp : kNegInf;
      }
      EM(a_stem, i, j) = stem;
      if (use_masks && stem != kNegInf) rowmask_set(rm, i, d);
      double temp = 0;
      bool flag = false;
      if (type != 0 && stem != kNegInf) {
        const double dng = dangle_energy(lds, v, type, i, j);
        temp = stem + MLintern + dng;
        flag = true;
        const double x = stem + dng; // (:235-236)
        reg_set<NP>(to_reg, pass, x + ao[i]);
      }
      const double prev = EM(a_multi2, i, j - 1);
      double m2;
      if (prev != kNegInf) {
        m2 = prev + MLbase;
        if (flag) m2 = ra_lse(lds, temp, m2);
      } else {
        m2 = flag ? temp : kNegInf;
      }
      EM(a_multi2, i, j) = m2;
    }
    wave_sync();
    RA_PROF(1);
    } // role == 0
    if constexpr (kF) __syncthreads(); // phase 1 (Alpha_stem, the row masks) before the helpers read them

    if (role == 0) {
    // phase 2: Alpha_multibif (:131-143) then Alpha_multi1 (:164-175)
    int pass = 0;
    for (int dtop = dmax; dtop >= kTurn; dtop -= kWave, pass++) {
      const int d = dtop - lane;
      const bool cell = d >= kTurn;
      const int i = cell ? j - d : j - kTurn;
      double temp = 0;
      bool flag = false;
      // (operands of eight terms fetched together, then folded in order: k = i + t, t = 1 .. d - 1)
      for (int t0 = 1; t0 < dtop; t0 += kRaAhead) {
        double vs[kRaAhead];
#pragma unroll
        for (int b = 0; b < kRaAhead; b++) {
          const bool in = cell && t0 + b < d;
          const int t = in ? t0 + b : 1; // (a lane without a term reads one of its own)
          const double m1 = EM(a_multi1, i, i + t), m2 = EM(a_multi2, i + t, j);
          vs[b] = (in && m1 != kNegInf && m2 != kNegInf) ? m1 + m2 : kNegInf;
        }
#pragma unroll
        for (int b = 0; b < kRaAhead; b++)
          if (vs[b] != kNegInf) {
            temp = flag ? ra_lse(lds, temp, vs[b]) : vs[b];
            flag = true;
          }
      }
      if (!cell) continue;
      const double mb = flag ? temp : kNegInf;
      reg_set<NP>(mb_reg, pass, mb);
      EM(a_multibif, i, j) = mb;
      const double m2 = EM(a_multi2, i, j);
      double m1;
      if (m2 != kNegInf && mb != kNegInf) m1 = ra_lse(lds, m2, mb);
      else if (m2 == kNegInf) m1 = mb;
      else m1 = m2;
      EM(a_multi1, i, j) = m1;
      SM(a_multi1t, i, j) = m1;
    }
    wave_sync();
    RA_PROF(2);

    // phase 3, two serial chains in ONE instruction stream on lanes 0 and 1: Alpha_multi (:177-191), i descending
    // (d ascending), and Alpha_outer[j] (:230-241), p ascending (d descending).  A step's inputs come from the
    // registers of the lanes that computed them (v_readlane), not through memory, so a step is one logsumexp.
    {
      const bool l0 = lane == 0;
      double acc = l0 ? kNegInf : ao[j - 1]; // lane 0: Alpha_multi of the previous cell (none before d = 3); lane 1: the running sum
      for (int k = 0; k + kTurn <= dmax; k++) {
        const int d0 = kTurn + k, d1 = dmax - k;
        const double mb = cell_value_desc<NP>(mb_reg, dmax, d0), to = cell_value_desc<NP>(to_reg, dmax, d1);
        if (lane < 2) {
          const double term = l0 ? mb : to;
          const double base = l0 ? acc + MLbase : acc;
          const double r = ra_lse(lds, base, term);
          if (l0) {
            const double m = acc != kNegInf ? (mb != kNegInf ? r : base) : mb;
            EM(a_multi, j - d0, j) = m;
            acc = m;
          } else if (to != kNegInf) {
            acc = r;
          }
        }
      }
      if (lane == 1) ao[j] = acc;
    }
    wave_sync();
    RA_PROF(3);
    } // role == 0

    // phase 4: Alpha_stemend (:193-226).  The enclosed stems (p = i + u1, q = j - u2) are folded in the
    // reference's order - p ascending, q ascending - by the lane that owns the cell; a lane walks only the
    // terms that exist (set bits of the rows of Alpha_stem, spans max(5, d - 30) .. d - u1), eight at a time:
    // their band entries and sequence codes are fetched together, then folded one by one.
    if constexpr (kF) {
      if (role != 0 && j != L) {
        const int bj = s[j];
        int pass = 0;
        for (int dtop = dmax; dtop >= kTurn; dtop -= kWave, pass++) {
          const int d = dtop - lane;
          const bool cell = d >= kTurn;
          const int i = cell ? j - d : 0;
          const int type = cell ? ra_bp(lds, s[i], s[j + 1]) : 0;
          const int bi1 = cell ? s[i + 1] : 0;
          double temp = type != 0 ? ra_hairpin_energy(lds, type, d, bi1, bj) : 0.0;
          const int m = type != 0 ? imin(kMaxLoop, d - (kTurn + 2)) : -1;
          const int span_lo = imax(kTurn + 2, d - kMaxLoop);
          int u1 = -1;
          uint32_t wbits = 0;
          bool more = m >= 0;
          bool pend = __ballot(more) != 0; // (the same in every wavefront: from the cells themselves)
          while (pend) {
            const int buf = seq & 1;
            if (role == 3) { // the fold
              if (!fold_wait(fsync, fsync.prod[0], seq + 1) || !fold_wait(fsync, fsync.prod[1], seq + 1)) return;
#pragma unroll
              for (int t = 0; t < kHB; t++) {
                const double x = xbuf[buf][t][lane];
                const double rr = ra_lse(lds, temp, x);
                temp = x != kNegInf ? rr : temp;
              }
              pend = xflag[buf] != 0;
              fold_post(fsync.cons, seq + 1, lane);
            } else { // a helper: which terms (both, identically), the values of its half of the batch
              if (!fold_wait(fsync, fsync.cons, seq - 1)) return; // (the batch that was in this buffer is folded)
              int su1[kHB], sspan[kHB];
#pragma unroll
              for (int t = 0; t < kHB; t++) {
                const bool adv = more && wbits == 0 && u1 < m;
                u1 += adv ? 1 : 0;
                const uint32_t wnew = rowmask_window(rm, i + (u1 < 0 ? 0 : u1), span_lo, d - u1 - (u1 == 0 ? 1 : 0));
                wbits = adv ? wnew : wbits;
                const bool has = more && wbits != 0;
                sspan[t] = span_lo + __builtin_ctz(wbits | 0x80000000u);
                su1[t] = has ? u1 : -1;
                wbits = has ? (wbits & (wbits - 1)) : wbits;
                more = more && (wbits != 0 || u1 < m);
              }
              const int t0 = role == 1 ? 0 : kHB / 2;
              double sts[kHB / 2];
              int su2[kHB / 2], sq[kHB / 2], sq1[kHB / 2], sp0[kHB / 2], sp1[kHB / 2];
#pragma unroll
              for (int t = 0; t < kHB / 2; t++) {
                const int u = su1[t0 + t];
                const bool ok = u >= 0;
                const int p = ok ? i + u : i, q = ok ? p + sspan[t0 + t] : j;
                su2[t] = j - q;
                sts[t] = EM(a_stem, p, q);
                sq[t] = s[q];
                sq1[t] = s[q + 1];
                sp0[t] = s[p];
                sp1[t] = s[p + 1];
              }
#pragma unroll
              for (int t = 0; t < kHB / 2; t++) {
                const int u = su1[t0 + t];
                const int type2 = ra_rtype(ra_bp(lds, sp1[t], sq[t]));
                const double z = ra_loop_energy_bf(lds, c.big, type, type2, u < 0 ? 0 : u, su2[t], bi1, bj, sp0[t], sq1[t]);
                xbuf[buf][t0 + t][lane] = (u >= 0 && sts[t] != kNegInf && type2 != 0) ? sts[t] + z : kNegInf;
              }
              pend = __ballot(more) != 0;
              if (role == 1 && lane == 0) xflag[buf] = pend ? 1 : 0;
              fold_post(fsync.prod[role - 1], seq + 1, lane);
            }
            seq++;
          }
          if (role == 3) fold_out[pass][lane] = temp;
        }
      }
      __syncthreads(); // the folds, and phases 2 - 3 (Alpha_multi) beside them
      if (role == 0 && j != L) { // the last term of phase 4
        const int bj = s[j];
        int pass = 0;
        for (int dtop = dmax; dtop >= kTurn; dtop -= kWave, pass++) {
          const int d = dtop - lane;
          if (d < kTurn) continue;
          const int i = j - d;
          const int type = ra_bp(lds, s[i], s[j + 1]);
          double out = kNegInf;
          if (type != 0) {
            const int tt = ra_rtype(type);
            out = ra_lse(lds, fold_out[pass][lane],
                         EM(a_multi, i, j) + MLclosing + MLintern + lds.small[SL::kDangle3 + tt * 5 + s[i + 1]] +
                             lds.small[SL::kDangle5 + tt * 5 + bj]);
          }
          EM(a_stemend, i, j) = out;
        }
      }
    } else if constexpr (kH > 0) {
      __syncthreads(); // the owner's phases 1 - 3 (Alpha_stem, the row masks) before the helpers read them
      if (j != L) {
        const int bj = s[j];
        for (int dtop = dmax; dtop >= kTurn; dtop -= kWave) {
          const int d = dtop - lane;
          const bool cell = d >= kTurn;
          const int i = cell ? j - d : 0;
          const int type = cell ? ra_bp(lds, s[i], s[j + 1]) : 0;
          const int bi1 = cell ? s[i + 1] : 0;
          double temp = type != 0 ? ra_hairpin_energy(lds, type, d, bi1, bj) : 0.0;
          const int m = type != 0 ? imin(kMaxLoop, d - (kTurn + 2)) : -1;
          const int span_lo = imax(kTurn + 2, d - kMaxLoop);
          int u1 = -1;
          uint32_t wbits = 0;
          bool more = m >= 0;
          // which terms (every wavefront, identically: integer work on the row masks in LDS)
          auto advance = [&](int (&su1)[kHB], int (&sspan)[kHB]) {
#pragma unroll
            for (int t = 0; t < kHB; t++) {
              const bool adv = more && wbits == 0 && u1 < m;
              u1 += adv ? 1 : 0;
              const uint32_t wnew = rowmask_window(rm, i + (u1 < 0 ? 0 : u1), span_lo, d - u1 - (u1 == 0 ? 1 : 0));
              wbits = adv ? wnew : wbits;
              const bool has = more && wbits != 0;
              sspan[t] = span_lo + __builtin_ctz(wbits | 0x80000000u);
              su1[t] = has ? u1 : -1;
              wbits = has ? (wbits & (wbits - 1)) : wbits;
              more = more && (wbits != 0 || u1 < m);
            }
          };
          // their values, terms [t0, t1) of the batch, into xbuf[buf] (a helper)
          auto values = [&](const int (&su1)[kHB], const int (&sspan)[kHB], int t0, int t1, int buf) {
            double sts[kHB];
            int su2[kHB], sq[kHB], sq1[kHB], sp0[kHB], sp1[kHB];
#pragma unroll
            for (int t = 0; t < kHB; t++) {
              if (t < t0 || t >= t1) continue;
              const bool ok = su1[t] >= 0;
              const int p = ok ? i + su1[t] : i, q = ok ? p + sspan[t] : j;
              su2[t] = j - q;
              sts[t] = EM(a_stem, p, q);
              sq[t] = s[q];
              sq1[t] = s[q + 1];
              sp0[t] = s[p];
              sp1[t] = s[p + 1];
            }
#pragma unroll
            for (int t = 0; t < kHB; t++) {
              if (t < t0 || t >= t1) continue;
              const int type2 = ra_rtype(ra_bp(lds, sp1[t], sq[t]));
              const double z = ra_loop_energy_bf(lds, c.big, type, type2, su1[t] < 0 ? 0 : su1[t], su2[t], bi1, bj, sp0[t], sq1[t]);
              xbuf[buf][t][lane] = (su1[t] >= 0 && sts[t] != kNegInf && type2 != 0) ? sts[t] + z : kNegInf;
            }
          };
          // (is there another batch?  Before the first: every wavefront sees it from the cells themselves; from then on
          // the first helper says so - the owner does not walk the masks at all, its rounds are the fold alone)
          bool pend = __ballot(more) != 0;
          for (int r = 0;; r++) {
            if (pend && role != 0) {
              int su1[kHB], sspan[kHB];
              advance(su1, sspan);
              if (kH == 1) {
                values(su1, sspan, 0, kHB, r & 1);
              } else {
                if (role == 1) values(su1, sspan, 0, kHB / 2, r & 1);
                if (role == 2) values(su1, sspan, kHB / 2, kHB, r & 1);
              }
              const bool nxt = __ballot(more) != 0; // (every lane of the helper takes part)
              if (role == 1 && lane == 0) xflag[(r + 1) & 1] = nxt ? 1 : 0;
            }
            if (role == 0 && r >= 1) { // the fold, one batch behind
#pragma unroll
              for (int t = 0; t < kHB; t++) {
                const double x = xbuf[(r - 1) & 1][t][lane];
                const double rr = ra_lse(lds, temp, x);
                temp = x != kNegInf ? rr : temp;
              }
            }
            __syncthreads();
            if (!pend) break;
            pend = xflag[(r + 1) & 1] != 0;
          }
          if (role == 0 && cell) {
            double out = kNegInf;
            if (type != 0) {
              const int tt = ra_rtype(type);
              out = ra_lse(lds, temp,
                           EM(a_multi, i, j) + MLclosing + MLintern + lds.small[SL::kDangle3 + tt * 5 + bi1] +
                               lds.small[SL::kDangle5 + tt * 5 + bj]);
            }
            EM(a_stemend, i, j) = out;
          }
        }
      }
    }
    if (kH == 0 && j != L && use_masks) {
      const int bj = s[j]; // = s[j'-1] for the closing pair (i, j' = j+1)
      for (int dtop = dmax; dtop >= kTurn; dtop -= kWave) {
        const int d = dtop - lane;
        const bool cell = d >= kTurn;
        const int i = cell ? j - d : 0;
        const int type = cell ? ra_bp(lds, s[i], s[j + 1]) : 0;
        const int bi1 = cell ? s[i + 1] : 0;
        double temp = type != 0 ? ra_hairpin_energy(lds, type, d, bi1, bj) : 0.0;
        // this lane's terms: rows u1 = 0 .. m, p <= j - 5
        const int m = type != 0 ? imin(kMaxLoop, d - (kTurn + 2)) : -1;
        const int span_lo = imax(kTurn + 2, d - kMaxLoop);
        int u1 = -1;
        uint32_t wbits = 0; // the terms left in row u1: bit b = span span_lo + b
        bool more = m >= 0;
        // One batch = the next eight terms of this lane, as values: (1) which terms - integer work and the row masks in
        // LDS, without a branch: a step that finds its row used up moves on ONE row and may come back empty-handed;
        // (2) their band entries and sequence codes, all fetched together (a lane without a term reads its own cell);
        // (3) their values - the table look-ups are independent of each other.
        auto next_batch = [&](double (&xs)[kRaAhead]) {
          int su1[kRaAhead], sspan[kRaAhead];
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const bool adv = more && wbits == 0 && u1 < m;
            u1 += adv ? 1 : 0;
            const uint32_t wnew = rowmask_window(rm, i + (u1 < 0 ? 0 : u1), span_lo, d - u1 - (u1 == 0 ? 1 : 0)); // (p, q) == (i, j) is excluded (:207)
            wbits = adv ? wnew : wbits;
            const bool has = more && wbits != 0;
            sspan[t] = span_lo + __builtin_ctz(wbits | 0x80000000u);
            su1[t] = has ? u1 : -1;
            wbits = has ? (wbits & (wbits - 1)) : wbits;
            more = more && (wbits != 0 || u1 < m);
          }
          double sts[kRaAhead];
          int su2[kRaAhead], sq[kRaAhead], sq1[kRaAhead], sp0[kRaAhead], sp1[kRaAhead];
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const bool ok = su1[t] >= 0;
            const int p = ok ? i + su1[t] : i, q = ok ? p + sspan[t] : j;
            su2[t] = j - q;
            sts[t] = EM(a_stem, p, q);
            sq[t] = s[q];
            sq1[t] = s[q + 1];
            sp0[t] = s[p];
            sp1[t] = s[p + 1];
          }
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const int type2 = ra_rtype(ra_bp(lds, sp1[t], sq[t])); // (not 0 where the stem exists)
            const double z = ra_loop_energy_bf(lds, c.big, type, type2, su1[t] < 0 ? 0 : su1[t], su2[t], bi1, bj, sp0[t], sq1[t]);
            xs[t] = (su1[t] >= 0 && sts[t] != kNegInf && type2 != 0) ? sts[t] + z : kNegInf;
          }
        };
        // The fold - a chain of dependent logsumexp, what a column costs - runs one batch behind: the next batch is
        // prepared in the same stretch of straight-line code, so its loads and look-ups fill the chain's stalls.
        double xs[kRaAhead];
        bool pending = __ballot(more) != 0;
        if (pending) next_batch(xs);
        while (pending) {
          double xn[kRaAhead];
          const bool again = __ballot(more) != 0;
          if (again) next_batch(xn);
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const double r = ra_lse(lds, temp, xs[t]);
            temp = xs[t] != kNegInf ? r : temp;
          }
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) xs[t] = xn[t];
          pending = again;
        }
        if (cell) {
          double out = kNegInf;
          if (type != 0) {
            const int tt = ra_rtype(type);
            out = ra_lse(lds, temp,
                         EM(a_multi, i, j) + MLclosing + MLintern + lds.small[SL::kDangle3 + tt * 5 + bi1] +
                             lds.small[SL::kDangle5 + tt * 5 + bj]);
          }
          EM(a_stemend, i, j) = out;
        }
      }
    }
    if (kH == 0 && j != L && !use_masks) {
      // (spans beyond the row masks: every (u1, u2) term of the window is enumerated wave-uniformly)
      const int bj = s[j]; // = s[j'-1] for the closing pair (i, j' = j+1)
      for (int dtop = dmax; dtop >= kTurn; dtop -= kWave) {
        const int d = dtop - lane;
        const bool cell = d >= kTurn;
        const int i = cell ? j - d : 0;
        const int type = cell ? ra_bp(lds, s[i], s[j + 1]) : 0;
        const int bi1 = cell ? s[i + 1] : 0;
        double temp = type != 0 ? ra_hairpin_energy(lds, type, d, bi1, bj) : 0.0;
        // p <= j-5 and q >= p+5  <=>  u1 + u2 <= d - 5; the pass bound uses dtop
        const int m = imin(kMaxLoop, dtop - (kTurn + 2));
        // (terms in blocks of kRaAhead: the band entries and sequence codes of a block are fetched
        // together, so their HBM / L2 latency is paid once per block, not once per term; the fold
        // itself stays in the reference's order)
        for (int u1 = 0; u1 <= m; u1++) {
          const int p = i + u1;
          const int bp1 = type != 0 ? s[p + 1] : 0, bp0 = type != 0 ? s[p] : 0;
          for (int u2hi = imin(kMaxLoop - u1, dtop - (kTurn + 2) - u1); u2hi >= 0; u2hi -= kRaAhead) {
            double sts[kRaAhead];
            int sq[kRaAhead], sq1[kRaAhead];
#pragma unroll
            for (int b = 0; b < kRaAhead; b++) {
              const int u2 = u2hi - b;
              // the (p,q) == (i,j) term is excluded (:207)
              const bool ok = u2 >= 0 && !(u1 == 0 && u2 == 0) && type != 0 && u1 + u2 <= d - (kTurn + 2);
              const int q = j - u2;
              sts[b] = ok ? EM(a_stem, p, q) : kNegInf;
              sq[b] = ok ? s[q] : 0;
              sq1[b] = ok ? s[q + 1] : 0;
            }
#pragma unroll
            for (int b = 0; b < kRaAhead; b++) {
              const double st = sts[b];
              if (st != kNegInf) {
                int type2 = ra_bp(lds, bp1, sq[b]);
                if (type2 != 0) {
                  type2 = ra_rtype(type2);
                  const double z = ra_loop_energy(lds, c.big, type, type2, u1, u2hi - b, bi1, bj, bp0, sq1[b]);
                  temp = ra_lse(lds, temp, st + z);
                }
              }
            }
          }
        }
        if (cell) {
          double out = kNegInf;
          if (type != 0) {
            const int tt = ra_rtype(type);
            out = ra_lse(lds, temp,
                         EM(a_multi, i, j) + MLclosing + MLintern + lds.small[SL::kDangle3 + tt * 5 + bi1] +
                             lds.small[SL::kDangle5 + tt * 5 + bj]);
          }
          EM(a_stemend, i, j) = out;
        }
      }
    }
    wave_sync();
    RA_PROF(4);
  }
}

// ----------------------------------------------------------------------------- outside
// (kH: helper wavefronts for the big fold, see k_inside)
// kH = 3: phase A, a barrier, then the big fold of phase E (helpers -> folding wavefront) beside phases B - D on the
// sequence's own wavefront, a barrier, the last two terms of phase E (they need Beta_multi2 of phase D) - see k_inside.
template <int NP, int kH>
__global__ __launch_bounds__(kH ? kWave * (1 + kH) : kBlock, kH ? 1 : PRB_RA_WPS) void k_outside(RaBatch b, RaConst c) {
  constexpr bool kF = kH == 3;
  __shared__ RaLds lds;
  __shared__ RowMasks rowmasks[kH ? 1 : kWavesPerBlock];
  __shared__ double xbuf[kH ? 2 : 1][kHB][kH ? kWave : 1];
  __shared__ int xflag[2];
  __shared__ FoldSync fsync;
  __shared__ double fold_out[kF ? NP : 1][kF ? kWave : 1];
  if (kF && threadIdx.x == 0) fsync.prod[0] = fsync.prod[1] = fsync.cons = fsync.abort = 0;
  int seq = 0;
  ra_load_lds(lds, c);
  const int lane = threadIdx.x & 63;
  const int role = kH ? (int)(threadIdx.x >> 6) : 0;
  const int idx = kH ? (int)blockIdx.x : (int)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (idx >= b.nseq) return;
  RA_PROF_DECL;
  RowMasks &rm = rowmasks[kH ? 0 : (threadIdx.x >> 6)]; // rows of Beta_stemend
  const bool use_masks = b.W + 2 + kMaxLoop <= 128;
  if (role == 0)
    for (int t = lane; t < 128; t += kWave) rowmask_clear(rm, t);
  using SL = RaSmallLayout;
  const SeqView v = make_view(b, idx);
  const int L = v.L, W = v.W, S = v.S;
  const unsigned char *s = v.s;
  const double *a_stem = v.tab(A_STEM), *a_multi2 = v.tab(A_MULTI2), *a_multi1t = v.tab(A_MULTI1_T);
  double *b_stem = v.tab(B_STEM), *b_stemend = v.tab(B_STEMEND), *b_multi = v.tab(B_MULTI);
  double *b_multibif = v.tab(B_MULTIBIF), *b_multi1 = v.tab(B_MULTI1), *b_multi2 = v.tab(B_MULTI2);
  const double *ao = v.v(V_AO);
  double *bo = v.v(V_BO);
  const double MLbase = lds.small[SL::kMLbase], MLintern = lds.small[SL::kMLintern],
               MLclosing = lds.small[SL::kMLclosing];

  // Beta_outer[i] (raccess.cpp:260-271) is a serial chain over i descending that only reads
  // Alpha_stem; it runs on lane 1 one position ahead of the column loop.
  auto beta_outer_at = [&](int i) {
    double temp = bo[i + 1];
    for (int p = i + 1; p <= imin(i + W + 1, L); p++) {
      const double st = EM(a_stem, i, p);
      if (st != kNegInf) {
        const int type = ra_bp(lds, s[i + 1], s[p]);
        const double x = st + dangle_energy(lds, v, type, i, p);
        temp = ra_lse(lds, temp, x + bo[p]);
      }
    }
    bo[i] = temp;
  };

  for (int q = L; q >= kTurn + 1; q--) {
    // cells of column q: p = max(0,q-W-1) .. q-3 ascending <=> d = q - p descending from dmax.
    // lane mapping: d = dbot + lane, dbot = 3, 67, ... (the last pass holds the large
    // spans, which are the cheap ones in every outside phase)
    const int dmax = imin(q, W + 1);
    double xb_reg[NP], tob_reg[NP];
    const int pend_o = imin(q + W, L); // Beta_outer[q - 1] sums p' = q .. pend_o

    if (role == 0) { // ---- phases A - D: the sequence's own wavefront ----
    // phase A: Beta_stemend (:278-279), copy from column q+1
    if (use_masks && lane == 0) rowmask_clear(rm, q - W); // the row whose first cell comes in column q - 1 (spans >= W never exist)
    // Also, in registers for the chains of phase B: per cell the term it contributes to Beta_multi (:296-300), and
    // - one per lane, p' = q + lane (+ 64) - the terms of Beta_outer[q - 1] (:262-269).
#pragma unroll
    for (int k = 0; k < NP; k++) xb_reg[k] = tob_reg[k] = kNegInf;
    {
      int pass = 0;
      for (int dbot = kTurn; dbot <= dmax; dbot += kWave, pass++) {
        const int d = dbot + lane;
        if (q == L || d > dmax) continue;
        const int p = q - d;
        if (p != 0) {
          const double se = d >= W ? kNegInf : EM(b_stem, p - 1, q + 1);
          EM(b_stemend, p, q) = se;
          if (use_masks && se != kNegInf) rowmask_set(rm, p, d);
          if (se != kNegInf) {
            const int tt = ra_rtype(ra_bp(lds, s[p], s[q + 1]));
            const double x = se + MLclosing + MLintern + lds.small[SL::kDangle3 + tt * 5 + s[p + 1]] +
                             lds.small[SL::kDangle5 + tt * 5 + s[q]];
            reg_set<NP>(xb_reg, pass, x);
          }
        }
      }
      for (pass = 0; pass < NP; pass++) {
        const int pp = q + lane + 64 * pass;
        if (pp > pend_o) continue;
        const int i = q - 1;
        const double st = EM(a_stem, i, pp);
        if (st != kNegInf) {
          const int type = ra_bp(lds, s[i + 1], s[pp]);
          const double x = st + dangle_energy(lds, v, type, i, pp);
          reg_set<NP>(tob_reg, pass, x + bo[pp]);
        }
      }
    }
    wave_sync();
    RA_PROF(8);
    } // role == 0
    if constexpr (kF) __syncthreads(); // phase A (Beta_stemend, the row masks) before the helpers read them

    if (role == 0) {
    // phase B, two serial chains in one instruction stream (see k_inside's phase 3): Beta_multi (:281-308), p
    // ascending (d descending), on lane 0; Beta_outer[q - 1] (:260-271), p' ascending, on lane 1; inputs by v_readlane.
    {
      const bool l0 = lane == 0;
      const int k0n = q != L ? dmax - kTurn + 1 : 0, k1n = pend_o - q + 1;
      double acc = l0 ? kNegInf : bo[q]; // lane 0: Beta_multi of the previous cell; lane 1: the running sum
      for (int k = 0; k < imax(k0n, k1n); k++) {
        const int d0 = dmax - k;
        const double xb = k < k0n ? lane_value_asc<NP>(xb_reg, d0 - kTurn) : kNegInf;
        const double to = k < k1n ? lane_value_asc<NP>(tob_reg, k) : kNegInf;
        if (lane < 2) {
          const double term = l0 ? xb : to;
          const double base = l0 ? acc + MLbase : acc;
          const double r = ra_lse(lds, base, term);
          if (l0) {
            const int p = q - d0;
            if (k < k0n && p != 0) {
              const bool flag = d0 + 1 <= W + 1 && acc != kNegInf; // (Beta_multi[p - 1][q - p + 1] exists)
              const double temp = xb != kNegInf ? (flag ? r : xb) : (flag ? base : kNegInf);
              EM(b_multi, p, q) = temp;
              acc = temp;
            }
          } else if (to != kNegInf) {
            acc = r;
          }
        }
      }
      if (lane == 1) bo[q - 1] = acc;
    }
    wave_sync();
    RA_PROF(9);

    if (q != L) {
      // phase C: Beta_multi1 (:310-324) then Beta_multibif (:354-364)
      for (int dbot = kTurn; dbot <= dmax; dbot += kWave) {
        const int d = dbot + lane;
        if (d > dmax) continue;
        const int p = q - d;
        if (p == 0) continue;
        double temp = 0;
        bool flag = false;
        const int kend = imin(L, p + W);
        // (operands of eight terms fetched together, then folded in order)
        for (int k0 = q + 1; k0 <= kend; k0 += kRaAhead) {
          double vs[kRaAhead];
#pragma unroll
          for (int b = 0; b < kRaAhead; b++) {
            const bool in = k0 + b <= kend;
            const int k = in ? k0 + b : q + 1;
            const double bb = EM(b_multibif, p, k), m2 = EM(a_multi2, q, k);
            vs[b] = (in && bb != kNegInf && m2 != kNegInf) ? bb + m2 : kNegInf;
          }
#pragma unroll
          for (int b = 0; b < kRaAhead; b++)
            if (vs[b] != kNegInf) {
              temp = flag ? ra_lse(lds, temp, vs[b]) : vs[b];
              flag = true;
            }
        }
        const double m1 = flag ? temp : kNegInf;
        EM(b_multi1, p, q) = m1;
        const double mm = EM(b_multi, p, q);
        double mb;
        if (m1 != kNegInf && mm != kNegInf) mb = ra_lse(lds, m1, mm);
        else if (mm == kNegInf) mb = m1;
        else mb = mm;
        EM(b_multibif, p, q) = mb;
      }
      wave_sync();
      RA_PROF(10);

      // phase D: Beta_multi2 (:326-352)
      for (int dbot = kTurn; dbot <= dmax; dbot += kWave) {
        const int d = dbot + lane;
        if (d > dmax) continue;
        const int p = q - d;
        if (p == 0) continue;
        double temp = 0;
        bool flag = false;
        const double m1 = EM(b_multi1, p, q);
        if (m1 != kNegInf) {
          temp = m1;
          flag = true;
        }
        if (d <= W) {
          const double nx = EM(b_multi2, p, q + 1);
          if (nx != kNegInf) {
            temp = flag ? ra_lse(lds, temp, nx + MLbase) : nx + MLbase;
            flag = true;
          }
        }
        for (int k0 = imax(0, q - W); k0 < p; k0 += kRaAhead) { // (eight terms' operands together, as in phase C)
          double vs[kRaAhead];
#pragma unroll
          for (int b = 0; b < kRaAhead; b++) {
            const bool in = k0 + b < p;
            const int k = in ? k0 + b : p - 1; // (p >= 1 here)
            const double bb = EM(b_multibif, k, q), a1 = SM(a_multi1t, k, p);
            vs[b] = (in && bb != kNegInf && a1 != kNegInf) ? bb + a1 : kNegInf;
          }
#pragma unroll
          for (int b = 0; b < kRaAhead; b++)
            if (vs[b] != kNegInf) {
              temp = flag ? ra_lse(lds, temp, vs[b]) : vs[b];
              flag = true;
            }
        }
        EM(b_multi2, p, q) = flag ? temp : kNegInf;
      }
      wave_sync();
      RA_PROF(11);
    }

    } // role == 0
    if constexpr (kF) {
      if (role != 0) {
        int pass = 0;
        for (int dbot = kTurn; dbot <= dmax; dbot += kWave, pass++) {
          const int d = dbot + lane;
          const bool cell = d <= dmax;
          const int p = cell ? q - d : 1;
          const int t2raw = cell ? ra_bp(lds, s[p + 1], s[q]) : 0;
          const int type2 = ra_rtype(t2raw);
          const int bp0 = cell ? s[p] : 0, bq1 = cell ? s[q + 1] : 0;
          double temp = 0;
          if (role == 3 && t2raw != 0) temp = ao[p] + bo[q] + dangle_energy(lds, v, t2raw, p, q);
          int u1 = t2raw != 0 ? imin(imin(kMaxLoop, p - 1), W + 1 - d) + 1 : 0; // (one above the first row)
          uint32_t wbits = 0;
          int span0 = 0;
          bool more = t2raw != 0 && u1 > 0;
          bool pend = __ballot(more) != 0;
          while (pend) {
            const int buf = seq & 1;
            if (role == 3) { // the fold
              if (!fold_wait(fsync, fsync.prod[0], seq + 1) || !fold_wait(fsync, fsync.prod[1], seq + 1)) return;
#pragma unroll
              for (int t = 0; t < kHB; t++) {
                const double x = xbuf[buf][t][lane];
                const double rr = ra_lse(lds, temp, x);
                temp = x != kNegInf ? rr : temp;
              }
              pend = xflag[buf] != 0;
              fold_post(fsync.cons, seq + 1, lane);
            } else {
              if (!fold_wait(fsync, fsync.cons, seq - 1)) return;
              int su1[kHB], sspan[kHB];
#pragma unroll
              for (int t = 0; t < kHB; t++) {
                const bool adv = more && wbits == 0 && u1 > 0;
                u1 -= adv ? 1 : 0;
                const int u2max = imin(imin(kMaxLoop - u1, L - 1 - q), W + 1 - d - u1);
                const int s0 = d + u1 + (u1 == 0 ? 1 : 0); // (i, j) == (p, q) is excluded (:377)
                const uint32_t wnew = rowmask_window(rm, p - u1, s0, d + u1 + u2max);
                wbits = adv ? wnew : wbits;
                span0 = adv ? s0 : span0;
                const bool has = more && wbits != 0;
                sspan[t] = span0 + __builtin_ctz(wbits | 0x80000000u);
                su1[t] = has ? u1 : -1;
                wbits = has ? (wbits & (wbits - 1)) : wbits;
                more = more && (wbits != 0 || u1 > 0);
              }
              const int t0 = role == 1 ? 0 : kHB / 2;
              double ses[kHB / 2];
              int su2[kHB / 2], sj[kHB / 2], sj1[kHB / 2], si0[kHB / 2], si1[kHB / 2];
#pragma unroll
              for (int t = 0; t < kHB / 2; t++) {
                const int u = su1[t0 + t];
                const bool ok = u >= 0;
                const int i = ok ? p - u : p, j = ok ? i + sspan[t0 + t] : q; // (a lane without a term reads its own cell)
                su2[t] = j - q;
                ses[t] = EM(b_stemend, i, j);
                sj[t] = s[j];
                sj1[t] = s[j + 1];
                si0[t] = s[i];
                si1[t] = s[i + 1];
              }
#pragma unroll
              for (int t = 0; t < kHB / 2; t++) {
                const int u = su1[t0 + t];
                const int type = ra_bp(lds, si0[t], sj1[t]);
                const double z = ra_loop_energy_bf(lds, c.big, type, type2, u < 0 ? 0 : u, su2[t], si1[t], sj[t], bp0, bq1);
                xbuf[buf][t0 + t][lane] = (u >= 0 && ses[t] != kNegInf && type != 0) ? ses[t] + z : kNegInf;
              }
              pend = __ballot(more) != 0;
              if (role == 1 && lane == 0) xflag[buf] = pend ? 1 : 0;
              fold_post(fsync.prod[role - 1], seq + 1, lane);
            }
            seq++;
          }
          if (role == 3) fold_out[pass][lane] = temp;
        }
      }
      __syncthreads(); // the folds, and phases B - D (Beta_multi2) beside them
      if (role == 0) { // the last terms of phase E
        int pass = 0;
        for (int dbot = kTurn; dbot <= dmax; dbot += kWave, pass++) {
          const int d = dbot + lane;
          if (d > dmax) continue;
          const int p = q - d;
          const int t2raw = ra_bp(lds, s[p + 1], s[q]);
          double out = kNegInf;
          if (t2raw != 0) {
            const int type2 = ra_rtype(t2raw);
            double temp = fold_out[pass][lane];
            if (p != 0 && q != L) {
              const int type = ra_bp(lds, s[p], s[q + 1]);
              if (type != 0 && d + 2 <= W + 1) {
                const double ps = EM(b_stem, p - 1, q + 1);
                if (ps != kNegInf) temp = ra_lse(lds, temp, ps + lds.small[SL::kStack + type * 7 + type2]);
              }
            }
            out = temp;
            const double m2 = EM(b_multi2, p, q);
            if (m2 != kNegInf) {
              const double x = m2 + MLintern + dangle_energy(lds, v, t2raw, p, q);
              out = ra_lse(lds, x, out);
            }
          }
          EM(b_stem, p, q) = out;
        }
      }
    } else {
    if constexpr (kH > 0) __syncthreads(); // the owner's phases A - D (Beta_stemend, the row masks) before the helpers read them

    // phase E: Beta_stem (:367-409).  Enclosing pairs (i = p - u1, j+1 = q + u2 + 1) in the reference's
    // order: i ascending, j ascending.  As in k_inside's phase 4, a lane walks only the terms whose
    // Beta_stemend entry exists (set bits of a window of the row masks), eight at a time: fetch, evaluate, fold.
    for (int dbot = kTurn; dbot <= dmax; dbot += kWave) {
      const int d = dbot + lane;
      const bool cell = d <= dmax;
      const int p = cell ? q - d : 1;
      const int t2raw = cell ? ra_bp(lds, s[p + 1], s[q]) : 0;
      const int type2 = ra_rtype(t2raw);
      const int bp0 = cell ? s[p] : 0, bq1 = cell ? s[q + 1] : 0;
      double temp = 0;
      if (t2raw != 0) temp = ao[p] + bo[q] + dangle_energy(lds, v, t2raw, p, q);
      if (use_masks) {
        // rows u1 = u1max .. 0 (i = p - u1 >= 1), spans d + u1 .. d + u1 + u2max with j <= L - 1 and j - i <= W + 1
        int u1 = t2raw != 0 ? imin(imin(kMaxLoop, p - 1), W + 1 - d) + 1 : 0; // (one above the first row)
        uint32_t wbits = 0;
        int span0 = 0; // span of bit 0 of wbits
        bool more = t2raw != 0 && u1 > 0;
        if constexpr (kH > 0) {
          // which terms (every wavefront of the workgroup, identically)
          auto advance = [&](int (&su1)[kHB], int (&sspan)[kHB]) {
#pragma unroll
            for (int t = 0; t < kHB; t++) {
              const bool adv = more && wbits == 0 && u1 > 0;
              u1 -= adv ? 1 : 0;
              const int u2max = imin(imin(kMaxLoop - u1, L - 1 - q), W + 1 - d - u1);
              const int s0 = d + u1 + (u1 == 0 ? 1 : 0); // (i, j) == (p, q) is excluded (:377)
              const uint32_t wnew = rowmask_window(rm, p - u1, s0, d + u1 + u2max);
              wbits = adv ? wnew : wbits;
              span0 = adv ? s0 : span0;
              const bool has = more && wbits != 0;
              sspan[t] = span0 + __builtin_ctz(wbits | 0x80000000u);
              su1[t] = has ? u1 : -1;
              wbits = has ? (wbits & (wbits - 1)) : wbits;
              more = more && (wbits != 0 || u1 > 0);
            }
          };
          // their values, terms [t0, t1) of the batch, into xbuf[buf] (a helper)
          auto values = [&](const int (&su1)[kHB], const int (&sspan)[kHB], int t0, int t1, int buf) {
            double ses[kHB];
            int su2[kHB], sj[kHB], sj1[kHB], si0[kHB], si1[kHB];
#pragma unroll
            for (int t = 0; t < kHB; t++) {
              if (t < t0 || t >= t1) continue;
              const bool ok = su1[t] >= 0;
              const int i = ok ? p - su1[t] : p, j = ok ? i + sspan[t] : q; // (a lane without a term reads its own cell)
              su2[t] = j - q;
              ses[t] = EM(b_stemend, i, j);
              sj[t] = s[j];
              sj1[t] = s[j + 1];
              si0[t] = s[i];
              si1[t] = s[i + 1];
            }
#pragma unroll
            for (int t = 0; t < kHB; t++) {
              if (t < t0 || t >= t1) continue;
              const int type = ra_bp(lds, si0[t], sj1[t]);
              const double z = ra_loop_energy_bf(lds, c.big, type, type2, su1[t] < 0 ? 0 : su1[t], su2[t], si1[t], sj[t], bp0, bq1);
              xbuf[buf][t][lane] = (su1[t] >= 0 && ses[t] != kNegInf && type != 0) ? ses[t] + z : kNegInf;
            }
          };
          // (is there another batch?  Before the first: every wavefront sees it from the cells themselves; from then on
          // the first helper says so - the owner does not walk the masks at all, its rounds are the fold alone)
          bool pend = __ballot(more) != 0;
          for (int r = 0;; r++) {
            if (pend && role != 0) {
              int su1[kHB], sspan[kHB];
              advance(su1, sspan);
              if (kH == 1) {
                values(su1, sspan, 0, kHB, r & 1);
              } else {
                if (role == 1) values(su1, sspan, 0, kHB / 2, r & 1);
                if (role == 2) values(su1, sspan, kHB / 2, kHB, r & 1);
              }
              const bool nxt = __ballot(more) != 0; // (every lane of the helper takes part)
              if (role == 1 && lane == 0) xflag[(r + 1) & 1] = nxt ? 1 : 0;
            }
            if (role == 0 && r >= 1) { // the fold, one batch behind
#pragma unroll
              for (int t = 0; t < kHB; t++) {
                const double x = xbuf[(r - 1) & 1][t][lane];
                const double rr = ra_lse(lds, temp, x);
                temp = x != kNegInf ? rr : temp;
              }
            }
            __syncthreads();
            if (!pend) break;
            pend = xflag[(r + 1) & 1] != 0;
          }
        } else {
        // (as in k_inside's phase 4: which terms, without a branch; fetch; values - then the fold one batch behind)
        auto next_batch = [&](double (&xs)[kRaAhead]) {
          int su1[kRaAhead], sspan[kRaAhead];
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const bool adv = more && wbits == 0 && u1 > 0;
            u1 -= adv ? 1 : 0;
            const int u2max = imin(imin(kMaxLoop - u1, L - 1 - q), W + 1 - d - u1);
            const int s0 = d + u1 + (u1 == 0 ? 1 : 0); // (i, j) == (p, q) is excluded (:377)
            const uint32_t wnew = rowmask_window(rm, p - u1, s0, d + u1 + u2max);
            wbits = adv ? wnew : wbits;
            span0 = adv ? s0 : span0;
            const bool has = more && wbits != 0;
            sspan[t] = span0 + __builtin_ctz(wbits | 0x80000000u);
            su1[t] = has ? u1 : -1;
            wbits = has ? (wbits & (wbits - 1)) : wbits;
            more = more && (wbits != 0 || u1 > 0);
          }
          double ses[kRaAhead];
          int su2[kRaAhead], sj[kRaAhead], sj1[kRaAhead], si0[kRaAhead], si1[kRaAhead];
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const bool ok = su1[t] >= 0;
            const int i = ok ? p - su1[t] : p, j = ok ? i + sspan[t] : q; // (a lane without a term reads its own cell)
            su2[t] = j - q;
            ses[t] = EM(b_stemend, i, j);
            sj[t] = s[j];
            sj1[t] = s[j + 1];
            si0[t] = s[i];
            si1[t] = s[i + 1];
          }
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const int type = ra_bp(lds, si0[t], sj1[t]);
            const double z = ra_loop_energy_bf(lds, c.big, type, type2, su1[t] < 0 ? 0 : su1[t], su2[t], si1[t], sj[t], bp0, bq1);
            xs[t] = (su1[t] >= 0 && ses[t] != kNegInf && type != 0) ? ses[t] + z : kNegInf;
          }
        };
        double xs[kRaAhead];
        bool pending = __ballot(more) != 0;
        if (pending) next_batch(xs);
        while (pending) {
          double xn[kRaAhead];
          const bool again = __ballot(more) != 0;
          if (again) next_batch(xn);
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) {
            const double r = ra_lse(lds, temp, xs[t]);
            temp = xs[t] != kNegInf ? r : temp;
          }
#pragma unroll
          for (int t = 0; t < kRaAhead; t++) xs[t] = xn[t];
          pending = again;
        }
        } // kH == 0
      } else {
        // (spans beyond the row masks: every (u1, u2) term of the window is enumerated wave-uniformly)
        // j - i = d + u1 + u2 <= W + 1; the pass bound uses dbot
        const int m = imin(kMaxLoop, W + 1 - dbot);
        const int u1top = imin(m, q - dbot - 1); // i >= 1 for the smallest span of the pass
        for (int u1 = u1top; u1 >= 0; u1--) {
          const int u2top = imin(imin(kMaxLoop - u1, L - 1 - q), W + 1 - dbot - u1);
          const bool row = t2raw != 0 && u1 <= p - 1;
          const int i = p - u1;
          const int bi0 = row ? s[i] : 0, bi1 = row ? s[i + 1] : 0;
          for (int u2lo = 0; u2lo <= u2top; u2lo += kRaAhead) { // blocks of terms, see k_inside
            double ses[kRaAhead];
            int sj[kRaAhead], sj1[kRaAhead];
#pragma unroll
            for (int b = 0; b < kRaAhead; b++) {
              const int u2 = u2lo + b;
              // (i,j) == (p,q) is excluded (:377)
              const bool ok = row && u2 <= u2top && !(u1 == 0 && u2 == 0) && d + u1 + u2 <= W + 1;
              const int j = q + u2;
              ses[b] = ok ? EM(b_stemend, i, j) : kNegInf;
              sj[b] = ok ? s[j] : 0;
              sj1[b] = ok ? s[j + 1] : 0;
            }
#pragma unroll
            for (int b = 0; b < kRaAhead; b++) {
              const double se = ses[b];
              if (se != kNegInf) {
                const int type = ra_bp(lds, bi0, sj1[b]);
                if (type != 0) {
                  const double z = ra_loop_energy(lds, c.big, type, type2, u1, u2lo + b, bi1, sj[b], bp0, bq1);
                  temp = ra_lse(lds, temp, se + z);
                }
              }
            }
          }
        }
      }
      if (cell) {
        double out = kNegInf;
        if (t2raw != 0) {
          if (p != 0 && q != L) {
            const int type = ra_bp(lds, bp0, bq1);
            if (type != 0 && d + 2 <= W + 1) {
              const double ps = EM(b_stem, p - 1, q + 1);
              if (ps != kNegInf) temp = ra_lse(lds, temp, ps + lds.small[SL::kStack + type * 7 + type2]);
            }
          }
          out = temp;
          const double m2 = EM(b_multi2, p, q);
          if (m2 != kNegInf) {
            const double x = m2 + MLintern + dangle_energy(lds, v, t2raw, p, q);
            out = ra_lse(lds, x, out);
          }
        }
        if (role == 0) EM(b_stem, p, q) = out;
      }
    }
    } // !kF
    wave_sync();
    RA_PROF(12);
  }
  // remaining Beta_outer positions (columns stop at q = 4)
  if (role == 0 && lane == 1)
    for (int i = (L >= kTurn + 1 ? kTurn - 1 : L - 1); i >= 0; i--) beta_outer_at(i);
}

// ------------------------------------------------------------- bulge / interior loops
// Ordered scatter of every interior-loop term into the positions it makes accessible
// (raccess.cpp:626-665 / :695-752).  The sums per position k are sequential in the
// reference, so tuples are applied one at a time in (i, j, p, q) order; the lanes own the
// positions (k mod 64, NS slots for windows up to 64 * NS positions), so one tuple updates all its
// positions in one predicated instruction.
template <int NS>
__device__ __forceinline__ double slot_get(const double (&a)[NS], int sl) {
  double r = a[0];
#pragma unroll
  for (int t = 1; t < NS; t++)
    if (sl == t) r = a[t];
  return r;
}
template <int NS>
__device__ __forceinline__ void slot_put(double (&a)[NS], int sl, double v) {
#pragma unroll
  for (int t = 0; t < NS; t++)
    if (sl == t) a[t] = v;
}

template <bool kLogSum, int NS>
__device__ void biloop_run(const RaLds &lds, const RaConst &c, const SeqView &v, int delta, int lane) {
  const int L = v.L, W = v.W, S = v.S;
  const unsigned char *s = v.s;
  const double *a_stem = v.tab(A_STEM), *b_stemend = v.tab(B_STEMEND);
  double *bp = v.v(V_BP), *cbp = v.v(V_CBP), *bfl = v.v(V_BFLAG), *cfl = v.v(V_CFLAG);
  // accumulators of position k live on lane k & 63, slot (k >> 6) & (NS - 1); "has a term" flags: one bit per slot
  double accb[NS], accc[NS];
#pragma unroll
  for (int t = 0; t < NS; t++) accb[t] = accc[t] = 0;
  unsigned fb = 0, fc = 0;

  for (int i = 1; i < L - kTurn - 2; i++) {
    const int kb = i + 1; // smallest position this and later i can touch
    const int jend = imin(i + W, L);
    for (int j = i + kTurn + 3; j <= jend; j++) {
      const int type = ra_bp(lds, s[i], s[j]);
      if (type == 0) continue;
      const double bs = EM(b_stemend, i, j - 1);
      if (bs == kNegInf) continue;
      const int D = j - i;
      const int m = imin(kMaxLoop, D - 6); // u1 + u2 <= m
      const int bi1 = s[i + 1], bj1 = s[j - 1];
      for (int u1 = 0; u1 <= m; u1++) {
        // lanes take the q of this p in the reference's order: q ascending <=> u2 descending
        const int p = i + 1 + u1;
        const int u2 = (m - u1) - lane;
        bool valid = u2 >= 0 && !(u1 == 0 && u2 == 0);
        double val = 0;
        int q = 0;
        if (valid) {
          q = j - 1 - u2;
          int type2 = ra_bp(lds, s[p], s[q]);
          const double as = EM(a_stem, p - 1, q);
          valid = type2 != 0 && as != kNegInf;
          if (valid) {
            type2 = ra_rtype(type2);
            const double e = bs + ra_loop_energy_bf(lds, c.big, type, type2, u1, u2, bi1, bj1, s[p - 1], s[q + 1]) + as;
            val = kLogSum ? e : ra_expd(lds, e);
          }
        }
        unsigned long long mask = __ballot(valid);
        const int kl1 = p - delta; // left range [i+1, p-delta]
        while (mask) {
          const int src = __builtin_ctzll(mask);
          mask &= mask - 1;
          const double tv = __shfl(val, src);
          const int tq = __shfl(q, src);
          const int kr0 = tq + 1, kr1 = j - delta; // right range [q+1, j-delta]
#pragma unroll
          for (int h = 0; h < NS; h++) {
            const int k = kb + ((lane - kb) & 63) + 64 * h;
            const bool inl = k <= kl1; // k >= kb = i+1 always
            const bool inr = k >= kr0 && k <= kr1;
            if (inl || inr) {
              const bool last = inl ? k == kl1 : k == kr1;
              const int sl = (k >> 6) & (NS - 1);
              if (!kLogSum) {
                if (last) slot_put<NS>(accb, sl, slot_get<NS>(accb, sl) + tv);
                else slot_put<NS>(accc, sl, slot_get<NS>(accc, sl) + tv);
              } else {
                if (last) {
                  const double cur = slot_get<NS>(accb, sl);
                  const double nv = ((fb >> sl) & 1) ? ra_lse(lds, cur, tv) : tv;
                  slot_put<NS>(accb, sl, nv);
                  fb |= 1u << sl;
                } else {
                  const double cur = slot_get<NS>(accc, sl);
                  const double nv = ((fc >> sl) & 1) ? ra_lse(lds, cur, tv) : tv;
                  slot_put<NS>(accc, sl, nv);
                  fc |= 1u << sl;
                }
              }
            }
          }
        }
      }
    }
    // position k = i+1 receives nothing from later i: store it and recycle the slot
    if ((kb & 63) == lane) {
      const int sl = (kb >> 6) & (NS - 1);
      bp[kb - 1] = slot_get<NS>(accb, sl);
      cbp[kb - 1] = slot_get<NS>(accc, sl);
      bfl[kb - 1] = ((fb >> sl) & 1) ? 1.0 : 0.0;
      cfl[kb - 1] = ((fc >> sl) & 1) ? 1.0 : 0.0;
      slot_put<NS>(accb, sl, 0.0);
      slot_put<NS>(accc, sl, 0.0);
      fb &= ~(1u << sl);
      fc &= ~(1u << sl);
    }
  }
  // positions beyond the last i keep their zero-initialised value: every position that
  // can receive a term (k <= L - delta, k >= 2) has been flushed when i passed k - 1,
  // except those with k - 1 >= L - 5; flush what the last windows left behind.
  for (int k = imax(2, L - kTurn - 2 + 1); k <= L; k++) {
    if ((k & 63) == lane) {
      const int sl = (k >> 6) & (NS - 1);
      bp[k - 1] = slot_get<NS>(accb, sl);
      cbp[k - 1] = slot_get<NS>(accc, sl);
      bfl[k - 1] = ((fb >> sl) & 1) ? 1.0 : 0.0;
      cfl[k - 1] = ((fc >> sl) & 1) ? 1.0 : 0.0;
      slot_put<NS>(accb, sl, 0.0);
      slot_put<NS>(accc, sl, 0.0);
      fb &= ~(1u << sl);
      fc &= ~(1u << sl);
    }
  }
}

// Linear branch, float-overflow regime: the un-normalised sums are only ever used as
// `(float)sum` under a log, and fmath::log(+inf) is the constant 88.722839.  A position whose sum
// contains ONE term expd(e) with e >= 89 (> FLT_MAX by a third) therefore has the same final value
// whatever the other terms and their order are, and a position without any non-zero term
// (e <= -708.39.. for all of them) stays exactly 0.  This pass only classifies: per position and
// per sum (bulge/interior "b", conditional "c") two bits - some term is that large / some term is
// non-zero - as order-independent ORs over the same (i, j, p, q) enumeration, without the
// tuple-by-tuple scatter.  It stores +inf / 0 where that decides the result (k_access's
// finalisation then takes the same branches as with the true sums) and returns false when some
// position needs its true sum, in which case the caller runs the ordered pass.  For random RNA
// of 1 kb (log Z ~ 270) every position is decided here.
template <int NS>
__device__ bool biloop_classify(const RaLds &lds, const RaConst &c, const SeqView &v, int delta, int lane) {
  const int L = v.L, W = v.W, S = v.S;
  const unsigned char *s = v.s;
  const double *a_stem = v.tab(A_STEM), *b_stemend = v.tab(B_STEMEND);
  double *bp = v.v(V_BP), *cbp = v.v(V_CBP);
  // flags of position k live on lane k & 63, in nibble (k >> 6) & (NS - 1) of one word: bit 0 b-big, 1 b-nonzero,
  // 2 c-big, 3 c-nonzero
  unsigned fl = 0;
  bool undecided = false;
  auto flush = [&](int k) {
    if ((k & 63) == lane) {
      const int sl = (k >> 6) & (NS - 1);
      const unsigned f = (fl >> (4 * sl)) & 15u;
      const bool bbig = f & 1, bnz = f & 2, cbig = f & 4, cnz = f & 8;
      if (!((cbig || !cnz) && (cbig || bbig || !bnz))) undecided = true;
      bp[k - 1] = bnz ? __builtin_huge_val() : 0.0;
      cbp[k - 1] = cnz ? __builtin_huge_val() : 0.0;
      fl &= ~(15u << (4 * sl));
    }
  };
  for (int i = 1; i < L - kTurn - 2; i++) {
    const int kb = i + 1; // smallest position this and later i can touch
    const int jend = imin(i + W, L);
    // the closing pairs (i, j) of this i that exist, found for all j at once: lane t looks at j = jb + t
    for (int jb = i + kTurn + 3; jb <= jend; jb += kWave) {
      const int jl = jb + lane;
      const int typ_l = jl <= jend ? ra_bp(lds, s[i], s[jl]) : 0;
      const double bs_l = typ_l != 0 ? EM(b_stemend, i, jl - 1) : kNegInf;
      unsigned long long jmask = __ballot(typ_l != 0 && bs_l != kNegInf);
      while (jmask) {
        const int jt = __builtin_ctzll(jmask);
        jmask &= jmask - 1;
        const int j = jb + jt;
        const int type = __builtin_amdgcn_readlane(typ_l, jt);
        const double bs = readlane_f64(bs_l, jt);
        const int D = j - i;
        const int m = imin(kMaxLoop, D - 6); // u1 + u2 <= m
        const int bi1 = s[i + 1], bj1 = s[j - 1];
        const int kr1 = j - delta;
        // The tuples (u1, u2), u2 = 0 .. m - u1, of row u1 make m - u1 + 1 lanes; rows a and m + 1 - a together make
        // m + 1 <= 31, so half a wavefront takes two rows and a pass four (row 0 and - m odd - the middle row go alone).
        // The flags are ORs, so only two things matter about a pass: which rows have a non-zero / a big term (the left
        // ranges [i + 1, p - delta] depend on the row only) and which columns q have one (the right ranges [q + 1,
        // j - delta] on the column only).  Both are collected from the ballots with scalar instructions; the positions
        // are updated once per (i, j), not once per row.
        unsigned long long col_nz = 0, col_big = 0; // bit = q - (j - 1 - m)
        unsigned row_nz = 0, row_big = 0;           // bit = u1
        const int nslots = 1 + (m + 1) / 2;         // (0), (1, m), (2, m - 1), ...
        for (int s0 = 0; s0 < nslots; s0 += 2) {
          const int half = lane >> 5, l5 = lane & 31;
          const int slot = s0 + half;
          const int ra_ = slot, rb_ = slot == 0 ? -1 : m + 1 - slot; // the slot's rows (rb_ <= ra_: none / the same row)
          const int na = slot < nslots ? m - ra_ + 1 : 0, nb = (slot < nslots && rb_ > ra_) ? m - rb_ + 1 : 0;
          const bool in_a = l5 < na, in_b = !in_a && l5 < na + nb;
          const int u1 = in_a ? ra_ : rb_, t5 = in_a ? l5 : l5 - na;
          const int u2 = (m - u1) - t5; // lane t of a row holds q = q0 + t, q0 = j - 1 - (m - u1)
          const bool in = (in_a || in_b) && !(u1 == 0 && u2 == 0);
          const int pp = in ? i + 1 + u1 : i + 1, q = in ? j - 1 - u2 : j - 1; // (a lane without a tuple reads inside the window)
          const int type2raw = ra_bp(lds, s[pp], s[q]);
          const double as = EM(a_stem, pp - 1, q);
          const double z = ra_loop_energy_bf(lds, c.big, type, ra_rtype(type2raw), in ? u1 : 0, in ? u2 : 0, bi1, bj1, s[pp - 1], s[q + 1]);
          const double e = bs + z + as;
          const bool ok = in && type2raw != 0 && as != kNegInf;
          const bool nz = ok && e > -708.39641853226408; // ra_expd(e) != 0
          const bool big = ok && e >= 89.0;
          const unsigned long long nzall = __ballot(nz);
          if (nzall == 0) continue;
          const unsigned long long bigall = __ballot(big);
#pragma unroll
          for (int h = 0; h < 2; h++) { // (wave-uniform: scalar instructions)
            const int sl = s0 + h;
            if (sl >= nslots) continue;
            const int a_ = sl, b_ = sl == 0 ? -1 : m + 1 - sl;
            const int n_a = m - a_ + 1, n_b = b_ > a_ ? m - b_ + 1 : 0;
            const unsigned hn = (unsigned)(nzall >> (32 * h)), hb = (unsigned)(bigall >> (32 * h));
            const unsigned an = hn & ((n_a >= 32 ? 0u : (1u << n_a)) - 1u), ab = hb & ((n_a >= 32 ? 0u : (1u << n_a)) - 1u);
            col_nz |= (unsigned long long)an << a_;
            col_big |= (unsigned long long)ab << a_;
            row_nz |= (an != 0 ? 1u : 0u) << a_;
            row_big |= (ab != 0 ? 1u : 0u) << a_;
            if (n_b > 0) {
              const unsigned bn = (hn >> n_a) & ((1u << n_b) - 1u), bb = (hb >> n_a) & ((1u << n_b) - 1u);
              col_nz |= (unsigned long long)bn << b_;
              col_big |= (unsigned long long)bb << b_;
              row_nz |= (bn != 0 ? 1u : 0u) << b_;
              row_big |= (bb != 0 ? 1u : 0u) << b_;
            }
          }
        }
        if (row_nz == 0) continue;
        // the positions: left ranges end at p - delta = i + 1 + u1 - delta, right ranges at j - delta
        const int qbase = j - 1 - m;
#pragma unroll
        for (int h = 0; h < NS; h++) {
          const int k = kb + ((lane - kb) & 63) + 64 * h;
          unsigned add = 0;
          const int ustar = k - (i + 1 - delta); // the row whose left range ends at k
          if (ustar >= 0 && ustar <= m) {
            if ((row_nz >> ustar) & 1) add |= 2u | (((row_big >> ustar) & 1) ? 1u : 0u);
          }
          if (ustar < m) { // rows above: k lies inside their left ranges
            const int sh = ustar < 0 ? 0 : ustar + 1;
            if (row_nz >> sh) add |= 8u | ((row_big >> sh) ? 4u : 0u);
          }
          if (k <= kr1) { // right ranges [q + 1, j - delta]: the tuples with q <= k - 1
            const int nl = k - qbase;
            if (nl > 0) {
              const unsigned long long below = nl >= 64 ? ~0ull : ((1ull << nl) - 1);
              if (col_nz & below) add |= (k == kr1 ? 2u : 8u) | ((col_big & below) ? (k == kr1 ? 1u : 4u) : 0u);
            }
          }
          fl |= add << (4 * ((k >> 6) & (NS - 1)));
        }
      }
    }
    flush(kb); // position i+1 receives nothing from later i
  }
  for (int k = imax(2, L - kTurn - 2 + 1); k <= L; k++) flush(k);
  return __ballot(undecided) == 0;
}

// The LOGSUM branch (|log Z| > 690, raccess.cpp:683-771) for a WINDOW of 64 positions: one logsumexp per (tuple,
// position) in the reference's (i, j, p, q) order makes the ordered pass above a chain of ~5,600 tuples per nucleotide
// on its one wavefront (2.35 ms per nucleotide: 47 s for a lone 20 kb sequence).  The sums of different positions do
// not depend on each other, only the order WITHIN a position matters: a wavefront owns positions k0 .. k0 + 63 (one per
// lane), walks the tuples of every i that can reach them (k - W + delta <= i <= k - 1) in the same order - the lanes
// evaluating the tuples of one (i, j, u1) side by side, as above -, and a lane folds a tuple in only if its position lies
// in one of the tuple's two ranges.  Every window costs what ~(W + 64) values of i cost, all windows of all sequences
// run at once: the time of a sequence no longer grows with its length.
// (kLogSum false: the linear branch, the same walk with plain sums of expd(e) - for the windows the classification below
// cannot decide, and for sequences with log Z < 120)
template <bool kLogSum>
__device__ void biloop_window(const RaLds &lds, const RaConst &c, const SeqView &v, int delta, int lane, int k0) {
  const int L = v.L, W = v.W, S = v.S;
  const int k = k0 + lane; // this lane's position
  const unsigned char *s = v.s;
  const double *a_stem = v.tab(A_STEM), *b_stemend = v.tab(B_STEMEND);
  double accb = 0, accc = 0;
  bool fb = false, fc = false;
  const int i_lo = imax(1, k0 - W + delta), i_hi = imin(L - kTurn - 3, k0 + kWave - 2); // i < L - kTurn - 2, i <= k - 1
  for (int i = i_lo; i <= i_hi; i++) {
    const int jend = imin(i + W, L);
    // (closing pairs whose right range ends before the window and whose left ranges cannot reach it either are skipped:
    // left ranges end at p - delta <= i + 1 + 30 - delta, right ranges at j - delta)
    for (int j = i + kTurn + 3; j <= jend; j++) {
      if (j - delta < k0 && i + 1 + kMaxLoop - delta < k0) continue;
      const int type = ra_bp(lds, s[i], s[j]);
      if (type == 0) continue;
      const double bs = EM(b_stemend, i, j - 1);
      if (bs == kNegInf) continue;
      const int D = j - i;
      const int m = imin(kMaxLoop, D - 6); // u1 + u2 <= m
      const int bi1 = s[i + 1], bj1 = s[j - 1];
      for (int u1 = 0; u1 <= m; u1++) {
        // lanes take the q of this p in the reference's order: q ascending <=> u2 descending
        const int p = i + 1 + u1;
        const int u2 = (m - u1) - lane;
        bool valid = u2 >= 0 && !(u1 == 0 && u2 == 0);
        double val = 0;
        int q = 0;
        if (valid) {
          q = j - 1 - u2;
          int type2 = ra_bp(lds, s[p], s[q]);
          const double as = EM(a_stem, p - 1, q);
          valid = type2 != 0 && as != kNegInf;
          if (valid) {
            type2 = ra_rtype(type2);
            const double e = bs + ra_loop_energy_bf(lds, c.big, type, type2, u1, u2, bi1, bj1, s[p - 1], s[q + 1]) + as;
            val = kLogSum ? e : ra_expd(lds, e);
          }
        }
        unsigned long long mask = __ballot(valid);
        const int kl1 = p - delta; // left range [i+1, p-delta]
        const bool inl = k >= i + 1 && k <= kl1;
        while (mask) {
          const int src = __builtin_ctzll(mask);
          mask &= mask - 1;
          const double tv = __shfl(val, src);
          const int tq = __shfl(q, src);
          const int kr0 = tq + 1, kr1 = j - delta; // right range [q+1, j-delta]
          const bool inr = k >= kr0 && k <= kr1;
          if (inl || inr) {
            const bool last = inl ? k == kl1 : k == kr1;
            if (!kLogSum) {
              if (last) accb += tv;
              else accc += tv;
            } else if (last) {
              accb = fb ? ra_lse(lds, accb, tv) : tv;
              fb = true;
            } else {
              accc = fc ? ra_lse(lds, accc, tv) : tv;
              fc = true;
            }
          }
        }
      }
    }
  }
  if (k <= L) {
    v.v(V_BP)[k - 1] = accb;
    v.v(V_CBP)[k - 1] = accc;
    v.v(V_BFLAG)[k - 1] = fb ? 1.0 : 0.0;
    v.v(V_CFLAG)[k - 1] = fc ? 1.0 : 0.0;
  }
}

// biloop_classify for a window of 64 positions (one per lane): the flags are order-independent ORs, so a window only
// has to look at the closing pairs whose ranges can reach it.  Stores +inf / 0 where that decides a position's sums and
// returns false when some position of the window needs its true sums.
__device__ bool biloop_classify_window(const RaLds &lds, const RaConst &c, const SeqView &v, int delta, int lane, int k0) {
  const int L = v.L, W = v.W, S = v.S;
  const int k = k0 + lane;
  const unsigned char *s = v.s;
  const double *a_stem = v.tab(A_STEM), *b_stemend = v.tab(B_STEMEND);
  unsigned fl = 0; // bit 0 b-big, 1 b-nonzero, 2 c-big, 3 c-nonzero
  const int i_lo = imax(1, k0 - W + delta), i_hi = imin(L - kTurn - 3, k0 + kWave - 2);
  for (int i = i_lo; i <= i_hi; i++) {
    const int jend = imin(i + W, L);
    for (int jb = i + kTurn + 3; jb <= jend; jb += kWave) {
      const int jl = jb + lane;
      const int typ_l = jl <= jend ? ra_bp(lds, s[i], s[jl]) : 0;
      const double bs_l = typ_l != 0 ? EM(b_stemend, i, jl - 1) : kNegInf;
      // (pairs that cannot reach the window: right ranges end at j - delta, left ranges at i + 1 + 30 - delta at the latest)
      unsigned long long jmask = __ballot(typ_l != 0 && bs_l != kNegInf && !(jl - delta < k0 && i + 1 + kMaxLoop - delta < k0));
      while (jmask) {
        const int jt = __builtin_ctzll(jmask);
        jmask &= jmask - 1;
        const int j = jb + jt;
        const int type = __builtin_amdgcn_readlane(typ_l, jt);
        const double bs = readlane_f64(bs_l, jt);
        const int D = j - i;
        const int m = imin(kMaxLoop, D - 6); // u1 + u2 <= m
        const int bi1 = s[i + 1], bj1 = s[j - 1];
        const int kr1 = j - delta;
        unsigned long long col_nz = 0, col_big = 0; // bit = q - (j - 1 - m)
        unsigned row_nz = 0, row_big = 0;           // bit = u1
        const int nslots = 1 + (m + 1) / 2;         // (0), (1, m), (2, m - 1), ... (see biloop_classify)
        for (int s0 = 0; s0 < nslots; s0 += 2) {
          const int half = lane >> 5, l5 = lane & 31;
          const int slot = s0 + half;
          const int ra_ = slot, rb_ = slot == 0 ? -1 : m + 1 - slot;
          const int na = slot < nslots ? m - ra_ + 1 : 0, nb = (slot < nslots && rb_ > ra_) ? m - rb_ + 1 : 0;
          const bool in_a = l5 < na, in_b = !in_a && l5 < na + nb;
          const int u1 = in_a ? ra_ : rb_, t5 = in_a ? l5 : l5 - na;
          const int u2 = (m - u1) - t5;
          const bool in = (in_a || in_b) && !(u1 == 0 && u2 == 0);
          const int pp = in ? i + 1 + u1 : i + 1, q = in ? j - 1 - u2 : j - 1;
          const int type2raw = ra_bp(lds, s[pp], s[q]);
          const double as = EM(a_stem, pp - 1, q);
          const double z = ra_loop_energy_bf(lds, c.big, type, ra_rtype(type2raw), in ? u1 : 0, in ? u2 : 0, bi1, bj1, s[pp - 1], s[q + 1]);
          const double e = bs + z + as;
          const bool ok = in && type2raw != 0 && as != kNegInf;
          const bool nz = ok && e > -708.39641853226408; // ra_expd(e) != 0
          const bool big = ok && e >= 89.0;
          const unsigned long long nzall = __ballot(nz);
          if (nzall == 0) continue;
          const unsigned long long bigall = __ballot(big);
#pragma unroll
          for (int h = 0; h < 2; h++) { // (wave-uniform: scalar instructions)
            const int sl = s0 + h;
            if (sl >= nslots) continue;
            const int a_ = sl, b_ = sl == 0 ? -1 : m + 1 - sl;
            const int n_a = m - a_ + 1, n_b = b_ > a_ ? m - b_ + 1 : 0;
            const unsigned hn = (unsigned)(nzall >> (32 * h)), hb = (unsigned)(bigall >> (32 * h));
            const unsigned an = hn & ((n_a >= 32 ? 0u : (1u << n_a)) - 1u), ab = hb & ((n_a >= 32 ? 0u : (1u << n_a)) - 1u);
            col_nz |= (unsigned long long)an << a_;
            col_big |= (unsigned long long)ab << a_;
            row_nz |= (an != 0 ? 1u : 0u) << a_;
            row_big |= (ab != 0 ? 1u : 0u) << a_;
            if (n_b > 0) {
              const unsigned bn = (hn >> n_a) & ((1u << n_b) - 1u), bb = (hb >> n_a) & ((1u << n_b) - 1u);
              col_nz |= (unsigned long long)bn << b_;
              col_big |= (unsigned long long)bb << b_;
              row_nz |= (bn != 0 ? 1u : 0u) << b_;
              row_big |= (bb != 0 ? 1u : 0u) << b_;
            }
          }
        }
        if (row_nz == 0) continue;
        // this lane's position: left ranges [i + 1, i + 1 + u1 - delta], right ranges [q + 1, j - delta]
        const int qbase = j - 1 - m;
        unsigned add = 0;
        if (k >= i + 1) {
          const int ustar = k - (i + 1 - delta); // the row whose left range ends at k
          if (ustar >= 0 && ustar <= m) {
            if ((row_nz >> ustar) & 1) add |= 2u | (((row_big >> ustar) & 1) ? 1u : 0u);
          }
          if (ustar < m) { // rows above: k lies inside their left ranges
            const int sh = ustar < 0 ? 0 : ustar + 1;
            if (row_nz >> sh) add |= 8u | ((row_big >> sh) ? 4u : 0u);
          }
        }
        if (k <= kr1) { // right ranges: the tuples with q <= k - 1
          const int nl = k - qbase;
          if (nl > 0) {
            const unsigned long long below = nl >= 64 ? ~0ull : ((1ull << nl) - 1);
            if (col_nz & below) add |= (k == kr1 ? 2u : 8u) | ((col_big & below) ? (k == kr1 ? 1u : 4u) : 0u);
          }
        }
        fl |= add;
      }
    }
  }
  const bool bbig = fl & 1, bnz = fl & 2, cbig = fl & 4, cnz = fl & 8;
  const bool undecided = k <= L && !((cbig || !cnz) && (cbig || bbig || !bnz));
  if (__ballot(undecided) != 0) return false;
  if (k <= L) {
    v.v(V_BP)[k - 1] = bnz ? __builtin_huge_val() : 0.0;
    v.v(V_CBP)[k - 1] = cnz ? __builtin_huge_val() : 0.0;
  }
  return true;
}

// The bulge / interior-loop sums by windows of 64 positions, a wavefront each.  RaBatch::windows_all: every sequence (a launch
// with few sequences: k_biloop is not launched at all); else only the sequences in the LOGSUM regime.
__global__ __launch_bounds__(kBlock) void k_biloop_win(RaBatch b, RaConst c) {
  __shared__ RaLds lds;
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.y;
  const SeqView v = make_view(b, idx);
  const int L = v.L;
  const double pf = v.v(V_AO)[L];
  const bool linear = pf >= -690 && pf <= 690; // raccess.cpp:500-507
  // (both tests are the same for every thread of the workgroup: nobody is left waiting at the barrier of the table load)
  if (linear && !b.windows_all) return; // the linear branch of this launch: k_biloop
  if (1 + kWave * (int)(blockIdx.x * kWavesPerBlock) > L) return;
  ra_load_lds(lds, c);
  const int k0 = 1 + kWave * (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (k0 > L) return;
  if (!linear) {
    biloop_window<true>(lds, c, v, b.delta, lane, k0);
  } else if (pf < 120.0 || !biloop_classify_window(lds, c, v, b.delta, lane, k0)) {
    biloop_window<false>(lds, c, v, b.delta, lane, k0);
  }
}

template <int NS>
__global__ __launch_bounds__(kBlock) void k_biloop(RaBatch b, RaConst c) {
  __shared__ RaLds lds;
  ra_load_lds(lds, c);
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (idx >= b.nseq) return;
  const SeqView v = make_view(b, idx);
  const double pf = v.v(V_AO)[v.L];
  if (pf >= -690 && pf <= 690) { // raccess.cpp:500-507
    // large log Z: nearly always decided by the classification alone
    if (pf < 120.0 || !biloop_classify<NS>(lds, c, v, b.delta, lane)) biloop_run<false, NS>(lds, c, v, b.delta, lane);
  } else if (b.logsum_windows == 0) { // (else: k_biloop_logsum, a wavefront per window of 64 positions)
    biloop_run<true, NS>(lds, c, v, b.delta, lane);
  }
}

// ---------------------------------------------------------------------- accessibility
// One lane per window start x (raccess.cpp:510-527); hairpin (:536-579), multi (:581-612)
// and exterior (:530-534) probabilities are gathers that are independent per x.
__device__ __forceinline__ double multi_prob(const RaLds &lds, const SeqView &v, int x, int w) {
  const int L = v.L, W = v.W, S = v.S;
  const double *a_multi = v.tab(A_MULTI), *a_multi2 = v.tab(A_MULTI2);
  const double *b_multi = v.tab(B_MULTI), *b_multi2 = v.tab(B_MULTI2);
  double temp = 0;
  bool flag = false;
  for (int i = x + w - 1; i <= imin(x + W, L); i++) {
    const double bb = EM(b_multi, x - 1, i), aa = EM(a_multi, x + w - 1, i);
    if (bb != kNegInf && aa != kNegInf) {
      temp = flag ? ra_lse(lds, temp, bb + aa) : bb + aa;
      flag = true;
    }
  }
  for (int i = imax(0, x + w - 1 - W); i < x; i++) {
    const double bb = EM(b_multi2, i, x + w - 1), aa = EM(a_multi2, i, x - 1);
    if (bb != kNegInf && aa != kNegInf) {
      temp = flag ? ra_lse(lds, temp, bb + aa) : bb + aa;
      flag = true;
    }
  }
  return flag ? ra_expd(lds, temp - v.v(V_AO)[L]) : 0.0;
}

// (a thread per window start: blockIdx.y = the sequence, blockIdx.x = its chunk of kBlock positions)
__global__ __launch_bounds__(kBlock) void k_access(RaBatch b, RaConst c) {
  __shared__ RaLds lds;
  const int idx = blockIdx.y;
  if (1 + (int)(blockIdx.x * kBlock) > b.desc[idx].L) return; // (the whole workgroup: before the barrier of the table load)
  ra_load_lds(lds, c);
  using SL = RaSmallLayout;
  const SeqView v = make_view(b, idx);
  const int L = v.L, W = v.W, S = v.S, delta = b.delta;
  const unsigned char *s = v.s;
  const double *b_stemend = v.tab(B_STEMEND);
  const double *ao = v.v(V_AO), *bo = v.v(V_BO);
  const double *bpv = v.v(V_BP), *cbpv = v.v(V_CBP), *bfl = v.v(V_BFLAG), *cfl = v.v(V_CFLAG);
  const double Z = ao[L];
  const double kT = lds.small[SL::kKT];
  const bool logsum = !(Z >= -690 && Z <= 690);
  float *acc = b.acc + v.out_off, *cond = b.cond + v.out_off;

  for (int x = 1 + (int)(blockIdx.x * kBlock + threadIdx.x); x <= L; x += L) { // (one position per thread)
    float acc_x = 0.0f;
    const bool has_acc = x + delta - 1 <= L;
    if (has_acc) {
      // hairpin loops enclosing [x, x+delta-1] (:536-579)
      double temp = 0, c_temp = 0;
      bool flag = false, c_flag = false;
      for (int i = imax(1, x - W); i < x; i++) {
        for (int j = x + delta; j <= imin(i + W, L); j++) {
          const double se = EM(b_stemend, i, j - 1);
          if (se != kNegInf) {
            const int type = ra_bp(lds, s[i], s[j]);
            const double h = se + ra_hairpin_energy(lds, type, j - i - 1, s[i + 1], s[j - 1]);
            if (j == x + delta) {
              temp = flag ? ra_lse(lds, temp, h) : h;
              flag = true;
            } else {
              c_temp = c_flag ? ra_lse(lds, c_temp, h) : h;
              c_flag = true;
            }
          }
        }
      }
      if (flag && c_flag) temp = ra_lse(lds, temp, c_temp);
      if (!flag && c_flag) {
        temp = c_temp;
        flag = true;
      }
      const double hp = flag ? ra_expd(lds, temp - Z) : 0.0;
      const double chp = c_flag ? ra_expd(lds, c_temp - Z) : 0.0;

      // bulge / interior loop terms: finalisation of the scatter sums (:667-680 / :754-770)
      double bl = bpv[x - 1], cbl = cbpv[x - 1];
      if (!logsum) {
        if (bl != 0) bl = ra_expd(lds, (double)ra_logf(lds, (float)(bl + cbl)) - Z);
        if (cbl != 0) cbl = ra_expd(lds, (double)ra_logf(lds, (float)cbl) - Z);
      } else {
        const bool f1 = bfl[x - 1] != 0, f2 = cfl[x - 1] != 0;
        if (f1 && f2) bl = ra_lse(lds, bl, cbl);
        if (!f1 && f2) bl = cbl;
        if (f1) bl = ra_expd(lds, bl - Z);
        if (f2) cbl = ra_expd(lds, cbl - Z);
      }

      double prob = 0.0;
      prob += ra_expd(lds, ao[x - 1] + bo[x + delta - 1] - Z);
      prob += hp;
      prob += bl;
      prob += multi_prob(lds, v, x, delta);
      acc_x = (float)(((double)(-ra_logf(lds, (float)prob)) * kT) / 1000);
      acc[x - 1] = acc_x;

      if (x + delta - 1 < L) {
        double cp = 0.0;
        cp += ra_expd(lds, ao[x - 1] + bo[x + delta] - Z);
        cp += chp;
        cp += cbl;
        cp += multi_prob(lds, v, x, delta + 1);
        cond[x + delta - 1] = (float)(((double)(-ra_logf(lds, (float)cp)) * kT) / 1000 - (double)acc_x);
      }
    }
  }
}

} // namespace

#ifdef PRB_GAP_PROFILE
extern "C" int prb_debug_ra_profile(unsigned long long *out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ra_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ra_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// whether a wait between the wavefronts of a workgroup (kH = 3) has given up since the last call; clears the flag.
// (After the launch has completed.)
hipError_t ra_watchdog_tripped(int *tripped, hipStream_t stream) {
  *tripped = 0;
  hipError_t e = hipMemcpyFromSymbolAsync(tripped, HIP_SYMBOL(g_ra_watchdog), sizeof(int), 0, hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  if (*tripped) {
    const int zero = 0;
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ra_watchdog), &zero, sizeof(int), 0, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  return e;
}

hipError_t ra_launch(const RaBatch &b, const RaConst &c, int64_t band_elems, int64_t vec_elems,
                     hipStream_t stream) {
  if (b.nseq <= 0) return hipSuccess;
  const int fill_blocks = 2048;
  hipLaunchKernelGGL(k_fill, dim3(fill_blocks), dim3(256), 0, stream, b.band, band_elems, kNegInf);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, stream, b.vec, vec_elems, 0.0);
  const int blocks = (b.nseq + kWavesPerBlock - 1) / kWavesPerBlock;
  // cells of a column (spans 3 .. W + 1) and terms of Beta_outer (W + 1) per pass of 64 lanes: two passes up to a
  // maximal span of 127 (the default is 70), four up to kRaMaxSpan
  if (b.W + 1 <= 128) {
    // Few sequences (a batch of queries): a workgroup per sequence, helper wavefronts for the big folds - the time is one
    // wavefront's chain, the GPU otherwise idle.  Many (a database): a wavefront per sequence, as many side by side as fit.
    const int helpers = b.helpers >= 0 ? b.helpers : 0;
    const bool masks = b.W + 2 + kMaxLoop <= 128;
    if (helpers == 3 && masks) {
      hipLaunchKernelGGL((k_inside<2, 3>), dim3(b.nseq), dim3(kWave * 4), 0, stream, b, c);
      hipLaunchKernelGGL((k_outside<2, 3>), dim3(b.nseq), dim3(kWave * 4), 0, stream, b, c);
    } else if (helpers == 2 && masks) {
      hipLaunchKernelGGL((k_inside<2, 2>), dim3(b.nseq), dim3(kWave * 3), 0, stream, b, c);
      hipLaunchKernelGGL((k_outside<2, 2>), dim3(b.nseq), dim3(kWave * 3), 0, stream, b, c);
    } else if (helpers == 1 && masks) {
      hipLaunchKernelGGL((k_inside<2, 1>), dim3(b.nseq), dim3(kWave * 2), 0, stream, b, c);
      hipLaunchKernelGGL((k_outside<2, 1>), dim3(b.nseq), dim3(kWave * 2), 0, stream, b, c);
    } else {
      hipLaunchKernelGGL((k_inside<2, 0>), dim3(blocks), dim3(kBlock), 0, stream, b, c);
      hipLaunchKernelGGL((k_outside<2, 0>), dim3(blocks), dim3(kBlock), 0, stream, b, c);
    }
    if (!(b.windows_all && b.logsum_windows > 0)) hipLaunchKernelGGL(k_biloop<2>, dim3(blocks), dim3(kBlock), 0, stream, b, c);
  } else {
    hipLaunchKernelGGL((k_inside<4, 0>), dim3(blocks), dim3(kBlock), 0, stream, b, c);
    hipLaunchKernelGGL((k_outside<4, 0>), dim3(blocks), dim3(kBlock), 0, stream, b, c);
    if (!(b.windows_all && b.logsum_windows > 0)) hipLaunchKernelGGL(k_biloop<4>, dim3(blocks), dim3(kBlock), 0, stream, b, c);
  }
  if (b.logsum_windows > 0) {
    const int wblocks = (b.logsum_windows + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(k_biloop_win, dim3(wblocks, b.nseq), dim3(kBlock), 0, stream, b, c);
  }
  hipLaunchKernelGGL(k_access, dim3((b.lmax + kBlock - 1) / kBlock, b.nseq), dim3(kBlock), 0, stream, b, c);
  return hipGetLastError();
}

} // namespace prb
