// C ABI, part 1: context, errors, timers, and stage 1 (accessibility).
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "../../include/priblast_hip.h"
#include "context.hpp"
#include "raccess_kernels.hpp"

namespace prb {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int hip_fail(hipError_t e, const char *what) {
  g_error = std::string("HIP error '") + hipGetErrorString(e) + "' in " + what;
  return PRB_ERR_HIP;
}

int DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return PRB_OK;
  if (p) {
    hipError_t e = hipFree(p);
    p = nullptr;
    cap = 0;
    if (e != hipSuccess) return hip_fail(e, "hipFree");
  }
  // (room to grow into, except for the giants: an eighth of 64 GB is what the next buffer lacks)
  size_t want = bytes > ((size_t)8 << 30) ? bytes : bytes + bytes / 8;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // (the failed attempt is not the next launch's error)
    e = hipMalloc(&p, bytes);
    want = bytes;
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    p = nullptr;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    set_error("out of device memory allocating " + std::to_string(bytes) + " bytes (" + std::to_string(free_b >> 20) + " MiB of " +
              std::to_string(total_b >> 20) + " free)");
    return PRB_ERR_NOMEM;
  }
  cap = want;
  return PRB_OK;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

static std::string default_param_file() {
  Dl_info info;
  if (dladdr((void *)&default_param_file, &info) && info.dli_fname) {
    std::string so = info.dli_fname; // .../priblast_amd/lib/libpriblast_hip.so
    size_t k = so.rfind('/');
    std::string dir = k == std::string::npos ? "." : so.substr(0, k);
    return dir + "/../params/rna_andronescu2007.par";
  }
  return "priblast_amd/params/rna_andronescu2007.par";
}

} // namespace prb

int prb_ctx::time_begin() {
  PRB_HIP(hipEventRecord(ev0, stream));
  return PRB_OK;
}
int prb_ctx::time_end(const char *stage, int64_t launches) {
  PRB_HIP(hipEventRecord(ev1, stream));
  PRB_HIP(hipEventSynchronize(ev1));
  float ms = 0;
  PRB_HIP(hipEventElapsedTime(&ms, ev0, ev1));
  auto &t = timers[stage];
  t.ms += ms;
  t.launches += launches;
  return PRB_OK;
}

using namespace prb;

extern "C" {

const char *prb_last_error(void) { return g_error.c_str(); }
const char *prb_version(void) { return "priblast-hip 0.1 (gfx950)"; }

void prb_ris_opts_default(prb_ris_opts *o) {
  o->max_seed_length = 20;
  o->hybrid_threshold = -6.0;
  o->interaction_threshold = -4.0;
  o->final_threshold = -8.0;
  o->drop_out_wo_gap = 5;
  o->drop_out_w_gap = 16;
  o->min_helix_length = 3;
  o->output_style = 0;
}

int prb_search_const_upload(prb_ctx *ctx); // capi_search.hip
void prb_search_const_free(prb_ctx *ctx);

} // extern "C" (reopened below)

// everything of prb_ctx_create that can fail half-way: the caller destroys the context then
static int ctx_init(prb_ctx *ctx, int device, const char *param_file) {
  PRB_HIP(hipSetDevice(device));
  ctx->device = device;
  std::string pf = param_file ? param_file : default_param_file();
  std::string err = load_energy_params(pf, ctx->params);
  if (!err.empty()) {
    set_error(err);
    return PRB_ERR_IO;
  }
  build_raccess_tables(ctx->params, ctx->ra_tables);
  build_fmath_tables(ctx->fm_tables);
  PRB_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  PRB_HIP(hipEventCreate(&ctx->ev0));
  PRB_HIP(hipEventCreate(&ctx->ev1));
  auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
    int rc = b.ensure(bytes);
    if (rc) return rc;
    PRB_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return PRB_OK;
  };
  int rc;
  if ((rc = up(ctx->d_expd, ctx->fm_tables.expd_tbl.data(), 2048 * sizeof(uint64_t)))) return rc;
  if ((rc = up(ctx->d_log, ctx->fm_tables.log_tbl.data(), 4096 * sizeof(float)))) return rc;
  if ((rc = up(ctx->d_small, ctx->ra_tables.small.data(), ctx->ra_tables.small.size() * sizeof(double)))) return rc;
  if ((rc = up(ctx->d_big, ctx->ra_tables.big.data(), ctx->ra_tables.big.size() * sizeof(double)))) return rc;
  RaConst &c = ctx->ra_const;
  c.expd_tbl = ctx->d_expd.as<uint64_t>();
  c.log_tbl = ctx->d_log.as<float>();
  c.small = ctx->d_small.as<double>();
  c.big = ctx->d_big.as<double>();
  c.expd_a = ctx->fm_tables.expd_a;
  c.expd_ra = ctx->fm_tables.expd_ra;
  c.c_log2 = ctx->fm_tables.c_log2;
  for (int a = 0; a < 5; a++)
    for (int b = 0; b < 5; b++) c.bp_pair[a * 5 + b] = (unsigned char)ctx->params.bp_pair[a][b];
  if ((rc = prb_search_const_upload(ctx))) return rc;
  size_t free_b = 0, total_b = 0;
  PRB_HIP(hipMemGetInfo(&free_b, &total_b));
  const char *env = getenv("PRB_RACCESS_WORKSPACE_GB");
  // default: room for one sequence per resident wavefront at 2 kb (256 CUs x 12 wavefronts x 2,000 nt x 7.5 KB): a launch
  // with fewer sequences than that is bound by the latency of one sequence, not by throughput
  double gb = env ? atof(env) : 48.0;
  ctx->ra_budget_bytes = (size_t)std::min(gb * (double)(1ull << 30), 0.5 * (double)free_b);
  return PRB_OK;
}

extern "C" {

int prb_ctx_create(int device, const char *param_file, prb_ctx **out) {
  if (!out) return PRB_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("no HIP device available (this library has no CPU path)");
    return PRB_ERR_HIP;
  }
  if (device < 0 || device >= ndev) {
    set_error("device index out of range");
    return PRB_ERR_ARG;
  }
  prb_ctx *ctx = nullptr;
  int rc;
  try {
    ctx = new prb_ctx();
    rc = ctx_init(ctx, device, param_file);
  } catch (const std::exception &ex) { // (no exception leaves the C ABI)
    set_error(std::string("prb_ctx_create: ") + ex.what());
    rc = PRB_ERR_NOMEM;
  }
  if (rc != PRB_OK) {
    const std::string keep = prb_last_error(); // (destroying must not lose the reason)
    if (ctx) prb_ctx_destroy(ctx);
    set_error(keep);
    return rc;
  }
  *out = ctx;
  return PRB_OK;
}


void prb_ctx_destroy(prb_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  prb_search_const_free(ctx);
  for (DevBuf *b : {&ctx->d_expd, &ctx->d_log, &ctx->d_small, &ctx->d_big, &ctx->ra_band, &ctx->ra_vec,
                    &ctx->ra_codes, &ctx->ra_desc, &ctx->ra_acc, &ctx->ra_cond})
    b->release();
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int prb_ctx_synchronize(prb_ctx *ctx) {
  if (!ctx) return PRB_ERR_ARG;
  PRB_HIP(hipStreamSynchronize(ctx->stream));
  return PRB_OK;
}

int prb_ctx_stage_ms(prb_ctx *ctx, const char *stage, double *ms, int64_t *launches) {
  if (!ctx || !stage) return PRB_ERR_ARG;
  if (std::strcmp(stage, "slow_hits") == 0) { // counter, not a timer
    if (ms) *ms = 0;
    if (launches) *launches = ctx->slow_hits;
    return PRB_OK;
  }
  auto it = ctx->timers.find(stage);
  if (ms) *ms = it == ctx->timers.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == ctx->timers.end() ? 0 : it->second.launches;
  return PRB_OK;
}
void prb_ctx_reset_timers(prb_ctx *ctx) {
  if (ctx) {
    ctx->timers.clear();
    ctx->slow_hits = 0;
  }
}

} // extern "C"

namespace prb {

// raccess.cpp:55-67: A/a=1, C/c=2, G/g=3, T/t/U/u=4, anything else 0
static inline unsigned char ra_code(char ch) {
  switch (ch) {
  case 'A': case 'a': return 1;
  case 'C': case 'c': return 2;
  case 'G': case 'g': return 3;
  case 'T': case 't': case 'U': case 'u': return 4;
  default: return 0;
  }
}

// Runs Raccess for `nseq` sequences (sequence i = seqs[in_off[i] .. in_off[i] + lens[i]));
// results go to the DEVICE arrays d_acc / d_cond, which the caller has zero-filled, sequence i
// at out_off[i].  Sequences are processed longest first in chunks that fit the workspace budget.
int run_accessibility(prb_ctx *ctx, int32_t nseq, const char *seqs, const int64_t *in_off, const int32_t *lens,
                      const int64_t *out_off, int W, int delta, float *d_acc, float *d_cond) {
  static_assert(kRaMaxSpan + 2 < RaSmallLayout::kHairpinN, "hairpin table covers every span");
  if (W < 1 || delta < 2 || W > kRaMaxSpan) {
    set_error("unsupported (maximal span -w, min accessible length -d): this build needs 1 <= w <= 255 and d >= 2");
    return PRB_ERR_ARG;
  }
  PRB_HIP(hipSetDevice(ctx->device));
  std::vector<int32_t> order(nseq);
  std::iota(order.begin(), order.end(), 0);
  auto len = [&](int i) { return (int64_t)lens[i]; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len(a) > len(b); });
  int32_t pos = 0;
  std::vector<RaSeqDesc> desc;
  std::vector<unsigned char> codes;
  while (pos < nseq) {
    desc.clear();
    codes.clear();
    int64_t band = 0, vec = 0;
    int32_t end = pos;
    while (end < nseq) {
      const int64_t L = len(order[end]);
      const size_t need = (size_t)(band + ra_band_elems((int)L, W) + vec + ra_vec_elems((int)L)) * sizeof(double);
      if (end > pos && need > ctx->ra_budget_bytes) break;
      RaSeqDesc d;
      d.band_off = band;
      d.vec_off = vec;
      d.code_off = (int64_t)codes.size();
      d.out_off = out_off[order[end]];
      d.L = (int32_t)L;
      d.pad = 0;
      desc.push_back(d);
      band += ra_band_elems((int)L, W);
      vec += ra_vec_elems((int)L);
      codes.push_back(0);
      const char *sp = seqs + in_off[order[end]];
      for (int64_t k = 0; k < L; k++) codes.push_back(ra_code(sp[k]));
      codes.push_back(0);
      end++;
    }
    const int32_t n = end - pos;
    int rc;
    if ((rc = ctx->ra_band.ensure((size_t)band * sizeof(double)))) return rc;
    if ((rc = ctx->ra_vec.ensure((size_t)vec * sizeof(double)))) return rc;
    if ((rc = ctx->ra_codes.ensure(codes.size()))) return rc;
    if ((rc = ctx->ra_desc.ensure(desc.size() * sizeof(RaSeqDesc)))) return rc;
    PRB_HIP(hipMemcpyAsync(ctx->ra_codes.p, codes.data(), codes.size(), hipMemcpyHostToDevice, ctx->stream));
    PRB_HIP(hipMemcpyAsync(ctx->ra_desc.p, desc.data(), desc.size() * sizeof(RaSeqDesc), hipMemcpyHostToDevice,
                           ctx->stream));
    RaBatch b;
    b.desc = ctx->ra_desc.as<RaSeqDesc>();
    b.nseq = n;
    b.lmax = (int32_t)len(order[pos]); // (longest first)
    b.W = W;
    b.delta = delta;
    b.band = ctx->ra_band.as<double>();
    b.vec = ctx->ra_vec.as<double>();
    b.codes = ctx->ra_codes.as<unsigned char>();
    b.acc = d_acc;
    b.cond = d_cond;
    {
      // the LOGSUM branch of the bulge / interior sums by windows of 64 positions (which sequences take it is only known on
      // the device, after the inside pass: the grid covers every sequence, the others' workgroups leave at once)
      const char *e = std::getenv("PRB_RACCESS_LOGSUM_WINDOWS"); // 0: the ordered pass on the sequence's own wavefront (tests)
      const int64_t lmax = len(order[pos]);                         // (longest first)
      b.logsum_windows = (e && std::atoi(e) == 0) ? 0 : (int32_t)((lmax + 63) / 64);
      // helper wavefronts for the big folds when the launch leaves most of the GPU idle anyway (three workgroups of three
      // wavefronts fit a compute unit); PRB_RACCESS_HELPERS = 0 / 1 / 2 says otherwise
      const char *h = std::getenv("PRB_RACCESS_HELPERS");
      b.helpers = h ? std::max(0, std::min(3, std::atoi(h))) : (n <= 512 ? 3 : 0);
      const char *wa = std::getenv("PRB_RACCESS_WINDOWS_ALL"); // (tests: 0 / 1 whatever the size of the launch)
      b.windows_all = wa ? (std::atoi(wa) != 0) : (n <= 512);
    }
    if ((rc = ctx->time_begin())) return rc;
    PRB_HIP(ra_launch(b, ctx->ra_const, band, vec, ctx->stream));
    if ((rc = ctx->time_end("raccess", 1))) return rc; // synchronises: host staging vectors may be reused
    if (b.helpers == 3) {
      int tripped = 0;
      PRB_HIP(ra_watchdog_tripped(&tripped, ctx->stream));
      if (tripped) {
        set_error("Raccess: a wavefront gave up waiting for its workgroup (PRB_RACCESS_HELPERS=2 avoids that form of the passes)");
        return PRB_ERR_STATE;
      }
    }
    pos = end;
  }
  return PRB_OK;
}

} // namespace prb

extern "C" int prb_accessibility(prb_ctx *ctx, int32_t nseq, const char *seqs, const int64_t *offsets,
                                 int32_t maximal_span, int32_t min_accessible_length, float *acc, float *cond) {
  if (!ctx || nseq < 0 || (nseq > 0 && (!seqs || !offsets)) || !acc || !cond) {
    set_error("prb_accessibility: bad argument");
    return PRB_ERR_ARG;
  }
  if (nseq == 0) return PRB_OK;
  const int64_t total = offsets[nseq] - offsets[0];
  const size_t bytes = (size_t)std::max<int64_t>(total, 1) * sizeof(float);
  int rc;
  PRB_HIP(hipSetDevice(ctx->device));
  if ((rc = ctx->ra_acc.ensure(bytes))) return rc;
  if ((rc = ctx->ra_cond.ensure(bytes))) return rc;
  PRB_HIP(hipMemsetAsync(ctx->ra_acc.p, 0, bytes, ctx->stream));
  PRB_HIP(hipMemsetAsync(ctx->ra_cond.p, 0, bytes, ctx->stream));
  std::vector<int32_t> lens(nseq);
  std::vector<int64_t> out_off(nseq);
  for (int32_t i = 0; i < nseq; i++) {
    if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > INT32_MAX - 4) {
      set_error("prb_accessibility: bad offsets");
      return PRB_ERR_ARG;
    }
    lens[i] = (int32_t)(offsets[i + 1] - offsets[i]);
    out_off[i] = offsets[i] - offsets[0];
  }
  if ((rc = run_accessibility(ctx, nseq, seqs, offsets, lens.data(), out_off.data(), maximal_span,
                              min_accessible_length, ctx->ra_acc.as<float>(), ctx->ra_cond.as<float>())))
    return rc;
  PRB_HIP(hipMemcpyAsync(acc, ctx->ra_acc.p, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PRB_HIP(hipMemcpyAsync(cond, ctx->ra_cond.p, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PRB_HIP(hipStreamSynchronize(ctx->stream));
  return PRB_OK;
}

extern "C" int prb_accessibility_tables(prb_ctx *ctx, const char *seq, int32_t len, int32_t maximal_span,
                                        int32_t min_accessible_length, float *acc, float *cond,
                                        double *alpha_outer, double *beta_outer, double *const tables[12]) {
  if (!ctx || !seq || len < 0 || !acc || !cond) return PRB_ERR_ARG;
  const int64_t offsets[2] = {0, len};
  int rc = prb_accessibility(ctx, 1, seq, offsets, maximal_span, min_accessible_length, acc, cond);
  if (rc) return rc;
  const int W = maximal_span, S = W + 2;
  const int64_t rows = (int64_t)len + 2;
  std::vector<double> band((size_t)kRaBands * rows * S), vec((size_t)kRaVecs * rows);
  PRB_HIP(hipMemcpy(band.data(), ctx->ra_band.p, band.size() * sizeof(double), hipMemcpyDeviceToHost));
  PRB_HIP(hipMemcpy(vec.data(), ctx->ra_vec.p, vec.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (alpha_outer) std::memcpy(alpha_outer, vec.data(), (size_t)(len + 1) * sizeof(double));
  if (beta_outer) std::memcpy(beta_outer, vec.data() + rows, (size_t)(len + 1) * sizeof(double));
  // band ids: 0..5 alpha (stem, stemend, multi, multibif, multi1, multi2), 6 = alpha_multi1 start-major copy,
  // 7..12 beta (stem, stemend, multi, multibif, multi1, multi2)
  static const int ids[12] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12};
  for (int t = 0; tables && t < 12; t++) {
    if (!tables[t]) continue;
    const double *src = band.data() + (size_t)ids[t] * rows * S;
    for (int64_t i = 0; i <= len; i++)
      for (int d = 0; d < S; d++) {
        const int64_t end = i + d;
        tables[t][i * S + d] = end <= len + 1 ? src[end * S + d] : kNegInf;
      }
  }
  return PRB_OK;
}
