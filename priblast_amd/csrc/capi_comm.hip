// C ABI, part 3: the final hit gather of a multi-GPU run - one process per GPU, RCCL over xGMI.
//
// Replaces the reference's output merge (MergeOutput, rna_interaction_search.cpp:426-487: every MPI
// rank appends its temporary file to the output in turn, a token passed around the ring).  Queries
// are independent end to end, so this is the only exchange of the whole `ris` step: every rank
// contributes the packed records of its batch as they lie in HBM after the search (k_pack_hits;
// nothing is uploaded again), rank `root` receives them over point-to-point sends, shifts query
// indices and pair offsets by what the lower ranks contributed, and copies the lot to pinned host
// memory once.  librccl is loaded at run time (dlopen), so single-GPU use does not depend on it.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/priblast_hip.h"
#include "context.hpp"
#include "hitset.hpp"

namespace prb {
namespace {

// the few RCCL entry points used, with the types of rccl.h (ncclComm_t is an opaque pointer,
// ncclUniqueId 128 bytes passed by value, ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4)
struct UniqueId {
  char internal[PRB_COMM_ID_BYTES];
};
enum : int { kNcclInt8 = 0, kNcclInt32 = 2, kNcclInt64 = 4 };
struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string error;
};

Rccl &rccl() {
  static Rccl r = [] {
    Rccl x;
    // a librccl that is already in the process (PyTorch ships its own) is reused: same soname
    const char *names[] = {std::getenv("PRB_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n || !*n) continue;
      x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (x.handle) break;
    }
    if (!x.handle) {
      x.error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
      return x;
    }
    auto sym = [&](const char *name) -> void * {
      void *p = dlsym(x.handle, name);
      if (!p && x.error.empty()) x.error = std::string("librccl lacks ") + name;
      return p;
    };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return r;
}

int rccl_fail(int code, const char *what) {
  Rccl &r = rccl();
  set_error(std::string("RCCL error '") + (r.GetErrorString ? r.GetErrorString(code) : "?") + "' in " + what);
  return PRB_ERR_HIP;
}
#define PRB_RCCL(call)                                   \
  do {                                                   \
    int e_ = (call);                                     \
    if (e_ != 0) return prb::rccl_fail(e_, #call);       \
  } while (0)

// query += qbase and bp_offset += bpbase for the n records at h (one launch per contributing rank)
__global__ __launch_bounds__(256) void k_rebase_hits(prb_hit *h, int64_t n, int32_t qbase, int64_t bpbase) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  h[i].query += qbase;
  h[i].bp_offset += bpbase;
}

} // namespace
} // namespace prb

using namespace prb;

// Pinned slots: a gathered hit set lives in one until it is freed (possibly by another thread, possibly after its
// communicator is gone: the pool belongs to the communicator AND to every hit set that borrows from it).
struct SlotPool {
  struct Slot {
    PinnedBuf hits, bp;
    bool busy = false;
  };
  std::vector<std::unique_ptr<Slot>> slots;
  std::mutex mu;
  int device = 0;
  int take() {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < slots.size(); i++)
      if (!slots[i]->busy) {
        slots[i]->busy = true;
        return (int)i;
      }
    slots.emplace_back(new Slot());
    slots.back()->busy = true;
    return (int)slots.size() - 1;
  }
  ~SlotPool() {
    (void)hipSetDevice(device);
    for (auto &sl : slots) {
      sl->hits.release();
      sl->bp.release();
    }
  }
  // prb_hitset::ext_release: `owner` is the hit set's own reference to the pool
  static void give_back(void *owner, int slot) {
    auto *ref = static_cast<std::shared_ptr<SlotPool> *>(owner);
    {
      std::lock_guard<std::mutex> lk((*ref)->mu);
      (*ref)->slots[(size_t)slot]->busy = false;
    }
    delete ref; // (the last reference frees the pinned buffers)
  }
};

struct prb_comm {
  prb_ctx *ctx = nullptr;
  void *comm = nullptr;
  int32_t nranks = 1, rank = 0;
  hipStream_t stream = nullptr; // the gather's own stream: it may run (from another host thread) beside the next search
  DevBuf meta, meta_all, qlen, rx_hits, rx_bp, rx_qlen;
  std::shared_ptr<SlotPool> pool = std::make_shared<SlotPool>();
};

extern "C" {

int prb_comm_unique_id(char id[PRB_COMM_ID_BYTES]) {
  if (!id) return PRB_ERR_ARG;
  Rccl &r = rccl();
  if (!r.error.empty()) {
    set_error(r.error);
    return PRB_ERR_STATE;
  }
  UniqueId u;
  PRB_RCCL(r.GetUniqueId(&u));
  std::memcpy(id, u.internal, PRB_COMM_ID_BYTES);
  return PRB_OK;
}

int prb_comm_create(prb_ctx *ctx, int32_t nranks, int32_t rank, const char id[PRB_COMM_ID_BYTES], prb_comm **out) {
  if (!ctx || !out || !id || nranks < 1 || rank < 0 || rank >= nranks) {
    set_error("prb_comm_create: bad argument");
    return PRB_ERR_ARG;
  }
  *out = nullptr;
  Rccl &r = rccl();
  if (!r.error.empty()) {
    set_error(r.error);
    return PRB_ERR_STATE;
  }
  PRB_HIP(hipSetDevice(ctx->device));
  auto *c = new prb_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->pool->device = ctx->device;
  UniqueId u;
  std::memcpy(u.internal, id, PRB_COMM_ID_BYTES);
  if (hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipStreamCreateWithFlags");
  }
  if (int e = r.CommInitRank(&c->comm, nranks, u, rank)) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return rccl_fail(e, "ncclCommInitRank");
  }
  ctx->keep_device_records = true; // final hit sets of this context keep their packed records in HBM from now on
  *out = c;
  return PRB_OK;
}

void prb_comm_destroy(prb_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  for (DevBuf *b : {&c->meta, &c->meta_all, &c->qlen, &c->rx_hits, &c->rx_bp, &c->rx_qlen}) b->release();
  delete c; // (the pinned slots go with the last gathered hit set that still borrows one)
}

int prb_gather_plan(int32_t nranks, const int64_t *counts, int64_t *bases) {
  if (nranks < 1 || !counts || !bases) {
    set_error("prb_gather_plan: bad argument");
    return PRB_ERR_ARG;
  }
  for (int j = 0; j < 3; j++) bases[j] = 0;
  for (int k = 0; k < nranks; k++)
    for (int j = 0; j < 3; j++) {
      if (counts[3 * k + j] < 0 || (j == 1 && (counts[3 * k + j] & 1))) {
        set_error("prb_gather_plan: negative count, or an odd number of pair ints");
        return PRB_ERR_ARG;
      }
      bases[3 * (k + 1) + j] = bases[3 * k + j] + counts[3 * k + j];
    }
  if (bases[3 * nranks + 2] > INT32_MAX) {
    set_error("prb_gather_hits: more than 2^31 queries in one gather");
    return PRB_ERR_ARG;
  }
  return PRB_OK;
}

static int gather_hits_impl(prb_comm *c, const prb_hitset *mine, int32_t nq, const int32_t *qlen_unmasked, int32_t root,
                            prb_hitset **out) {
  if (!c || !out || nq < 0 || (nq && !qlen_unmasked) || root < 0 || root >= c->nranks) {
    set_error("prb_gather_hits: bad argument");
    return PRB_ERR_ARG;
  }
  *out = nullptr;
  if (mine && !mine->on_device) {
    set_error("prb_gather_hits: the hit set has no device copy (it must come from prb_search_page, last_stage 3, "
              "called after prb_comm_create on the same context)");
    return PRB_ERR_STATE;
  }
  Rccl &r = rccl();
  prb_ctx *ctx = c->ctx;
  hipStream_t s = c->stream; // (the records of `mine` are complete: prb_search_page returns after its last copy)
  PRB_HIP(hipSetDevice(ctx->device));
  const int n = c->nranks;
  int rc;
  // 1. what every rank brings: hits, pair-array ints, queries
  int64_t my[3] = {mine ? (int64_t)(mine->d_hits.used / sizeof(prb_hit)) : 0, mine ? (int64_t)(mine->d_bp.used / 4) : 0, nq};
  std::vector<int64_t> all((size_t)3 * n);
  if ((rc = c->meta.ensure(sizeof my)) || (rc = c->meta_all.ensure(sizeof(int64_t) * 3 * n))) return rc;
  PRB_HIP(hipMemcpyAsync(c->meta.p, my, sizeof my, hipMemcpyHostToDevice, s));
  PRB_RCCL(r.AllGather(c->meta.p, c->meta_all.p, 3, kNcclInt64, c->comm, s));
  PRB_HIP(hipMemcpyAsync(all.data(), c->meta_all.p, sizeof(int64_t) * 3 * n, hipMemcpyDeviceToHost, s));
  PRB_HIP(hipStreamSynchronize(s));
  std::vector<int64_t> bases((size_t)3 * (n + 1));
  if ((rc = prb_gather_plan(n, all.data(), bases.data()))) return rc;
  std::vector<int64_t> hbase((size_t)n + 1), bbase((size_t)n + 1), qbase((size_t)n + 1);
  for (int k = 0; k <= n; k++) {
    hbase[k] = bases[3 * k];
    bbase[k] = bases[3 * k + 1];
    qbase[k] = bases[3 * k + 2];
  }
  // 2. point-to-point: every rank sends what it has to the root, which receives at the final offsets
  if ((rc = c->qlen.ensure(std::max<size_t>((size_t)nq * 4, 16)))) return rc;
  if (nq) PRB_HIP(hipMemcpyAsync(c->qlen.p, qlen_unmasked, (size_t)nq * 4, hipMemcpyHostToDevice, s));
  const bool is_root = c->rank == root;
  if (is_root) {
    if ((rc = c->rx_hits.ensure(std::max<size_t>((size_t)hbase[n] * sizeof(prb_hit), 16))) ||
        (rc = c->rx_bp.ensure(std::max<size_t>((size_t)bbase[n] * 4, 16))) || (rc = c->rx_qlen.ensure(std::max<size_t>((size_t)qbase[n] * 4, 16))))
      return rc;
  }
  PRB_RCCL(r.GroupStart());
  // (an error inside the group must not leave it open: the sends / receives are issued by a function of their own,
  // the group is closed whatever it returns, and its error is the one reported)
  auto issue = [&]() -> int {
    if (!is_root) {
      if (my[0]) PRB_RCCL(r.Send(mine->d_hits.b.p, (size_t)my[0] * sizeof(prb_hit), kNcclInt8, root, c->comm, s));
      if (my[1]) PRB_RCCL(r.Send(mine->d_bp.b.p, (size_t)my[1], kNcclInt32, root, c->comm, s));
      if (my[2]) PRB_RCCL(r.Send(c->qlen.p, (size_t)my[2], kNcclInt32, root, c->comm, s));
      return PRB_OK;
    }
    for (int k = 0; k < n; k++) {
      if (k == root) continue;
      if (all[3 * k])
        PRB_RCCL(r.Recv(c->rx_hits.as<prb_hit>() + hbase[k], (size_t)all[3 * k] * sizeof(prb_hit), kNcclInt8, k, c->comm, s));
      if (all[3 * k + 1]) PRB_RCCL(r.Recv(c->rx_bp.as<int32_t>() + bbase[k], (size_t)all[3 * k + 1], kNcclInt32, k, c->comm, s));
      if (all[3 * k + 2]) PRB_RCCL(r.Recv(c->rx_qlen.as<int32_t>() + qbase[k], (size_t)all[3 * k + 2], kNcclInt32, k, c->comm, s));
    }
    return PRB_OK;
  };
  rc = issue();
  const int ge = r.GroupEnd();
  if (rc) return rc;
  if (ge) return rccl_fail(ge, "ncclGroupEnd");
  if (!is_root) {
    PRB_HIP(hipStreamSynchronize(s)); // `mine` may be freed by the caller right away
    return PRB_OK;
  }
  if (my[0])
    PRB_HIP(hipMemcpyAsync(c->rx_hits.as<prb_hit>() + hbase[root], mine->d_hits.b.p, (size_t)my[0] * sizeof(prb_hit), hipMemcpyDeviceToDevice, s));
  if (my[1]) PRB_HIP(hipMemcpyAsync(c->rx_bp.as<int32_t>() + bbase[root], mine->d_bp.b.p, (size_t)my[1] * 4, hipMemcpyDeviceToDevice, s));
  if (my[2]) PRB_HIP(hipMemcpyAsync(c->rx_qlen.as<int32_t>() + qbase[root], c->qlen.p, (size_t)my[2] * 4, hipMemcpyDeviceToDevice, s));
  // 3. one hit set: query indices and pair offsets continue across the ranks
  for (int k = 0; k < n; k++) {
    const int64_t m = all[3 * k];
    if (m == 0 || (qbase[k] == 0 && bbase[k] == 0)) continue;
    hipLaunchKernelGGL(k_rebase_hits, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, c->rx_hits.as<prb_hit>() + hbase[k], m,
                       (int32_t)qbase[k], bbase[k] / 2);
    PRB_HIP(hipGetLastError());
  }
  // 4. to pinned host memory, once
  const int slot = c->pool->take();
  SlotPool::Slot &sl = *c->pool->slots[(size_t)slot];
  auto *hs = new prb_hitset(); // (owns the slot from here on: freed on every error path below)
  hs->device = ctx->device;
  hs->ext_release = &SlotPool::give_back;
  hs->ext_owner = new std::shared_ptr<SlotPool>(c->pool);
  hs->ext_slot = slot;
  std::unique_ptr<prb_hitset> guard(hs);
  if ((rc = sl.hits.ensure(std::max<size_t>((size_t)hbase[n] * sizeof(prb_hit), 16))) ||
      (rc = sl.bp.ensure(std::max<size_t>((size_t)bbase[n] * 4, 16))))
    return rc;
  if (hbase[n]) PRB_HIP(hipMemcpyAsync(sl.hits.p, c->rx_hits.p, (size_t)hbase[n] * sizeof(prb_hit), hipMemcpyDeviceToHost, s));
  if (bbase[n]) PRB_HIP(hipMemcpyAsync(sl.bp.p, c->rx_bp.p, (size_t)bbase[n] * 4, hipMemcpyDeviceToHost, s));
  hs->g_qlen.resize((size_t)qbase[n]);
  if (qbase[n]) PRB_HIP(hipMemcpyAsync(hs->g_qlen.data(), c->rx_qlen.p, (size_t)qbase[n] * 4, hipMemcpyDeviceToHost, s));
  PRB_HIP(hipStreamSynchronize(s));
  hs->ext_hits = static_cast<const prb_hit *>(sl.hits.p);
  hs->ext_nhits = hbase[n];
  hs->ext_bp = static_cast<const int32_t *>(sl.bp.p);
  hs->ext_bp_ints = bbase[n];
  if (mine)
    for (int i = 0; i < 3; i++) hs->counts[i] = mine->counts[i]; // (the root's own; the stage counts of the others do not travel)
  hs->counts[2] = hbase[n];
  for (int k = 0; k < n; k++) hs->g_nq_of_rank.push_back((int32_t)all[3 * k + 2]);
  *out = guard.release();
  return PRB_OK;
}

int prb_gather_hits(prb_comm *c, const prb_hitset *mine, int32_t nq, const int32_t *qlen_unmasked, int32_t root,
                    prb_hitset **out) {
  try {
    return gather_hits_impl(c, mine, nq, qlen_unmasked, root, out);
  } catch (const std::exception &e) { // (no exception leaves the C ABI)
    if (out) *out = nullptr;
    set_error(std::string("prb_gather_hits: ") + e.what());
    return PRB_ERR_NOMEM;
  }
}

int prb_hitset_gathered_queries(const prb_hitset *hs, int32_t *nranks, const int32_t **nq_of_rank, const int32_t **qlen_unmasked) {
  if (!hs || hs->g_nq_of_rank.empty()) {
    set_error("prb_hitset_gathered_queries: not a gathered hit set");
    return PRB_ERR_ARG;
  }
  if (nranks) *nranks = (int32_t)hs->g_nq_of_rank.size();
  if (nq_of_rank) *nq_of_rank = hs->g_nq_of_rank.data();
  if (qlen_unmasked) *qlen_unmasked = hs->g_qlen.data();
  return PRB_OK;
}

void prb_ctx_keep_device_records(prb_ctx *ctx, int32_t on) {
  if (ctx) ctx->keep_device_records = on != 0;
}

} // extern "C"
