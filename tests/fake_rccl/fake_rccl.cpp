// TEST INFRASTRUCTURE ONLY - never part of the product, never loaded unless a test sets PRB_RCCL_LIB.
//
// A stand-in for the nine librccl entry points priblast_amd/csrc/capi_comm.hip binds with dlsym, so that the
// multi-rank half of prb_gather_hits (the grouped ncclSend / ncclRecv at their final offsets, k_rebase_hits with
// non-zero bases, a root that is not rank 0) can be EXECUTED by two or more processes that share the one GPU of a
// test box: real RCCL refuses two ranks on one device.  Same signatures and call semantics as rccl.h (ncclComm_t an
// opaque pointer, ncclUniqueId 128 bytes by value, sends and receives between the same pair of ranks match in the
// order they were issued, a group is carried out at ncclGroupEnd); the transport is files in a directory every rank
// sees (PRB_FAKE_RCCL_DIR, default /dev/shm), staged through host memory.  Nothing here is fast, nor meant to be.
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

struct UniqueId {
  char internal[128];
};

struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

struct Comm {
  int nranks = 0, rank = 0;
  std::string base; // directory + token
  uint64_t ag_seq = 0;
  std::map<std::pair<int, int>, uint64_t> seq; // (src, dst) -> messages so far
};

thread_local int g_group_depth = 0;
thread_local std::vector<std::pair<Comm *, Op>> g_ops;

enum { kOk = 0, kHipError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kRemoteError = 6 };

size_t type_size(int t) {
  switch (t) {
  case 0: case 1: return 1;                 // int8 / uint8
  case 2: case 3: case 7: return 4;         // int32 / uint32 / float32
  case 4: case 5: case 8: return 8;         // int64 / uint64 / float64
  case 6: case 9: return 2;                 // float16 / bfloat16
  default: return 0;
  }
}

std::string dir() {
  const char *d = std::getenv("PRB_FAKE_RCCL_DIR");
  return d && *d ? d : "/dev/shm";
}

int timeout_s() {
  const char *t = std::getenv("PRB_FAKE_RCCL_TIMEOUT");
  return t ? std::atoi(t) : 120;
}

bool write_file(const std::string &path, const void *p, size_t n) {
  const std::string tmp = path + ".tmp";
  std::FILE *f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = (n == 0 || std::fwrite(p, 1, n, f) == n);
  return (std::fclose(f) == 0) && ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}

// waits for `path` (written atomically by a peer) and reads exactly n bytes of it
int read_file(const std::string &path, void *p, size_t n, bool remove_after) {
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s())) {
      std::fprintf(stderr, "fake_rccl: timed out waiting for %s\n", path.c_str());
      return kRemoteError;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  if ((size_t)st.st_size != n) {
    std::fprintf(stderr, "fake_rccl: %s holds %zu bytes, the receive asks for %zu\n", path.c_str(), (size_t)st.st_size, n);
    return kInvalidArgument;
  }
  std::FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return kSystemError;
  const bool ok = n == 0 || std::fread(p, 1, n, f) == n;
  std::fclose(f);
  if (remove_after) std::remove(path.c_str());
  return ok ? kOk : kSystemError;
}

int run(Comm *c, const Op &op) {
  if (op.peer < 0 || op.peer >= c->nranks || op.peer == c->rank) return kInvalidArgument;
  std::vector<char> host(op.bytes);
  if (op.send) {
    if (hipStreamSynchronize(op.stream) != hipSuccess) return kHipError;
    if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kHipError;
    const uint64_t k = c->seq[{c->rank, op.peer}]++;
    const std::string path = c->base + "_msg_" + std::to_string(c->rank) + "_" + std::to_string(op.peer) + "_" + std::to_string(k);
    return write_file(path, host.data(), op.bytes) ? kOk : kSystemError;
  }
  const uint64_t k = c->seq[{op.peer, c->rank}]++;
  const std::string path = c->base + "_msg_" + std::to_string(op.peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(k);
  if (int e = read_file(path, host.data(), op.bytes, true)) return e;
  if (hipStreamSynchronize(op.stream) != hipSuccess) return kHipError;
  if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return kHipError;
  return kOk;
}

int submit(Comm *c, const Op &op) {
  if (g_group_depth > 0) {
    g_ops.emplace_back(c, op);
    return kOk;
  }
  return run(c, op);
}

} // namespace

extern "C" {

int ncclGetUniqueId(UniqueId *id) {
  if (!id) return kInvalidArgument;
  std::memset(id->internal, 0, sizeof id->internal);
  unsigned long long r = 0;
  if (std::FILE *f = std::fopen("/dev/urandom", "rb")) {
    if (std::fread(&r, sizeof r, 1, f) != 1) r = 0;
    std::fclose(f);
  }
  std::snprintf(id->internal, sizeof id->internal, "prbfake_%d_%016llx", (int)getpid(), r);
  return kOk;
}

int ncclCommInitRank(void **comm, int nranks, UniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[sizeof id.internal - 1] != 0) return kInvalidArgument;
  auto *c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  c->base = dir() + "/" + id.internal;
  // every rank announces itself and waits for the others (the real call is a collective as well)
  char one = 1;
  if (!write_file(c->base + "_join_" + std::to_string(rank), &one, 1)) {
    delete c;
    return kSystemError;
  }
  for (int k = 0; k < nranks; k++)
    if (int e = read_file(c->base + "_join_" + std::to_string(k), &one, 1, false)) {
      delete c;
      return e;
    }
  *comm = c;
  return kOk;
}

int ncclCommDestroy(void *comm) {
  delete static_cast<Comm *>(comm);
  return kOk;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
  auto *c = static_cast<Comm *>(comm);
  const size_t bytes = count * type_size(dtype);
  if (!c || !type_size(dtype) || g_group_depth > 0) return kInvalidArgument;
  std::vector<char> host(bytes * (size_t)c->nranks);
  if (hipStreamSynchronize(stream) != hipSuccess) return kHipError;
  if (bytes && hipMemcpy(host.data() + bytes * (size_t)c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kHipError;
  const uint64_t k = c->ag_seq++;
  const std::string stem = c->base + "_ag_" + std::to_string(k) + "_";
  if (!write_file(stem + std::to_string(c->rank), host.data() + bytes * (size_t)c->rank, bytes)) return kSystemError;
  for (int r = 0; r < c->nranks; r++)
    if (int e = read_file(stem + std::to_string(r), host.data() + bytes * (size_t)r, bytes, false)) return e;
  if (bytes && hipMemcpy(recv, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return kHipError;
  return kOk;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  if (!comm || !type_size(dtype)) return kInvalidArgument;
  return submit(static_cast<Comm *>(comm), Op{true, const_cast<void *>(buf), count * type_size(dtype), peer, stream});
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  if (!comm || !type_size(dtype)) return kInvalidArgument;
  return submit(static_cast<Comm *>(comm), Op{false, buf, count * type_size(dtype), peer, stream});
}

int ncclGroupStart() {
  g_group_depth++;
  return kOk;
}

int ncclGroupEnd() {
  if (g_group_depth <= 0) return kInvalidArgument;
  if (--g_group_depth > 0) return kOk;
  // the sends first: they only write files, so no order of the ranks' groups can deadlock
  int rc = kOk;
  for (int pass = 0; pass < 2 && rc == kOk; pass++)
    for (auto &co : g_ops)
      if (co.second.send == (pass == 0) && rc == kOk) rc = run(co.first, co.second);
  g_ops.clear();
  return rc;
}

const char *ncclGetErrorString(int e) {
  switch (e) {
  case kOk: return "no error";
  case kHipError: return "unhandled hip error (fake_rccl)";
  case kSystemError: return "unhandled system error (fake_rccl)";
  case kInvalidArgument: return "invalid argument (fake_rccl)";
  case kRemoteError: return "remote process timed out (fake_rccl)";
  default: return "internal error (fake_rccl)";
  }
}

} // extern "C"
