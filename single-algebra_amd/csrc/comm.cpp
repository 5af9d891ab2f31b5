#include "comm.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace sapca {

namespace {

// Minimal slice of the RCCL (NCCL-compatible) C API, bound by name at run time.
struct RcclId { char internal[128]; };
using ncclComm_t = void*;
struct RcclApi {
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;
  int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;
  int (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* so = nullptr;
};
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0;

RcclApi& api() {
  static RcclApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    // (-DSAPCA_DEBUG_SWITCHES builds only: the tests' stand-in for librccl, tests/fake_rccl -- RCCL mode with several ranks on
    // one GPU.  Opened by its full path and without RTLD_GLOBAL: the process may carry the real RCCL for someone else.)
    if (const char* fake = dbg_env("SAPCA_RCCL_LIBRARY")) a.so = dlopen(fake, RTLD_NOW | RTLD_LOCAL);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (a.so) break;
      a.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!a.so) return;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.so, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.so, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.so, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.so, "ncclCommDestroy"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.so, "ncclCommAbort"));
    a.CommGetAsyncError = reinterpret_cast<decltype(a.CommGetAsyncError)>(dlsym(a.so, "ncclCommGetAsyncError"));
    a.CommSplit = reinterpret_cast<decltype(a.CommSplit)>(dlsym(a.so, "ncclCommSplit"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.so, "ncclGetErrorString"));
  });
  if (!a.so || !a.GetUniqueId || !a.CommInitRank || !a.AllReduce)
    throw Error(SAPCA_ERR_COMM, "RCCL (librccl.so.1) could not be loaded; use sapca_comm_set_callback instead");
  return a;
}

void check(int rc, const char* what) {
  if (rc != 0) {
    const char* msg = api().GetErrorString ? api().GetErrorString(rc) : "?";
    throw Error(SAPCA_ERR_COMM, std::string(what) + ": " + msg);
  }
}

}  // namespace

bool Comm::rccl_available() {
  try {
    (void)api();
    return true;
  } catch (const Error&) {
    return false;
  }
}

void Comm::unique_id(uint8_t id[128]) {
  RcclId u;
  check(api().GetUniqueId(&u), "ncclGetUniqueId");
  std::memcpy(id, u.internal, 128);
}

void Comm::init_rccl(uint32_t nranks_, uint32_t rank_, const uint8_t id[128]) {
  destroy();
  SAPCA_CHECK(nranks_ >= 1 && rank_ < nranks_, SAPCA_ERR_ARG, "comm: rank out of range");
  nranks = nranks_;
  rank = rank_;
  // a one-rank job needs no collective; SAPCA_COMM_FORCE_RCCL=1 still builds the communicator and
  // routes every all-reduce site through RCCL (a copy), which is how a one-GPU box tests the binding
  if (nranks == 1 && !dbg_env("SAPCA_COMM_FORCE_RCCL")) { mode = NONE; return; }
  RcclId u;
  std::memcpy(u.internal, id, 128);
  ncclComm_t c = nullptr;
  check(api().CommInitRank(&c, (int)nranks, u, (int)rank), "ncclCommInitRank");
  rccl_comm.store(c);
  mode = RCCL;
  aborted.store(false);
  // The side stream's communicator (a collective split with one colour) is only made where the two-piece A^T sweep is asked
  // for (SAPCA_AT_OVERLAP=1: opt-in under RCCL, engine.cpp) -- and only if EVERY rank asks: ranks count the ones that do not
  // with one all-reduce on the main communicator (a rank entering ncclCommSplit alone would wait for ever), split, and count
  // the ranks whose split failed the same way; any of them drops the duplicate everywhere.
  rccl_comm2.store(nullptr);
  float* d_flag = nullptr;
  SAPCA_HIP(hipMalloc(&d_flag, sizeof(float)));
  int rc = 0;
  hipError_t he = hipSuccess;
  const auto missing_anywhere = [&](bool mine_missing) {   // true when any rank says "missing" (or the exchange itself failed)
    const float missing = mine_missing ? 1.f : 0.f;
    float total = 1.f;
    he = hipMemcpy(d_flag, &missing, sizeof(float), hipMemcpyHostToDevice);
    if (he == hipSuccess) rc = api().AllReduce(d_flag, d_flag, 1, kNcclFloat32, kNcclSum, c, nullptr);
    if (he == hipSuccess && rc == 0) he = hipStreamSynchronize(nullptr);
    if (he == hipSuccess && rc == 0) he = hipMemcpy(&total, d_flag, sizeof(float), hipMemcpyDeviceToHost);
    return he != hipSuccess || rc != 0 || total != 0.f;
  };
  const char* ov = getenv("SAPCA_AT_OVERLAP");
  const bool want = ov != nullptr && atoi(ov) != 0 && api().CommSplit && dbg_env("SAPCA_COMM_NO_SPLIT") == nullptr;
  ncclComm_t c2 = nullptr;
  if (!missing_anywhere(!want)) {
    if (api().CommSplit(c, 0, (int)rank, &c2, nullptr) != 0) c2 = nullptr;
    if (missing_anywhere(c2 == nullptr)) {
      if (c2 && api().CommDestroy) (void)api().CommDestroy(c2);
      c2 = nullptr;
    }
  }
  (void)hipFree(d_flag);
  rccl_comm2.store(c2);
  check(rc, "ncclAllReduce (side-lane agreement)");
  SAPCA_HIP(he);
}

void Comm::set_callback(uint32_t nranks_, uint32_t rank_, sapca_allreduce_fn f, void* c) {
  destroy();
  SAPCA_CHECK(nranks_ >= 1 && rank_ < nranks_, SAPCA_ERR_ARG, "comm: rank out of range");
  SAPCA_CHECK(f != nullptr || nranks_ == 1, SAPCA_ERR_ARG, "comm: null all-reduce callback");
  nranks = nranks_;
  rank = rank_;
  fn = f;
  ctx = c;
  mode = nranks > 1 ? CALLBACK : NONE;
}

void Comm::allreduce(void* buf, uint64_t count, int dtype, hipStream_t s, int lane) {
  if (!active() || count == 0) return;
  if (aborted.load()) throw Error(SAPCA_ERR_COMM, "communicator aborted: a peer of this fit failed");
  auto t0 = std::chrono::steady_clock::now();
  if (mode == RCCL) {
    std::lock_guard<std::timed_mutex> issue(issue_mu);   // abort() swaps the communicators out only between two enqueues
    void* const c = (lane == 0 ? rccl_comm : rccl_comm2).load();
    if (aborted.load() || (lane == 0 && c == nullptr)) throw Error(SAPCA_ERR_COMM, "communicator aborted: a peer of this fit failed");
    SAPCA_CHECK(c != nullptr, SAPCA_ERR_COMM, "internal: no communicator for the side stream");
    check(api().AllReduce(buf, buf, (size_t)count, dtype == 1 ? kNcclFloat64 : kNcclFloat32, kNcclSum, c, s), "ncclAllReduce");
  } else if (mode == CALLBACK) {
    if (fn(ctx, buf, count, dtype, (void*)s) != 0) throw Error(SAPCA_ERR_COMM, "all-reduce callback reported failure");
  } else {
    throw Error(SAPCA_ERR_COMM, "multi-rank handle without a collective");
  }
  host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void Comm::abort() {
  if (aborted.exchange(true)) return;
  if (mode != RCCL || !api().CommAbort) return;
  // (ncclCommAbort frees the communicator: nobody may touch it again -- the pointers are taken out under the enqueue lock,
  // or without it when the owning thread has been inside an enqueue for 250 ms: that is the hang this call ends)
  std::lock_guard<std::mutex> state(state_mu);
  std::unique_lock<std::timed_mutex> issue(issue_mu, std::defer_lock);
  (void)issue.try_lock_for(std::chrono::milliseconds(250));
  void* c2 = rccl_comm2.exchange(nullptr);
  void* c1 = rccl_comm.exchange(nullptr);
  if (issue.owns_lock()) issue.unlock();
  if (c2) (void)api().CommAbort(c2);
  if (c1) (void)api().CommAbort(c1);
}

int Comm::async_error() {
  if (aborted.load()) return -1;
  if (mode != RCCL || !api().CommGetAsyncError) return 0;
  int e = 0;
  std::lock_guard<std::mutex> state(state_mu);   // (abort() frees the communicators: not while they are being asked)
  void* const c1 = rccl_comm.load();
  void* const c2 = rccl_comm2.load();
  if (aborted.load()) return -1;
  if (c1 && api().CommGetAsyncError(c1, &e) == 0 && e != 0) return e;
  if (c2 && api().CommGetAsyncError(c2, &e) == 0 && e != 0) return e;
  return 0;
}

void Comm::destroy() {
  void* c2;
  void* c1;
  {
    std::lock_guard<std::mutex> state(state_mu);
    std::lock_guard<std::timed_mutex> issue(issue_mu);
    c2 = rccl_comm2.exchange(nullptr);
    c1 = rccl_comm.exchange(nullptr);
  }
  if (mode == RCCL && c2 && api().CommDestroy) (void)api().CommDestroy(c2);
  if (mode == RCCL && c1 && api().CommDestroy) (void)api().CommDestroy(c1);
  aborted.store(false);
  fn = nullptr;
  ctx = nullptr;
  mode = NONE;
  nranks = 1;
  rank = 0;
}

}  // namespace sapca
