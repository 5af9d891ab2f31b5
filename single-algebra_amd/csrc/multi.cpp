// One handle, several GPUs, one calling thread: the form in which a single-process caller (the reference's
// `SparsePCABuilder...build().fit_transform(&CsrMatrix)`, /root/reference/src/dimred/pca/sparse/mod.rs:355-358, one call
// from one thread) reaches the row-sharded path of SURVEY.md §8e without becoming an SPMD program.
//
// A sapca_multi owns one ordinary handle per device.  A call partitions the HOST CsrMatrix into nnz-balanced contiguous
// row ranges (sapca_partition_rows), starts one host thread per device, and each thread runs the ordinary host entry
// point on its shard; the handles are the ranks of one communicator, so the three all-reduce sites inside a fit (column
// statistics, the l x l Gram of a row-sharded panel, the n x l panel of an A^T sweep) connect them exactly as they
// connect the processes of the one-process-per-GPU deployment.  Omega, the n-side panels, the small SVD and every fitted
// quantity are replicated and bitwise identical on all devices: the getters of any member handle answer for the model.
//
// Transport: RCCL (one ncclCommInitRank per device thread over an id made in this process) when the devices are
// distinct and librccl resolves; otherwise an in-process all-reduce through page-locked host memory (devices listed
// twice -- how a one-GPU box tests this path -- or no RCCL): every rank copies its buffer out, the ranks sum disjoint
// slices of all copies in rank order (the same result on every device), every rank copies the sum back.
//
// Everything here sits on the public C ABI (include/sapca.h): no engine internals.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sapca.h"

namespace {

// Barrier of the device threads of one call that can be abandoned: a thread whose fit failed outside a collective
// releases the others (they report SAPCA_ERR_COMM) instead of leaving them blocked.
struct Rendezvous {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, arrived = 0;
  uint64_t generation = 0;
  bool failed = false;
  bool wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const uint64_t g = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return generation != g || failed; });
    return !failed;
  }
  void abandon() {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
  }
  void reset(int ranks) {
    std::lock_guard<std::mutex> lk(mu);
    n = ranks;
    arrived = 0;
    failed = false;
  }
};

struct Pinned {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return false;
    cap = bytes;
    return true;
  }
  ~Pinned() { if (p) (void)hipHostFree(p); }
};

struct InProcess {
  Rendezvous rv;
  std::vector<Pinned> stage;   // one copy-out area per rank
  Pinned result;
  bool result_ok = true;
};

struct RankCtx {
  InProcess* shared = nullptr;
  int rank = 0, nranks = 1;
};

template <typename T>
void sum_slice(const std::vector<Pinned>& stage, void* result, int nranks, uint64_t lo, uint64_t hi) {
  T* out = static_cast<T*>(result);
  for (uint64_t i = lo; i < hi; ++i) {
    T a = static_cast<const T*>(stage[0].p)[i];
    for (int r = 1; r < nranks; ++r) a += static_cast<const T*>(stage[(size_t)r].p)[i];   // rank order: the same sum everywhere
    out[i] = a;
  }
}

// sapca_allreduce_fn of the in-process transport
int inprocess_allreduce(void* ctx, void* buf, uint64_t count, int32_t dtype, void* stream) {
  RankCtx* c = static_cast<RankCtx*>(ctx);
  InProcess& sh = *c->shared;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = (size_t)count * (dtype == 1 ? 8 : 4);
  bool ok = sh.stage[(size_t)c->rank].ensure(bytes);
  if (c->rank == 0) sh.result_ok = sh.result.ensure(bytes);
  ok = ok && hipMemcpyAsync(sh.stage[(size_t)c->rank].p, buf, bytes, hipMemcpyDeviceToHost, s) == hipSuccess &&
       hipStreamSynchronize(s) == hipSuccess;
  if (!ok) { sh.rv.abandon(); return 1; }
  if (!sh.rv.wait()) return 1;
  if (!sh.result_ok) { sh.rv.abandon(); return 1; }
  const uint64_t per = (count + (uint64_t)c->nranks - 1) / (uint64_t)c->nranks;
  const uint64_t lo = std::min<uint64_t>(count, per * (uint64_t)c->rank), hi = std::min<uint64_t>(count, lo + per);
  if (dtype == 1) sum_slice<double>(sh.stage, sh.result.p, c->nranks, lo, hi);
  else sum_slice<float>(sh.stage, sh.result.p, c->nranks, lo, hi);
  if (!sh.rv.wait()) return 1;
  ok = hipMemcpyAsync(buf, sh.result.p, bytes, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  if (!ok) { sh.rv.abandon(); return 1; }
  return sh.rv.wait() ? 0 : 1;   // (the copies are about to be overwritten by the next collective)
}

}  // namespace

// a member's resident shard (sapca_multi_upload_csr_*): arrays owned by the member handle
struct Shard {
  uint64_t row0 = 0, rows = 0, nnz = 0;
  const int64_t* ptr = nullptr;
  const int32_t* idx = nullptr;
  void* val = nullptr;
};

struct sapca_multi_s {
  std::vector<sapca_handle> h;
  std::vector<int32_t> devices;
  std::vector<RankCtx> ctx;
  InProcess inproc;
  bool rccl = false;
  bool comm_broken = false;   // a member failed under RCCL: every communicator was aborted, the next call builds new ones
  bool connecting = false;    // inside the collective ncclCommInitRank: nothing to abort yet
  uint64_t k = 0;
  std::string err;
  // resident shards
  std::vector<Shard> shard;
  uint64_t res_m = 0, res_n = 0;
  int res_dtype = -1;   // 0 f32, 1 f64, -1 nothing uploaded
};

namespace {

thread_local std::string g_multi_create_error;

// A member that fails outside a collective never joins the ones its peers wait in.  In-process transport: the rendezvous
// is abandoned.  RCCL: every member's communicators are aborted (ncclCommAbort, callable from this thread while the
// peers' threads sit behind their streams) -- their collectives return, their next one fails at once with
// SAPCA_ERR_COMM -- and the sapca_multi builds new communicators at its next call.
void release_peers(sapca_multi_s* mh) {
  mh->inproc.rv.abandon();
  if (!mh->rccl) return;
  mh->comm_broken = true;
  if (mh->connecting) return;
  for (sapca_handle h : mh->h) (void)sapca_comm_abort(h);
}

// one host thread per device; returns the first failing status (and that device's message)
template <typename F>
sapca_status on_every_device(sapca_multi_s* mh, F&& f) {
  const int nd = (int)mh->h.size();
  std::vector<sapca_status> st((size_t)nd, SAPCA_OK);
  mh->inproc.rv.reset(nd);
  std::mutex release_mu;
  bool released = false;
  std::vector<std::thread> th;
  for (int i = 0; i < nd; ++i)
    th.emplace_back([&, i] {
      st[(size_t)i] = f(i);
      if (st[(size_t)i] != SAPCA_OK && nd > 1) {   // (nobody waits for a rank that has already left)
        std::lock_guard<std::mutex> lk(release_mu);
        if (!released) release_peers(mh);
        released = true;
      }
    });
  for (auto& t : th) t.join();
  // the device that failed first in its own right (not because a peer abandoned the collective) explains the failure best
  int bad = -1;
  for (int i = 0; i < nd; ++i)
    if (st[(size_t)i] != SAPCA_OK && (bad < 0 || (st[(size_t)bad] == SAPCA_ERR_COMM && st[(size_t)i] != SAPCA_ERR_COMM))) bad = i;
  if (bad < 0) { mh->err.clear(); return SAPCA_OK; }
  const char* msg = sapca_last_error(mh->h[(size_t)bad]);
  mh->err = std::string("device ") + std::to_string(mh->devices[(size_t)bad]) + " (shard " + std::to_string(bad) + "): " + (msg ? msg : "");
  return st[(size_t)bad];
}

void use_inprocess(sapca_multi_s* mh) {
  const uint32_t nd = (uint32_t)mh->h.size();
  mh->rccl = false;
  mh->inproc.stage.resize(nd);
  mh->ctx.resize(nd);
  for (uint32_t i = 0; i < nd; ++i) {
    mh->ctx[i].shared = &mh->inproc;
    mh->ctx[i].rank = (int)i;
    mh->ctx[i].nranks = (int)nd;
    // (set_callback destroys whatever communicator the member still holds)
    (void)sapca_comm_set_callback(mh->h[i], nd, i, &inprocess_allreduce, &mh->ctx[i]);
  }
}

// RCCL between the members: ncclCommInitRank is collective, one thread per device enters it.  All members get their
// communicators or none keeps one (the in-process transport takes over).
bool connect_rccl(sapca_multi_s* mh) {
  const uint32_t nd = (uint32_t)mh->h.size();
  uint8_t id[128];
  if (sapca_comm_unique_id(id) != SAPCA_OK) return false;
  mh->rccl = true;   // (a member whose init fails aborts the others' communicators: release_peers)
  mh->comm_broken = false;
  mh->connecting = true;
  const sapca_status st = on_every_device(mh, [&](int i) { return sapca_comm_init_rank(mh->h[(size_t)i], nd, (uint32_t)i, id); });
  mh->connecting = false;
  if (st == SAPCA_OK && !mh->comm_broken) return true;
  mh->comm_broken = false;
  use_inprocess(mh);
  return false;
}

// before every sharded call: communicators that were aborted by the previous call's failure are replaced
void ensure_comm(sapca_multi_s* mh) {
  if (mh->h.size() > 1 && mh->rccl && mh->comm_broken) (void)connect_rccl(mh);
}

// nnz-balanced row ranges of a host CSR, one per device (even row counts if a shard would be empty)
sapca_status shard_bounds(sapca_multi_s* mh, uint64_t m, const uint64_t* ro, std::vector<uint64_t>& bounds) {
  const uint32_t nd = (uint32_t)mh->h.size();
  bounds.assign((size_t)nd + 1, 0);
  const sapca_status st = sapca_partition_rows(m, ro, nd, bounds.data());
  if (st != SAPCA_OK) { mh->err = "row_offsets are not a CSR offset array"; return st; }
  for (uint32_t i = 0; i < nd; ++i)
    if (bounds[i] == bounds[i + 1]) {   // (a shard without rows: a few very long rows; even row counts then)
      for (uint32_t j = 0; j <= nd; ++j) bounds[j] = m * j / nd;
      break;
    }
  return SAPCA_OK;
}

template <typename T> struct HostAbi;
template <> struct HostAbi<float> {
  static sapca_status fit(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const float* v) { return sapca_fit_csr_f32(h, m, n, z, p, i, v); }
  static sapca_status transform(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const float* v, float* o) { return sapca_transform_csr_f32(h, m, n, z, p, i, v, o); }
  static sapca_status fit_transform(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const float* v, float* o) { return sapca_fit_transform_csr_f32(h, m, n, z, p, i, v, o); }
};
template <> struct HostAbi<double> {
  static sapca_status fit(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const double* v) { return sapca_fit_csr_f64(h, m, n, z, p, i, v); }
  static sapca_status transform(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const double* v, double* o) { return sapca_transform_csr_f64(h, m, n, z, p, i, v, o); }
  static sapca_status fit_transform(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const double* v, double* o) { return sapca_fit_transform_csr_f64(h, m, n, z, p, i, v, o); }
};

// op: 0 fit, 1 transform, 2 fit_transform
template <typename T>
sapca_status sharded(sapca_multi_s* mh, int op, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const T* v, T* out) {
  if (!mh) return SAPCA_ERR_ARG;
  const uint32_t nd = (uint32_t)mh->h.size();
  if (!ro || (nnz && (!ci || !v)) || (op != 0 && !out && m)) { mh->err = "null CSR array or output buffer"; return SAPCA_ERR_ARG; }
  if (m < nd) { mh->err = "fewer rows than devices"; return SAPCA_ERR_ARG; }
  ensure_comm(mh);
  mh->res_dtype = -1;   // (the host entry points reuse the members' upload buffers: resident shards are gone)
  std::vector<uint64_t> bounds;
  sapca_status st = shard_bounds(mh, m, ro, bounds);
  if (st != SAPCA_OK) return st;
  if (mh->rccl && nnz) {
    // A rank that refuses its shard (a column index out of range) would leave its peers inside an RCCL collective for good:
    // the whole matrix is checked before anyone starts.  (The in-process transport releases the peers instead.)
    std::atomic<bool> bad{false};
    std::vector<std::thread> th;
    const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const uint64_t per = (nnz + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        bool b = false;
        for (uint64_t e = std::min<uint64_t>(nnz, per * t), e1 = std::min<uint64_t>(nnz, e + per); e < e1; ++e) b |= ci[e] >= n;
        if (b) bad.store(true, std::memory_order_relaxed);
      });
    for (auto& t : th) t.join();
    if (bad.load()) { mh->err = "column index out of range"; return SAPCA_ERR_ARG; }
  }
  return on_every_device(mh, [&](int i) {
    const uint64_t r0 = bounds[(size_t)i], r1 = bounds[(size_t)i + 1], e0 = ro[r0];
    std::vector<uint64_t> rebased((size_t)(r1 - r0) + 1);   // the host entry points take offsets that start at 0
    for (uint64_t r = r0; r <= r1; ++r) rebased[(size_t)(r - r0)] = ro[r] - e0;
    const uint64_t mi = r1 - r0, zi = ro[r1] - e0;
    const uint64_t* cii = ci ? ci + e0 : nullptr;
    const T* vi = v ? v + e0 : nullptr;
    T* oi = out ? out + r0 * mh->k : nullptr;
    sapca_handle h = mh->h[(size_t)i];
    if (op == 0) return HostAbi<T>::fit(h, mi, n, zi, rebased.data(), cii, vi);
    if (op == 1) return HostAbi<T>::transform(h, mi, n, zi, rebased.data(), cii, vi, oi);
    return HostAbi<T>::fit_transform(h, mi, n, zi, rebased.data(), cii, vi, oi);
  });
}

template <typename T> struct ResidentAbi;
template <> struct ResidentAbi<float> {
  static constexpr int dtype = 0;
  static sapca_status upload(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const float* v, Shard& sh) {
    float* dv = nullptr;
    const sapca_status st = sapca_upload_csr_f32(h, m, n, z, p, i, v, &sh.ptr, &sh.idx, &dv);
    sh.val = dv;
    return st;
  }
  static sapca_status fit(sapca_handle h, uint64_t n, const Shard& sh) { return sapca_fit_csr_device_f32(h, sh.rows, n, sh.nnz, sh.ptr, sh.idx, static_cast<const float*>(sh.val)); }
  static sapca_status transform(sapca_handle h, uint64_t n, const Shard& sh, float* o) { return sapca_transform_csr_device_to_host_f32(h, sh.rows, n, sh.nnz, sh.ptr, sh.idx, static_cast<const float*>(sh.val), o); }
  static sapca_status fit_transform(sapca_handle h, uint64_t n, const Shard& sh, float* o) { return sapca_fit_transform_csr_device_to_host_f32(h, sh.rows, n, sh.nnz, sh.ptr, sh.idx, static_cast<const float*>(sh.val), o); }
};
template <> struct ResidentAbi<double> {
  static constexpr int dtype = 1;
  static sapca_status upload(sapca_handle h, uint64_t m, uint64_t n, uint64_t z, const uint64_t* p, const uint64_t* i, const double* v, Shard& sh) {
    double* dv = nullptr;
    const sapca_status st = sapca_upload_csr_f64(h, m, n, z, p, i, v, &sh.ptr, &sh.idx, &dv);
    sh.val = dv;
    return st;
  }
  static sapca_status fit(sapca_handle h, uint64_t n, const Shard& sh) { return sapca_fit_csr_device_f64(h, sh.rows, n, sh.nnz, sh.ptr, sh.idx, static_cast<const double*>(sh.val)); }
  static sapca_status transform(sapca_handle h, uint64_t n, const Shard& sh, double* o) { return sapca_transform_csr_device_to_host_f64(h, sh.rows, n, sh.nnz, sh.ptr, sh.idx, static_cast<const double*>(sh.val), o); }
  static sapca_status fit_transform(sapca_handle h, uint64_t n, const Shard& sh, double* o) { return sapca_fit_transform_csr_device_to_host_f64(h, sh.rows, n, sh.nnz, sh.ptr, sh.idx, static_cast<const double*>(sh.val), o); }
};

// every shard to its device, side by side (one host thread per device: the uploads share nothing but the host's DRAM)
template <typename T>
sapca_status upload_shards(sapca_multi_s* mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const T* v) {
  if (!mh) return SAPCA_ERR_ARG;
  const uint32_t nd = (uint32_t)mh->h.size();
  mh->res_dtype = -1;
  if (!ro || (nnz && (!ci || !v))) { mh->err = "null CSR array"; return SAPCA_ERR_ARG; }
  if (m < nd) { mh->err = "fewer rows than devices"; return SAPCA_ERR_ARG; }
  std::vector<uint64_t> bounds;
  sapca_status st = shard_bounds(mh, m, ro, bounds);
  if (st != SAPCA_OK) return st;
  mh->shard.assign(nd, Shard());
  // (no collective in here: a member that fails -- a column index out of range, no memory -- leaves nobody waiting; the
  //  rendezvous / communicators are not touched)
  std::vector<sapca_status> sts(nd, SAPCA_OK);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < nd; ++i)
    th.emplace_back([&, i] {
      const uint64_t r0 = bounds[i], r1 = bounds[i + 1], e0 = ro[r0];
      std::vector<uint64_t> rebased((size_t)(r1 - r0) + 1);
      for (uint64_t r = r0; r <= r1; ++r) rebased[(size_t)(r - r0)] = ro[r] - e0;
      Shard& sh = mh->shard[i];
      sh.row0 = r0; sh.rows = r1 - r0; sh.nnz = ro[r1] - e0;
      sts[i] = ResidentAbi<T>::upload(mh->h[i], sh.rows, n, sh.nnz, rebased.data(), ci ? ci + e0 : nullptr, v ? v + e0 : nullptr, sh);
    });
  for (auto& t : th) t.join();
  for (uint32_t i = 0; i < nd; ++i)
    if (sts[i] != SAPCA_OK) {
      const char* msg = sapca_last_error(mh->h[i]);
      mh->err = std::string("device ") + std::to_string(mh->devices[i]) + " (shard " + std::to_string(i) + "): " + (msg ? msg : "");
      return sts[i];
    }
  mh->res_m = m; mh->res_n = n; mh->res_dtype = ResidentAbi<T>::dtype;
  mh->err.clear();
  return SAPCA_OK;
}

// op: 0 fit, 1 transform, 2 fit_transform on the resident shards
template <typename T>
sapca_status resident(sapca_multi_s* mh, int op, T* out) {
  if (!mh) return SAPCA_ERR_ARG;
  if (mh->res_dtype != ResidentAbi<T>::dtype) {
    mh->err = mh->res_dtype < 0 ? "no resident matrix: call sapca_multi_upload_csr_* first" : "the resident matrix has the other value type";
    return SAPCA_ERR_ARG;
  }
  if (op != 0 && !out) { mh->err = "null output buffer"; return SAPCA_ERR_ARG; }
  ensure_comm(mh);
  return on_every_device(mh, [&](int i) {
    const Shard& sh = mh->shard[(size_t)i];
    T* oi = out ? out + sh.row0 * mh->k : nullptr;
    sapca_handle h = mh->h[(size_t)i];
    if (op == 0) return ResidentAbi<T>::fit(h, mh->res_n, sh);
    if (op == 1) return ResidentAbi<T>::transform(h, mh->res_n, sh, oi);
    return ResidentAbi<T>::fit_transform(h, mh->res_n, sh, oi);
  });
}

}  // namespace

extern "C" {

sapca_status sapca_multi_create(const sapca_options* opts, const int32_t* device_ids, uint32_t n_devices, sapca_multi* out) {
  if (!out) return SAPCA_ERR_ARG;
  *out = nullptr;
  if (!device_ids || n_devices == 0 || n_devices > 64) { g_multi_create_error = "device_ids: 1 to 64 device ordinals"; return SAPCA_ERR_ARG; }
  sapca_options o;
  sapca_options_default(&o);
  if (opts) {
    if (opts->struct_size != sizeof(sapca_options)) { g_multi_create_error = "sapca_options.struct_size mismatch (ABI)"; return SAPCA_ERR_ARG; }
    o = *opts;
  }
  o.stream = nullptr;   // every member runs on a stream of its own device
  sapca_multi_s* mh = new sapca_multi_s();
  mh->k = o.n_components;
  auto fail = [&](sapca_status st, const char* msg) {
    g_multi_create_error = msg ? msg : "";
    sapca_multi_destroy(mh);
    return st;
  };
  for (uint32_t i = 0; i < n_devices; ++i) {
    o.device_id = device_ids[i];
    sapca_handle h = nullptr;
    const sapca_status st = sapca_create(&o, &h);
    if (st != SAPCA_OK) return fail(st, sapca_last_error(nullptr));
    mh->h.push_back(h);
    mh->devices.push_back(device_ids[i]);
  }
  if (n_devices > 1) {
    const std::set<int32_t> distinct(mh->devices.begin(), mh->devices.end());
    const bool want_rccl = distinct.size() == n_devices && sapca_comm_rccl_available() != 0 && getenv("SAPCA_MULTI_INPROCESS") == nullptr;
    if (!want_rccl || !connect_rccl(mh)) use_inprocess(mh);
  }
  *out = mh;
  return SAPCA_OK;
}

void sapca_multi_destroy(sapca_multi mh) {
  if (!mh) return;
  for (sapca_handle h : mh->h) sapca_destroy(h);
  delete mh;
}

const char* sapca_multi_last_error(sapca_multi mh) { return mh ? mh->err.c_str() : g_multi_create_error.c_str(); }
uint32_t sapca_multi_n_devices(sapca_multi mh) { return mh ? (uint32_t)mh->h.size() : 0; }
sapca_handle sapca_multi_member(sapca_multi mh, uint32_t i) { return (mh && i < mh->h.size()) ? mh->h[i] : nullptr; }
/* 1: RCCL between the devices; 0: the in-process all-reduce through page-locked host memory */
int sapca_multi_uses_rccl(sapca_multi mh) { return (mh && mh->rccl) ? 1 : 0; }

sapca_status sapca_multi_set_mask(sapca_multi mh, const uint8_t* mask, size_t n) {
  if (!mh) return SAPCA_ERR_ARG;
  for (size_t i = 0; i < mh->h.size(); ++i) {
    const sapca_status st = sapca_set_mask(mh->h[i], mask, n);
    if (st != SAPCA_OK) { mh->err = sapca_last_error(mh->h[i]); return st; }
  }
  return SAPCA_OK;
}

sapca_status sapca_multi_fit_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const float* v) {
  return sharded<float>(mh, 0, m, n, nnz, ro, ci, v, nullptr);
}
sapca_status sapca_multi_fit_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const double* v) {
  return sharded<double>(mh, 0, m, n, nnz, ro, ci, v, nullptr);
}
sapca_status sapca_multi_transform_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const float* v, float* out) {
  return sharded<float>(mh, 1, m, n, nnz, ro, ci, v, out);
}
sapca_status sapca_multi_transform_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const double* v, double* out) {
  return sharded<double>(mh, 1, m, n, nnz, ro, ci, v, out);
}
sapca_status sapca_multi_fit_transform_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const float* v, float* out) {
  return sharded<float>(mh, 2, m, n, nnz, ro, ci, v, out);
}
sapca_status sapca_multi_fit_transform_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const double* v, double* out) {
  return sharded<double>(mh, 2, m, n, nnz, ro, ci, v, out);
}

sapca_status sapca_multi_upload_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const float* v) {
  return upload_shards<float>(mh, m, n, nnz, ro, ci, v);
}
sapca_status sapca_multi_upload_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci, const double* v) {
  return upload_shards<double>(mh, m, n, nnz, ro, ci, v);
}
sapca_status sapca_multi_fit_resident(sapca_multi mh) {
  if (!mh) return SAPCA_ERR_ARG;
  return mh->res_dtype == 1 ? resident<double>(mh, 0, nullptr) : resident<float>(mh, 0, nullptr);
}
sapca_status sapca_multi_transform_resident_f32(sapca_multi mh, float* out) { return resident<float>(mh, 1, out); }
sapca_status sapca_multi_transform_resident_f64(sapca_multi mh, double* out) { return resident<double>(mh, 1, out); }
sapca_status sapca_multi_fit_transform_resident_f32(sapca_multi mh, float* out) { return resident<float>(mh, 2, out); }
sapca_status sapca_multi_fit_transform_resident_f64(sapca_multi mh, double* out) { return resident<double>(mh, 2, out); }

sapca_status sapca_multi_resident_shard(sapca_multi mh, uint32_t i, uint64_t* first_row, uint64_t* rows, uint64_t* nnz,
                                        const int64_t** d_row_offsets, const int32_t** d_col_indices, void** d_values) {
  if (!mh || mh->res_dtype < 0 || i >= mh->shard.size()) return SAPCA_ERR_ARG;
  const Shard& sh = mh->shard[i];
  if (first_row) *first_row = sh.row0;
  if (rows) *rows = sh.rows;
  if (nnz) *nnz = sh.nnz;
  if (d_row_offsets) *d_row_offsets = sh.ptr;
  if (d_col_indices) *d_col_indices = sh.idx;
  if (d_values) *d_values = sh.val;
  return SAPCA_OK;
}

}  // extern "C"
