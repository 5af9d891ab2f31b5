// All-reduce(sum) over the ranks of a row-sharded fit.  One process per GPU.
// Built-in transport: RCCL, resolved at run time (dlopen "librccl.so.1") so that a host process
// that already carries an RCCL (e.g. PyTorch's) shares it and a single-GPU user needs none.
// Alternative: a caller-supplied callback (e.g. torch.distributed.all_reduce).
#pragma once
#include "common.h"

#include <atomic>
#include <mutex>

namespace sapca {

struct Comm {
  uint32_t nranks = 1, rank = 0;
  enum Mode { NONE, RCCL, CALLBACK } mode = NONE;
  // The communicators are read by the owning thread (allreduce, destroy), by whoever polls async_error(), and swapped out by
  // abort() from ANY thread: atomics; `issue_mu` is held across an enqueue so that abort() frees a communicator only between
  // two of them (or after waiting 250 ms for an enqueue that never returns -- the case ncclCommAbort exists for); `state_mu`
  // keeps abort() / destroy() and a concurrent async_error() apart (a poll never waits for an enqueue).
  std::atomic<void*> rccl_comm{nullptr};
  // a duplicate of the communicator (ncclCommSplit, all ranks one colour) for collectives issued on a second stream: the
  // first piece's all-reduce of a two-piece A^T sweep runs behind the second piece's sweep (engine.cpp); two streams must
  // not issue on one communicator.  Null when the library has no ncclCommSplit: the sweep then runs in one piece.
  std::atomic<void*> rccl_comm2{nullptr};
  std::timed_mutex issue_mu;
  std::mutex state_mu;
  std::atomic<bool> aborted{false};   // abort() was called: every later collective fails at once with SAPCA_ERR_COMM
  sapca_allreduce_fn fn = nullptr;
  void* ctx = nullptr;
  double host_ms = 0;  // accumulated host-observed time inside collectives

  bool active() const { return mode != NONE; }
  void init_rccl(uint32_t nranks_, uint32_t rank_, const uint8_t id[128]);
  void set_callback(uint32_t nranks_, uint32_t rank_, sapca_allreduce_fn f, void* c);
  // dtype: 0 = f32, 1 = f64.  In place on a device buffer, ordered on `s`.
  // lane 1: the side stream's communicator (RCCL); the callback transports take any stream
  void allreduce(void* buf, uint64_t count, int dtype, hipStream_t s, int lane = 0);
  bool has_side_lane() const { return mode == CALLBACK || (mode == RCCL && rccl_comm2.load() != nullptr); }
  // Ends every collective this rank has in flight or will issue (ncclCommAbort on both communicators): kernels of a
  // collective a failed peer never joins return, and so do the stream waits behind them.  Callable from any thread.  The
  // communicator is unusable afterwards (init again).
  void abort();
  // 0: no asynchronous error on the communicators; otherwise the ncclResult_t RCCL reports (a peer died, a network error)
  int async_error();
  void destroy();
  static void unique_id(uint8_t id[128]);
  static bool rccl_available();   // librccl and the four entry points resolve in this process
};

}  // namespace sapca
