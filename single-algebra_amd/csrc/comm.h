// All-reduce(sum) over the ranks of a row-sharded fit.  One process per GPU.
// Built-in transport: RCCL, resolved at run time (dlopen "librccl.so.1") so that a host process
// that already carries an RCCL (e.g. PyTorch's) shares it and a single-GPU user needs none.
// Alternative: a caller-supplied callback (e.g. torch.distributed.all_reduce).
#pragma once
#include "common.h"

namespace sapca {

struct Comm {
  uint32_t nranks = 1, rank = 0;
  enum Mode { NONE, RCCL, CALLBACK } mode = NONE;
  void* rccl_comm = nullptr;
  sapca_allreduce_fn fn = nullptr;
  void* ctx = nullptr;
  double host_ms = 0;  // accumulated host-observed time inside collectives

  bool active() const { return mode != NONE; }
  void init_rccl(uint32_t nranks_, uint32_t rank_, const uint8_t id[128]);
  void set_callback(uint32_t nranks_, uint32_t rank_, sapca_allreduce_fn f, void* c);
  // dtype: 0 = f32, 1 = f64.  In place on a device buffer, ordered on `s`.
  void allreduce(void* buf, uint64_t count, int dtype, hipStream_t s);
  void destroy();
  static void unique_id(uint8_t id[128]);
  static bool rccl_available();   // librccl and the four entry points resolve in this process
};

}  // namespace sapca
