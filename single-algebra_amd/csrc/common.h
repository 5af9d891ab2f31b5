// Shared host-side plumbing for libsapca: error propagation, device buffers, views.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sapca.h"
#include "switches.h"

namespace sapca {

struct Error : std::runtime_error {
  sapca_status code;
  Error(sapca_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define SAPCA_HIP(expr)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      throw ::sapca::Error(SAPCA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

#define SAPCA_CHECK(cond, code, msg)                 \
  do {                                               \
    if (!(cond)) throw ::sapca::Error((code), (msg)); \
  } while (0)

inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

// Grow-only device buffer: repeated fits of the same shape never re-allocate.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  void* ensure(size_t bytes) {
    if (bytes > cap) {
      release();
      size_t want = bytes + bytes / 16 + 256;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) {
        p = nullptr;
        throw Error(SAPCA_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
      }
      cap = want;
    }
    return p;
  }
  template <typename U>
  U* as(size_t count) { return static_cast<U*>(ensure(count * sizeof(U))); }
  template <typename U>
  U* ptr() const { return static_cast<U*>(p); }
};

// Non-owning view of a device CSR matrix (rows x cols).
template <typename T>
struct CsrView {
  int64_t rows = 0, cols = 0, nnz = 0;
  const int64_t* ptr = nullptr;  // [rows+1]
  const int32_t* idx = nullptr;  // [nnz] ascending within a row
  const T* val = nullptr;        // [nnz]
};

// Tile-major companion of an f32 CSR operator for the LDS-staged sweep (spmm_tiled.hip).
// Dynamic LDS above the default limit has to be enabled per kernel AND per device: a process may drive
// handles on several devices.  `state` is one static per call site (per template instantiation).
struct LdsAttrState {
  std::atomic<int> bytes[64];
  LdsAttrState() { for (auto& b : bytes) b.store(0, std::memory_order_relaxed); }
};
inline void ensure_dynamic_lds(const void* kernel, size_t bytes, LdsAttrState& state) {
  int dev = 0;
  SAPCA_HIP(hipGetDevice(&dev));
  std::atomic<int>& have = state.bytes[dev & 63];
  if (have.load(std::memory_order_acquire) >= (int)bytes) return;
  SAPCA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have.store((int)bytes, std::memory_order_release);
}

// Page-locked host staging buffer (grow-only), for the chunked upload of the host entry points.
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void* ensure(size_t bytes) {
    if (bytes > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr;
      cap = 0;
      hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
      if (e != hipSuccess) {
        p = nullptr;
        throw Error(SAPCA_ERR_NOMEM, std::string("hipHostMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
      }
      cap = bytes;
    }
    return p;
  }
};

struct TiledOp {
  bool valid = false;
  int64_t rows = 0, cols = 0, total_entries = 0;
  int ldp = 0;              // panel leading dimension (elements) the format was built for: 64 or 128 (f32), 64 (f64)
  int elem = 4;             // bytes per value and panel element: 4 (f32) or 8 (f64)
  int tc = 0, nct = 0;      // panel rows per column tile, number of tiles
  int nrb = 0;              // row blocks (one workgroup each)
  int block_rows = 0;       // most rows a block holds (quad format: 256 f64, 512, or 1024 for the DPP-fed sweep with 16 row slots)
  int nsplit = 1, tiles_per_split = 0;
  int slots = 2;            // lane groups per wave the entry stream was padded for
  int fmt = 0;              // 0: two half-waves share a row; 1: "quad", one row per 16-lane group
  int tile_bytes = 0;       // LDS bytes of one panel tile (the entry staging takes the rest of the 160 KiB)
  int64_t max_chunk = 0;    // most entries of one (row block, tile): the staged-entry kernels need it to fit their LDS staging
  const int32_t* blk_row0 = nullptr;   // [nrb+1] slot positions
  int64_t mid_row0 = -1;               // blk_row0[nrb / 2] on the host (the cut of a sweep in two pieces); -1: not recorded
  const uint32_t* row_perm = nullptr;  // [rows] slot position -> row (rows sorted by length, longest first); null = identity
  const int64_t* chunk_off = nullptr;  // [nrb*nct+1] entry offsets
  const uint32_t* wave_off = nullptr;  // [nrb*nct][8]
  const uint8_t* steps = nullptr;      // [nrb*nct][256]
  const void* ent = nullptr;           // {u32 lds byte offset, f32 value} or {u32 offset, u32 pad, f64 value}
  // DPP-fed sweep (spmm_dq.hip): per (row block, wave, tile) {entry offset / 8, 16-step chunks}
  bool dq = false;
  const uint32_t* dq_info = nullptr;
};
struct TiledBuffers {
  DevBuf blk, seg, steps, wave_off, chunk_off, ent, tmp, misc, run, rank, perm, lens, dq_info, bounds;
  PinnedBuf host;   // page-locked staging of the builder's small host <-> device exchanges
  // the DPP-fed sweep's tables are a latency-bound kernel over the counts: it runs on this stream beside the
  // bandwidth-bound fill (fork / join events on the build's own stream)
  hipStream_t aux = nullptr;
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  TiledBuffers() = default;
  TiledBuffers(const TiledBuffers&) = delete;
  TiledBuffers& operator=(const TiledBuffers&) = delete;
  ~TiledBuffers() {
    if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); }
    if (aux_fork) (void)hipEventDestroy(aux_fork);
    if (aux_join) (void)hipEventDestroy(aux_join);
  }
};

struct Stream {
  hipStream_t s = nullptr;
  bool owned = false;
};

// HIP-event stopwatch on a stream; a no-op unless enabled.
struct EventTimer {
  bool enabled = false;
  hipStream_t s = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
  struct Span { size_t i; };
  void begin_collect(hipStream_t st, bool on) {
    enabled = on;
    s = st;
    used = 0;
  }
  int start() {
    if (!enabled) return -1;
    if (used == pool.size()) {
      hipEvent_t a, b;
      SAPCA_HIP(hipEventCreate(&a));
      SAPCA_HIP(hipEventCreate(&b));
      pool.emplace_back(a, b);
    }
    SAPCA_HIP(hipEventRecord(pool[used].first, s));
    return (int)used++;
  }
  void stop(int i) {
    if (i >= 0) SAPCA_HIP(hipEventRecord(pool[i].second, s));
  }
  double ms(int i) {
    if (i < 0) return 0.0;
    float t = 0;
    SAPCA_HIP(hipEventElapsedTime(&t, pool[i].first, pool[i].second));
    return t;
  }
  ~EventTimer() {
    for (auto& p : pool) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
  }
};

}  // namespace sapca
