// extern "C" surface of libsapca.so (include/sapca.h).  Every function: set device, try,
// translate exceptions into a status + per-handle message.  No compute lives here.
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>

#include "engine.h"

using sapca::CsrView;
using sapca::Engine;
using sapca::Error;

namespace {

thread_local std::string g_create_error;

template <typename F>
sapca_status guarded(sapca_handle h, F&& f) {
  if (!h) return SAPCA_ERR_ARG;
  try {
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) throw Error(SAPCA_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    f();
    h->err.clear();
    return SAPCA_OK;
  } catch (const Error& e) {
    h->err = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    h->err = "out of host memory";
    return SAPCA_ERR_NOMEM;
  } catch (const std::exception& e) {
    h->err = e.what();
    return SAPCA_ERR_ARG;
  }
}

// Host CSR (nalgebra layout, usize indices) -> device CSR in the handle's upload buffers.
// PCIe is the bound of this path, so the usize column indices are narrowed to int32 on the HOST, chunk by
// chunk into a page-locked ring by a few threads, while the previous chunk's DMA (and, first, the values')
// is in flight: 8 bytes per stored entry cross the bus instead of 12.
namespace {
constexpr size_t kUpChunk = (size_t)16 << 20;   // indices per chunk (64 MiB of int32)

// out[i] = (int32) in[i]; returns true if any in[i] >= n
bool narrow_chunk(const uint64_t* in, int32_t* out, size_t count, uint64_t n, unsigned nthreads) {
  std::atomic<bool> bad{false};
  auto work = [&](size_t lo, size_t hi) {
    bool b = false;
    for (size_t i = lo; i < hi; ++i) {
      const uint64_t c = in[i];
      b |= c >= n;
      out[i] = (int32_t)c;
    }
    if (b) bad.store(true, std::memory_order_relaxed);
  };
  if (nthreads <= 1 || count < (size_t)1 << 16) {
    work(0, count);
  } else {
    std::vector<std::thread> th;
    const size_t per = (count + nthreads - 1) / nthreads;
    for (unsigned t = 1; t < nthreads; ++t) {
      const size_t lo = std::min(count, t * per), hi = std::min(count, lo + per);
      if (lo < hi) th.emplace_back(work, lo, hi);
    }
    work(0, std::min(count, per));
    for (auto& x : th) x.join();
  }
  return bad.load();
}
}  // namespace

// side stream shared by prepare() (A's format beside the transposition) and upload() (statistics beside the DMA)
void ensure_side_stream(sapca_handle h) {
  if (h->stream2) return;
  SAPCA_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  SAPCA_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  SAPCA_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
}

// with_stats: the matrix is about to be fitted -- accumulate its column statistics (R1/R2 and the per-column counts)
// on the side stream, chunk by chunk behind the DMA, so prepare() finds them ready
template <typename T>
CsrView<T> upload(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* row_offsets,
                  const uint64_t* col_indices, const T* values, bool with_stats = false) {
  SAPCA_CHECK(row_offsets != nullptr && (nnz == 0 || (col_indices && values)), SAPCA_ERR_ARG, "null CSR array");
  SAPCA_CHECK(row_offsets[0] == 0 && row_offsets[m] == nnz, SAPCA_ERR_ARG, "row_offsets do not span [0, nnz]");
  {   // with monotone offsets and in-range columns (checked below) no kernel can leave the arrays
    bool monotone = true;
    for (uint64_t i = 0; i < m; ++i) monotone &= row_offsets[i] <= row_offsets[i + 1];
    SAPCA_CHECK(monotone, SAPCA_ERR_ARG, "row_offsets must be non-decreasing");
  }
  SAPCA_CHECK(n < (1ull << 31) && m < (1ull << 31), SAPCA_ERR_ARG, "more than 2^31-1 rows or columns is not supported");
  hipStream_t s = h->stream;
  h->prep_key.valid = false;  // the upload buffers are about to hold a different matrix
  h->up_stats.valid = false;
  auto t0 = std::chrono::steady_clock::now();
  int64_t* d_ptr = h->in_ptr.as<int64_t>(m + 1);
  int32_t* d_idx = h->in_idx.as<int32_t>(std::max<uint64_t>(nnz, 1));
  T* d_val = h->in_val.as<T>(std::max<uint64_t>(nnz, 1));
  uint64_t* d64 = h->up64.as<uint64_t>(m + 1 + 8);
  int* flag = reinterpret_cast<int*>(d64 + m + 1);
  SAPCA_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  SAPCA_HIP(hipMemcpyAsync(d64, row_offsets, (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  static const bool on_device = sapca::dbg_env("SAPCA_UPLOAD_NARROW_ON_DEVICE") != nullptr;   // the first version: ship u64, narrow on the GPU
  bool bad_host = false, stats_here = false;
  void* stats_work = nullptr;
  if (nnz == 0 || !on_device) sapca::k::narrow_indices(d64, d64, (int64_t)m, 0, (int64_t)n, d_ptr, d_idx, flag, s);   // row offsets only
  if (nnz) {
    SAPCA_HIP(hipMemcpyAsync(d_val, values, nnz * sizeof(T), hipMemcpyHostToDevice, s));
    if (on_device) {
      uint64_t* d64i = h->up64i.as<uint64_t>(nnz);
      SAPCA_HIP(hipMemcpyAsync(d64i, col_indices, nnz * sizeof(uint64_t), hipMemcpyHostToDevice, s));
      sapca::k::narrow_indices(d64, d64i, (int64_t)m, (int64_t)nnz, (int64_t)n, d_ptr, d_idx, flag, s);
    } else {
      const unsigned nthreads = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
      const size_t chunk = std::min<size_t>(kUpChunk, nnz);
      for (int b = 0; b < 2; ++b) {
        h->up_stage[b].ensure(chunk * sizeof(int32_t));
        if (!h->up_done[b]) SAPCA_HIP(hipEventCreateWithFlags(&h->up_done[b], hipEventDisableTiming));
      }
      static const bool stats_off = sapca::dbg_env("SAPCA_UPLOAD_STATS_OFF") != nullptr;
      stats_here = with_stats && !stats_off && n > 0 && sapca::k::exact_colstats_bytes<T>((int64_t)n) <= ((size_t)1 << 30);
      if (stats_here) {
        ensure_side_stream(h);
        stats_work = h->up_stats.work.ensure(sapca::k::exact_colstats_bytes<T>((int64_t)n));
        sapca::k::exact_colstats_reset<T>(stats_work, (int64_t)n, h->stream2);
        SAPCA_HIP(hipEventRecord(h->ev_fork, s));                        // row offsets and values are on the device
        SAPCA_HIP(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        sapca::k::exact_colstats_scan_values<T>(d_val, (int64_t)nnz, (int64_t)n, stats_work, h->stream2);
      }
      bool used[2] = {false, false};
      int b = 0;
      for (size_t off = 0; off < nnz; off += chunk, b ^= 1) {
        const size_t cnt = std::min<size_t>(chunk, nnz - off);
        if (used[b]) SAPCA_HIP(hipEventSynchronize(h->up_done[b]));   // its DMA has drained
        int32_t* stage = static_cast<int32_t*>(h->up_stage[b].p);
        bad_host |= narrow_chunk(col_indices + off, stage, cnt, n, nthreads);
        SAPCA_HIP(hipMemcpyAsync(d_idx + off, stage, cnt * sizeof(int32_t), hipMemcpyHostToDevice, s));
        SAPCA_HIP(hipEventRecord(h->up_done[b], s));
        used[b] = true;
        if (stats_here) {   // this chunk's entries (the values all went first) feed the accumulators while the next chunk crosses
          SAPCA_HIP(hipStreamWaitEvent(h->stream2, h->up_done[b], 0));
          const uint64_t* ro_end = row_offsets + m + 1;
          const int64_t r_lo = (int64_t)(std::upper_bound(row_offsets, ro_end, (uint64_t)off) - row_offsets) - 1;
          const int64_t r_hi = (int64_t)(std::lower_bound(row_offsets, ro_end, (uint64_t)(off + cnt)) - row_offsets);
          sapca::k::exact_colstats_add<T>(d_ptr, d_idx, d_val, r_lo, std::min<int64_t>(r_hi, (int64_t)m), (int64_t)off,
                                          (int64_t)(off + cnt), (int64_t)n, stats_work, h->stream2);
        }
      }
    }
  }
  int bad = 0;
  if (stats_here) {
    // the last chunk's share and the final rounding stay in flight behind this call: their consumer (prepare(), the column
    // statistics helper) waits for `up_stats_done` and reads the "a value was inf/nan" flag then
    double* out = h->up_stats.out.as<double>(3 * n);
    int* nonfinite = static_cast<int*>(h->up_stats.flag.ensure(sizeof(int)));
    sapca::k::exact_colstats_finish<T>(stats_work, (int64_t)n, out, nonfinite, h->stream2);
    if (!h->up_stats_done) SAPCA_HIP(hipEventCreateWithFlags(&h->up_stats_done, hipEventDisableTiming));
    SAPCA_HIP(hipEventRecord(h->up_stats_done, h->stream2));
  }
  SAPCA_HIP(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, s));
  SAPCA_HIP(hipStreamSynchronize(s));
  SAPCA_CHECK(bad == 0 && !bad_host, SAPCA_ERR_ARG, "column index out of range");
  if (stats_here) {
    h->up_stats.m = m; h->up_stats.n = n; h->up_stats.nnz = nnz; h->up_stats.dtype = sizeof(T) == 4 ? 0 : 1;
    h->up_stats.valid = true;
  }
  h->timings.upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CsrView<T> v;
  v.rows = (int64_t)m; v.cols = (int64_t)n; v.nnz = (int64_t)nnz;
  v.ptr = d_ptr; v.idx = d_idx; v.val = d_val;
  return v;
}

template <typename T>
CsrView<T> device_view(uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p, const int32_t* i, const T* v) {
  SAPCA_CHECK(p != nullptr && (nnz == 0 || (i && v)), SAPCA_ERR_ARG, "null CSR array");
  SAPCA_CHECK(n < (1ull << 31) && m < (1ull << 31), SAPCA_ERR_ARG, "more than 2^31-1 rows or columns is not supported");
  CsrView<T> a;
  a.rows = (int64_t)m; a.cols = (int64_t)n; a.nnz = (int64_t)nnz; a.ptr = p; a.idx = i; a.val = v;
  return a;
}

// dst <- src on a few host threads (a pageable destination at DRAM speed instead of one core's)
void parallel_copy(void* dst, const void* src, size_t bytes, unsigned nthreads) {
  if (nthreads <= 1 || bytes < ((size_t)1 << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = ((bytes + nthreads - 1) / nthreads + 63) & ~(size_t)63;
  for (unsigned t = 1; t < nthreads; ++t) {
    const size_t lo = std::min(bytes, t * per), hi = std::min(bytes, lo + per);
    if (lo < hi) th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, hi - lo); });
  }
  std::memcpy(dst, src, std::min(bytes, per));
  for (auto& x : th) x.join();
}

// Device -> caller's (pageable) host array.  A direct copy to pageable memory runs at about 10 GB/s; results of more than a
// few MB (the m x k projection) go through the page-locked ring of the upload instead: DMA into one half while a few
// threads copy the other half out.
template <typename T>
void download_out(sapca_handle h, const T* d, T* out, size_t count) {
  hipStream_t s = h->stream;
  const size_t bytes = count * sizeof(T);
  constexpr size_t kPiece = (size_t)8 << 20;
  if (bytes < 2 * kPiece) {
    SAPCA_HIP(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipStreamSynchronize(s));
    return;
  }
  const unsigned nthreads = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  for (int b = 0; b < 2; ++b) {
    h->up_stage[b].ensure(kPiece);
    if (!h->up_done[b]) SAPCA_HIP(hipEventCreateWithFlags(&h->up_done[b], hipEventDisableTiming));
  }
  const char* src = reinterpret_cast<const char*>(d);
  char* dst = reinterpret_cast<char*>(out);
  size_t pending_off[2] = {0, 0}, pending_len[2] = {0, 0};
  int b = 0;
  for (size_t off = 0; off < bytes; off += kPiece, b ^= 1) {
    if (pending_len[b]) {   // the piece that used this half: drained, copy it out
      SAPCA_HIP(hipEventSynchronize(h->up_done[b]));
      parallel_copy(dst + pending_off[b], h->up_stage[b].p, pending_len[b], nthreads);
    }
    const size_t len = std::min(kPiece, bytes - off);
    SAPCA_HIP(hipMemcpyAsync(h->up_stage[b].p, src + off, len, hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipEventRecord(h->up_done[b], s));
    pending_off[b] = off;
    pending_len[b] = len;
  }
  for (int i = 0; i < 2; ++i, b ^= 1)   // oldest first
    if (pending_len[b]) {
      SAPCA_HIP(hipEventSynchronize(h->up_done[b]));
      parallel_copy(dst + pending_off[b], h->up_stage[b].p, pending_len[b], nthreads);
      pending_len[b] = 0;
    }
  SAPCA_HIP(hipStreamSynchronize(s));
}

template <typename T>
sapca_status fit_host(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci,
                      const T* v) {
  return guarded(h, [&] { Engine<T>::fit(*h, upload<T>(h, m, n, nnz, ro, ci, v, true)); });
}

template <typename T>
sapca_status transform_host(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,
                            const uint64_t* ci, const T* v, T* out, bool fit_first) {
  return guarded(h, [&] {
    SAPCA_CHECK(out != nullptr || m == 0, SAPCA_ERR_ARG, "null output buffer");
    if (!fit_first) {
      if (!h->mask.empty() && h->mask.size() != n)
        throw Error(SAPCA_ERR_MASK_LEN, "The mask vector length and the number of features (columns) have to be the same!");
      if (!h->fitted) throw Error(SAPCA_ERR_NOT_FITTED, "Must be fitted before transform!");
    }
    CsrView<T> A = upload<T>(h, m, n, nnz, ro, ci, v, fit_first);
    if (fit_first) Engine<T>::fit(*h, A, true);   // (its host-side tail runs once the projection is queued)
    try {
      T* d_out = h->out_tmp.as<T>(std::max<uint64_t>(m * h->k, 1));
      Engine<T>::transform(*h, A, d_out);
      download_out(h, d_out, out, (size_t)(m * h->k));
    } catch (...) {
      Engine<T>::finish_fit(*h);
      throw;
    }
  });
}

template <typename T>
sapca_status fit_device(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p, const int32_t* i,
                        const T* v) {
  return guarded(h, [&] { Engine<T>::fit(*h, device_view<T>(m, n, nnz, p, i, v)); });
}

template <typename T>
sapca_status transform_device(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p,
                              const int32_t* i, const T* v, T* d_out, bool fit_first, bool out_on_host = false) {
  return guarded(h, [&] {
    SAPCA_CHECK(d_out != nullptr || m == 0, SAPCA_ERR_ARG, "null output buffer");
    CsrView<T> A = device_view<T>(m, n, nnz, p, i, v);
    T* host_out = out_on_host ? d_out : nullptr;
    // (the staging holds what transform() writes and download_out() copies: m x k of the model that projects -- the one
    // being fitted here has at most n_components, a fitted one has h->k)
    if (out_on_host) d_out = h->out_tmp.as<T>(std::max<uint64_t>(m * std::max<uint64_t>(h->opt.n_components, fit_first ? 0 : h->k), 1));
    if (fit_first) {
      Engine<T>::fit(*h, A, true);
    } else if (static_cast<const void*>(v) != h->in_val.p) {
      // A separate transform call on caller-owned device arrays: pointer identity does not prove the matrix is the fitted
      // one (values edited in place, an allocator handing the same addresses to another matrix), so the preparation kept
      // from fit() -- tile-major copy of the values, per-column counts -- is not reused.  It is reused inside
      // fit_transform (one call, the matrix is borrowed unmodified) and for the library-owned arrays of sapca_upload_csr_*,
      // which only the library's own entry points modify (each of them drops the preparation).
      h->prep_key.valid = false;
    }
    try {
      Engine<T>::transform(*h, A, d_out);
      if (host_out) download_out(h, d_out, host_out, (size_t)(m * h->k));
    } catch (...) {
      Engine<T>::finish_fit(*h);   // (a fit whose tail was held back for the projection still completes)
      throw;
    }
  });
}

template <typename T>
void copy_out(const std::vector<double>& src, T* out, size_t cap) {
  SAPCA_CHECK(out != nullptr && cap >= src.size(), SAPCA_ERR_ARG, "output buffer too small");
  for (size_t i = 0; i < src.size(); ++i) out[i] = (T)src[i];
}

void need_fitted(sapca_handle h) {
  if (!h->fitted) throw Error(SAPCA_ERR_NOT_FITTED, "Model must be fitted first!");  // sparse/mod.rs:299,316
}

// components live on the device in the fitted dtype; fetched (and converted) on demand
template <typename T>
void get_components(sapca_handle h, T* out, size_t cap, bool squared) {
  need_fitted(h);
  const size_t count = (size_t)(h->k * h->n_used);
  SAPCA_CHECK(out != nullptr && cap >= count, SAPCA_ERR_ARG, "output buffer too small");
  if (h->dtype == 0) {
    std::vector<float> tmp(count);
    download_out(h, h->components_dev.ptr<float>(), tmp.data(), count);
    for (size_t i = 0; i < count; ++i) out[i] = squared ? (T)(tmp[i] * tmp[i]) : (T)tmp[i];
  } else {
    std::vector<double> tmp(count);
    download_out(h, h->components_dev.ptr<double>(), tmp.data(), count);
    for (size_t i = 0; i < count; ++i) out[i] = squared ? (T)(tmp[i] * tmp[i]) : (T)tmp[i];
  }
}

template <typename T>
void get_ratio(sapca_handle h, T* out, size_t cap, bool cumulative) {
  need_fitted(h);
  SAPCA_CHECK(out != nullptr && cap >= h->k, SAPCA_ERR_ARG, "output buffer too small");
  // sparse/mod.rs:318-319: ratio_i = ev_i / sum over the k computed components, in T
  T total = 0;
  for (uint64_t i = 0; i < h->k; ++i) total += (T)h->expl_var[i];
  T run = 0;
  for (uint64_t i = 0; i < h->k; ++i) {
    const T r = (T)h->expl_var[i] / total;
    run += r;  // sparse/mod.rs:336-340
    out[i] = cumulative ? run : r;
  }
}

// stage-level helpers -------------------------------------------------------------------
template <typename T>
sapca_status colstats_host(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,
                           const uint64_t* ci, const T* v, T* sum_col, T* sum_sq, uint64_t* cnt) {
  return guarded(h, [&] {
    CsrView<T> A = upload<T>(h, m, n, nnz, ro, ci, v, true);
    hipStream_t s = h->stream;
    std::vector<double> host(3 * n);
    if (h->up_stats.valid) SAPCA_HIP(hipEventSynchronize(h->up_stats_done));
    if (h->up_stats.valid && *static_cast<const int*>(h->up_stats.flag.p) == 0) {   // gathered behind the DMA (exact sums, rounded once)
      download_out(h, h->up_stats.out.ptr<double>(), host.data(), 3 * n);
    } else {                   // row sums of A^T: what prepare() runs on device-resident input
      int64_t* at_ptr = h->at_ptr.as<int64_t>(n + 1);
      int32_t* at_idx = h->at_idx.as<int32_t>(std::max<uint64_t>(nnz, 1));
      T* at_val = h->at_val.as<T>(std::max<uint64_t>(nnz, 1));
      sapca::k::transpose_csr(A, at_ptr, at_idx, at_val, h->scratch, s);
      CsrView<T> At;
      At.rows = (int64_t)n; At.cols = (int64_t)m; At.nnz = (int64_t)nnz; At.ptr = at_ptr; At.idx = at_idx; At.val = at_val;
      double* d = h->stats.as<double>(3 * n + 1);
      sapca::k::row_sums(At, d, d + n, s);
      sapca::k::row_lengths_f64(at_ptr, (int64_t)n, d + 2 * n, s);
      download_out(h, d, host.data(), 3 * n);
    }
    for (uint64_t j = 0; j < n; ++j) {
      if (sum_col) sum_col[j] = (T)host[j];
      if (sum_sq) sum_sq[j] = (T)host[n + j];
      if (cnt) cnt[j] = (uint64_t)host[2 * n + j];
    }
  });
}

// ---- device-resident workflow: upload once, preprocess and analyse in HBM -------------
template <typename T>
sapca_status upload_host(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci,
                         const T* v, const int64_t** d_ptr, const int32_t** d_idx, T** d_val) {
  return guarded(h, [&] {
    SAPCA_CHECK(d_ptr && d_idx && d_val, SAPCA_ERR_ARG, "null output pointer");
    CsrView<T> A = upload<T>(h, m, n, nnz, ro, ci, v, true);   // (a later fit of the untouched arrays finds its statistics ready)
    *d_ptr = A.ptr;
    *d_idx = A.idx;
    *d_val = const_cast<T*>(A.val);
  });
}

template <typename T>
sapca_status normalize_device(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p, const int32_t* i, T* v,
                              const double* sums, uint64_t sums_len, double target, int32_t direction) {
  return guarded(h, [&] {
    SAPCA_CHECK(direction == 0 || direction == 1, SAPCA_ERR_ARG, "direction must be 0 (ROW) or 1 (COLUMN)");
    const uint64_t want = direction == 1 ? n : m;
    SAPCA_CHECK(sums != nullptr || want == 0, SAPCA_ERR_ARG, "null sums");
    SAPCA_CHECK(sums_len == want, SAPCA_ERR_ARG,
                direction == 1 ? "Length of sums must match number of columns" : "Length of sums must match number of rows");
    CsrView<T> A = device_view<T>(m, n, nnz, p, i, v);
    if (want == 0 || nnz == 0) return;
    h->prep_key.valid = false;   // the values change under any cached preparation
    h->up_stats.valid = false;   // .. and under the statistics gathered at upload
    double* d = h->stats.as<double>(2 * want);
    SAPCA_HIP(hipMemcpyAsync(d, sums, want * sizeof(double), hipMemcpyHostToDevice, h->stream));
    sapca::k::normalize_csr(A, v, d, target, direction == 1, d + want, h->stream);
    SAPCA_HIP(hipStreamSynchronize(h->stream));   // `sums` may go out of scope
  });
}

template <typename T>
sapca_status log1p_device(sapca_handle h, uint64_t nnz, T* v) {
  return guarded(h, [&] {
    SAPCA_CHECK(v != nullptr || nnz == 0, SAPCA_ERR_ARG, "null values");
    h->prep_key.valid = false;
    h->up_stats.valid = false;
    sapca::k::log1p_values(v, (int64_t)nnz, h->stream);
    SAPCA_HIP(hipStreamSynchronize(h->stream));
  });
}

// direction 0: per row (sum_row, sum_row_squared, nonzero_row, min_max_row); 1: per column (the same on A^T)
template <typename T>
sapca_status stats_device(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p, const int32_t* i, const T* v,
                          int32_t direction, double* sum, double* sumsq, uint64_t* nonzero, T* minv, T* maxv) {
  return guarded(h, [&] {
    SAPCA_CHECK(direction == 0 || direction == 1, SAPCA_ERR_ARG, "direction must be 0 (ROW) or 1 (COLUMN)");
    CsrView<T> A = device_view<T>(m, n, nnz, p, i, v);
    hipStream_t s = h->stream;
    const uint64_t len = direction == 1 ? n : m;
    if (len == 0) return;
    h->prep_key.valid = false;   // the statistics and transposition buffers are shared with prepare()
    CsrView<T> R = A;
    if (direction == 1) {
      // (the transposition buffers are shared with prepare())
      int64_t* at_ptr = h->at_ptr.as<int64_t>(n + 1);
      int32_t* at_idx = h->at_idx.as<int32_t>(std::max<uint64_t>(nnz, 1));
      T* at_val = h->at_val.as<T>(std::max<uint64_t>(nnz, 1));
      sapca::k::transpose_csr(A, at_ptr, at_idx, at_val, h->scratch, s);
      R.rows = (int64_t)n; R.cols = (int64_t)m; R.ptr = at_ptr; R.idx = at_idx; R.val = at_val;
    }
    double* d = h->stats.as<double>(3 * len + 1);
    T* dmm = h->out_tmp.as<T>(2 * len);
    sapca::k::row_stats(R, d, d + len, dmm, dmm + len, s);
    sapca::k::row_lengths_f64(R.ptr, (int64_t)len, d + 2 * len, s);
    std::vector<double> host(3 * len);
    std::vector<T> mm(2 * len);
    SAPCA_HIP(hipMemcpyAsync(host.data(), d, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipMemcpyAsync(mm.data(), dmm, mm.size() * sizeof(T), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipStreamSynchronize(s));
    for (uint64_t j = 0; j < len; ++j) {
      if (sum) sum[j] = host[j];
      if (sumsq) sumsq[j] = host[len + j];
      if (nonzero) nonzero[j] = (uint64_t)host[2 * len + j];
      if (minv) minv[j] = mm[j];
      if (maxv) maxv[j] = mm[len + j];
    }
  });
}

template <typename T>
sapca_status spmm_host(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro, const uint64_t* ci,
                       const T* v, const T* mu, uint64_t l, const T* In, T* Out, bool transposed) {
  return guarded(h, [&] {
    SAPCA_CHECK(l >= 1 && l <= (uint64_t)sapca::k::kMaxPanelWidth && In && Out, SAPCA_ERR_ARG, "spmm: panel width must be in [1, 1024]");
    CsrView<T> A = upload<T>(h, m, n, nnz, ro, ci, v);
    hipStream_t s = h->stream;
    // spmm_variant == 2 exercises the LDS-staged sweep on its own (f32 only): panels padded to 64/128
    const bool want_tiled = h->opt.spmm_variant == 2;
    const int ld = l > 128 ? (int)sapca::round_up((int64_t)l, 64) : want_tiled ? (l <= 64 ? 64 : 128) : (int)sapca::round_up((int64_t)l, 16);
    const uint64_t in_rows = transposed ? m : n, out_rows = transposed ? n : m;
    T* stage = h->scratch2.as<T>(std::max<uint64_t>(in_rows * l, out_rows * l) + n + 2 * (uint64_t)std::max(ld, 128));
    T* X = h->panel_x.as<T>(std::max<uint64_t>(in_rows, 1) * ld);
    T* Y = h->panel_y.as<T>(std::max<uint64_t>(out_rows, 1) * ld);
    SAPCA_HIP(hipMemcpyAsync(stage, In, in_rows * l * sizeof(T), hipMemcpyHostToDevice, s));
    sapca::k::add_padding(stage, (int64_t)in_rows, (int)l, X, ld, s);
    T* d_mu = nullptr;
    if (mu) {
      d_mu = h->mean_used_dev.as<T>(n);
      SAPCA_HIP(hipMemcpyAsync(d_mu, mu, n * sizeof(T), hipMemcpyHostToDevice, s));
    }
    const size_t W = (size_t)std::max(ld, 128);   // (the layout of engine.cpp's SmallLayout)
    double* small = h->small.as<double>(6 * W * W + 64 + 4 * W);
    T* cvec = reinterpret_cast<T*>(small + 6 * W * W + 64);
    T* svec = cvec + W;
    SAPCA_HIP(hipStreamSynchronize(s));
    if (!transposed) {
      if (mu) sapca::k::weighted_colsum(X, (int64_t)n, ld, d_mu, cvec, h->scratch, s);
      const sapca::TiledOp* top = nullptr;
      if constexpr (sizeof(T) == 4) {
        if (want_tiled) {
          if (sapca::k::build_tiled(A, false, sapca::k::tiled_geometry((int)l), h->tiled_a, h->tb_a, s)) top = &h->tiled_a;   // else: row kernel
        }
      } else {
        if (want_tiled && sapca::k::build_tiled(A, 64, h->tiled_a, h->tb_a, s)) top = &h->tiled_a;
      }
      sapca::k::spmm(A, top, X, ld, Y, ld, ld, mu ? cvec : nullptr, h->opt.spmm_variant, h->split_scratch, s);
    } else {
      int64_t* at_ptr = h->at_ptr.as<int64_t>(n + 1);
      int32_t* at_idx = h->at_idx.as<int32_t>(std::max<uint64_t>(nnz, 1));
      T* at_val = h->at_val.as<T>(std::max<uint64_t>(nnz, 1));
      sapca::k::transpose_csr(A, at_ptr, at_idx, at_val, h->scratch, s);
      CsrView<T> At;
      At.rows = (int64_t)n; At.cols = (int64_t)m; At.nnz = (int64_t)nnz; At.ptr = at_ptr; At.idx = at_idx; At.val = at_val;
      const sapca::TiledOp* top = nullptr;
      if constexpr (sizeof(T) == 4) {
        if (want_tiled) {
          if (sapca::k::build_tiled(At, false, sapca::k::tiled_geometry((int)l), h->tiled_at, h->tb_at, s)) top = &h->tiled_at;
        }
      } else {
        if (want_tiled && sapca::k::build_tiled(At, 64, h->tiled_at, h->tb_at, s)) top = &h->tiled_at;
      }
      sapca::k::spmm(At, top, X, ld, Y, ld, ld, (const T*)nullptr, h->opt.spmm_variant, h->split_scratch, s);
      if (mu) {
        sapca::k::weighted_colsum(X, (int64_t)m, ld, (const T*)nullptr, svec, h->scratch, s);
        sapca::k::rank1_subtract(Y, (int64_t)n, ld, d_mu, svec, s);
      }
    }
    sapca::k::strip_padding(Y, (int64_t)out_rows, ld, (int)l, stage, s);
    download_out(h, stage, Out, (size_t)(out_rows * l));
  });
}

template <typename T>
sapca_status normalize_host(sapca_handle h, int32_t normalizer, uint64_t rows, uint64_t l, T* panel) {
  return guarded(h, [&] {
    SAPCA_CHECK(l >= 1 && l <= (uint64_t)sapca::k::kMaxPanelWidth && panel && rows >= 1, SAPCA_ERR_ARG, "normalize: panel width must be in [1, 1024]");
    SAPCA_CHECK(normalizer >= 0 && normalizer <= 2, SAPCA_ERR_ARG, "unknown normalizer");
    hipStream_t s = h->stream;
    const int ld = (int)sapca::round_up((int64_t)l, l > 128 ? 64 : 16);
    T* stage = h->scratch.as<T>(rows * l);
    T* P = h->panel_y.as<T>(rows * ld);
    SAPCA_HIP(hipMemcpyAsync(stage, panel, rows * l * sizeof(T), hipMemcpyHostToDevice, s));
    sapca::k::add_padding(stage, (int64_t)rows, (int)l, P, ld, s);
    Engine<T>::normalize(*h, P, (int64_t)rows, (int)l, ld, normalizer, false, nullptr, nullptr);
    sapca::k::strip_padding(P, (int64_t)rows, ld, (int)l, stage, s);
    download_out(h, stage, panel, (size_t)(rows * l));
  });
}

template <typename T>
sapca_status omega_host(sapca_handle h, uint64_t rows, uint64_t l, T* out) {
  return guarded(h, [&] {
    SAPCA_CHECK(l >= 1 && l <= (uint64_t)sapca::k::kMaxPanelWidth && out, SAPCA_ERR_ARG, "omega: panel width must be in [1, 1024]");
    const int ld = (int)sapca::round_up((int64_t)l, 16);
    T* P = h->panel_x.as<T>(std::max<uint64_t>(rows, 1) * ld);
    T* stage = h->scratch.as<T>(std::max<uint64_t>(rows * l, 1));
    sapca::k::gaussian_panel(P, (int64_t)rows, (int)l, ld, h->opt.random_seed, h->stream);
    sapca::k::strip_padding(P, (int64_t)rows, ld, (int)l, stage, h->stream);
    download_out(h, stage, out, (size_t)(rows * l));
  });
}

}  // namespace

extern "C" {

int sapca_abi_version(void) { return SAPCA_ABI_VERSION; }

void sapca_options_default(sapca_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (uint32_t)sizeof(sapca_options);
  o->random_seed = 42;        // sparse/mod.rs:397
  o->n_components = 50;       // :394
  o->alpha = 1.0;             // :395
  o->tolerance = 1e-6;        // :396
  o->center = 1;              // :398
  o->verbose = 0;             // :399
  o->method = SAPCA_LANCZOS;  // pca/mod.rs:64-68
  o->n_oversamples = 10;
  o->n_power_iterations = 4;
  o->normalizer = SAPCA_NORM_QR;
  o->transform_semantics = SAPCA_TRANSFORM_REFERENCE;
  o->device_id = -1;
  o->spmm_variant = 0;
  o->stream = nullptr;
}

sapca_status sapca_create(const sapca_options* opts, sapca_handle* out) {
  if (!out) return SAPCA_ERR_ARG;
  *out = nullptr;
  try {
    sapca_options o;
    sapca_options_default(&o);
    if (opts) {
      SAPCA_CHECK(opts->struct_size == sizeof(sapca_options), SAPCA_ERR_ARG, "sapca_options.struct_size mismatch (ABI)");
      o = *opts;
    }
    SAPCA_CHECK(o.method == SAPCA_LANCZOS || o.method == SAPCA_RANDOM, SAPCA_ERR_ARG, "unknown SVD method");
    SAPCA_CHECK(o.normalizer >= 0 && o.normalizer <= 2, SAPCA_ERR_ARG, "unknown normalizer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
      throw Error(SAPCA_ERR_HIP, "no HIP device available: libsapca has no CPU path");
    sapca_handle h = new sapca_handle_s();
    h->opt = o;
    if (o.device_id >= 0) h->device = o.device_id;
    else SAPCA_HIP(hipGetDevice(&h->device));
    SAPCA_HIP(hipSetDevice(h->device));
    if (o.stream) {
      h->stream = static_cast<hipStream_t>(o.stream);
    } else {
      SAPCA_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
      h->own_stream = true;
    }
    *out = h;
    return SAPCA_OK;
  } catch (const Error& e) {
    g_create_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return SAPCA_ERR_NOMEM;
  }
}

void sapca_destroy(sapca_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->comm.destroy();
  for (int b = 0; b < 2; ++b)
    if (h->up_done[b]) (void)hipEventDestroy(h->up_done[b]);
  if (h->stream2) {
    (void)hipStreamSynchronize(h->stream2);
    (void)hipStreamDestroy(h->stream2);
  }
  if (h->stream_comm) {
    (void)hipStreamSynchronize(h->stream_comm);
    (void)hipStreamDestroy(h->stream_comm);
    if (h->ev_piece) (void)hipEventDestroy(h->ev_piece);
    if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
  }
  if (h->stream3) {
    (void)hipStreamSynchronize(h->stream3);
    (void)hipStreamDestroy(h->stream3);
  }
  if (h->ev_small) (void)hipEventDestroy(h->ev_small);
  if (h->ev_kept) (void)hipEventDestroy(h->ev_kept);
  if (h->ev_stats) (void)hipEventDestroy(h->ev_stats);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_drop) (void)hipEventDestroy(h->ev_drop);
  if (h->up_stats_done) (void)hipEventDestroy(h->up_stats_done);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* sapca_last_error(sapca_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

sapca_status sapca_set_mask(sapca_handle h, const uint8_t* mask, size_t len) {
  if (!h) return SAPCA_ERR_ARG;
  if (len && !mask) return SAPCA_ERR_ARG;
  h->mask.assign(mask, mask + len);
  for (auto& b : h->mask) b = b ? 1 : 0;
  h->mask_version++;
  h->prep_key.valid = false;
  return SAPCA_OK;
}

sapca_status sapca_set_omega_f32(sapca_handle h, const float* omega, size_t rows, size_t cols) {
  if (!h || ((rows * cols) && !omega)) return SAPCA_ERR_ARG;
  h->omega.assign(omega, omega + rows * cols);
  h->omega_rows = rows;
  h->omega_cols = cols;
  return SAPCA_OK;
}
sapca_status sapca_set_omega_f64(sapca_handle h, const double* omega, size_t rows, size_t cols) {
  if (!h || ((rows * cols) && !omega)) return SAPCA_ERR_ARG;
  h->omega.assign(omega, omega + rows * cols);
  h->omega_rows = rows;
  h->omega_cols = cols;
  return SAPCA_OK;
}

#define SAPCA_DEFINE_TYPED(SUF, T)                                                                                       \
  sapca_status sapca_fit_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,             \
                                   const uint64_t* ci, const T* v) {                                                     \
    return fit_host<T>(h, m, n, nnz, ro, ci, v);                                                                         \
  }                                                                                                                      \
  sapca_status sapca_transform_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,       \
                                         const uint64_t* ci, const T* v, T* out) {                                       \
    return transform_host<T>(h, m, n, nnz, ro, ci, v, out, false);                                                       \
  }                                                                                                                      \
  sapca_status sapca_fit_transform_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,   \
                                             const uint64_t* ci, const T* v, T* out) {                                   \
    return transform_host<T>(h, m, n, nnz, ro, ci, v, out, true);                                                        \
  }                                                                                                                      \
  sapca_status sapca_fit_csr_device_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p,        \
                                          const int32_t* i, const T* v) {                                                \
    return fit_device<T>(h, m, n, nnz, p, i, v);                                                                         \
  }                                                                                                                      \
  sapca_status sapca_transform_csr_device_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p,  \
                                                const int32_t* i, const T* v, T* out) {                                  \
    return transform_device<T>(h, m, n, nnz, p, i, v, out, false);                                                       \
  }                                                                                                                      \
  sapca_status sapca_fit_transform_csr_device_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,                \
                                                    const int64_t* p, const int32_t* i, const T* v, T* out) {            \
    return transform_device<T>(h, m, n, nnz, p, i, v, out, true);                                                        \
  }                                                                                                                      \
  sapca_status sapca_transform_csr_device_to_host_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,            \
                                                        const int64_t* p, const int32_t* i, const T* v, T* out) {        \
    return transform_device<T>(h, m, n, nnz, p, i, v, out, false, true);                                                 \
  }                                                                                                                      \
  sapca_status sapca_fit_transform_csr_device_to_host_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,        \
                                                            const int64_t* p, const int32_t* i, const T* v, T* out) {    \
    return transform_device<T>(h, m, n, nnz, p, i, v, out, true, true);                                                  \
  }                                                                                                                      \
  sapca_status sapca_get_components_##SUF(sapca_handle h, T* out, size_t cap) {                                          \
    return guarded(h, [&] { get_components<T>(h, out, cap, false); });                                                   \
  }                                                                                                                      \
  sapca_status sapca_get_feature_importances_##SUF(sapca_handle h, T* out, size_t cap) {                                 \
    return guarded(h, [&] { get_components<T>(h, out, cap, true); });                                                    \
  }                                                                                                                      \
  sapca_status sapca_get_singular_values_##SUF(sapca_handle h, T* out, size_t cap) {                                     \
    return guarded(h, [&] { need_fitted(h); copy_out<T>(h->sing, out, cap); });                                          \
  }                                                                                                                      \
  sapca_status sapca_get_explained_variance_##SUF(sapca_handle h, T* out, size_t cap) {                                  \
    return guarded(h, [&] { need_fitted(h); copy_out<T>(h->expl_var, out, cap); });                                      \
  }                                                                                                                      \
  sapca_status sapca_get_mean_##SUF(sapca_handle h, T* out, size_t cap) {                                                \
    return guarded(h, [&] { need_fitted(h); copy_out<T>(h->mean, out, cap); });                                          \
  }                                                                                                                      \
  sapca_status sapca_get_explained_variance_ratio_##SUF(sapca_handle h, T* out, size_t cap) {                            \
    return guarded(h, [&] { get_ratio<T>(h, out, cap, false); });                                                        \
  }                                                                                                                      \
  sapca_status sapca_get_cumulative_explained_variance_ratio_##SUF(sapca_handle h, T* out, size_t cap) {                 \
    return guarded(h, [&] { get_ratio<T>(h, out, cap, true); });                                                         \
  }                                                                                                                      \
  sapca_status sapca_colstats_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,        \
                                        const uint64_t* ci, const T* v, T* sc, T* ssq, uint64_t* cnt) {                  \
    return colstats_host<T>(h, m, n, nnz, ro, ci, v, sc, ssq, cnt);                                                      \
  }                                                                                                                      \
  sapca_status sapca_upload_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,          \
                                      const uint64_t* ci, const T* v, const int64_t** dp, const int32_t** di, T** dv) {  \
    return upload_host<T>(h, m, n, nnz, ro, ci, v, dp, di, dv);                                                          \
  }                                                                                                                      \
  sapca_status sapca_normalize_csr_device_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p,  \
                                                const int32_t* i, T* v, const double* sums, uint64_t sums_len,            \
                                                double target, int32_t direction) {                                      \
    return normalize_device<T>(h, m, n, nnz, p, i, v, sums, sums_len, target, direction);                                \
  }                                                                                                                      \
  sapca_status sapca_log1p_csr_device_##SUF(sapca_handle h, uint64_t nnz, T* v) { return log1p_device<T>(h, nnz, v); }    \
  sapca_status sapca_stats_csr_device_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p,      \
                                            const int32_t* i, const T* v, int32_t direction, double* sum, double* sumsq, \
                                            uint64_t* nonzero, T* minv, T* maxv) {                                       \
    return stats_device<T>(h, m, n, nnz, p, i, v, direction, sum, sumsq, nonzero, minv, maxv);                           \
  }                                                                                                                      \
  sapca_status sapca_spmm_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,            \
                                    const uint64_t* ci, const T* v, const T* mu, uint64_t l, const T* X, T* Y) {         \
    return spmm_host<T>(h, m, n, nnz, ro, ci, v, mu, l, X, Y, false);                                                    \
  }                                                                                                                      \
  sapca_status sapca_spmmt_csr_##SUF(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const uint64_t* ro,           \
                                     const uint64_t* ci, const T* v, const T* mu, uint64_t l, const T* Y, T* Z) {        \
    return spmm_host<T>(h, m, n, nnz, ro, ci, v, mu, l, Y, Z, true);                                                     \
  }                                                                                                                      \
  sapca_status sapca_normalize_panel_##SUF(sapca_handle h, int32_t normalizer, uint64_t rows, uint64_t l, T* panel) {    \
    return normalize_host<T>(h, normalizer, rows, l, panel);                                                             \
  }                                                                                                                      \
  sapca_status sapca_generate_omega_##SUF(sapca_handle h, uint64_t rows, uint64_t l, T* out) {                           \
    return omega_host<T>(h, rows, l, out);                                                                               \
  }

SAPCA_DEFINE_TYPED(f32, float)
SAPCA_DEFINE_TYPED(f64, double)
#undef SAPCA_DEFINE_TYPED

sapca_status sapca_get_dims(sapca_handle h, uint64_t* k, uint64_t* n_used, uint64_t* n_cols) {
  return guarded(h, [&] {
    need_fitted(h);
    if (k) *k = h->k;
    if (n_used) *n_used = h->n_used;
    if (n_cols) *n_cols = h->n_cols;
  });
}

sapca_status sapca_get_total_variance(sapca_handle h, double* out) {
  return guarded(h, [&] {
    need_fitted(h);
    SAPCA_CHECK(out != nullptr, SAPCA_ERR_ARG, "null output");
    *out = h->total_var;
  });
}

sapca_status sapca_get_mask_index_maps(sapca_handle h, uint64_t* cols_to_use, size_t cap_cols, int64_t* orig_to_masked,
                                       size_t cap_map) {
  return guarded(h, [&] {
    // maps follow the mask currently set (sparse_masked/mod.rs:264-271, :455-466); no fit needed
    std::vector<uint64_t> cols;
    std::vector<int64_t> o2m(h->mask.size(), -1);
    for (size_t j = 0; j < h->mask.size(); ++j)
      if (h->mask[j]) {
        o2m[j] = (int64_t)cols.size();
        cols.push_back((uint64_t)j);
      }
    if (cols_to_use) {
      SAPCA_CHECK(cap_cols >= cols.size(), SAPCA_ERR_ARG, "cols_to_use buffer too small");
      std::copy(cols.begin(), cols.end(), cols_to_use);
    }
    if (orig_to_masked) {
      SAPCA_CHECK(cap_map >= o2m.size(), SAPCA_ERR_ARG, "orig_to_masked buffer too small");
      std::copy(o2m.begin(), o2m.end(), orig_to_masked);
    }
  });
}

sapca_status sapca_get_timings(sapca_handle h, sapca_timings* out) {
  if (!h || !out) return SAPCA_ERR_ARG;
  *out = h->timings;
  return SAPCA_OK;
}

sapca_status sapca_partition_rows(uint64_t m, const uint64_t* row_offsets, uint32_t nparts, uint64_t* bounds) {
  if (!row_offsets || !bounds || nparts == 0) return SAPCA_ERR_ARG;
  // contiguous ranges balanced by stored entries: boundary p is the first row whose
  // prefix reaches p/nparts of nnz, kept monotone.
  const uint64_t nnz = row_offsets[m] - row_offsets[0];
  bounds[0] = 0;
  for (uint32_t p = 1; p < nparts; ++p) {
    const uint64_t target = row_offsets[0] + (uint64_t)((long double)nnz * p / nparts);
    const uint64_t* it = std::lower_bound(row_offsets, row_offsets + m + 1, target);
    uint64_t r = (uint64_t)(it - row_offsets);
    if (r > 0 && r <= m && target - row_offsets[r - 1] < row_offsets[r] - target) r -= 1;  // nearer boundary
    if (nnz == 0) r = m * p / nparts;
    if (r > m) r = m;
    if (r < bounds[p - 1]) r = bounds[p - 1];
    bounds[p] = r;
  }
  bounds[nparts] = m;
  return SAPCA_OK;
}

sapca_status sapca_upload_values_changed(sapca_handle h) {
  return guarded(h, [&] {
    h->prep_key.valid = false;
    h->up_stats.valid = false;
  });
}

sapca_status sapca_measure_copy_gbs(sapca_handle h, uint64_t bytes, uint32_t reps, double* gbs) {
  return guarded(h, [&] {
    SAPCA_CHECK(gbs != nullptr && bytes >= 4096 && reps >= 1, SAPCA_ERR_ARG, "measure_copy_gbs: bytes >= 4096, reps >= 1");
    bytes &= ~(uint64_t)15;
    void *src = nullptr, *dst = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    auto cleanup = [&] {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      if (src) (void)hipFree(src);
      if (dst) (void)hipFree(dst);
    };
    try {
      SAPCA_HIP(hipMalloc(&src, bytes));
      SAPCA_HIP(hipMalloc(&dst, bytes));
      SAPCA_HIP(hipMemsetAsync(src, 1, bytes, h->stream));
      SAPCA_HIP(hipEventCreate(&a));
      SAPCA_HIP(hipEventCreate(&b));
      float best = 0;
      for (uint32_t r = 0; r <= reps; ++r) {   // (the first pass also faults the pages in)
        SAPCA_HIP(hipEventRecord(a, h->stream));
        sapca::k::stream_copy16(src, dst, (int64_t)bytes, h->stream);
        SAPCA_HIP(hipEventRecord(b, h->stream));
        SAPCA_HIP(hipEventSynchronize(b));
        float ms = 0;
        SAPCA_HIP(hipEventElapsedTime(&ms, a, b));
        if (r > 0 && (best == 0 || ms < best)) best = ms;
      }
      *gbs = 2.0 * (double)bytes / ((double)best * 1e-3) / 1e9;
    } catch (...) {
      cleanup();
      throw;
    }
    cleanup();
  });
}

int sapca_comm_rccl_available(void) { return sapca::Comm::rccl_available() ? 1 : 0; }

sapca_status sapca_comm_unique_id(uint8_t id[128]) {
  if (!id) return SAPCA_ERR_ARG;
  try {
    sapca::Comm::unique_id(id);
    return SAPCA_OK;
  } catch (const Error& e) {
    g_create_error = e.what();
    return e.code;
  }
}

sapca_status sapca_comm_init_rank(sapca_handle h, uint32_t nranks, uint32_t rank, const uint8_t id[128]) {
  return guarded(h, [&] {
    SAPCA_CHECK(id != nullptr, SAPCA_ERR_ARG, "null unique id");
    h->comm.init_rccl(nranks, rank, id);
  });
}

sapca_status sapca_comm_set_callback(sapca_handle h, uint32_t nranks, uint32_t rank, sapca_allreduce_fn fn, void* ctx) {
  return guarded(h, [&] { h->comm.set_callback(nranks, rank, fn, ctx); });
}

sapca_status sapca_comm_abort(sapca_handle h) {
  if (!h) return SAPCA_ERR_ARG;
  try {   // (not `guarded`: callable from another thread while a fit of this handle is blocked behind a collective)
    h->comm.abort();
    return SAPCA_OK;
  } catch (...) {
    return SAPCA_ERR_COMM;
  }
}

sapca_status sapca_comm_async_error(sapca_handle h, int32_t* state) {
  if (!h || !state) return SAPCA_ERR_ARG;
  try {
    *state = h->comm.async_error();
    return SAPCA_OK;
  } catch (...) {
    return SAPCA_ERR_COMM;
  }
}

int sapca_comm_has_side_lane(sapca_handle h) { return (h && h->comm.has_side_lane()) ? 1 : 0; }

sapca_status sapca_comm_allreduce(sapca_handle h, void* buf, uint64_t count, int32_t dtype) {
  return guarded(h, [&] {
    h->comm.allreduce(buf, count, dtype, h->stream);
    SAPCA_HIP(hipStreamSynchronize(h->stream));
  });
}

}  // extern "C"
