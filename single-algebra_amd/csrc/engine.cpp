// Host orchestration of the sparse-PCA hot path on one GPU (one rank of a row-sharded job).
// Mirrors, step by step:
//   SparsePCA::fit            /root/reference/src/dimred/pca/sparse/mod.rs:102-242
//   MaskedSparsePCA::fit      /root/reference/src/dimred/pca/sparse_masked/mod.rs:255-419
//   SparsePCA::transform      sparse/mod.rs:255-285   (quirk Q2)
//   MaskedSparsePCA::transform sparse_masked/mod.rs:438-546 (quirk Q3)
// with the single-svdlib calls (randomized_svd, svd_las2, svd_flip, MaskedCSRMatrix) realised
// by the kernels in this directory.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <thread>

#include "lanczos.h"
#include "small_svd.h"
#include "spmm_dq.h"

namespace sapca {

namespace {

// The handle's small f64 buffer: six ld x ld matrices (Gram, R^-1, scratch R, R1, R2, the small factor M), an int, and the
// centring vectors c | s of the sweeps (T, W entries each) -- laid out for panels of W = max(ld, 128) columns.
struct SmallLayout {
  size_t W;
  explicit SmallLayout(int ld) : W((size_t)std::max(ld, 128)) {}
  size_t doubles() const { return 6 * W * W + 64 + 4 * W; }
  size_t info_at() const { return 6 * W * W; }
  size_t cvec_at() const { return 6 * W * W + 64; }
  size_t wsum_at() const { return 6 * W * W + 64 + 2 * W; }   // W doubles behind c | s: the column sums a Gram pass gathers
};


// Does the LDS-staged sweep pay on an operator of `rows` x `cols` with `entries` stored entries and panels of l columns?  It
// refills an 80 KiB panel tile per (row block, column tile) chunk and only beats the row-gather kernel when a chunk carries
// enough entries to amortise that fill (measured: 64 tile bytes per entry or less -> clearly faster; 164 -> no gain, 4x the
// preparation), and small operators stay on the row kernel whatever their density: their panels sit in L2 / Infinity Cache
// (tools/crossover.py, k = 50: f32 ties at 1e7 entries and the staged path wins by 18 % at 1.6e7; f64 wins by 27 % at 1e7).
// Returns the panel leading dimension of the tile geometry, 0 for the row kernel.
template <typename T>
int staged_sweep_ldp(int64_t rows, int64_t cols, double entries, int64_t l, int spmm_variant) {
  if (spmm_variant == 1 || rows <= 0 || cols <= 0 || l < 1 || l > k::kMaxPanelWidth) return 0;
  const int ldp = sizeof(T) == 4 ? k::tiled_geometry((int)l) : 64;
  const double row_bytes = (double)ldp * sizeof(T);
  const double tile_bytes = 80.0 * 1024.0, block_rows = row_bytes == 256.0 ? 512.0 : 256.0;
  const double chunks = std::ceil((double)rows / block_rows) * std::ceil((double)cols * row_bytes / tile_bytes);
  double floor_entries = sizeof(T) == 4 ? 1e7 : 5e6;
  if (const char* e = dbg_env("SAPCA_TILED_MIN_ENTRIES")) floor_entries = atof(e);   // tests: shards either side of the floor
  const bool dense_enough = entries * (sizeof(T) == 4 ? 64.0 : 96.0) >= chunks * tile_bytes && entries >= floor_entries;
  return (spmm_variant == 2 || dense_enough) ? ldp : 0;
}

enum Cat { C_PREPARE, C_STATS, C_SPMM, C_SPMMT, C_ORTHO, C_SMALL, C_LANCZOS, C_TRANSFORM, C_COMM, C_COUNT };

struct Scope {
  sapca_handle_s& h;
  int ev;
  Scope(sapca_handle_s& h_, int cat) : h(h_), ev(h_.timer.start()) {
    if (ev >= 0) h.spans.emplace_back(cat, ev);
  }
  ~Scope() {
    try { h.timer.stop(ev); } catch (...) {}
  }
};

void collect_timings(sapca_handle_s& h, bool is_fit) {
  if (!h.timer.enabled) return;
  sapca_timings& t = h.timings;
  if (is_fit) {
    const double keep_upload = t.upload_ms;
    const uint64_t keep_steps = t.lanczos_steps;
    std::memset(&t, 0, sizeof(t));
    t.upload_ms = keep_upload;
    t.lanczos_steps = keep_steps;
  } else {
    t.transform_ms = 0;
    for (int ev : h.small_in_transform) t.transform_ms -= h.timer.ms(ev);   // the held-back small SVD counts as small_svd_ms only
  }
  double comm_dev_ms = 0;
  for (auto& sp : h.spans) {
    if (is_fit == (sp.first == C_TRANSFORM)) continue;  // fit spans on fit, transform spans on transform
    const double ms = h.timer.ms(sp.second);
    switch (sp.first) {
      case C_PREPARE: t.prepare_ms += ms; break;
      case C_STATS: t.stats_ms += ms; break;
      case C_SPMM:
        t.spmm_ms += ms;
        if (t.n_spmm < 32) t.spmm_sweep_ms[t.n_spmm] = ms;
        t.n_spmm++;
        break;
      case C_SPMMT:
        t.spmmt_ms += ms;
        if (t.n_spmmt < 32) t.spmmt_sweep_ms[t.n_spmmt] = ms;
        t.n_spmmt++;
        break;
      case C_ORTHO: t.ortho_ms += ms; break;
      case C_SMALL: t.small_svd_ms += ms; break;
      case C_LANCZOS: t.lanczos_ms += ms; break;
      case C_TRANSFORM: t.transform_ms += ms; break;
      case C_COMM: comm_dev_ms += ms; break;
      default: break;
    }
  }
  // collectives: device time between events around every all-reduce on the library stream (what the GPU waited);
  // the host-observed time is what remains when timings are not collected
  if (is_fit) t.comm_ms = comm_dev_ms > 0 ? comm_dev_ms : h.comm.host_ms;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// prepare: A^T, column statistics, mask compaction.
// ------------------------------------------------------------------------------------------
template <typename T>
void Engine<T>::prepare(H& h, const CsrView<T>& A) {
  hipStream_t s = h.stream;
  const int64_t m = A.rows, n = A.cols, nnz = A.nnz;
  const bool masked = !h.mask.empty();
  if (masked && (int64_t)h.mask.size() != n)  // sparse_masked/mod.rs:258-262
    throw Error(SAPCA_ERR_MASK_LEN, "The mask vector length and the number of features (columns) have to be the same!");
  h.prep_key.valid = false;
  // A masked fit that failed after its prepare() (n_components check, SVD failure, no Lanczos convergence) leaves its
  // statistics chain queued on the third stream: that chain sorts in at_* and copies into the pinned statistics, which
  // every kind of fit reuses from here on.  Idle in the normal case.
  if (h.stream3) SAPCA_HIP(hipStreamSynchronize(h.stream3));
  h.stats_pending = false;
  h.stats_on_side = false;
  h.lz_scatter = false;

  // LDS-staged sweep (randomized fits): decided here because it fixes the order in which the transposed rows are produced
  // (staged_sweep_ldp above has the break-even).
  int tiled_ldp = 0;
  if (h.opt.method == SAPCA_RANDOM && m > 0 && n > 0) {
    const int64_t n_kept = masked ? (int64_t)std::count_if(h.mask.begin(), h.mask.end(), [](uint8_t b) { return b != 0; }) : n;
    const int64_t l = std::min<int64_t>((int64_t)(h.opt.n_components + h.opt.n_oversamples), std::min<int64_t>(m, n_kept));
    if (n_kept > 0) tiled_ldp = staged_sweep_ldp<T>(m, n_kept, (double)nnz * ((double)n_kept / (double)n), l, h.opt.spmm_variant);
  }
  const bool from_at = dbg_env("SAPCA_TILED_FROM_A") == nullptr;   // A^T's format from the transposed CSR (default) or straight from A
  const bool at_tile_major = tiled_ldp != 0 && from_at && dbg_env("SAPCA_AT_NATURAL") == nullptr;
  const int tiled_ldp_words = tiled_ldp * (int)sizeof(T) / 4;   // panel row in 4-byte words: what the tile arithmetic counts in

  // mask index maps (sparse_masked/mod.rs:264-271 and the HashMap of :462-466)
  h.cols_to_use.clear();
  h.orig_to_masked.clear();
  h.has_mask_maps = masked;
  std::vector<int32_t> o2m32, sel;   // (alive until the copies below have been synchronised)
  int32_t *d_o2m = nullptr, *d_sel = nullptr;
  if (masked) {
    h.orig_to_masked.assign((size_t)n, -1);
    for (int64_t j = 0; j < n; ++j)
      if (h.mask[(size_t)j]) {
        h.orig_to_masked[(size_t)j] = (int64_t)h.cols_to_use.size();
        h.cols_to_use.push_back((uint64_t)j);
      }
  }
  const int64_t n_used = masked ? (int64_t)h.cols_to_use.size() : n;
  if (masked) {
    o2m32.resize((size_t)n);
    sel.resize((size_t)std::max<int64_t>(n_used, 1));
    for (int64_t j = 0; j < n; ++j) o2m32[(size_t)j] = (int32_t)h.orig_to_masked[(size_t)j];
    for (int64_t j = 0; j < n_used; ++j) sel[(size_t)j] = (int32_t)h.cols_to_use[(size_t)j];
    d_o2m = h.o2m_dev.as<int32_t>((size_t)std::max<int64_t>(n, 1));
    d_sel = h.sel_rows_dev.as<int32_t>((size_t)std::max<int64_t>(n_used, 1));
    SAPCA_HIP(hipMemcpyAsync(d_o2m, o2m32.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    SAPCA_HIP(hipMemcpyAsync(d_sel, sel.data(), (size_t)n_used * sizeof(int32_t), hipMemcpyHostToDevice, s));
  }

  CsrView<T> At;
  const uint64_t* at_packed = nullptr;
  bool at_seg_ready = false;
  // a host matrix that came through upload() brought its statistics along (gathered behind the DMA, exact sums)
  bool from_upload = h.up_stats.valid && A.ptr == h.in_ptr.p && A.idx == h.in_idx.p && A.val == h.in_val.p &&
                     h.up_stats.m == (uint64_t)m && h.up_stats.n == (uint64_t)n && h.up_stats.nnz == (uint64_t)nnz &&
                     h.up_stats.dtype == kDtype && n > 0 && !h.comm.active();   // (ranks must not differ in their collectives)

  if (from_upload) {
    // the last chunk's share of those statistics may still be in flight; the accumulators refuse inf/nan (the flag is
    // final once the side stream has passed up_stats_done): the sums of the transposed matrix take over then
    SAPCA_HIP(hipEventSynchronize(h.up_stats_done));
    if (*static_cast<const int*>(h.up_stats.flag.p) != 0) {
      h.up_stats.valid = false;
      from_upload = false;
    }
  }

  // Lanczos fits whose transposed side fits a workgroup's LDS (at most ~19k kept columns: C3 keeps 18k) never build A^T: the
  // second product of a step scatters into LDS (scatter.hip, fixed-point sums: reproducible) and the column statistics
  // come from the same kind of pass over A, on the third stream beside the iterations (a Lanczos fit does not centre:
  // nothing reads them before fit() ends).  What is left of the preparation is the mask compaction.
  // SAPCA_LANCZOS_TRANSPOSE=1 brings the transposed operator (radix sort) back.
  const bool serial_early = dbg_env("SAPCA_PREPARE_SERIAL") != nullptr;
  bool lz_scatter = h.opt.method == SAPCA_LANCZOS && n_used > 0 && n_used <= m && m >= 4096 && nnz > 0 && k::scatter_fits(n_used) &&
                    (!masked || !serial_early) && dbg_env("SAPCA_LANCZOS_TRANSPOSE") == nullptr;
  bool lz_side = false;   // its statistics run on the third stream
  if (lz_scatter && !masked) {
    // max |a| for the fixed-point scales (masked fits: the compaction gathers it on its way through the values).  Values that
    // are not finite turn the scale, and with it every product and sum, into nan: the fit then fails to converge, as it does
    // on the floating-point route.
    k::absmax(A.val, nnz, h.lz_scalars.as<unsigned long long>(4), s);
  }
  h.lz_scatter = lz_scatter;

  // A's side of the preparation runs beside the main stream.  Its pieces synchronise with the host (entry counts come
  // back), so where the main thread has its own synchronising work a helper thread drives them on the side stream:
  //  * masked fits: the column compaction (MaskedCSRMatrix::new, sparse_masked/mod.rs:313), then the compacted matrix's format;
  //  * unmasked f32 fits on the staged sweep: A's format, while this thread builds A^T's straight from A (below).
  bool a_built_aside = false, ok_a_aside = false;
  std::thread aside;
  std::exception_ptr aside_err;
  int64_t nnz_used = 0;
  struct Joiner {   // (an exception on the main path must not leave the helper running into freed state)
    std::thread& t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } joiner{aside};
  const bool serial = dbg_env("SAPCA_PREPARE_SERIAL") != nullptr;
  const bool masked_aside = masked && n_used > 0 && !serial;
  const bool try_direct = sizeof(T) == 4 && at_tile_major && !masked && !serial;   // A^T's format without a transposed CSR
  // Masked fits compact first (MaskedCSRMatrix::new, sparse_masked/mod.rs:313) and transpose only what the mask keeps.  The
  // entries the compaction drops leave as (column, value) pairs: the sums of the masked-out columns (mean_ is full width,
  // sparse_masked/mod.rs:279-286) come from a stable sort of those pairs by column.  On the staged sweep (f32) the
  // compacted matrix then takes the bucket route to A^T's format; otherwise it is transposed into a CSR.
  const bool try_masked_direct = sizeof(T) == 4 && at_tile_major && masked_aside && n_used <= 65536 && dbg_env("SAPCA_AT_SORT") == nullptr;
  bool compaction_done = false;
  bool side_stats = false;   // the masked-out columns' sums (and the host copy of all statistics) finish on stream3, behind the fit
  // SAPCA_MASK_SUMS_SCATTER=1: single-rank masked fits take those sums straight from A on the third stream (scatter.hip:
  // fixed-point sums in LDS) instead of sorting the (column, value) pairs the compaction writes out for it.  Measured slower
  // on randomized fits (C3's matrix in f32: 14.0 against 12.9 ms): the scatter kernels take a CU's whole LDS, so the sweeps
  // cannot share the chip with them the way they do with the sort's light kernels.  Not the default.
  const bool scatter_sums = masked && !from_upload && !lz_scatter && !h.comm.active() && dbg_env("SAPCA_MASK_STATS_INLINE") == nullptr &&
                            dbg_env("SAPCA_MASK_SUMS_SCATTER") != nullptr;
  auto scatter_sums_on_stream3 = [&](double* d_drop) {   // sum | sumsq of EVERY column of A (the kept ones are overwritten later)
    unsigned long long* sc = h.lz_scalars.as<unsigned long long>(4);
    k::absmax(A.val, nnz, sc, h.stream3);
    k::colstats_scatter(A, sc, d_drop, d_drop + n, (double*)nullptr, h.drop_tmp, h.stream3);
  };
  int32_t* drop_col = nullptr;
  T* drop_val = nullptr;
  if (masked_aside && dbg_env("SAPCA_MASK_TRANSPOSE_FIRST") == nullptr) {
    Scope sc(h, C_PREPARE);
    int64_t* ca_ptr = h.ca_ptr.as<int64_t>((size_t)m + 1);
    int32_t* ca_idx = h.ca_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
    T* ca_val = h.ca_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
    drop_col = (from_upload || lz_scatter || scatter_sums) ? nullptr : h.drop_col.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
    drop_val = (from_upload || lz_scatter || scatter_sums) ? nullptr : h.drop_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
    k::compact_columns(A, d_o2m, ca_ptr, ca_idx, ca_val, &nnz_used, h.scratch, s, drop_col, drop_val,
                       lz_scatter ? h.lz_scalars.as<unsigned long long>(4) : nullptr);
    h.a_used = {m, n_used, nnz_used, ca_ptr, ca_idx, ca_val};
    compaction_done = true;
    if (!from_upload && !lz_scatter) {
      // the masked-out columns' sums on their own stream, beside the transposition / the bucket route of the kept part and
      // A's format build (sum | sumsq of every column, zero where a column is kept).  Single-rank fits (`side_stats`) keep
      // them in their own arrays and never wait for them on the main stream: see the statistics below.
      if (!h.stream3) {
        SAPCA_HIP(hipStreamCreateWithFlags(&h.stream3, hipStreamNonBlocking));
        SAPCA_HIP(hipEventCreateWithFlags(&h.ev_kept, hipEventDisableTiming));
        SAPCA_HIP(hipEventCreateWithFlags(&h.ev_stats, hipEventDisableTiming));
      }
      if (!h.ev_drop) SAPCA_HIP(hipEventCreateWithFlags(&h.ev_drop, hipEventDisableTiming));
      side_stats = !h.comm.active() && dbg_env("SAPCA_MASK_STATS_INLINE") == nullptr;
      double* d_stats = h.stats.as<double>((size_t)3 * n + 1 + H::kStatsTail);
      SAPCA_HIP(hipMemsetAsync(d_stats + 2 * n, 0, (size_t)n * sizeof(double), s));
      if (!side_stats) {
        SAPCA_HIP(hipEventRecord(h.ev_drop, s));   // (the compaction synchronised: this only orders the side stream after it)
        SAPCA_HIP(hipStreamWaitEvent(h.stream3, h.ev_drop, 0));
        k::sums_by_column(drop_col, drop_val, nnz - nnz_used, n, h.at_ptr.as<int64_t>((size_t)n + 1),
                          h.at_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1)), h.at_val.as<T>((size_t)std::max<int64_t>(nnz, 1)),
                          d_stats, d_stats + n, h.drop_tmp, h.stream3);
        SAPCA_HIP(hipEventRecord(h.ev_drop, h.stream3));
      }
      // side_stats, randomized fits: queued at the end of prepare(), behind the format builds -- they are the ones the first
      // sweep waits for and the sort shares HBM badly with them, while the sweeps leave most of the HBM rate unused.
      // Lanczos fits (HBM-bound steps, no formats): now, beside the transposition.
      if (side_stats && h.opt.method != SAPCA_RANDOM) {
        double* d_drop = h.drop_stats.as<double>((size_t)2 * n);
        SAPCA_HIP(hipEventRecord(h.ev_drop, s));
        SAPCA_HIP(hipStreamWaitEvent(h.stream3, h.ev_drop, 0));
        if (scatter_sums)
          scatter_sums_on_stream3(d_drop);
        else
          k::sums_by_column(drop_col, drop_val, nnz - nnz_used, n, h.at_ptr.as<int64_t>((size_t)n + 1),
                            h.at_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1)), h.at_val.as<T>((size_t)std::max<int64_t>(nnz, 1)),
                            d_drop, d_drop + n, h.drop_tmp, h.stream3);
      }
    }
  }
  if (masked_aside || try_direct) {
    if (!h.stream2) {
      SAPCA_HIP(hipStreamCreateWithFlags(&h.stream2, hipStreamNonBlocking));
      SAPCA_HIP(hipEventCreateWithFlags(&h.ev_fork, hipEventDisableTiming));
      SAPCA_HIP(hipEventCreateWithFlags(&h.ev_join, hipEventDisableTiming));
    }
    SAPCA_HIP(hipEventRecord(h.ev_fork, s));   // A (and the index maps) are on the device
    h.tiled_a = TiledOp();
    a_built_aside = true;
    aside = std::thread([&, n_used, tiled_ldp, compaction_done] {
      try {
        SAPCA_HIP(hipSetDevice(h.device));
        SAPCA_HIP(hipStreamWaitEvent(h.stream2, h.ev_fork, 0));
        CsrView<T> src = A;
        if (masked && compaction_done) {
          src = view(h.a_used);
        } else if (masked) {
          int64_t* ca_ptr = h.ca_ptr.as<int64_t>((size_t)m + 1);
          int32_t* ca_idx = h.ca_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
          T* ca_val = h.ca_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
          k::compact_columns(A, d_o2m, ca_ptr, ca_idx, ca_val, &nnz_used, h.tb_a.tmp, h.stream2);
          h.a_used = {m, n_used, nnz_used, ca_ptr, ca_idx, ca_val};
          src = view(h.a_used);
        }
        if (tiled_ldp != 0) {
          if constexpr (sizeof(T) == 4) ok_a_aside = k::build_tiled(src, false, tiled_ldp, h.tiled_a, h.tb_a, h.stream2);
          else ok_a_aside = k::build_tiled(src, tiled_ldp, h.tiled_a, h.tb_a, h.stream2);
        }
        SAPCA_HIP(hipEventRecord(h.ev_join, h.stream2));
      } catch (...) {
        aside_err = std::current_exception();
      }
    });
  }

  // A^T's tile-major format straight from A (spmm_tiled.hip, "bucket route"): no transposed CSR, no sort; the column
  // statistics come out of the same pass.  Outside its limits (more than 65536 columns, ...) the transposition takes over.
  bool at_direct = false;
  if constexpr (sizeof(T) == 4) {
    if (try_direct) {
      if (dbg_env("SAPCA_PREPARE_ASIDE_FIRST") && aside.joinable()) {   // timing runs: A's format alone on the chip, then A^T's
        aside.join();
        SAPCA_HIP(hipStreamSynchronize(h.stream2));
      }
      Scope sc(h, C_PREPARE);
      int64_t* at_ptr = h.at_ptr.as<int64_t>((size_t)n + 1);
      double* d_stats = h.stats.as<double>((size_t)3 * n + 1 + H::kStatsTail);
      h.tiled_at = TiledOp();
      at_direct = k::build_tiled_at_direct(A, tiled_ldp, h.tiled_at, h.tb_at, at_ptr, from_upload ? nullptr : d_stats, h.scratch, s);
      if (at_direct) { At.rows = n; At.cols = m; At.nnz = nnz; At.ptr = at_ptr; At.idx = nullptr; At.val = nullptr; }
    }
  }

  bool masked_direct = false;
  if constexpr (sizeof(T) == 4) {
    if (compaction_done && try_masked_direct && nnz_used > 0) {
      Scope sc(h, C_PREPARE);
      double* d_stats = h.stats.as<double>((size_t)3 * n + 1 + H::kStatsTail);
      int64_t* cat_ptr = h.cat_ptr.as<int64_t>((size_t)n_used + 1);
      double* d_part = h.scratch2.as<double>((size_t)2 * n_used + 2);   // sums | sums of squares of the kept columns, compact numbering
      h.tiled_at = TiledOp();
      masked_direct = k::build_tiled_at_direct(view(h.a_used), tiled_ldp, h.tiled_at, h.tb_at, cat_ptr, from_upload ? nullptr : d_part, h.scratch, s);
      if (masked_direct) {
        h.at_used = {n_used, m, nnz_used, cat_ptr, nullptr, nullptr};
        if (!from_upload) {
          // sums of every column from the pairs the compaction dropped (zero where a column is kept), then the kept
          // columns' sums from the bucket route on top; the per-column counts are only read by the unmasked projection
          if (!side_stats) SAPCA_HIP(hipStreamWaitEvent(s, h.ev_drop, 0));   // (the dropped pairs' sums, from the side stream)
          k::scatter_pairs(d_part, d_part + n_used, d_sel, n_used, d_stats, d_stats + n, s);
        }
        At.rows = n; At.cols = m; At.nnz = nnz; At.ptr = nullptr; At.idx = nullptr; At.val = nullptr;
      }
    }
  }

  // masked, off the bucket route: the compacted matrix transposed into a CSR (tile-major rows where the staged sweep's format
  // is built from them), the kept columns' sums as its row sums, the masked-out columns' from the dropped pairs
  bool masked_compact = false;
  if (compaction_done && !masked_direct && !lz_scatter) {
    Scope sc(h, C_PREPARE);
    double* d_stats = h.stats.as<double>((size_t)3 * n + 1 + H::kStatsTail);
    int64_t* cat_ptr = h.cat_ptr.as<int64_t>((size_t)n_used + 1);
    int32_t* cat_idx = h.cat_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
    T* cat_val = h.cat_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
    k::transpose_csr(view(h.a_used), cat_ptr, cat_idx, cat_val, h.scratch, s, at_tile_major ? k::tiled_tile_count(m, tiled_ldp_words) : 0,
                     nullptr);
    h.at_used = {n_used, m, nnz_used, cat_ptr, cat_idx, cat_val};
    if (!from_upload) {
      double* d_part = h.scratch2.as<double>((size_t)2 * n_used + 2);
      k::row_sums(view(h.at_used), d_part, d_part + n_used, s);
      if (!side_stats) SAPCA_HIP(hipStreamWaitEvent(s, h.ev_drop, 0));   // (the dropped pairs' sums, from the side stream)
      k::scatter_pairs(d_part, d_part + n_used, d_sel, n_used, d_stats, d_stats + n, s);
    }
    At.rows = n; At.cols = m; At.nnz = nnz; At.ptr = nullptr; At.idx = nullptr; At.val = nullptr;
    masked_compact = true;
  }

  // (f64, or f32 off the bucket route: A's format is built on the side stream from this thread once the transposition is
  // queued; the side stream forks HERE, so the two run side by side on the GPU)
  const bool a_aside_late = !a_built_aside && tiled_ldp != 0 && !masked && !serial;
  if (a_aside_late) {
    if (!h.stream2) {
      SAPCA_HIP(hipStreamCreateWithFlags(&h.stream2, hipStreamNonBlocking));
      SAPCA_HIP(hipEventCreateWithFlags(&h.ev_fork, hipEventDisableTiming));
      SAPCA_HIP(hipEventCreateWithFlags(&h.ev_join, hipEventDisableTiming));
    }
    SAPCA_HIP(hipEventRecord(h.ev_fork, s));               // A is ready on the main stream at this point
    SAPCA_HIP(hipStreamWaitEvent(h.stream2, h.ev_fork, 0));
  }

  if (!at_direct && !masked_direct && !masked_compact && !lz_scatter) {
    Scope sc(h, C_PREPARE);
    int64_t* at_ptr = h.at_ptr.as<int64_t>((size_t)n + 1);
    int32_t* at_idx = h.at_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
    T* at_val = h.at_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
    // rows of A^T grouped by the interleaved tile of the A row they came from: its format fill streams
    // unmasked: the statistics and the format builder read the sort's packed rows directly and the
    // unpack pass into (at_idx, at_val) is skipped (done lazily below if the row kernel has to take over)
    k::transpose_csr(A, at_ptr, at_idx, at_val, h.scratch, s, at_tile_major ? k::tiled_tile_count(m, tiled_ldp_words) : 0,
                     (at_tile_major && !masked && dbg_env("SAPCA_AT_UNPACK") == nullptr) ? &at_packed : nullptr);
    At.rows = n; At.cols = m; At.nnz = nnz; At.ptr = at_ptr; At.idx = at_idx; At.val = at_val;
  }

  if (a_aside_late) {
    h.tiled_a = TiledOp();
    if constexpr (sizeof(T) == 4) ok_a_aside = k::build_tiled(A, false, tiled_ldp, h.tiled_a, h.tb_a, h.stream2);
    else ok_a_aside = k::build_tiled(A, tiled_ldp, h.tiled_a, h.tb_a, h.stream2);
    SAPCA_HIP(hipEventRecord(h.ev_join, h.stream2));       // ...and the main stream waits for it at the end of prepare()
    a_built_aside = true;
  }

  // R1/R2 (csr.rs:259-312, 558-608) as row sums of A^T, plus the per-column stored-entry count.
  double* sums = static_cast<double*>(h.stats_host.ensure(((size_t)2 * n + 1) * sizeof(double)));
  auto column_statistics = [&](bool uploaded) {
    Scope sc(h, C_STATS);
    double* d_stats = h.stats.as<double>((size_t)3 * n + 1 + H::kStatsTail);
    if (uploaded) {
      SAPCA_HIP(hipMemcpyAsync(d_stats, h.up_stats.out.p, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToDevice, s));
    } else if (lz_scatter) {
      // every column of A (masked-out ones included: mean_ is full width, sparse_masked/mod.rs:279-286) straight from A
      const unsigned long long* sc = h.lz_scalars.ptr<unsigned long long>();
      if (!h.comm.active()) {
        // one rank: on the third stream, with the copy to the host behind it; fit() waits for ev_stats at its end
        if (!h.stream3) {
          SAPCA_HIP(hipStreamCreateWithFlags(&h.stream3, hipStreamNonBlocking));
          SAPCA_HIP(hipEventCreateWithFlags(&h.ev_kept, hipEventDisableTiming));
          SAPCA_HIP(hipEventCreateWithFlags(&h.ev_stats, hipEventDisableTiming));
        }
        SAPCA_HIP(hipEventRecord(h.ev_kept, s));   // (A and max |a| are in place on the main stream)
        SAPCA_HIP(hipStreamWaitEvent(h.stream3, h.ev_kept, 0));
        k::colstats_scatter(A, sc, d_stats, d_stats + n, d_stats + 2 * n, h.drop_tmp, h.stream3);
        SAPCA_HIP(hipMemcpyAsync(sums, d_stats, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, h.stream3));
        SAPCA_HIP(hipEventRecord(h.ev_stats, h.stream3));
        sums[(size_t)2 * n] = (double)m;
        h.stats_on_side = true;
        lz_side = true;
        return;
      }
      k::colstats_scatter(A, sc, d_stats, d_stats + n, d_stats + 2 * n, h.scratch, s);
    } else if (at_direct) {
      k::row_lengths_f64(At.ptr, n, d_stats + 2 * n, s);   // (the sums came out of the format build)
    } else if (masked_direct || masked_compact) {
      // (sums in place; the per-column counts are only read by the unmasked projection)
    } else {
      if constexpr (sizeof(T) == 4) {
        // packed tile-major rows: the statistics pass also leaves the A^T builder's per-row tile index behind
        if (at_packed) {
          k::at_stats_index(At.ptr, at_packed, n, m, tiled_ldp, h.tb_at, d_stats, d_stats + n, s);
          at_seg_ready = true;
        }
      }
      if (!at_seg_ready) k::row_sums(At, d_stats, d_stats + n, s);
      k::row_lengths_f64(At.ptr, n, d_stats + 2 * n, s);
    }
    h.stats_on_side = false;
    if (side_stats && !uploaded && (masked_direct || masked_compact)) {
      // single rank, masked: the main stream holds the kept columns' sums (all the sweeps' centring reads); stream3 puts
      // them over the dropped columns' arrays once both are there and copies the lot to the host -- the main stream
      // never waits for the sort of the dropped pairs
      sums[(size_t)2 * n] = (double)m;
      h.stats_on_side = true;   // (the chain itself is queued at the end of prepare())
      return;
    }
    if (h.comm.active()) {
      // the global row count rides along in the statistics' all-reduce
      // ... and so does this rank's vote on where the two-piece A^T sweep cuts the panel (fit_randomized), when its A^T format
      // exists by now (the bucket / gather route builds it in front of the statistics): tail = {rows, ranks that voted, votes}
      double* tail = h.stats_tail;   // (a member: the copy may still be reading it when this function has returned)
      const uint32_t nr = h.comm.nranks;
      const bool ride = vote_rides(h);
      const size_t ntail = 1 + (ride ? (size_t)nr + 1 : 0);
      std::fill(tail, tail + ntail, 0.0);
      tail[0] = (double)m;
      if (ride && h.tiled_at.valid) {
        tail[1] = 1.0;
        tail[2 + h.comm.rank] = (double)piece_vote(h, h.tiled_at.ldp);
      }
      SAPCA_HIP(hipMemcpyAsync(d_stats + 3 * n, tail, ntail * sizeof(double), hipMemcpyHostToDevice, s));
      { Scope cs(h, C_COMM); h.comm.allreduce(d_stats, (uint64_t)3 * n + ntail, 1, s); }
      SAPCA_HIP(hipMemcpyAsync(sums, d_stats, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, s));
      SAPCA_HIP(hipMemcpyAsync(tail, d_stats + 3 * n, ntail * sizeof(double), hipMemcpyDeviceToHost, s));
    } else {
      // one rank: the count is m -- no 8-byte copies in either direction in front of the first sweep (each a ~15 us hole)
      sums[(size_t)2 * n] = (double)m;
      SAPCA_HIP(hipMemcpyAsync(sums, d_stats, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, s));
    }
  };
  column_statistics(from_upload);
  h.stats_cols = n;
  h.vote_ready = false;
  if (h.comm.active()) {   // the global row count is the sum over the ranks: needed on the host now
    SAPCA_HIP(hipStreamSynchronize(s));
    sums[(size_t)2 * n] = h.stats_tail[0];
    if (vote_rides(h) && (uint32_t)std::llround(h.stats_tail[1]) == h.comm.nranks) {
      h.vote_cut = (int64_t)*std::min_element(h.stats_tail + 2, h.stats_tail + 2 + h.comm.nranks);
      h.vote_ready = true;
    }
    h.m_global = (uint64_t)std::llround(sums[(size_t)2 * n]);
    h.stats_pending = false;
    finish_statistics(h);
  } else {
    // one rank: the count is m; the sums reach the host by the time fit() ends (the sweeps centre with means computed on
    // the device), so nothing waits here
    h.m_global = (uint64_t)m;
    h.stats_pending = true;
  }

  // operator seen by the SVD engines: MaskedCSRMatrix::new (sparse_masked/mod.rs:313)
  int64_t nnz_used_t = 0;
  if (lz_scatter) {
    // no transposed operator: a_used is A or its compaction (set above), at_used only carries the shape
    if (!masked) { h.a_used = {m, n, nnz, A.ptr, A.idx, A.val}; nnz_used = nnz; }
    SAPCA_CHECK(!masked || compaction_done, SAPCA_ERR_HIP, "internal: Lanczos scatter route without its compaction");
    h.at_used = {n_used, m, nnz_used, nullptr, nullptr, nullptr};
    nnz_used_t = nnz_used;
  } else if (masked_direct || masked_compact) {
    nnz_used_t = nnz_used;   // (a_used and at_used were set above)
  } else if (masked) {
    Scope sc(h, C_PREPARE);
    if (!masked_aside && !compaction_done) {
      int64_t* ca_ptr = h.ca_ptr.as<int64_t>((size_t)m + 1);
      int32_t* ca_idx = h.ca_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
      T* ca_val = h.ca_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
      k::compact_columns(A, d_o2m, ca_ptr, ca_idx, ca_val, &nnz_used, h.scratch, s);
      h.a_used = {m, n_used, nnz_used, ca_ptr, ca_idx, ca_val};
    }
    int64_t* cat_ptr = h.cat_ptr.as<int64_t>((size_t)n_used + 1);
    int32_t* cat_idx = h.cat_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
    T* cat_val = h.cat_val.as<T>((size_t)std::max<int64_t>(nnz, 1));
    k::select_rows(At, d_sel, n_used, cat_ptr, cat_idx, cat_val, &nnz_used_t, h.scratch, s);
    h.at_used = {n_used, m, nnz_used_t, cat_ptr, cat_idx, cat_val};
  } else {
    h.a_used = {m, n, nnz, A.ptr, A.idx, A.val};
    h.at_used = {n, m, nnz, At.ptr, At.idx, At.val};
  }
  // (the helper thread is joined below, after A^T's format has been enqueued on the main stream)
  auto join_aside = [&] {
    if (aside.joinable()) aside.join();
    if (aside_err) std::rethrow_exception(aside_err);
    if (masked) SAPCA_CHECK(nnz_used == nnz_used_t, SAPCA_ERR_HIP, "internal: mask compaction of A and A^T disagree");
  };

  // tile-major companions for the LDS-staged sweep
  if (!a_built_aside) h.tiled_a = TiledOp();
  if (!at_direct && !masked_direct) h.tiled_at = TiledOp();
  if constexpr (sizeof(T) == 4) {
    if (tiled_ldp != 0 && n_used > 0) {
      Scope sc(h, C_PREPARE);
      if (!a_built_aside) ok_a_aside = k::build_tiled(view(h.a_used), false, tiled_ldp, h.tiled_a, h.tb_a, s);
      if (!from_at) join_aside();   // (this route reads the compacted A)
      bool ok_at = at_direct || masked_direct || (!from_at && k::build_tiled(view(h.a_used), true, tiled_ldp, h.tiled_at, h.tb_at, s)) ||
                   k::build_tiled(view(h.at_used), false, tiled_ldp, h.tiled_at, h.tb_at, s, at_tile_major, at_packed, true, at_seg_ready);
      if (at_packed && !ok_at) {   // someone needs the transposed CSR after all
        k::unpack_transposed(at_packed, nnz, const_cast<int32_t*>(At.idx), reinterpret_cast<float*>(const_cast<T*>(At.val)), s);
        at_packed = nullptr;
        ok_at = k::build_tiled(view(h.at_used), false, tiled_ldp, h.tiled_at, h.tb_at, s, at_tile_major);
      }
      join_aside();
      const bool ok_a = ok_a_aside;
      ok_at = ok_at && ok_a;
      if (h.opt.verbose)
        fprintf(stderr, "sapca: tile-major formats: A %s (nrb %d, nct %d, split %d, %lld entries), A^T %s (nrb %d, nct %d, split %d, %lld entries)\n",
                ok_a ? "ok" : "no", h.tiled_a.nrb, h.tiled_a.nct, h.tiled_a.nsplit, (long long)h.tiled_a.total_entries,
                ok_at ? "ok" : "no", h.tiled_at.nrb, h.tiled_at.nct, h.tiled_at.nsplit, (long long)h.tiled_at.total_entries);
      if (!ok_a || !ok_at) {
        h.tiled_a = TiledOp();
        h.tiled_at = TiledOp();
        if (at_direct || masked_direct) {   // the row kernel reads a transposed CSR, which the bucket route never made
          const CsrView<T> src = view(h.a_used);   // (the compacted matrix on the masked route)
          int64_t* t_ptr = masked_direct ? h.cat_ptr.as<int64_t>((size_t)n_used + 1) : h.at_ptr.as<int64_t>((size_t)n + 1);
          int32_t* t_idx = (masked_direct ? h.cat_idx : h.at_idx).template as<int32_t>((size_t)std::max<int64_t>(nnz, 1));
          T* t_val = (masked_direct ? h.cat_val : h.at_val).template as<T>((size_t)std::max<int64_t>(nnz, 1));
          k::transpose_csr(src, t_ptr, t_idx, t_val, h.scratch, s, 0, nullptr);
          h.at_used = {src.cols, src.rows, src.nnz, t_ptr, t_idx, t_val};
        }
      }
    }
  } else {
    if (tiled_ldp != 0 && n_used > 0) {
      Scope sc(h, C_PREPARE);
      if (!a_built_aside) ok_a_aside = k::build_tiled(view(h.a_used), tiled_ldp, h.tiled_a, h.tb_a, s);
      bool ok_at = k::build_tiled(view(h.at_used), tiled_ldp, h.tiled_at, h.tb_at, s, at_tile_major);
      join_aside();
      const bool ok_a = ok_a_aside;
      ok_at = ok_at && ok_a;
      if (h.opt.verbose)
        fprintf(stderr, "sapca: tile-major formats (f64): A %s (nrb %d, nct %d, split %d, %lld entries), A^T %s (nrb %d, nct %d, split %d, %lld entries)\n",
                ok_a ? "ok" : "no", h.tiled_a.nrb, h.tiled_a.nct, h.tiled_a.nsplit, (long long)h.tiled_a.total_entries,
                ok_at ? "ok" : "no", h.tiled_at.nrb, h.tiled_at.nct, h.tiled_at.nsplit, (long long)h.tiled_at.total_entries);
      if (!ok_a || !ok_at) { h.tiled_a = TiledOp(); h.tiled_at = TiledOp(); }
    }
  }

  join_aside();   // (fits without tile-major formats)
  if (a_built_aside) SAPCA_HIP(hipStreamWaitEvent(s, h.ev_join, 0));

  if (h.stats_on_side && !lz_side) {
    // everything the first sweep needs is queued: now the masked-out columns' sums, the kept columns' sums over them, the
    // copy of all statistics to the host (read at the end of fit())
    double* d_stats = h.stats.ptr<double>();
    double* d_drop = h.drop_stats.as<double>((size_t)2 * n);
    SAPCA_HIP(hipEventRecord(h.ev_kept, s));
    SAPCA_HIP(hipStreamWaitEvent(h.stream3, h.ev_kept, 0));
    if (h.opt.method == SAPCA_RANDOM) {
      if (scatter_sums)
        scatter_sums_on_stream3(d_drop);
      else
        k::sums_by_column(drop_col, drop_val, nnz - nnz_used, n, h.at_ptr.as<int64_t>((size_t)n + 1),
                          h.at_idx.as<int32_t>((size_t)std::max<int64_t>(nnz, 1)), h.at_val.as<T>((size_t)std::max<int64_t>(nnz, 1)),
                          d_drop, d_drop + n, h.drop_tmp, h.stream3);
    }
    k::copy_selected(d_stats, d_stats + n, d_sel, n_used, d_drop, d_drop + n, h.stream3);
    SAPCA_HIP(hipMemcpyAsync(sums, d_drop, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, h.stream3));
    SAPCA_HIP(hipEventRecord(h.ev_stats, h.stream3));
  }

  h.prep_key.ptr = A.ptr; h.prep_key.idx = A.idx; h.prep_key.val = A.val;
  h.prep_key.m = (uint64_t)m; h.prep_key.n = (uint64_t)n; h.prep_key.nnz = (uint64_t)nnz;
  h.prep_key.mask_version = h.mask_version; h.prep_key.dtype = kDtype; h.prep_key.valid = true;
}

// R3: mean and total variance (sparse/mod.rs:106-131; masked :273-311, over cols_to_use only) from the column sums on the host
template <typename T>
void Engine<T>::finish_statistics(H& h) {
  const int64_t n = h.stats_cols;
  const double* sums = static_cast<const double*>(h.stats_host.p);
  const double mg = (double)h.m_global;
  const bool masked = h.has_mask_maps;
  h.prep_mean.assign((size_t)n, 0.0);
  h.prep_total_var = 0;
  if (h.opt.center) {
    for (int64_t j = 0; j < n; ++j) h.prep_mean[(size_t)j] = (double)(T)(sums[(size_t)j] / mg);
    auto var_of = [&](int64_t j) {
      const double mean = sums[(size_t)j] / mg;
      return (sums[(size_t)n + j] - mean * sums[(size_t)j]) / (mg - 1.0);
    };
    if (masked) for (uint64_t j : h.cols_to_use) h.prep_total_var += var_of((int64_t)j);
    else for (int64_t j = 0; j < n; ++j) h.prep_total_var += var_of(j);
  }
  // Q3 through two sweeps (A'W - P diag(mu) W, transform()) subtracts two f32 sums that cancel where stored values sit
  // close to their column's mean, i.e. in a well-filled column of small spread: (sum a)^2 / (m sum a^2) -> 1.  The
  // reference subtracts entry by entry (sparse_masked/mod.rs:488-494) and keeps those digits; above 1/4 the projection
  // takes the row kernel, which does the same.
  h.q3_cancels = false;
  if (h.opt.center && masked)
    for (uint64_t j : h.cols_to_use) {
      const double sj = sums[(size_t)j], qj = sums[(size_t)n + j];
      if (qj > 0 && sj * sj > 0.25 * mg * qj) { h.q3_cancels = true; break; }
    }
  h.stats_pending = false;
}

// ------------------------------------------------------------------------------------------
// R10: PowerIterationNormalizer on a rows x ld panel (CholeskyQR, f64 Gram on MFMA).
// ------------------------------------------------------------------------------------------
template <typename T>
void Engine<T>::normalize(H& h, T* P, int64_t rows, int l, int ld, int normalizer, bool sharded, double* R1, double* R2,
                          int passes_hint, const k::PanelSource<T>* src, const T* w, T* vec_out) {
  // src (nullable): the panel is still as its producer left it (slabs of a split sweep, the centring term not yet
  // subtracted): the first Gram applies that on its way through.  vec_out (nullable): receives sum_r w[r] Q[r][:] of the
  // normalised panel Q = P R^-1 (w null: ones) -- the centring vector of the sweep that reads Q next -- computed as
  // R^-T (P^T w) from sums gathered in the Gram's read pass, not from another pass over Q.
  hipStream_t s = h.stream;
  if (normalizer == SAPCA_NORM_NONE) {
    if (src) k::materialize(P, rows, ld, *src, s);
    if (vec_out) k::weighted_colsum(P, rows, ld, w, vec_out, h.scratch2, s);
    return;
  }
  Scope sc(h, C_ORTHO);
  const SmallLayout lay(ld);
  double* base = h.small.as<double>(lay.doubles());
  double* G = base;
  double* Rinv = base + (size_t)ld * ld;
  double* Rtmp = base + (size_t)2 * ld * ld;
  int* info = reinterpret_cast<int*>(base + lay.info_at());
  double* wsum = vec_out ? base + lay.wsum_at() : nullptr;
  // QR -> CholeskyQR2 (orthonormal to working precision); LU -> one pass: a well-conditioned
  // basis of the same span, which is all the reference's LU normaliser provides.  Between power
  // iterations only the span matters (the next sweep re-mixes the basis), so the intermediate QR
  // normalisations run the single pass too (`passes_hint` = 1); the final range basis Q and the
  // factorisation of B^T always get both passes.
  const int passes = passes_hint > 0 ? passes_hint : (normalizer == SAPCA_NORM_QR ? 2 : 1);
  for (int pass = 0; pass < passes; ++pass) {
    k::gram(P, rows, ld, G, h.scratch2, s, pass == 0 ? src : nullptr, w, wsum);
    if (sharded && h.comm.active()) { Scope cs(h, C_COMM); h.comm.allreduce(G, (uint64_t)ld * ld, 1, s); }
    double* Rout = pass == 0 ? (R1 ? R1 : Rtmp) : (R2 ? R2 : Rtmp);
    k::chol_inv(G, l, ld, Rout, Rinv, info, s, wsum, sizeof(T) == 4 ? reinterpret_cast<float*>(vec_out) : nullptr,
                sizeof(T) == 8 ? reinterpret_cast<double*>(vec_out) : nullptr);
    if (ld <= 128) {
      k::panel_gemm(P, rows, ld, Rinv, ld, P, s, true);   // (R^-1 is upper triangular: its zero blocks are skipped)
    } else {   // wide panels: block by block into a second panel, then back (R^-1 is upper triangular)
      T* Q = h.panel_wide.as<T>((size_t)std::max<int64_t>(rows, 1) * ld);
      k::panel_gemm(P, rows, ld, Rinv, ld, Q, s, true);
      SAPCA_HIP(hipMemcpyAsync(P, Q, (size_t)rows * ld * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
  }
}

template <typename T>
bool Engine<T>::vote_rides(const H& h) {
  return sizeof(T) == 4 && h.comm.active() && h.opt.method == SAPCA_RANDOM && h.opt.spmm_variant != 1 &&
         (size_t)h.comm.nranks + 2 <= H::kStatsTail;
}

template <typename T>
int64_t Engine<T>::piece_vote(H& h, int ld) {
  const char* ov = getenv("SAPCA_AT_OVERLAP");   // (read per fit: the tests switch it)
  const bool enabled = ov != nullptr ? atoi(ov) != 0 : h.comm.mode != Comm::RCCL;
  if (!enabled || !h.comm.has_side_lane() || !k::spmm_tiled_pieces_ok(h.tiled_at, 2, ld)) return 0;
  std::vector<int64_t> b;
  k::spmm_tiled_piece_bounds(h.tiled_at, 2, b, h.stream);
  return b[1];
}

// ------------------------------------------------------------------------------------------
// R7-R11, R13: randomized SVD of the (implicitly centred) prepared operator.
// ------------------------------------------------------------------------------------------
template <typename T>
void Engine<T>::fit_randomized(H& h) {
  hipStream_t s = h.stream;
  const CsrView<T> A = view(h.a_used), At = view(h.at_used);
  const int64_t m = A.rows, n_used = A.cols;
  const int k = (int)h.opt.n_components;
  const int64_t l_req = (int64_t)h.opt.n_components + (int64_t)h.opt.n_oversamples;
  const int l = (int)std::min<int64_t>(l_req, std::min<int64_t>((int64_t)h.m_global, n_used));
  SAPCA_CHECK(l <= k::kMaxPanelWidth, SAPCA_ERR_ARG, "n_components + n_oversamples above 1024 is not supported");
  const bool tiled = h.tiled_a.valid && h.tiled_at.valid;
  // The panel leading dimension enters the element counts of the all-reduces below, so with more than one rank it must
  // not depend on anything a rank decides locally (its own entry count against the staged-sweep floor, whether its
  // format build succeeded): row-sharded fits always use the staged sweep's panel geometry, whichever kernel a rank
  // picks for its shard (the row kernel takes any multiple of 16).
  // (above 128 columns every panel is a multiple of 64 wide: column passes of the sweeps, 64 / 128-column blocks of the dense kernels)
  const int ld = l > 128 ? (int)round_up(l, 64)
                         : (tiled || h.comm.active()) ? std::max(tiled ? h.tiled_a.ldp : 0, l <= 64 ? 64 : 128) : (int)round_up(l, 16);
  const int q = (int)h.opt.n_power_iterations;
  const int norm = h.opt.normalizer;
  const bool center = h.opt.center != 0;
  const int variant = h.opt.spmm_variant;

  T* X = h.panel_x.as<T>(((size_t)std::max<int64_t>(n_used, 1) + 1) * ld);   // + one row: the column sums ride along in the all-reduce
  T* Y = h.panel_y.as<T>((size_t)std::max<int64_t>(m, 1) * ld);
  const SmallLayout lay(ld);
  double* small = h.small.as<double>(lay.doubles());
  double* R1 = small + (size_t)3 * ld * ld;
  double* R2 = small + (size_t)4 * ld * ld;
  double* Mdev = small + (size_t)5 * ld * ld;
  int* info = reinterpret_cast<int*>(small + lay.info_at());
  T* cvec = reinterpret_cast<T*>(small + lay.cvec_at());
  T* svec = cvec + lay.W;
  const T* mu = center ? h.mean_used_dev.ptr<T>() : nullptr;
  SAPCA_HIP(hipMemsetAsync(info, 0, sizeof(int), s));

  // Omega: injected (parity tests) or generated from the seed
  if (!h.omega.empty()) {
    SAPCA_CHECK((int64_t)h.omega_rows == n_used && (int64_t)h.omega_cols >= l, SAPCA_ERR_ARG,
                "injected Omega must be (features seen by the SVD) x (n_components + n_oversamples)");
    std::vector<T> tmp((size_t)n_used * l);
    for (int64_t r = 0; r < n_used; ++r)
      for (int j = 0; j < l; ++j) tmp[(size_t)r * l + j] = (T)h.omega[(size_t)r * h.omega_cols + j];
    T* stage = h.scratch2.as<T>(tmp.size());
    SAPCA_HIP(hipMemcpyAsync(stage, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s));
    k::add_padding(stage, n_used, l, X, ld, s);
    SAPCA_HIP(hipStreamSynchronize(s));  // tmp goes out of scope
  } else {
    k::gaussian_panel(X, n_used, l, ld, h.opt.random_seed, s);
  }

  // Centring vectors.  c = X^T mu (sweep_A) and the column sums 1^T Y (sweep_At) belong to a panel that has just been
  // normalised: normalize() delivers them from sums gathered in its Gram pass (vec_out), so no kernel re-reads the
  // normalised panel for them; only the very first c (of Omega) is summed here.
  bool cvec_ready = false;
  auto sweep_A = [&]() {  // Y = Ac X   (R8)
    if (center && !cvec_ready) k::weighted_colsum(X, n_used, ld, mu, cvec, h.scratch2, s);
    cvec_ready = false;
    Scope sc(h, C_SPMM);
    k::spmm(A, &h.tiled_a, X, ld, Y, ld, ld, center ? cvec : nullptr, variant, h.split_scratch, s);
  };
  std::vector<int64_t> piece_rows;
  bool overlap = false;
  int64_t cut = 0;   // rows [0, cut) of the panel are all-reduced behind the first piece, the rest behind the second
  int piece_wgs = 240;
  h.at_sweep_pieces = 1u;
  if constexpr (sizeof(T) == 4) {
    // On by default wherever the side stream has a lane of its own and the path has run with several ranks: the callback /
    // in-process transports.  Under RCCL (the duplicate communicator made at init: two streams never issue on one
    // communicator) it is opt-in, SAPCA_AT_OVERLAP=1, until it has run on more than one GPU; SAPCA_AT_OVERLAP=0 switches it
    // off everywhere.  Whether a rank takes part is ITS decision, so every rank of a multi-rank fit votes (0 = one piece)
    // and nobody waits in a collective a peer never joins.
    if (h.comm.active() && variant != 1) {
      // The pieces are whole row blocks of this rank's operator, and ranks cut their blocks differently (the block count
      // follows the shard's own tile count): the ranks agree on one row count -- the smallest first piece, 0 if any rank
      // cannot or will not sweep in pieces -- so that every rank's collectives have the same sizes.  The votes came with the
      // column statistics when every rank had its A^T format by then (no extra collective, no host round trip here);
      // otherwise one small all-reduce.
      // (only A^T's format matters for the pieces -- the vote may have been cast before A's own format was known to be built)
      const bool mine = h.tiled_at.valid && k::spmm_tiled_pieces_ok(h.tiled_at, 2, ld);
      if (h.vote_ready) {
        cut = h.vote_cut;
        SAPCA_CHECK(cut == 0 || mine, SAPCA_ERR_COMM, "internal: the ranks agreed on a two-piece A^T sweep this rank cannot run");
      } else {
        const uint32_t nr = h.comm.nranks;
        std::vector<double> votes((size_t)nr, 0.0);
        votes[h.comm.rank] = mine ? (double)piece_vote(h, ld) : 0.0;
        double* d_votes = h.votes.as<double>(nr);
        SAPCA_HIP(hipMemcpyAsync(d_votes, votes.data(), nr * sizeof(double), hipMemcpyHostToDevice, s));
        h.comm.allreduce(d_votes, nr, 1, s);
        SAPCA_HIP(hipMemcpyAsync(votes.data(), d_votes, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        SAPCA_HIP(hipStreamSynchronize(s));
        cut = (int64_t)*std::min_element(votes.begin(), votes.end());
      }
      overlap = mine && cut > 0 && cut < n_used;
      if (overlap) k::spmm_tiled_piece_bounds(h.tiled_at, 2, piece_rows, s);
    }
    h.at_sweep_pieces = overlap ? 2u : 1u;
    if (overlap) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, h.device) == hipSuccess) piece_wgs = std::max(16, pr.multiProcessorCount - 16);
      if (!h.stream_comm) {
        SAPCA_HIP(hipStreamCreateWithFlags(&h.stream_comm, hipStreamNonBlocking));
        SAPCA_HIP(hipEventCreateWithFlags(&h.ev_piece, hipEventDisableTiming));
        SAPCA_HIP(hipEventCreateWithFlags(&h.ev_comm, hipEventDisableTiming));
      }
    }
  }
  // one collective carries the l column sums of this rank's Y too: they live in the row behind the panel then
  T* const sv = h.comm.active() ? X + (size_t)n_used * ld : svec;
  // X = Ac^T Y   (R9); partial products are summed over ranks.  The panel is left as the sweep produced it: `src` says what
  // the next pass over X -- the Gram of its normalisation, or of the small SVD -- still has to apply (slabs of a sweep
  // whose tile range was split over workgroups, the centring term mu (1^T Y)^T).  sv_ready: the normaliser of Y delivered 1^T Y.
  auto sweep_At = [&](bool sv_ready, k::PanelSource<T>& src) {
    src = k::PanelSource<T>();
    if constexpr (sizeof(T) == 4) {
      if (overlap) {
        const int wgs = piece_wgs;
        const int64_t r1 = piece_rows[1];
        {
          Scope sc(h, C_SPMMT);
          k::spmm_tiled_piece(h.tiled_at, 0, 2, wgs, 0, r1, Y, ld, X, ld, ld, h.split_scratch, s);
          SAPCA_HIP(hipEventRecord(h.ev_piece, s));
          k::spmm_tiled_piece(h.tiled_at, 1, 2, wgs, r1, n_used - r1, Y, ld, X, ld, ld, h.split_scratch2, s);
        }
        if (center && !sv_ready) k::weighted_colsum(Y, m, ld, (const T*)nullptr, sv, h.scratch2, s);
        {
          Scope cs(h, C_COMM);   // (device time from the first piece's collective being possible to the last one's end)
          SAPCA_HIP(hipStreamWaitEvent(h.stream_comm, h.ev_piece, 0));
          h.comm.allreduce(X, (uint64_t)cut * ld, kDtype, h.stream_comm, 1);   // (cut <= r1: rows this rank has finished)
          SAPCA_HIP(hipEventRecord(h.ev_comm, h.stream_comm));
          h.comm.allreduce(X + (size_t)cut * ld, (uint64_t)(n_used - cut) * ld + (center ? (uint64_t)ld : 0), kDtype, s);
          SAPCA_HIP(hipStreamWaitEvent(s, h.ev_comm, 0));
        }
        src.parts = X; src.nsplit = 1;
        if (center) { src.mu = mu; src.sv = sv; }
        return;
      }
    }
    {
      Scope sc(h, C_SPMMT);
      // (one rank: the slabs may stay unsummed; several: the collective needs the sum)
      k::spmm(At, &h.tiled_at, Y, ld, X, ld, ld, (const T*)nullptr, variant, h.split_scratch, s, h.comm.active() ? nullptr : &src);
    }
    if (!src.parts) { src.parts = X; src.nsplit = 1; }
    if (center && !sv_ready) k::weighted_colsum(Y, m, ld, (const T*)nullptr, sv, h.scratch2, s);
    if (h.comm.active()) { Scope cs(h, C_COMM); h.comm.allreduce(X, (uint64_t)n_used * ld + (center ? (uint64_t)ld : 0), kDtype, s); }
    if (center) { src.mu = mu; src.sv = sv; }
  };

  k::PanelSource<T> src;
  for (int it = 0; it < q; ++it) {
    sweep_A();
    normalize(h, Y, m, l, ld, norm, true, nullptr, nullptr, 1, nullptr, nullptr, center ? sv : nullptr);
    sweep_At(true, src);
    normalize(h, X, n_used, l, ld, norm, false, nullptr, nullptr, 1, &src, mu, center ? cvec : nullptr);
    cvec_ready = center;
  }
  sweep_A();
  normalize(h, Y, m, l, ld, SAPCA_NORM_QR, true, nullptr, nullptr, 0, nullptr, nullptr, center ? sv : nullptr);  // Q = qr(Y): always orthonormal
  sweep_At(true, src);                                              // X = B^T = Ac^T Q  (n_used x l), completed by the Gram below

  // R11 (f32): SVD of B through the l x l Gram of B^T: G = B B^T = Uh S^2 Uh^T (f64, MFMA + a host
  // eigensolver), vt = (B^T Uh S^-1)^T.  The Gram squares the condition number, so sigma_i is good to
  // eps_f64 (sigma_1/sigma_i)^2 relative: 1e-8 even for a 1e4 decay, below what the f32 panels carry; the
  // f64 instantiation keeps the QR + Jacobi route below.  Saves two CholeskyQR passes over the n x l
  // panel and 0.4 ms of host time per fit.
  int info_host = 0;
  const bool gram_route = sizeof(T) == 4 && dbg_env("SAPCA_SMALL_SVD_QR") == nullptr;
  if (gram_route) {
    Scope sc(h, C_SMALL);
    double* G = small;
    k::gram(X, n_used, ld, G, h.scratch2, s, &src);
    const int ldk = (int)round_up(k, 16);
    if (k::sym_eig_device_ok(l)) {
      // (opt-in experiment, SAPCA_EIG_DEVICE=1: slower than the host path below -- see sym_eig_device_ok)
      // the eigenproblem stays on the device (one workgroup, parallel Jacobi: dense.hip) and writes the factor M itself:
      // no wait for the host anywhere in the small SVD.  The singular values (and the solver's status) cross in page-locked
      // memory behind everything else; finish_fit() reads them after the fit's last wait.
      double* d_sigma = small + (size_t)ld * ld;                  // (the normaliser's R^-1 slot: free here)
      int* d_status = reinterpret_cast<int*>(d_sigma + ld);
      k::sym_eig_device(G, l, ld, k, ldk, Mdev, d_sigma, d_status, s);
      double* host = static_cast<double*>(h.small_host.ensure(((size_t)l + 4) * sizeof(double)));
      int* host_i = reinterpret_cast<int*>(host + l);
      SAPCA_HIP(hipMemcpyAsync(host, d_sigma, (size_t)l * sizeof(double), hipMemcpyDeviceToHost, s));
      SAPCA_HIP(hipMemcpyAsync(host_i, d_status, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
      SAPCA_HIP(hipMemcpyAsync(host_i + 2, info, sizeof(int), hipMemcpyDeviceToHost, s));
      T* VtT = h.panel_w.as<T>((size_t)std::max<int64_t>(n_used, 1) * ldk);
      k::panel_gemm(X, n_used, ld, Mdev, ldk, VtT, s);
      T* comps = h.components_dev.as<T>((size_t)k * std::max<int64_t>(n_used, 1));
      k::flip_transpose(VtT, n_used, ldk, k, comps, h.scratch2, s);  // R13
      h.sing_pending = l;
      return;
    }
    // page-locked staging owned by the handle: G comes back and M goes out without the runtime's bounce buffers, and M
    // outlives this call -- nothing here waits for the device after the eigensolver
    double* g = static_cast<double*>(h.small_host.ensure(((size_t)ld * ld + (size_t)ld * ldk + 2) * sizeof(double)));
    int* info_pinned = reinterpret_cast<int*>(g + (size_t)ld * ld + (size_t)ld * ldk);
    SAPCA_HIP(hipMemcpyAsync(g, G, (size_t)ld * ld * sizeof(double), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipMemcpyAsync(info_pinned, info, sizeof(int), hipMemcpyDeviceToHost, s));
    h.small_l = l;
    h.small_ld = ld;
    if (h.defer_small) {
      // fit_transform: the host eigensolver runs while the GPU sweeps the projection with the un-rotated panel (transform():
      // T = [Ac diag(cnt) B^T] M), instead of in front of an idle GPU
      if (!h.ev_small) SAPCA_HIP(hipEventCreateWithFlags(&h.ev_small, hipEventDisableTiming));
      SAPCA_HIP(hipEventRecord(h.ev_small, s));
      h.small_pending = true;
      return;
    }
    finish_small_svd(h, nullptr);
    return;
  }

  // R11 (f64): SVD of B through a QR of B^T: B^T = Qz Rz, Rz = Ur S Vr^T  =>  vt = (Qz Ur)^T.
  std::vector<double> r1((size_t)ld * ld), r2((size_t)ld * ld);
  {
    Scope sc(h, C_SMALL);
    normalize(h, X, n_used, l, ld, SAPCA_NORM_QR, false, R1, R2, 0, &src);
    SAPCA_HIP(hipMemcpyAsync(r1.data(), R1, r1.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipMemcpyAsync(r2.data(), R2, r2.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipMemcpyAsync(&info_host, info, sizeof(int), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipStreamSynchronize(s));
    std::vector<double> Rz((size_t)l * l, 0.0), Ur, sv;
    for (int i = 0; i < l; ++i)
      for (int j = i; j < l; ++j) {
        double acc = 0;
        for (int t = i; t <= j; ++t) acc += r2[(size_t)i * ld + t] * r1[(size_t)t * ld + j];
        Rz[(size_t)i * l + j] = acc;
      }
    const auto tj0 = std::chrono::steady_clock::now();
    jacobi_svd(Rz, l, Ur, sv);
    if (h.opt.verbose)
      fprintf(stderr, "sapca: host Jacobi SVD of the %d x %d factor: %.3f ms\n", l, l,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count());
    for (int i = 0; i < l; ++i)
      SAPCA_CHECK(std::isfinite(sv[i]), SAPCA_ERR_SVD, "Randomized SVD computation failed: non-finite singular value");
    const int ldk = (int)round_up(k, 16);
    std::vector<double> M((size_t)ld * ldk, 0.0);
    for (int i = 0; i < l; ++i)
      for (int j = 0; j < k; ++j) M[(size_t)i * ldk + j] = Ur[(size_t)i * l + j];
    SAPCA_HIP(hipMemcpyAsync(Mdev, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice, s));
    T* VtT = h.panel_w.as<T>((size_t)std::max<int64_t>(n_used, 1) * ldk);
    k::panel_gemm(X, n_used, ld, Mdev, ldk, VtT, s);
    T* comps = h.components_dev.as<T>((size_t)k * std::max<int64_t>(n_used, 1));
    k::flip_transpose(VtT, n_used, ldk, k, comps, h.scratch2, s);  // R13
    SAPCA_HIP(hipStreamSynchronize(s));                            // M goes out of scope
    h.sing.assign(sv.begin(), sv.begin() + k);
  }
  h.chol_regularised = info_host;
}

// R11 (f32), host half: eigen-decomposition of the l x l Gram staged in small_host by fit_randomized, the factor
// M = Uh S^-1 back to the device, vt = (B^T M)^T, svd_flip (R13).  Called at the end of fit_randomized, or -- fit_transform,
// `small_pending` -- from transform() once the projection sweep is queued.  sign_out: the device address of the k flip signs.
template <typename T>
void Engine<T>::finish_small_svd(H& h, const double** sign_out) {
  hipStream_t s = h.stream;
  const bool deferred = h.small_pending;
  h.small_pending = false;
  const int l = h.small_l, ld = h.small_ld, k = (int)h.opt.n_components;
  const int64_t n_used = h.a_used.cols;
  const int ldk = (int)round_up(k, 16);
  const SmallLayout lay(ld);
  double* small = h.small.as<double>(lay.doubles());
  double* Mdev = small + (size_t)5 * ld * ld;
  T* X = h.panel_x.ptr<T>();
  try {
    std::unique_ptr<Scope> sc(deferred ? new Scope(h, C_SMALL) : nullptr);   // (not deferred: inside fit_randomized's own span)
    if (sc && sc->ev >= 0) h.small_in_transform.push_back(sc->ev);             // (it lies inside the projection's span: taken out of transform_ms)
    double* g = static_cast<double*>(h.small_host.p);
    double* M = g + (size_t)ld * ld;
    const int* info_pinned = reinterpret_cast<const int*>(M + (size_t)ld * ldk);
    if (deferred) SAPCA_HIP(hipEventSynchronize(h.ev_small));
    else SAPCA_HIP(hipStreamSynchronize(s));
    const int info_host = *info_pinned;
    std::vector<double> Gl((size_t)l * l), w, Vt;
    for (int i = 0; i < l; ++i)
      for (int j = 0; j < l; ++j) Gl[(size_t)i * l + j] = g[(size_t)i * ld + j];
    const auto tj0 = std::chrono::steady_clock::now();
    SAPCA_CHECK(sym_eigh_desc(Gl, l, w, Vt), SAPCA_ERR_SVD, "Randomized SVD computation failed: eigensolver did not converge");
    if (h.opt.verbose)
      fprintf(stderr, "sapca: host eigensolver of the %d x %d Gram: %.3f ms\n", l, l,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count());
    std::vector<double> sv((size_t)l);
    for (int i = 0; i < l; ++i) {
      SAPCA_CHECK(std::isfinite(w[i]), SAPCA_ERR_SVD, "Randomized SVD computation failed: non-finite singular value");
      sv[i] = std::sqrt(std::max(w[i], 0.0));
    }
    std::fill(M, M + (size_t)ld * ldk, 0.0);
    const double tiny = sv[0] * 1e-12;
    for (int j = 0; j < k; ++j) {
      if (!(sv[j] > tiny)) continue;   // numerically rank-deficient direction: a zero component, sigma ~ 0
      const double inv = 1.0 / sv[j];
      for (int i = 0; i < l; ++i) M[(size_t)i * ldk + j] = Vt[(size_t)j * l + i] * inv;
    }
    SAPCA_HIP(hipMemcpyAsync(Mdev, M, (size_t)ld * ldk * sizeof(double), hipMemcpyHostToDevice, s));
    T* VtT = h.panel_w.as<T>((size_t)std::max<int64_t>(n_used, 1) * ldk);
    k::panel_gemm(X, n_used, ld, Mdev, ldk, VtT, s);
    T* comps = h.components_dev.as<T>((size_t)k * std::max<int64_t>(n_used, 1));
    k::flip_transpose(VtT, n_used, ldk, k, comps, h.scratch2, s, sign_out);  // R13
    h.sing.assign(sv.begin(), sv.begin() + k);
    h.chol_regularised = info_host;
  } catch (...) {
    if (deferred) {   // the fit had been reported as done: take that back
      h.fitted = false;
      h.finish_pending = false;
    }
    throw;
  }
}

// ------------------------------------------------------------------------------------------
// R12: Lanczos on the raw operator (no centring: quirk Q1).
// ------------------------------------------------------------------------------------------
template <typename T>
void Engine<T>::fit_lanczos(H& h) {
  Scope sc(h, C_LANCZOS);
  lanczos_fit<T>(h);
}

// ------------------------------------------------------------------------------------------
// fit
// ------------------------------------------------------------------------------------------
template <typename T>
void Engine<T>::fit(H& h, const CsrView<T>& A, bool defer_finish) {
  hipStream_t s = h.stream;
  h.finish_pending = false;
  h.sing_pending = 0;
  h.small_pending = false;
  // fit_transform of an unmasked f32 randomized fit: the small SVD's host half is held back until transform() has queued
  // the projection sweep (masked fits finish first, see below)
  // ... where that pays: the rotation is one more m x l by l x k panel product (0.2 ms per million rows at l <= 64, four
  // times that at 128 columns) against a host stall of 0.25 ms (l = 60) to 1 ms (l = 110) -- C2 and the shards of a
  // strong-scaled fit gain 0.2 ms of a 9.5 ms step, a million rows on one GPU would lose as much (measured, round 4).
  {
    const double lw = (double)(h.opt.n_components + h.opt.n_oversamples) <= 64 ? 64.0 : 128.0;
    // (decided from what every rank of a sharded fit sees alike -- the ranks' shards of one projection take one route;
    // m_global is not known yet, the shard is about 1 / nranks of it)
    h.defer_small = defer_finish && h.mask.empty() && h.opt.method == SAPCA_RANDOM && sizeof(T) == 4 &&
                    (h.comm.active() || (double)A.rows * lw * lw <= 400e3 * 64.0 * 64.0);
  }
  h.spans.clear();
  h.small_in_transform.clear();
  h.comm.host_ms = 0;
  h.timer.begin_collect(s, h.opt.collect_timings != 0);
  const int total_ev = h.timer.start();
  h.fitted = false;
  h.timings.lanczos_steps = 0;
  SAPCA_CHECK(h.opt.n_components > 0, SAPCA_ERR_ARG, "n_components must be positive");
  SAPCA_CHECK(A.rows > 0 && A.cols > 0, SAPCA_ERR_ARG, "empty matrix");
  prepare(h, A);
  const int64_t n_used = h.a_used.cols;
  if (n_used == 0) throw Error(SAPCA_ERR_SVD, "SVD computation failed: the mask selects no feature");
  if ((int64_t)h.opt.n_components > std::min<int64_t>((int64_t)h.m_global, n_used))
    throw Error(SAPCA_ERR_SVD, std::string(h.opt.method == SAPCA_RANDOM ? "Randomized SVD" : "SVD") +
                                   " computation failed: n_components exceeds the matrix dimensions");
  SAPCA_CHECK(h.m_global >= 2, SAPCA_ERR_ARG, "need at least two samples");

  // column means of the features the SVD sees, in T (the centring vector of the sweeps and of transform)
  auto device_means = [&] {
    T* d_mu = h.mean_used_dev.as<T>((size_t)n_used);
    if (h.opt.center)
      k::mean_from_sums(h.stats.ptr<double>(), (double)h.m_global, h.has_mask_maps ? h.sel_rows_dev.ptr<int32_t>() : nullptr, n_used, d_mu, s);
    else
      SAPCA_HIP(hipMemsetAsync(d_mu, 0, (size_t)n_used * sizeof(T), s));
  };

  if (h.opt.method == SAPCA_RANDOM) {
    device_means();
    fit_randomized(h);
  } else {
    // the Lanczos branch never centres (Q1): only transform reads the means.  On the scatter route the column sums arrive
    // from the third stream, beside the iterations: the main stream meets them here, behind the last step
    const bool sums_on_side = h.lz_scatter && h.stats_on_side;
    if (!sums_on_side) device_means();
    fit_lanczos(h);
    if (sums_on_side) {
      SAPCA_HIP(hipStreamWaitEvent(s, h.ev_stats, 0));
      device_means();
    }
  }

  h.k = h.opt.n_components;
  h.n_used = (uint64_t)n_used;
  h.n_cols = (uint64_t)A.cols;
  h.m_fit = h.m_global;
  h.dtype = kDtype;
  h.fitted = true;
  h.timer.stop(total_ev);
  h.fit_total_ev = total_ev;
  h.finish_pending = true;
  // fit_transform: the host-side tail (statistics, timings: two waits for the device) runs once the projection is queued --
  // the model the projection reads is all on the device by now
  // (masked fits finish first: their projection asks the finished statistics whether its two sweeps would cancel, Q3)
  if (!defer_finish || !h.mask.empty()) finish_fit(h);
}

template <typename T>
void Engine<T>::finish_fit(H& h) {
  if (!h.finish_pending) return;
  if (h.small_pending) finish_small_svd(h, nullptr);   // (a projection that failed before it reached the held-back half)
  h.finish_pending = false;
  hipStream_t s = h.stream;
  const int total_ev = h.fit_total_ev;
  const int64_t n_used = (int64_t)h.n_used;
  if (h.sing_pending) {   // the device eigensolver's singular values and status (fit_randomized)
    const int l = h.sing_pending;
    h.sing_pending = 0;
    SAPCA_HIP(hipStreamSynchronize(s));
    const double* host = static_cast<const double*>(h.small_host.p);
    const int* host_i = reinterpret_cast<const int*>(host + l);
    bool finite = true;
    for (int i = 0; i < l; ++i) finite = finite && std::isfinite(host[i]);
    if ((host_i[0] & 1) || (host_i[0] & 2) || !finite) {
      h.fitted = false;
      throw Error(SAPCA_ERR_SVD, (host_i[0] & 1) ? "Randomized SVD computation failed: eigensolver did not converge"
                                                 : "Randomized SVD computation failed: non-finite singular value");
    }
    if (h.opt.verbose) fprintf(stderr, "sapca: device eigensolver of the %d x %d Gram: %d sweeps\n", l, l, host_i[1]);
    h.sing.assign(host, host + h.k);
    h.chol_regularised = host_i[2];
  }
  // sparse/mod.rs:106-117: mean_ = col_sums / n when centring, zeros otherwise (the reference
  // allocates zeros(n_samples) there -- a length bug that is never read; n_cols zeros here).
  if (h.stats_pending) {   // (the copy was queued in prepare(); every path through the SVD engines has synchronised since)
    SAPCA_HIP(hipStreamSynchronize(s));
    if (h.stats_on_side) SAPCA_HIP(hipEventSynchronize(h.ev_stats));   // (masked fits: the copy came from the third stream)
    finish_statistics(h);
  }
  h.mean = h.prep_mean;
  const double nm1 = (double)(h.m_fit - 1);
  h.expl_var.resize(h.k);
  for (uint64_t i = 0; i < h.k; ++i) {  // sparse/mod.rs:210-216
    const T sv = (T)h.sing[i];
    h.expl_var[i] = (double)(T)((T)(sv * sv) / (T)nm1);
  }
  if (h.opt.center) {
    h.total_var = h.prep_total_var;
  } else {  // sparse/mod.rs:218-223
    h.total_var = 0;
    for (uint64_t i = 0; i < h.k; ++i) h.total_var += h.expl_var[i];
  }
  SAPCA_HIP(hipStreamSynchronize(s));
  collect_timings(h, true);
  if (total_ev >= 0) h.timings.fit_total_ms = h.timer.ms(total_ev);
  {
    // ALGORITHMIC bytes of one sparse x dense sweep
    const double l = (double)std::min<uint64_t>(h.opt.n_components + h.opt.n_oversamples,
                                                std::min<uint64_t>(h.m_global, (uint64_t)n_used));
    h.timings.bytes_per_sweep = (double)h.a_used.nnz * (sizeof(T) + 4) + ((double)h.a_used.rows + 1) * 8 +
                                (double)n_used * l * sizeof(T) + (double)h.a_used.rows * l * sizeof(T);
    const bool tiled = h.opt.method == SAPCA_RANDOM && h.tiled_a.valid && h.tiled_at.valid;
    h.timings.sweep_kernel = !tiled ? 0u : (k::dq_usable(h.tiled_a, 64) && k::dq_usable(h.tiled_at, 64) && h.opt.spmm_variant != 1 ? 2u : 1u);
    h.timings.at_sweep_pieces = h.opt.method == SAPCA_RANDOM ? h.at_sweep_pieces : 0u;
    h.timings.sweep_slots_a = tiled ? (uint64_t)h.tiled_a.total_entries : 0;
    h.timings.sweep_slots_at = tiled ? (uint64_t)h.tiled_at.total_entries : 0;
  }
}

// ------------------------------------------------------------------------------------------
// transform
// ------------------------------------------------------------------------------------------
template <typename T>
void Engine<T>::transform(H& h, const CsrView<T>& A, T* d_out) {
  hipStream_t s = h.stream;
  const bool masked = !h.mask.empty();
  if (masked && (int64_t)h.mask.size() != A.cols)  // sparse_masked/mod.rs:440-444
    throw Error(SAPCA_ERR_MASK_LEN, "The mask vector length and the number of features (columns) have to be the same!");
  if (!h.fitted) throw Error(SAPCA_ERR_NOT_FITTED, "Must be fitted before transform!");  // sparse/mod.rs:259,263
  SAPCA_CHECK(h.dtype == kDtype, SAPCA_ERR_ARG, "transform dtype differs from the fitted model's");
  SAPCA_CHECK((uint64_t)A.cols == h.n_cols, SAPCA_ERR_ARG, "transform: column count differs from the fitted matrix");
  SAPCA_CHECK(masked == h.has_mask_maps, SAPCA_ERR_ARG, "transform: mask changed since fit");
  // keep the fit's spans (their events stay valid); drop those of an earlier transform
  h.spans.erase(std::remove_if(h.spans.begin(), h.spans.end(), [](const std::pair<int, int>& p) { return p.first == C_TRANSFORM; }),
                h.spans.end());
  if (h.spans.empty()) h.timer.begin_collect(s, h.opt.collect_timings != 0);
  const int64_t m = A.rows, n = A.cols, n_used = (int64_t)h.n_used;
  const int k = (int)h.k;
  int ldk = (int)round_up(k, 16);
  const bool center = h.opt.center != 0;
  const bool ref_sem = h.opt.transform_semantics == SAPCA_TRANSFORM_REFERENCE;
  if (m == 0) {   // (a rank with an empty shard still completes a fit whose tail was held back)
    finish_fit(h);
    return;
  }
  {
    Scope sc(h, C_TRANSFORM);
    H::PrepKey key;
    key.ptr = A.ptr; key.idx = A.idx; key.val = A.val; key.m = (uint64_t)m; key.n = (uint64_t)n;
    key.nnz = (uint64_t)A.nnz; key.mask_version = h.mask_version; key.dtype = kDtype; key.valid = true;
    const bool prepared = h.prep_key == key;
    // the fitted matrix's tile-major format serves the projection sweep too (one row block per workgroup)
    // (operators with few row blocks -- a shard of a strong-scaled fit -- split their tile range over workgroups: the sweep sums
    //  the slabs itself; only the masked Q3 projection insists on an unsplit operator and checks that on its own)
    const TiledOp* top = (prepared && h.tiled_a.valid && k <= k::kMaxPanelWidth) ? &h.tiled_a : nullptr;
    if (top) ldk = k > 128 ? (int)round_up(k, 64) : std::max(top->ldp, k <= 64 ? 64 : 128);
    CsrView<T> Au;
    double* d_cnt = nullptr;
    if (prepared) {
      Au = view(h.a_used);
      d_cnt = h.stats.ptr<double>() + 2 * n;
    } else if (masked) {
      h.prep_key.valid = false;  // the compaction buffers are about to be reused
      std::vector<int32_t> o2m32((size_t)n);
      for (int64_t j = 0; j < n; ++j) o2m32[(size_t)j] = (int32_t)h.orig_to_masked[(size_t)j];
      int32_t* d_o2m = h.o2m_dev.as<int32_t>((size_t)n);
      SAPCA_HIP(hipMemcpyAsync(d_o2m, o2m32.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
      int64_t* ca_ptr = h.ca_ptr.as<int64_t>((size_t)m + 1);
      int32_t* ca_idx = h.ca_idx.as<int32_t>((size_t)std::max<int64_t>(A.nnz, 1));
      T* ca_val = h.ca_val.as<T>((size_t)std::max<int64_t>(A.nnz, 1));
      int64_t nnz_used = 0;
      k::compact_columns(A, d_o2m, ca_ptr, ca_idx, ca_val, &nnz_used, h.scratch, s);
      Au.rows = m; Au.cols = n_used; Au.nnz = nnz_used; Au.ptr = ca_ptr; Au.idx = ca_idx; Au.val = ca_val;
    } else {
      Au = A;
      if (ref_sem) {
        h.prep_key.valid = false;
        d_cnt = h.stats.as<double>((size_t)3 * n + 1 + H::kStatsTail) + 2 * n;
        k::column_counts_f64(A.idx, A.nnz, n, d_cnt, h.scratch, s);
        if (h.comm.active()) h.comm.allreduce(d_cnt, (uint64_t)n, 1, s);
      }
    }
    // A matrix the handle holds no preparation of (a separate transform of new rows, or of caller-owned arrays): above the
    // staged sweep's break-even its tile-major format is built for this one sweep -- 0.9 ms + a 0.55 ms sweep at C2's size
    // against 5 ms through the row kernel.  (The masked Q3 projection checks on its own whether it can use it.)
    if (!top && !prepared && k <= k::kMaxPanelWidth && Au.nnz > 0) {
      const int ldp_t = staged_sweep_ldp<T>(Au.rows, Au.cols, (double)Au.nnz, k, h.opt.spmm_variant);
      if (ldp_t != 0) {
        h.prep_key.valid = false;   // (tiled_a no longer belongs to the fitted matrix)
        h.tiled_at = TiledOp();
        h.tiled_a = TiledOp();
        bool ok;
        if constexpr (sizeof(T) == 4) ok = k::build_tiled(Au, false, ldp_t, h.tiled_a, h.tb_a, s);
        else ok = k::build_tiled(Au, ldp_t, h.tiled_a, h.tb_a, s);
        if (ok && h.tiled_a.valid) {
          top = &h.tiled_a;
          ldk = k > 128 ? (int)round_up(k, 64) : std::max(top->ldp, k <= 64 ? 64 : 128);
        } else {
          h.tiled_a = TiledOp();
        }
      }
    }
    const T* mu = h.mean_used_dev.ptr<T>();
    // fit_transform, f32 randomized: the fit stopped in front of the host eigensolver.  The projection is linear in the
    // components: with vt^T = B^T M (M = Uh S^-1 diag(sign), l x k) the reference's t = Ac diag(cnt) vt^T (Q2; cnt = 1 for the
    // centred semantics) is [Ac diag(cnt) B^T] M -- the sweep runs on the un-rotated n x l panel while the host solves the
    // l x l eigenproblem, and one m x l by l x k panel product rotates its result.  (B^T = panel_x; Q = panel_y is free.)
    auto project_unrotated = [&] {
      const int ld = h.small_ld;
      const SmallLayout lay(ld);
      double* small = h.small.as<double>(lay.doubles());
      T* cvec = reinterpret_cast<T*>(small + lay.cvec_at());
      double* Mdev = small + (size_t)5 * ld * ld;
      const T* X = h.panel_x.ptr<T>();
      T* Xs = h.panel_xs.as<T>((size_t)n_used * ld);
      T* Tp = h.panel_y.as<T>((size_t)m * ld);
      const TiledOp* topl = h.tiled_a.valid ? &h.tiled_a : nullptr;
      k::scale_rows(X, n_used, ld, ref_sem ? d_cnt : nullptr, Xs, s);
      if (center) k::weighted_colsum(Xs, n_used, ld, mu, cvec, h.scratch2, s);
      k::spmm(Au, topl, Xs, ld, Tp, ld, ld, center ? cvec : nullptr, h.opt.spmm_variant, h.split_scratch, s);
      const double* sign = nullptr;
      finish_small_svd(h, &sign);   // (waits for the Gram's copy only; the sweep above is running)
      const int ldm = (int)round_up(k, 16);
      k::scale_columns(Mdev, ld, ldm, k, sign, s);
      k::panel_gemm(Tp, m, ld, Mdev, ldm, d_out, s, false, k, k);
    };
    // the projection with the fitted components (a separate transform, masked fits, f64, Lanczos, large m)
    auto project_with_components = [&] {
      const SmallLayout lay(ldk);
      double* small = h.small.as<double>(lay.doubles());
      T* cvec = reinterpret_cast<T*>(small + lay.cvec_at());
      const T* comps = h.components_dev.ptr<T>();
      T* W = h.panel_w.as<T>((size_t)n_used * ldk);
      if (ref_sem && !masked) {
        // Q2 (sparse/mod.rs:268-282): t_ik = sum_j cnt_j (x_ij - [center] mu_j) V_kj
        k::scaled_transpose(comps, n_used, k, d_cnt, W, ldk, s);
        if (center) k::weighted_colsum(W, n_used, ldk, mu, cvec, h.scratch2, s);
        k::spmm(Au, top, W, ldk, d_out, k, k, center ? cvec : nullptr, h.opt.spmm_variant, h.split_scratch, s);
      } else if (ref_sem && masked) {
        // Q3 (sparse_masked/mod.rs:488-529): mean subtracted at stored, kept entries only
        k::scaled_transpose(comps, n_used, k, (const double*)nullptr, W, ldk, s);
        bool done = false;
        if constexpr (sizeof(T) == 4) {
          // the fitted matrix's tile-major format: A'W - P diag(mu) W as two sweeps (spmm_dq.hip)
          // (only for the fitted matrix: `q3_cancels` was decided from ITS column statistics; another matrix takes the row
          // kernel, which subtracts entry by entry like the reference)
          if (center && top && prepared && h.opt.spmm_variant != 1 && !h.q3_cancels && dbg_env("SAPCA_Q3_ROWKERNEL") == nullptr) {
            float* W2 = h.scratch2.as<float>((size_t)n_used * ldk);
            float* tmp = h.panel_y.as<float>((size_t)m * std::max(k, 1));
            done = k::q3_projection_dq(Au, *top, W, ldk, mu, W2, tmp, d_out, k, s);
          }
        }
        if (!done) {
          if (center) k::spmm_rows_shifted(Au, W, ldk, d_out, k, k, mu, s);   // (the row kernel subtracts mu_j entry by entry)
          else k::spmm(Au, top, W, ldk, d_out, k, k, (const T*)nullptr, h.opt.spmm_variant, h.split_scratch, s);
        }
      } else {
        // opt-in: the mathematically centred projection (A - 1 mu^T) V^T
        k::scaled_transpose(comps, n_used, k, (const double*)nullptr, W, ldk, s);
        if (center) k::weighted_colsum(W, n_used, ldk, mu, cvec, h.scratch2, s);
        k::spmm(Au, top, W, ldk, d_out, k, k, center ? cvec : nullptr, h.opt.spmm_variant, h.split_scratch, s);
      }
    };
    if (h.small_pending && prepared && !masked) {
      project_unrotated();
    } else {
      if (h.small_pending) finish_small_svd(h, nullptr);   // (a held-back small SVD whose projection cannot take the un-rotated route)
      project_with_components();
    }
  }
  SAPCA_HIP(hipStreamSynchronize(s));
  finish_fit(h);   // (fit_transform: the fit's host-side tail, held back until the projection was queued)
  collect_timings(h, false);
}

template struct Engine<float>;
template struct Engine<double>;

}  // namespace sapca
