// The handle behind the C ABI and the typed engine that drives the kernels.
#pragma once
#include <string>
#include <vector>

#include "comm.h"
#include "common.h"
#include "kernels.h"

struct sapca_handle_s {
  sapca_options opt{};
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t stream2 = nullptr;        // side stream: A's format is built beside the transposition (prepare)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_drop = nullptr;
  // third stream: the sums of the columns a mask drops (a sort of the dropped pairs).  Only mean_ reads them, at the end of
  // fit(): single-rank fits let that chain run beside the format builds AND the sweeps, and it copies the statistics to
  // the host itself (ev_kept: the kept columns' sums are in place on the main stream; ev_stats: the host copy has landed)
  hipStream_t stream3 = nullptr;
  hipStream_t stream_comm = nullptr;   // the first piece's all-reduce of an A^T sweep swept in two pieces (multi-rank fits)
  hipEvent_t ev_piece = nullptr, ev_comm = nullptr;
  hipEvent_t ev_kept = nullptr, ev_stats = nullptr;
  bool stats_on_side = false;

  // builder state
  std::vector<uint8_t> mask;
  uint64_t mask_version = 0;
  std::vector<double> omega;  // injected test matrix (rows x cols, row-major)
  size_t omega_rows = 0, omega_cols = 0;

  // fitted state
  bool fitted = false;
  int dtype = 0;  // 0 = f32, 1 = f64
  uint64_t k = 0, n_used = 0, n_cols = 0, m_fit = 0;
  std::vector<double> sing, expl_var, mean;  // k, k, n_cols
  double total_var = 0;
  std::vector<uint64_t> cols_to_use;
  std::vector<int64_t> orig_to_masked;
  bool has_mask_maps = false;
  int chol_regularised = 0;

  // prepared-operator cache key (the matrix the device-side companions were built from)
  struct PrepKey {
    const void *ptr = nullptr, *idx = nullptr, *val = nullptr;
    uint64_t m = 0, n = 0, nnz = 0, mask_version = 0;
    int dtype = -1;
    bool valid = false;
    bool operator==(const PrepKey& o) const {
      return valid && o.valid && ptr == o.ptr && idx == o.idx && val == o.val && m == o.m && n == o.n && nnz == o.nnz &&
             mask_version == o.mask_version && dtype == o.dtype;
    }
  } prep_key;
  struct RawCsr {
    int64_t rows = 0, cols = 0, nnz = 0;
    const void *ptr = nullptr, *idx = nullptr, *val = nullptr;
  } a_used, at_used;
  uint64_t m_global = 0;
  std::vector<double> prep_mean;  // column means of the prepared matrix (n)
  double prep_total_var = 0;
  uint32_t at_sweep_pieces = 1;   // 2: the last randomized fit swept A^T in two pieces (multi-rank overlap)
  bool lz_scatter = false;        // the prepared Lanczos fit has no transposed operator: its second product scatters into LDS (scatter.hip)
  bool q3_cancels = false;        // a kept column is well filled and of small spread: the masked projection subtracts entry by entry
  // the column sums on the host (sum | sumsq | row count), copied asynchronously: single-rank fits read them at the end
  // of fit() instead of stalling the stream between the preparation and the first sweep
  sapca::PinnedBuf stats_host;
  sapca::PinnedBuf small_host;    // the l x l Gram on its way to the host eigensolver, its factor on the way back
  int fit_total_ev = -1;
  bool finish_pending = false;    // fit() returned with its host-side tail still to run (fit_transform)
  // fit_transform (unmasked f32 randomized fits): fit() returns in front of the host eigensolver of the l x l Gram and
  // transform() runs it once the projection sweep is queued (engine.cpp, finish_small_svd)
  bool defer_small = false, small_pending = false;
  int small_l = 0, small_ld = 0;
  hipEvent_t ev_small = nullptr;
  int sing_pending = 0;           // l: the singular values of the device eigensolver are still on their way (read in finish_fit)
  sapca::PinnedBuf lanczos_host;  // alpha | beta of the Lanczos tridiagonal, read back at each convergence check
  bool stats_pending = false;
  int64_t stats_cols = 0;
  // multi-rank fits: what rides behind the column statistics in their all-reduce -- {this rank's rows, ranks whose vote is
  // in, one vote per rank on the cut of the two-piece A^T sweep} (engine.cpp, column_statistics / fit_randomized)
  static constexpr size_t kStatsTail = 72;
  double stats_tail[kStatsTail] = {};
  bool vote_ready = false;        // every rank's vote came with the statistics: vote_cut is the agreed cut (0: one piece)
  int64_t vote_cut = 0;

  // device buffers (grow-only)
  sapca::DevBuf in_ptr, in_idx, in_val, up64, up64i, out_tmp;    // host-entry uploads
  sapca::PinnedBuf up_stage[2];                                   // page-locked ring of the chunked index upload
  hipEvent_t up_done[2] = {nullptr, nullptr};
  // column statistics accumulated chunk by chunk while the upload is in flight (upstats.hip); valid for exactly the
  // matrix in (in_ptr, in_idx, in_val) until one of the library's entry points rewrites its values
  struct UpStats {
    sapca::DevBuf work, out;            // long accumulators; sum | sumsq | count (f64, n each)
    sapca::PinnedBuf flag;              // 1: a value was inf/nan and `out` is void (readable once up_stats_done has passed)
    uint64_t m = 0, n = 0, nnz = 0;
    int dtype = -1;
    bool valid = false;
  } up_stats;
  hipEvent_t up_stats_done = nullptr;
  sapca::DevBuf at_ptr, at_idx, at_val;                          // A^T
  sapca::DevBuf ca_ptr, ca_idx, ca_val, cat_ptr, cat_idx, cat_val;  // mask-compacted A, A^T
  sapca::DevBuf drop_stats, drop_tmp;                              // their sums (sum | sumsq, full width) and the sort's work space
  sapca::DevBuf drop_col, drop_val;                                // the entries the compaction dropped, as (column, value) pairs
  sapca::DevBuf scratch, scratch2;
  sapca::DevBuf panel_x, panel_y, panel_w, panel_xs, panel_wide;   // (panel_wide: the out-of-place product of a panel wider than 128 columns)
  sapca::DevBuf small;                                           // G, R1, R2, Rinv, M, cvec, svec, info
  sapca::DevBuf stats;                                           // sum, sumsq, cnt (f64, n each)
  sapca::DevBuf mean_used_dev, o2m_dev, sel_rows_dev;
  sapca::DevBuf components_dev;                                  // k x n_used, T
  sapca::DevBuf lanczos_buf;
  sapca::DevBuf idx16_a, idx16_b;                                 // 2-byte index copies of the two operators of a Lanczos step
  sapca::TiledBuffers tb_a, tb_at;                               // tile-major formats for the LDS-staged sweep
  sapca::TiledOp tiled_a, tiled_at;
  sapca::DevBuf split_scratch, split_scratch2, votes, lz_scalars;

  sapca::EventTimer timer;
  std::vector<std::pair<int, int>> spans;  // (category, event index) of the last fit/transform
  std::vector<int> small_in_transform;      // events of a held-back small SVD that ran inside the projection's span
  sapca_timings timings{};
  sapca::Comm comm;
};

namespace sapca {

template <typename T>
struct Engine {
  using H = sapca_handle_s;
  static constexpr int kDtype = sizeof(T) == 8 ? 1 : 0;
  static void prepare(H& h, const CsrView<T>& A);
  static void finish_statistics(H& h);   // host side of R3 from stats_host (mean, total variance)
  static void fit(H& h, const CsrView<T>& A, bool defer_finish = false);
  static void finish_small_svd(H& h, const double** sign_out);   // host half of the f32 small SVD (R11) + svd_flip
  static void finish_fit(H& h);          // host-side tail of fit(): statistics, timings (after the last wait for the device)
  static void transform(H& h, const CsrView<T>& A, T* d_out);
  static void fit_randomized(H& h);
  static void fit_lanczos(H& h);
  static bool vote_rides(const H& h);          // the cut of the two-piece A^T sweep is agreed inside the statistics' all-reduce
  static int64_t piece_vote(H& h, int ld);     // this rank's vote: the first output row of its second piece, 0 = one piece
  // normaliser on a rows x ld panel; R_out (ld x ld f64 device, may be null) receives the
  // accumulated upper factor of the last CholeskyQR2.
  static void normalize(H& h, T* P, int64_t rows, int l, int ld, int normalizer, bool sharded, double* R1, double* R2,
                        int passes_hint = 0, const k::PanelSource<T>* src = nullptr, const T* w = nullptr, T* vec_out = nullptr);
  static CsrView<T> view(const H::RawCsr& r) {
    CsrView<T> v;
    v.rows = r.rows; v.cols = r.cols; v.nnz = r.nnz;
    v.ptr = static_cast<const int64_t*>(r.ptr);
    v.idx = static_cast<const int32_t*>(r.idx);
    v.val = static_cast<const T*>(r.val);
    return v;
  }
};

}  // namespace sapca
