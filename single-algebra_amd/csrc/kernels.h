// Launchers of the hand-written gfx950 kernels.  Everything takes device pointers and a
// stream; nothing here allocates except through the DevBuf scratch arguments.
#pragma once
#include <vector>
#include "common.h"

namespace sapca {
namespace k {

// How the producer of a panel left it: P = sum of nsplit slabs (slab sp at parts + sp * slab_stride, row stride = the panel's)
// minus the centring term mu sv^T (mu per row, sv per column; null: none).  gram() applies it on its way through the panel
// and writes P; materialize() only writes P.  parts may be P itself (nsplit = 1).
template <typename T>
struct PanelSource {
  const T* parts = nullptr;
  int nsplit = 0;
  int64_t slab_stride = 0;
  const T* mu = nullptr;
  const T* sv = nullptr;
};

// ---- prep.hip --------------------------------------------------------------------------
// nalgebra's usize indices -> int64 row offsets + int32 column indices; *flag |= 1 on col >= n.
void narrow_indices(const uint64_t* ptr64, const uint64_t* idx64, int64_t m, int64_t nnz, int64_t n,
                    int64_t* ptr, int32_t* idx, int* flag, hipStream_t s);
// CSR(A) -> CSR(A^T), entries of each A^T row in ascending A-row order (stable, deterministic).
template <typename T>
void transpose_csr(const CsrView<T>& A, int64_t* t_ptr, int32_t* t_idx, T* t_val, DevBuf& scratch, hipStream_t s,
                   int tile_major_nct = 0,   // f32, > 1: entries of a transposed row ordered by (row mod nct, row / nct)
                   const uint64_t** packed_rows_out = nullptr);   // with tile_major_nct: leave the rows packed as
                                                                   // (row << 32 | value bits) in `scratch`, skip t_idx / t_val
void unpack_transposed(const uint64_t* packed, int64_t nnz, int32_t* t_idx, float* t_val, hipStream_t s);
// Row sums of a CSR (applied to A^T: the reference's sum_col / sum_col_squared), f64 accumulation.
template <typename T>
void row_sums(const CsrView<T>& At, double* sum, double* sumsq, hipStream_t s);
// Mask compaction of A: keep entries with o2m[col] >= 0, renumbered.  Two passes around a scan.
template <typename T>
void compact_columns(const CsrView<T>& A, const int32_t* o2m, int64_t* new_ptr, int32_t* new_idx, T* new_val,
                     int64_t* new_nnz_host, DevBuf& scratch, hipStream_t s, int32_t* drop_col = nullptr, T* drop_val = nullptr,
                     unsigned long long* amax_bits = nullptr);
// (amax_bits: receives the bit pattern of max |value| over ALL stored entries of A as a double, a quiet nan's if one is not finite)
// (drop_col / drop_val, A.nnz each, receive the entries that were NOT kept as (original column, value) pairs, row after row)
// sum[c], sumsq[c] (c < n) over (column, value) pairs through a stable sort by column; seg (n + 1), keys_out, vals_out: work arrays
template <typename T>
void sums_by_column(const int32_t* cols, const T* vals, int64_t count, int64_t n, int64_t* seg, int32_t* keys_out, T* vals_out,
                    double* sum, double* sumsq, DevBuf& scratch, hipStream_t s);
// Row selection of A^T: rows listed in `rows` (ascending), column indices untouched.
template <typename T>
void select_rows(const CsrView<T>& At, const int32_t* rows, int64_t n_sel, int64_t* new_ptr, int32_t* new_idx,
                 T* new_val, int64_t* new_nnz_host, DevBuf& scratch, hipStream_t s);
// Exact, order-independent column statistics of a CSR whose entries land chunk by chunk (upstats.hip).  `work` holds the
// long accumulators.  scan_values (once all values are on the device) fixes the limb window kept in LDS; add takes the
// entries [e_lo, e_hi) of rows [r_lo, r_hi); finish writes out[0..n) = sum, out[n..2n) = sum of squares, out[2n..3n) =
// stored-entry count, each the correctly rounded exact value, and copies the "a value was inf/nan" flag to the host
// asynchronously.
template <typename T> size_t exact_colstats_bytes(int64_t n);
template <typename T> void exact_colstats_reset(void* work, int64_t n, hipStream_t s);
template <typename T> void exact_colstats_scan_values(const T* val, int64_t count, int64_t n, void* work, hipStream_t s);
template <typename T>
void exact_colstats_add(const int64_t* ptr, const int32_t* idx, const T* val, int64_t r_lo, int64_t r_hi, int64_t e_lo, int64_t e_hi,
                        int64_t n, void* work, hipStream_t s);
template <typename T> void exact_colstats_finish(void* work, int64_t n, double* out, int* nonfinite_host, hipStream_t s);
// out_a[where[j]] = a[where[j]], out_b[where[j]] = b[where[j]]   (the selected positions of two full-width arrays)
void copy_selected(const double* a, const double* b, const int32_t* where, int64_t count, double* out_a, double* out_b, hipStream_t s);
// out_a[where[j]] = a[j], out_b[where[j]] = b[j]   (column statistics from a compacted numbering back to the full width)
void scatter_pairs(const double* a, const double* b, const int32_t* where, int64_t count, double* out_a, double* out_b, hipStream_t s);
// mu[j] = T(sum[sel ? sel[j] : j] / count): the column means the sweeps centre with, from the device-side column sums
template <typename T>
void mean_from_sums(const double* sum, double count, const int32_t* sel, int64_t n_used, T* mu, hipStream_t s);
// out[r] = ptr[r+1] - ptr[r] as f64 (column counts when applied to A^T's row offsets)
void row_lengths_f64(const int64_t* ptr, int64_t rows, double* out, hipStream_t s);
// out[j] = number of stored entries with column j (integer atomics; transform of a matrix that
// was not the fitted one)
void column_counts_f64(const int32_t* idx, int64_t nnz, int64_t n, double* out, DevBuf& scratch, hipStream_t s);
// per-row segment table for the LDS-tiled sweep: seg[r][t] = first entry of row r with col >= t*tile_cols
template <typename T>
void build_tile_index(const CsrView<T>& A, int tile_cols, int n_tiles, int32_t* seg, hipStream_t s);

// ---- preproc.hip (preprocessing and statistics on a device-resident CSR) ----------
// Normalize<T> for CsrMatrix (csr.rs:1012-1066): values *= target / sums[row or column] where the sum is > 0.
// d_sums: device, length rows or cols; d_scale: device scratch of the same length.
template <typename T>
void normalize_csr(const CsrView<T>& A, T* values, const double* d_sums, double target, bool by_column, double* d_scale,
                   hipStream_t s);
// Log1P (csr.rs:1069-1078): values = ln(1 + values), in T.
template <typename T>
void log1p_values(T* values, int64_t nnz, hipStream_t s);
// sum_row, sum_row_squared, min_max_row of a CSR (applied to A^T: the column versions); any output may be null.
template <typename T>
void row_stats(const CsrView<T>& A, double* sum, double* sumsq, T* minv, T* maxv, hipStream_t s);

// dst[0 .. bytes) = src[0 .. bytes) by a 16-byte-per-lane streaming kernel (the attainable-HBM-rate probe of sapca_measure_copy_gbs)
void stream_copy16(const void* src, void* dst, int64_t bytes, hipStream_t s);

// ---- spmm.hip --------------------------------------------------------------------------
// Y[r][j] = sum_e val_e X[col_e][j] - cvec[j]   for j < ncols; X has leading dimension ldx
// (multiple of 16/sizeof(T)... see spmm.hip), Y leading dimension ldy.  cvec may be null.
// keep (nullable): a staged sweep whose tile range is split over workgroups may leave its partial slabs unsummed and describe
// them there (nsplit > 1; the caller's next pass over Y -- gram() -- sums them); otherwise keep = {Y, 1, 0}.  Only with cvec null.
template <typename T>
void spmm(const CsrView<T>& A, const TiledOp* tiled, const T* X, int ldx, T* Y, int ldy, int ncols,
          const T* cvec, int variant, DevBuf& scratch, hipStream_t s, PanelSource<T>* keep = nullptr);

// Y[r][j] = sum over the stored entries of row r of (value - shift[column]) X[column][j]   (quirk Q3 through the row kernel)
template <typename T>
void spmm_rows_shifted(const CsrView<T>& A, const T* X, int ldx, T* Y, int ldy, int ncols, const T* shift, hipStream_t s);

// ---- scatter.hip: column-wise reductions of a CSR without its transpose (fixed-point sums in LDS) ----------
bool scatter_fits(int64_t cols);   // one output vector of `cols` 64-bit words fits a workgroup's LDS
// *out_bits = bit pattern of max |v| as a double (a quiet NaN's pattern if any value is inf / nan)
template <typename T>
void absmax(const T* v, int64_t count, unsigned long long* out_bits, hipStream_t s);
// *out_bits = max(*out_bits, max |y|): the slot must have been cleared
void vecmax(const double* y, int64_t len, unsigned long long* out_bits, hipStream_t s);
// z = A^T y  (A: rows x cols, cols fitting LDS).  amax_bits / ymax_bits: max |a|, max |y| (the fixed-point scale follows from
// them); clear_bits (nullable): a slot to reset to 0 for the next product.  idx16 (nullable): 2-byte copy of A.idx.
template <typename T>
void spmvt_scatter(const CsrView<T>& A, const uint16_t* idx16, const double* y, const unsigned long long* amax_bits,
                   const unsigned long long* ymax_bits, unsigned long long* clear_bits, double* z, DevBuf& scratch, hipStream_t s);
// sum[c], sumsq[c], cnt[c] (nullable) of every column of A in ceil(cols / 7680) passes over the matrix
template <typename T>
void colstats_scatter(const CsrView<T>& A, const unsigned long long* amax_bits, double* sum, double* sumsq, double* cnt, DevBuf& scratch,
                      hipStream_t s);

// ---- spmm_tiled.hip ---------------------------------------------------------------------
// Builds the tile-major format of an f32 operator for panels of leading dimension ldp (64/128).
// Returns false (op.valid == false) when the operator does not fit the LDS staging; callers then
// stay on the row kernel.
// transposed: the operator is S^T, built straight from S (no transposed CSR needed); false when the
// format cannot be built (the caller stays on the row kernel)
// rows_tile_major: S's rows were produced by transpose_csr(..., tile_major_nct = tiled_tile_count(S.cols, ldp))
bool build_tiled(const CsrView<float>& S, bool transposed, int ldp, TiledOp& op, TiledBuffers& buf, hipStream_t s,
                 bool rows_tile_major = false, const uint64_t* packed_rows = nullptr,   // packed_rows: S.idx / S.val are not
                 bool allow_big_tile = true,                                             // filled, read (row << 32 | value) instead
                 bool seg_ready = false);   // buf.seg already holds the per-row tile index (at_stats_index)
// The same format with f64 values for panels of 64 f64 columns (512-byte rows: the tile geometry of the
// 128-float panels); built from a CSR in natural row order by the direct fill.
bool build_tiled(const CsrView<double>& S, int ldp, TiledOp& op, TiledBuffers& buf, hipStream_t s, bool rows_tile_major = false);
// The format of A^T (op: A.cols x A.rows) straight from A, without a transposed CSR: a histogram per (tile of A rows,
// column), a scatter of A's entries into per-chunk buckets, one workgroup per chunk for the format.  Byte-identical to
// build_tiled on the tile-major transposition.  at_ptr (A.cols + 1) receives A^T's row offsets, stats (2 * A.cols, may
// be null) the column sums and sums of squares of A, added per tile and then in tile order.  Returns false (nothing
// usable built) when the operator is outside this route's limits: the caller transposes instead.
bool build_tiled_at_direct(const CsrView<float>& A, int ldp, TiledOp& op, TiledBuffers& buf, int64_t* at_ptr, double* stats,
                           DevBuf& scratch, hipStream_t s);
// Column statistics of A (row sums / sums of squares of the packed tile-major A^T rows, same summation order as
// row_sums) and, in the same pass, the per-row tile index build_tiled(..., rows_tile_major, packed, ., seg_ready) needs.
void at_stats_index(const int64_t* ptr, const uint64_t* packed, int64_t rows, int64_t cols, int ldp, TiledBuffers& buf,
                    double* sum, double* sumsq, hipStream_t s);
// number of interleaved column tiles the format uses for an operator with `cols` columns
int tiled_tile_count(int64_t cols, int ldp);
// tile geometry (panel columns held per LDS tile row) for a panel of l columns: 64, two column passes when l > 64
int tiled_geometry(int l);
void spmm_tiled(const TiledOp& op, const float* X, int ldx, float* Y, int ldy, int ncols, const float* cvec, DevBuf& scratch,
                hipStream_t s, PanelSource<float>* keep = nullptr);
// the DPP-fed sweep in pieces of its output rows (spmm_tiled.hip): can the operator be swept so, the pieces' row bounds, one piece
bool spmm_tiled_pieces_ok(const TiledOp& op, int npieces, int ldx);
void spmm_tiled_piece_bounds(const TiledOp& op, int npieces, std::vector<int64_t>& bounds, hipStream_t s);
void spmm_tiled_piece(const TiledOp& op, int piece, int npieces, int wgs, int64_t first_row, int64_t row_count, const float* X, int ldx, float* Y,
                      int ldy, int ncols, DevBuf& scratch, hipStream_t s);
void spmm_tiled(const TiledOp& op, const double* X, int ldx, double* Y, int ldy, int ncols, const double* cvec, DevBuf& scratch,
                hipStream_t s, PanelSource<double>* keep = nullptr);

// ---- dense.hip -------------------------------------------------------------------------
// widest panel (n_components + n_oversamples, padded) the dense kernels take: up to 128 columns in one launch, beyond that in
// blocks of 64 / 128 columns (correct, not tuned)
constexpr int kMaxPanelWidth = 1024;

// G = P^T P (ld x ld, f64, full symmetric) for a rows x ld panel with ld % 16 == 0 (ld <= 128) or ld % 64 == 0 (ld <= 1024).
template <typename T>
void materialize(T* P, int64_t rows, int ld, const PanelSource<T>& src, hipStream_t s);
// wsum (nullable, ld doubles): sum_r w[r] P[r][:] of the same pass (w null: ones)
template <typename T>
void gram(T* P, int64_t rows, int ld, double* G, DevBuf& scratch, hipStream_t s, const PanelSource<T>* src = nullptr, const T* w = nullptr,
          double* wsum = nullptr);
// Upper Cholesky G = R^T R on the leading l x l block, Rinv = R^{-1}; both ld x ld, zero padded.
// *info += number of pivots that had to be regularised.
// wsum / vec32 / vec64 (nullable): vec[j] = sum_i Rinv[i][j] wsum[i] -- the column sums of P R^-1 from those of P
void chol_inv(const double* G, int l, int ld, double* R, double* Rinv, int* info, hipStream_t s, const double* wsum = nullptr,
              float* vec32 = nullptr, double* vec64 = nullptr);

// Eigen-decomposition of the symmetric l x l matrix G (row stride ld) on the device, one workgroup (parallel Jacobi in LDS,
// l <= 112): M (ld x ldk, zero padded) = leading k eigenvectors in columns, each divided by sigma_j = sqrt(lambda_j), in
// descending order; sigma[0..l) = all of them; status[0] = 0 / 1 (not converged) | 2 (non-finite), status[1] = sweeps.
bool sym_eig_device_ok(int l);
void sym_eig_device(const double* G, int l, int ld, int k, int ldk, double* M, double* sigma, int* status, hipStream_t s);
// out[rows x ldo] = P[rows x ld] * M[ld x ldo]  (M f64, row-major, ldo % 16 == 0); out may alias P
// when ldo == ld.
// out_stride (0: ldo) is the row stride of `out`, out_cols (0: ldo) the number of leading columns written.
template <typename T>
void panel_gemm(const T* P, int64_t rows, int ld, const double* M, int ldo, T* out, hipStream_t s, bool upper = false, int out_stride = 0,
                int out_cols = 0);
// out[j] = sum_r w[r] P[r][j]  (w null => ones), j < ld, f64 accumulation.
template <typename T>
void weighted_colsum(const T* P, int64_t rows, int ld, const T* w, T* out, DevBuf& scratch, hipStream_t s);
// Z[r][j] -= mu[r] * svec[j]
template <typename T>
void rank1_subtract(T* Z, int64_t rows, int ld, const T* mu, const T* svec, hipStream_t s);
// components[r][j] = sign_r * VtT[j][r] for r < k, with sign_r making the largest-|.| entry of
// row r positive (first index on ties): single_svdlib::randomized::svd_flip, v-based.
// sign_out (nullable): receives the device address of the k signs (in `scratch`)
template <typename T>
void flip_transpose(const T* VtT, int64_t n, int ld, int k, T* components, DevBuf& scratch, hipStream_t s, const double** sign_out = nullptr);
// out[j][c] = scale[j] * P[j][c] (scale null: copy); M[i][j] *= sign[j] for j < k
template <typename T>
void scale_rows(const T* P, int64_t rows, int ld, const double* scale, T* out, hipStream_t s);
void scale_columns(double* M, int rows, int ld, int k, const double* sign, hipStream_t s);
// W[j][r] = scale[j] * comps[r][j]  (r < k; zero for k <= r < ld); scale may be null.
template <typename T>
void scaled_transpose(const T* comps, int64_t n, int k, const double* scale, T* W, int ld, hipStream_t s);
template <typename T>
void fill_zero(T* p, int64_t count, hipStream_t s);
template <typename T>
void convert_from_f64(const double* in, T* out, int64_t count, hipStream_t s);
template <typename T>
void strip_padding(const T* in, int64_t rows, int ld, int ncols, T* out, hipStream_t s);
template <typename T>
void add_padding(const T* in, int64_t rows, int ncols, T* out, int ld, hipStream_t s);

// ---- rng.hip ---------------------------------------------------------------------------
// Omega[r][j] ~ N(0,1) for j < l (zero for l <= j < ld), a pure function of (seed, r*l+j).
template <typename T>
void gaussian_panel(T* out, int64_t rows, int l, int ld, uint32_t seed, hipStream_t s);

// ---- lanczos.hip (BLAS-1/2 helpers on f64 vectors) ---------------------------------------
template <typename T>
void spmv(const CsrView<T>& A, const double* x, double* y, hipStream_t s);

}  // namespace k
}  // namespace sapca
