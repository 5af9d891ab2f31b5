/*
 * sapca.h -- C ABI of the MI355X-native sparse-PCA hot path (libsapca.so).
 *
 * Drop-in boundary for single-algebra's src/dimred/pca (reference v0.9.2).  The
 * reference has no FFI of its own: its boundary is the Rust generic API
 * (SparsePCABuilder / MaskedSparsePCABuilder / fit / transform / fit_transform)
 * and, one level down, the single-svdlib calls it makes.  Every entry point below
 * names the reference interface it replaces (paths relative to the reference
 * repository root); INTEGRATION.md shows the Rust `extern "C"` binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch/HIP types in signatures
 *     (a HIP stream is passed as void*).
 *   - every function returns a sapca_status; nothing throws or aborts across the
 *     ABI.  sapca_last_error(h) gives the message (the reference's anyhow text
 *     where one exists).
 *   - _f32/_f64 pairs mirror the reference's generic T (f32/f64 in practice).
 *   - host CSR inputs use nalgebra_sparse::CsrMatrix's own layout: row_offsets
 *     and col_indices are usize (uint64_t), zero-copy from Rust.  The host entry
 *     points check what keeps every kernel inside the arrays (offsets start at 0,
 *     end at nnz and never decrease; columns < n) and return SAPCA_ERR_ARG otherwise;
 *     columns sorted and unique within a row is CsrMatrix's own invariant and is
 *     relied upon, not checked (unsorted rows give wrong numbers, not stray accesses).
 *   - "device" entry points take HBM-resident CSR (int64 row offsets, int32 column
 *     indices) and device output buffers; they are what bench.py times.  Device
 *     arrays are trusted (no validation pass).
 *   - inputs are borrowed for the duration of a call; outputs are written into
 *     caller-allocated buffers.
 *   - a handle is not thread-safe; distinct handles may be used concurrently.
 *
 * Deviations from the reference (everything else, quirks included, is the reference's behaviour):
 *   1. a fit whose SVD returns fewer than k values gives SAPCA_ERR_SVD where the reference panics on s[i]
 *      (sparse/mod.rs:213-215);
 *   2. mean_ has n_cols zeros when center = false (the reference: n_samples zeros, never read, sparse/mod.rs:116);
 *   3. no unconditional stdout (the reference prints from MaskedSparsePCA::fit, sparse_masked/mod.rs:373-378);
 *   4. n_components + n_oversamples is limited to 1024 (tuned to 128);
 *   5. SVDMethod::Lanczos: svd_las2 of single-svdlib is SVDLIBC's las2, a single-vector Lanczos with SELECTIVE
 *      re-orthogonalisation; csrc/lanczos.hip keeps the recurrence, the end interval [-1e-30, 1e30], kappa and the iteration
 *      cap of the call sites (sparse/mod.rs:135-143, sparse_masked/mod.rs:316-331) but re-orthogonalises every new vector
 *      against all previous ones (two passes of classical Gram-Schmidt on the device: one GEMV pair instead of las2's
 *      bookkeeping of which Ritz vectors have converged).  Converged triplets agree to kappa; the number of Lanczos steps
 *      taken, and triplets that have NOT converged at the cap, can differ.
 */
#ifndef SAPCA_H
#define SAPCA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAPCA_ABI_VERSION 4   /* 2: sapca_timings grew sweep_kernel / sweep_slots_*; sapca_comm_rccl_available
                                 3: sapca_multi_* (one handle, several GPUs), sapca_upload_values_changed
                                 4: *_csr_device_to_host_*, sapca_comm_abort / _async_error / _has_side_lane,
                                    sapca_multi_upload_csr_* and the sapca_multi_*_resident calls        */

typedef struct sapca_handle_s* sapca_handle;

typedef enum sapca_status {
  SAPCA_OK = 0,
  SAPCA_ERR_ARG = 1,        /* bad argument / unsupported size                                  */
  SAPCA_ERR_MASK_LEN = 2,   /* "The mask vector length and the number of features (columns)
                               have to be the same!"   sparse_masked/mod.rs:258-262, 440-444    */
  SAPCA_ERR_NOT_FITTED = 3, /* "Must be fitted before transform!" sparse/mod.rs:259,263;
                               "Model must be fitted first!"      sparse/mod.rs:299,316          */
  SAPCA_ERR_SVD = 4,        /* "SVD computation failed: .." / "Randomized SVD computation
                               failed: .." sparse/mod.rs:144,180; also returned where the
                               reference would panic on s[i] (rank < k), sparse/mod.rs:213-215   */
  SAPCA_ERR_HIP = 5,        /* HIP runtime error (message carries hipGetErrorString)            */
  SAPCA_ERR_COMM = 6,       /* RCCL / collective error                                          */
  SAPCA_ERR_NOMEM = 7
} sapca_status;

/* SVDMethod                                     src/dimred/pca/mod.rs:49-68 (default Lanczos) */
typedef enum sapca_method { SAPCA_LANCZOS = 0, SAPCA_RANDOM = 1 } sapca_method;
/* PowerIterationNormalizer (re-export)          src/dimred/pca/mod.rs:41                      */
typedef enum sapca_normalizer { SAPCA_NORM_QR = 0, SAPCA_NORM_LU = 1, SAPCA_NORM_NONE = 2 } sapca_normalizer;
/* transform semantics: REFERENCE reproduces quirks Q2/Q3; CENTERED is the
 * mathematically centred projection (A - 1 mu^T) V^T, offered as an opt-in superset.          */
typedef enum sapca_transform_semantics { SAPCA_TRANSFORM_REFERENCE = 0, SAPCA_TRANSFORM_CENTERED = 1 } sapca_transform_semantics;

/* Builder fields.  SparsePCABuilder sparse/mod.rs:375-484 (defaults :392-401);
 * MaskedSparsePCABuilder sparse_masked/mod.rs:37-160 (defaults :55-66).
 * alpha and tolerance are stored and never read by the reference; kept for drop-in.          */
typedef struct sapca_options {
  uint32_t struct_size;          /* = sizeof(sapca_options); ABI growth guard                   */
  uint32_t random_seed;          /* .random_seed(u32), default 42                               */
  uint64_t n_components;         /* .n_components(usize), default 50                            */
  double alpha;                  /* .alpha(T), default 1.0 (unused)                             */
  double tolerance;              /* .tolerance(T), default 1e-6 (unused)                        */
  uint8_t center;                /* .center(bool), default 1                                    */
  uint8_t verbose;               /* .verbose(bool), default 0                                   */
  uint8_t collect_timings;       /* record per-stage HIP-event timings (sapca_get_timings)      */
  uint8_t reserved0;
  int32_t method;                /* sapca_method, default SAPCA_LANCZOS                         */
  uint64_t n_oversamples;        /* SVDMethod::Random.n_oversamples (n_components + this: at most 1024; above 128 block-wise, untuned) */
  uint64_t n_power_iterations;   /* SVDMethod::Random.n_power_iterations                        */
  int32_t normalizer;            /* sapca_normalizer                                            */
  int32_t transform_semantics;   /* sapca_transform_semantics                                   */
  int32_t device_id;             /* HIP device ordinal; -1 = current device                     */
  int32_t spmm_variant;          /* 0 = auto; 1 = L2-gather row kernel; 2 = LDS-tiled kernel    */
  void* stream;                  /* hipStream_t to run on; NULL = library-owned stream          */
} sapca_options;

/* per-stage device timings of the last fit/transform (ms, HIP events on the handle's stream) */
typedef struct sapca_timings {
  double upload_ms;          /* narrowing + H2D (host entry points only)                        */
  double prepare_ms;         /* mask compaction, transpose, tile formats                        */
  double stats_ms;           /* column statistics                                               */
  double spmm_ms;            /* sum over the A*X sweeps                                         */
  double spmmt_ms;           /* sum over the A^T*Y sweeps                                       */
  double ortho_ms;           /* Gram / Cholesky / panel GEMM                                    */
  double small_svd_ms;       /* final factorisation incl. host Jacobi                           */
  double lanczos_ms;
  double transform_ms;
  double comm_ms;            /* collectives: device time (events around every all-reduce) when
                              * timings are collected, host-observed time otherwise            */
  double fit_total_ms;
  uint32_t n_spmm;           /* number of A*X sweeps timed                                      */
  uint32_t n_spmmt;
  double spmm_sweep_ms[32];  /* individual sweeps, in launch order                              */
  double spmmt_sweep_ms[32];
  double bytes_per_sweep;    /* ALGORITHMIC bytes of one sweep (SURVEY.md §8d formula)          */
  uint64_t lanczos_steps;
  uint32_t sweep_kernel;     /* randomized fits: 0 row-gather kernel, 1 staged-entry quad sweep,
                              * 2 DPP-fed quad sweep (spmm_dq.hip)                              */
  uint32_t at_sweep_pieces;  /* multi-rank randomized fits: 2 when the A^T sweeps ran in two pieces with the first
                              * piece's panel all-reduce behind the second piece's sweep; else 1 (0: no sweep) */
  uint64_t sweep_slots_a;    /* entry slots (stored entries + padding) one A*X sweep walks; 0 on
                              * the row-gather kernel.  x 256 B (f32, 64 columns) = LDS gather bytes */
  uint64_t sweep_slots_at;   /* the same for one A^T*Y sweep                                    */
} sapca_timings;

void sapca_options_default(sapca_options* o);
int sapca_abi_version(void);

/* SparsePCABuilder::build / MaskedSparsePCABuilder::build   sparse/mod.rs:470-483, masked :144-159 */
sapca_status sapca_create(const sapca_options* opts, sapca_handle* out);
void sapca_destroy(sapca_handle h);
const char* sapca_last_error(sapca_handle h);   /* valid until the next call on h; h may be NULL (create errors) */

/* MaskedSparsePCABuilder::mask(Vec<bool>)       sparse_masked/mod.rs:111-114.  len is checked
 * against ncols at fit/transform time like the reference (:258-262).  len == 0 clears it.    */
sapca_status sapca_set_mask(sapca_handle h, const uint8_t* mask, size_t len);

/* Test hook: inject the Gaussian test matrix Omega (n_used x (k+p), row-major, host) used by
 * the next randomized fit instead of the built-in generator -- the reference's rand-0.9
 * stream is not reproducible, so parity is checked with a shared Omega.       */
sapca_status sapca_set_omega_f32(sapca_handle h, const float* omega, size_t rows, size_t cols);
sapca_status sapca_set_omega_f64(sapca_handle h, const double* omega, size_t rows, size_t cols);

/* SparsePCA::fit / MaskedSparsePCA::fit          sparse/mod.rs:102-242; masked :255-419
 * Host matrices: the column statistics of the fit (sum_col, sum_col_squared, csr.rs:259-312 and
 * 558-608, and the per-column counts) are gathered behind the upload's DMA as exact sums rounded
 * once -- independent of summation order, hence bit-reproducible; inf/nan values switch to the
 * row sums of the transposed matrix, which device-resident inputs always use.                  */
sapca_status sapca_fit_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                               const uint64_t* row_offsets, const uint64_t* col_indices, const float* values);
sapca_status sapca_fit_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                               const uint64_t* row_offsets, const uint64_t* col_indices, const double* values);
/* SparsePCA::transform / MaskedSparsePCA::transform   sparse/mod.rs:255-285; masked :438-546.
 * out: m x n_components, row-major (ndarray Array2 standard layout).                          */
sapca_status sapca_transform_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                     const uint64_t* row_offsets, const uint64_t* col_indices, const float* values, float* out);
sapca_status sapca_transform_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                     const uint64_t* row_offsets, const uint64_t* col_indices, const double* values, double* out);
/* fit_transform                                   sparse/mod.rs:355-358; masked :616-619        */
sapca_status sapca_fit_transform_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                         const uint64_t* row_offsets, const uint64_t* col_indices, const float* values, float* out);
sapca_status sapca_fit_transform_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                         const uint64_t* row_offsets, const uint64_t* col_indices, const double* values, double* out);

/* Same three operations on HBM-resident CSR (device pointers: int64 row offsets [m+1],
 * int32 column indices [nnz], values [nnz]); `out` is a device buffer m x n_components.
 * The matrix is borrowed: it must stay valid and unmodified until the call returns.
 * Nothing derived from caller-owned arrays outlives the call: a later sapca_transform_csr_device_*
 * on the same pointers re-derives what it needs (they may hold different values by then); only
 * fit_transform, and the library-owned arrays of sapca_upload_csr_*, reuse the fit's preparation.
 * When the handle belongs to a multi-rank communicator (sapca_comm_*), (m, row_offsets, ..)
 * describe THIS rank's row shard and n is the global column count.                            */
sapca_status sapca_fit_csr_device_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                      const int64_t* d_row_offsets, const int32_t* d_col_indices, const float* d_values);
sapca_status sapca_fit_csr_device_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                      const int64_t* d_row_offsets, const int32_t* d_col_indices, const double* d_values);
sapca_status sapca_transform_csr_device_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                            const int64_t* d_row_offsets, const int32_t* d_col_indices, const float* d_values, float* d_out);
sapca_status sapca_transform_csr_device_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                            const int64_t* d_row_offsets, const int32_t* d_col_indices, const double* d_values, double* d_out);
sapca_status sapca_fit_transform_csr_device_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                                const int64_t* d_row_offsets, const int32_t* d_col_indices, const float* d_values, float* d_out);
sapca_status sapca_fit_transform_csr_device_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                                const int64_t* d_row_offsets, const int32_t* d_col_indices, const double* d_values, double* d_out);
/* The same two calls with the m x n_components projection delivered to HOST memory (through the
 * handle's page-locked ring): what a caller whose matrix is resident (sapca_upload_csr_*) but whose
 * consumer is host code -- the reference's Array2<T> result, sparse/mod.rs:255-285 -- wants.     */
sapca_status sapca_transform_csr_device_to_host_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                                    const int64_t* d_row_offsets, const int32_t* d_col_indices, const float* d_values, float* out);
sapca_status sapca_transform_csr_device_to_host_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                                    const int64_t* d_row_offsets, const int32_t* d_col_indices, const double* d_values, double* out);
sapca_status sapca_fit_transform_csr_device_to_host_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                                        const int64_t* d_row_offsets, const int32_t* d_col_indices, const float* d_values, float* out);
sapca_status sapca_fit_transform_csr_device_to_host_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                                        const int64_t* d_row_offsets, const int32_t* d_col_indices, const double* d_values, double* out);

/* Fitted state.  The reference keeps components_/explained_variance_/mean_ private
 * (sparse/mod.rs:41-43); a Rust wrapper needs them to populate those fields.
 * dims: k = n_components, n_used = columns seen by the SVD (n, or n' under a mask),
 * n_cols = columns of the fitted matrix (length of mean_).                                    */
sapca_status sapca_get_dims(sapca_handle h, uint64_t* k, uint64_t* n_used, uint64_t* n_cols);
sapca_status sapca_get_components_f32(sapca_handle h, float* out, size_t cap);            /* k x n_used   sparse/mod.rs:208 */
sapca_status sapca_get_components_f64(sapca_handle h, double* out, size_t cap);
sapca_status sapca_get_singular_values_f32(sapca_handle h, float* out, size_t cap);       /* k            res.s             */
sapca_status sapca_get_singular_values_f64(sapca_handle h, double* out, size_t cap);
sapca_status sapca_get_explained_variance_f32(sapca_handle h, float* out, size_t cap);    /* k            sparse/mod.rs:210-216 */
sapca_status sapca_get_explained_variance_f64(sapca_handle h, double* out, size_t cap);
sapca_status sapca_get_mean_f32(sapca_handle h, float* out, size_t cap);                  /* n_cols       sparse/mod.rs:106-117 */
sapca_status sapca_get_mean_f64(sapca_handle h, double* out, size_t cap);
sapca_status sapca_get_total_variance(sapca_handle h, double* out);                       /* sparse/mod.rs:119-131, 218-223 */
/* explained_variance_ratio()                      sparse/mod.rs:312-322; masked :574-582       */
sapca_status sapca_get_explained_variance_ratio_f32(sapca_handle h, float* out, size_t cap);
sapca_status sapca_get_explained_variance_ratio_f64(sapca_handle h, double* out, size_t cap);
/* cumulative_explained_variance_ratio()           sparse/mod.rs:333-343; masked :593-603       */
sapca_status sapca_get_cumulative_explained_variance_ratio_f32(sapca_handle h, float* out, size_t cap);
sapca_status sapca_get_cumulative_explained_variance_ratio_f64(sapca_handle h, double* out, size_t cap);
/* feature_importances()                           sparse/mod.rs:295-302; masked :557-564       */
sapca_status sapca_get_feature_importances_f32(sapca_handle h, float* out, size_t cap);   /* k x n_used */
sapca_status sapca_get_feature_importances_f64(sapca_handle h, double* out, size_t cap);
/* cols_to_use (ascending, sparse_masked/mod.rs:264-271) and the col -> masked-index map of
 * :462-466 as a dense table (-1 = column dropped).  Integer, bit-exact.  Either may be NULL.  */
sapca_status sapca_get_mask_index_maps(sapca_handle h, uint64_t* cols_to_use, size_t cap_cols,
                                       int64_t* orig_to_masked, size_t cap_map);
sapca_status sapca_get_timings(sapca_handle h, sapca_timings* out);

/* ---- stage-level operators (host buffers in and out), used by the parity tests ------------ */
/* <CsrMatrix as MatrixSum>::sum_col / sum_col_squared   src/sparse/csr.rs:259-312, 558-608.
 * One fused device pass yields both plus the per-column stored-entry count (any may be NULL). */
sapca_status sapca_colstats_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                    const uint64_t* row_offsets, const uint64_t* col_indices, const float* values,
                                    float* sum_col, float* sum_col_squared, uint64_t* nonzero_col);
sapca_status sapca_colstats_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                    const uint64_t* row_offsets, const uint64_t* col_indices, const double* values,
                                    double* sum_col, double* sum_col_squared, uint64_t* nonzero_col);
/* The two sweeps inside single_svdlib::randomized::randomized_svd (call sites
 * sparse/mod.rs:170-180): Y = (A - 1 mu^T) X   (X: n x l, Y: m x l) and
 * Z = (A - 1 mu^T)^T Y (Y: m x l, Z: n x l); row-major, mu may be NULL (uncentred).         */
sapca_status sapca_spmm_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                const uint64_t* row_offsets, const uint64_t* col_indices, const float* values,
                                const float* mu, uint64_t l, const float* X, float* Y);
sapca_status sapca_spmm_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                const uint64_t* row_offsets, const uint64_t* col_indices, const double* values,
                                const double* mu, uint64_t l, const double* X, double* Y);
sapca_status sapca_spmmt_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                 const uint64_t* row_offsets, const uint64_t* col_indices, const float* values,
                                 const float* mu, uint64_t l, const float* Y, float* Z);
sapca_status sapca_spmmt_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                 const uint64_t* row_offsets, const uint64_t* col_indices, const double* values,
                                 const double* mu, uint64_t l, const double* Y, double* Z);
/* PowerIterationNormalizer applied to a rows x l row-major panel in place (QR: orthonormal
 * basis of the same span; LU: well-conditioned basis of the same span; NONE: untouched).     */
sapca_status sapca_normalize_panel_f32(sapca_handle h, int32_t normalizer, uint64_t rows, uint64_t l, float* panel);
sapca_status sapca_normalize_panel_f64(sapca_handle h, int32_t normalizer, uint64_t rows, uint64_t l, double* panel);
/* The built-in Omega generator (rows x l standard normal from (seed)), for inspection.        */
sapca_status sapca_generate_omega_f32(sapca_handle h, uint64_t rows, uint64_t l, float* out);
sapca_status sapca_generate_omega_f64(sapca_handle h, uint64_t rows, uint64_t l, double* out);

/* ---- device-resident workflow: upload once, preprocess and analyse in HBM ----
 * The consumer's typical pipeline is normalize -> log1p -> PCA (/root/reference/src/lib.rs:28-33).
 * sapca_upload_csr_* copies a host CsrMatrix (usize indices) into buffers owned by the handle and
 * returns the device arrays (valid until the next host-matrix call on this handle or its
 * destruction); the *_device_* entry points below and sapca_fit*_csr_device_* then work on them
 * without touching PCIe again.  The column statistics gathered during the upload serve a fit of
 * the arrays as uploaded; normalize / log1p on them drop those statistics.                      */
sapca_status sapca_upload_csr_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                  const uint64_t* row_offsets, const uint64_t* col_indices, const float* values,
                                  const int64_t** d_row_offsets, const int32_t** d_col_indices, float** d_values);
sapca_status sapca_upload_csr_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                  const uint64_t* row_offsets, const uint64_t* col_indices, const double* values,
                                  const int64_t** d_row_offsets, const int32_t** d_col_indices, double** d_values);
/* <CsrMatrix<T> as Normalize<T>>::normalize::<f64>(&sums, target, &direction)
 * (/root/reference/src/sparse/csr.rs:1012-1066): every stored value of row (direction 0) or
 * column (direction 1) i becomes T(f64(value) * (target / sums[i])) where sums[i] > 0; others
 * are left alone.  `sums` is a HOST array of length m (ROW) or n (COLUMN); a wrong length is
 * SAPCA_ERR_ARG (the dense twin's message, src/dense/mod.rs; the CSR impl would index out of
 * bounds).  `values` is the DEVICE value array, modified in place.                             */
sapca_status sapca_normalize_csr_device_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                            const int64_t* row_offsets, const int32_t* col_indices, float* values,
                                            const double* sums, uint64_t sums_len, double target, int32_t direction);
sapca_status sapca_normalize_csr_device_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                            const int64_t* row_offsets, const int32_t* col_indices, double* values,
                                            const double* sums, uint64_t sums_len, double target, int32_t direction);
/* <CsrMatrix<T> as Log1P<T>>::log1p_normalize (csr.rs:1069-1078): value = ln(1 + value) in T,
 * in place on the DEVICE value array.                                                           */
sapca_status sapca_log1p_csr_device_f32(sapca_handle h, uint64_t nnz, float* values);
sapca_status sapca_log1p_csr_device_f64(sapca_handle h, uint64_t nnz, double* values);
/* d_values of sapca_upload_csr_* is writable: a caller that edits the uploaded values with its own
 * kernel (anything but the two entry points above) says so here BEFORE the next fit, so that the
 * column statistics gathered during the upload and any cached preparation are dropped and the
 * fit re-derives mean_ / the total variance from the values as they are now.  The reference has
 * no counterpart (a &CsrMatrix is immutable while borrowed, sparse/mod.rs:102).                  */
sapca_status sapca_upload_values_changed(sapca_handle h);
/* The per-row (direction 0) or per-column (direction 1) statistics of the MatrixSum /
 * MatrixNonZero / MatrixMinMax traits in one call on a device-resident CSR: sum_row|col
 * (csr.rs:259-392), sum_row|col_squared (:558-630), nonzero_row|col (:23-134, stored entries),
 * min_max_row|col (:917-1008: over the stored entries; a row/column without any keeps
 * (T::MAX, -T::MAX), the reference's initial values).  Outputs are HOST arrays of length m or n;
 * any may be NULL.  Sums are accumulated in f64 (the reference accumulates in the caller's T, in
 * storage order).  var_row|col (csr.rs:632-726) is host arithmetic on these:
 *   var = (sumsq/N - (sum/N)^2) * N/(N-1), N = the other dimension.                             */
sapca_status sapca_stats_csr_device_f32(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                        const int64_t* row_offsets, const int32_t* col_indices, const float* values,
                                        int32_t direction, double* sum, double* sum_squared, uint64_t* nonzero,
                                        float* min_out, float* max_out);
sapca_status sapca_stats_csr_device_f64(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz,
                                        const int64_t* row_offsets, const int32_t* col_indices, const double* values,
                                        int32_t direction, double* sum, double* sum_squared, uint64_t* nonzero,
                                        double* min_out, double* max_out);

/* Measurement support: the rate (GB/s, read + write counted) of a 16-byte-per-lane streaming copy of `bytes`
 * bytes on the handle's device, best of `reps` -- the HBM rate a kernel of this library can attain, reported by
 * bench.py beside the data-sheet peak.  Allocates and frees its two buffers.                       */
sapca_status sapca_measure_copy_gbs(sapca_handle h, uint64_t bytes, uint32_t reps, double* gbs);

/* ---- multi-GPU: one process per GPU, rows range-partitioned ---------------- */
/* nnz-balanced contiguous row ranges: bounds[0]=0 <= ... <= bounds[nparts]=m.  Pure host code. */
sapca_status sapca_partition_rows(uint64_t m, const uint64_t* row_offsets, uint32_t nparts, uint64_t* bounds);
/* Built-in collective = RCCL (resolved at run time from librccl.so.1).  Rank 0 creates the
 * 128-byte id and the host program distributes it (torch.distributed broadcast, MPI, a file).  */
/* 1 when librccl and its entry points resolve in this process: every rank checks this BEFORE the
 * collective ncclCommInitRank inside sapca_comm_init_rank, which must not fail on some ranks only. */
int sapca_comm_rccl_available(void);
sapca_status sapca_comm_unique_id(uint8_t id[128]);
sapca_status sapca_comm_init_rank(sapca_handle h, uint32_t nranks, uint32_t rank, const uint8_t id[128]);
/* Or bring your own all-reduce(sum) over all ranks: `buf` is a DEVICE pointer holding `count`
 * elements of dtype (0 = f32, 1 = f64), reduced in place, ordered on `stream` (hipStream_t).
 * Return 0 on success.                                                                        */
typedef int (*sapca_allreduce_fn)(void* ctx, void* buf, uint64_t count, int32_t dtype, void* stream);
sapca_status sapca_comm_set_callback(sapca_handle h, uint32_t nranks, uint32_t rank, sapca_allreduce_fn fn, void* ctx);
/* Invokes the handle's collective once on a caller buffer (plumbing self-test).               */
sapca_status sapca_comm_allreduce(sapca_handle h, void* buf, uint64_t count, int32_t dtype);
/* A rank that fails OUTSIDE a collective (out of memory, a HIP error, a refused argument) never
 * joins the collectives its peers are waiting in.  Under the built-in RCCL transport the way out
 * is ncclCommAbort: sapca_comm_abort ends every collective of THIS handle that is in flight or
 * still to come -- a fit blocked behind one returns SAPCA_ERR_COMM -- and may be called from any
 * thread while a fit of the handle is running.  sapca_multi_* does this for its members; in the
 * one-process-per-GPU deployment the HOST program must: when a rank's call fails, tell the peers
 * (its own control channel) and have each call sapca_comm_abort on its handle, or poll
 * sapca_comm_async_error (0 = healthy; otherwise RCCL's ncclResult_t, -1 after an abort) from a
 * watchdog thread.  The communicator is gone afterwards: sapca_comm_init_rank again.            */
sapca_status sapca_comm_abort(sapca_handle h);
sapca_status sapca_comm_async_error(sapca_handle h, int32_t* state);
/* 1 when collectives issued on the library's side stream have a lane of their own (a callback
 * transport, or RCCL with the duplicate communicator made by ncclCommSplit at init): the A^T
 * sweep of a row-sharded fit then runs in two pieces with the first piece's all-reduce behind
 * the second piece's sweep.  Every rank must see the same answer.                               */
int sapca_comm_has_side_lane(sapca_handle h);


/* ---- one handle, several GPUs, one calling thread (SURVEY.md §8b "Threading", §8e) ----------
 * The reference call is ONE fit_transform(&CsrMatrix) from one thread (sparse/mod.rs:355-358;
 * masked :616-619).  A sapca_multi owns one member handle per listed device; every call below
 * splits the HOST CsrMatrix into nnz-balanced contiguous row ranges (sapca_partition_rows),
 * runs the matching sapca_*_csr_* entry point on every shard from a host thread per device, and
 * blocks until all are done.  The members are the ranks of one communicator: RCCL between
 * distinct devices, an in-process all-reduce through page-locked host memory when a device is
 * listed more than once (a one-GPU box rehearsing the path) or librccl does not resolve.
 * A member that fails (out of memory, a refused shard, a HIP error) ends the call for all: the
 * peers' collectives are abandoned (in-process) or aborted (ncclCommAbort on every member's
 * communicators), every member returns, the call reports the first failure, and the next call
 * builds new communicators -- no call blocks on a peer that has left.
 * `out` (m x n_components, row-major, HOST) receives every shard's rows in place.  The fitted
 * state is replicated and bitwise identical on all members: read it from any member with the
 * sapca_get_* functions (sapca_multi_member(mh, 0)); sapca_set_omega_* must be applied to every
 * member.  A sapca_multi is not thread-safe for concurrent calls, like a handle.                */
typedef struct sapca_multi_s* sapca_multi;
sapca_status sapca_multi_create(const sapca_options* opts /* device_id and stream are ignored */,
                                const int32_t* device_ids, uint32_t n_devices, sapca_multi* out);
void sapca_multi_destroy(sapca_multi mh);
const char* sapca_multi_last_error(sapca_multi mh /* NULL: the last failed sapca_multi_create of this thread */);
uint32_t sapca_multi_n_devices(sapca_multi mh);
sapca_handle sapca_multi_member(sapca_multi mh, uint32_t i);
int sapca_multi_uses_rccl(sapca_multi mh);
sapca_status sapca_multi_set_mask(sapca_multi mh, const uint8_t* mask, size_t n);
sapca_status sapca_multi_fit_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                     const uint64_t* row_offsets, const uint64_t* col_indices, const float* values);
sapca_status sapca_multi_fit_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                     const uint64_t* row_offsets, const uint64_t* col_indices, const double* values);
sapca_status sapca_multi_transform_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                           const uint64_t* row_offsets, const uint64_t* col_indices, const float* values, float* out);
sapca_status sapca_multi_transform_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                           const uint64_t* row_offsets, const uint64_t* col_indices, const double* values, double* out);
sapca_status sapca_multi_fit_transform_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                               const uint64_t* row_offsets, const uint64_t* col_indices, const float* values, float* out);
sapca_status sapca_multi_fit_transform_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                               const uint64_t* row_offsets, const uint64_t* col_indices, const double* values, double* out);
/* Resident shards (SURVEY.md §8f-1 for several devices): sapca_multi_upload_csr_* partitions the HOST
 * CsrMatrix as above and uploads every shard to its device ONCE (the uploads run side by side, one
 * host thread per device); the *_resident calls then fit / project the uploaded shards without
 * touching PCIe for the matrix again -- repeated fits (another k, another mask, another seed)
 * skip the upload, which is where the wall-clock of a one-off call goes (10.8 GB of usize CSR at
 * BASELINE configs[3]).  `out` is HOST memory, m x n_components, as above.  The shards stay valid
 * until the next sapca_multi_upload_csr_* / non-resident call on this sapca_multi.
 * sapca_multi_resident_shard: the row range and the device arrays of member i's shard (for
 * callers that preprocess in HBM with sapca_normalize_csr_device_* / sapca_log1p_csr_device_* on
 * sapca_multi_member(mh, i)); any output may be NULL.                                           */
sapca_status sapca_multi_upload_csr_f32(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                        const uint64_t* row_offsets, const uint64_t* col_indices, const float* values);
sapca_status sapca_multi_upload_csr_f64(sapca_multi mh, uint64_t m, uint64_t n, uint64_t nnz,
                                        const uint64_t* row_offsets, const uint64_t* col_indices, const double* values);
sapca_status sapca_multi_fit_resident(sapca_multi mh);
sapca_status sapca_multi_transform_resident_f32(sapca_multi mh, float* out);
sapca_status sapca_multi_transform_resident_f64(sapca_multi mh, double* out);
sapca_status sapca_multi_fit_transform_resident_f32(sapca_multi mh, float* out);
sapca_status sapca_multi_fit_transform_resident_f64(sapca_multi mh, double* out);
sapca_status sapca_multi_resident_shard(sapca_multi mh, uint32_t i, uint64_t* first_row, uint64_t* rows, uint64_t* nnz,
                                        const int64_t** d_row_offsets, const int32_t** d_col_indices, void** d_values);

#ifdef __cplusplus
}
#endif
#endif /* SAPCA_H */
