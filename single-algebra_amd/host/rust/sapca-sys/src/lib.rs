//! Raw bindings to `include/sapca.h` (ABI version 3).  UNTESTED SOURCE: written against the header,
//! never compiled in the build image (no rustc).  One `extern "C"` item per header declaration that
//! the safe wrapper uses; the `_f64` twins mirror the `_f32` ones.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct sapca_handle_s {
    _private: [u8; 0],
}
pub type sapca_handle = *mut sapca_handle_s;
#[repr(C)]
pub struct sapca_multi_s {
    _private: [u8; 0],
}
pub type sapca_multi = *mut sapca_multi_s;

pub const SAPCA_OK: c_int = 0;
pub const SAPCA_ERR_ARG: c_int = 1;
pub const SAPCA_ERR_MASK_LEN: c_int = 2;
pub const SAPCA_ERR_NOT_FITTED: c_int = 3;
pub const SAPCA_ERR_SVD: c_int = 4;
pub const SAPCA_ERR_HIP: c_int = 5;
pub const SAPCA_ERR_COMM: c_int = 6;
pub const SAPCA_ERR_NOMEM: c_int = 7;

pub const SAPCA_LANCZOS: i32 = 0;
pub const SAPCA_RANDOM: i32 = 1;
pub const SAPCA_NORM_QR: i32 = 0;
pub const SAPCA_NORM_LU: i32 = 1;
pub const SAPCA_NORM_NONE: i32 = 2;

/// `sapca_options` (include/sapca.h): the builder fields of sparse/mod.rs:375-484.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct sapca_options {
    pub struct_size: u32,
    pub random_seed: u32,
    pub n_components: u64,
    pub alpha: f64,
    pub tolerance: f64,
    pub center: u8,
    pub verbose: u8,
    pub collect_timings: u8,
    pub reserved0: u8,
    pub method: i32,
    pub n_oversamples: u64,
    pub n_power_iterations: u64,
    pub normalizer: i32,
    pub transform_semantics: i32,
    pub device_id: i32,
    pub spmm_variant: i32,
    pub stream: *mut c_void,
}

extern "C" {
    pub fn sapca_options_default(o: *mut sapca_options);
    pub fn sapca_abi_version() -> c_int;
    pub fn sapca_create(opts: *const sapca_options, out: *mut sapca_handle) -> c_int;
    pub fn sapca_destroy(h: sapca_handle);
    pub fn sapca_last_error(h: sapca_handle) -> *const c_char;
    pub fn sapca_set_mask(h: sapca_handle, mask: *const u8, len: usize) -> c_int;

    // nalgebra_sparse::CsrMatrix<T>: row_offsets()/col_indices() are &[usize] == *const u64 on 64-bit targets
    pub fn sapca_fit_csr_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                             col_indices: *const u64, values: *const f32) -> c_int;
    pub fn sapca_fit_csr_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                             col_indices: *const u64, values: *const f64) -> c_int;
    pub fn sapca_transform_csr_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                   col_indices: *const u64, values: *const f32, out: *mut f32) -> c_int;
    pub fn sapca_transform_csr_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                   col_indices: *const u64, values: *const f64, out: *mut f64) -> c_int;
    pub fn sapca_fit_transform_csr_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                       col_indices: *const u64, values: *const f32, out: *mut f32) -> c_int;
    pub fn sapca_fit_transform_csr_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                       col_indices: *const u64, values: *const f64, out: *mut f64) -> c_int;

    pub fn sapca_get_dims(h: sapca_handle, k: *mut u64, n_used: *mut u64, n_cols: *mut u64) -> c_int;
    pub fn sapca_get_mask_index_maps(h: sapca_handle, cols_to_use: *mut u64, cols_cap: usize, orig_to_masked: *mut i64,
                                     map_cap: usize) -> c_int;
    pub fn sapca_comm_rccl_available() -> c_int;

    // one handle, several GPUs, one calling thread (ABI 3): the host CsrMatrix is split into nnz-balanced row ranges, one
    // member handle and one host thread per device; fitted state is read from any member (sapca_multi_member(mh, 0))
    pub fn sapca_multi_create(opts: *const sapca_options, device_ids: *const i32, n_devices: u32, out: *mut sapca_multi) -> c_int;
    pub fn sapca_multi_destroy(mh: sapca_multi);
    pub fn sapca_multi_last_error(mh: sapca_multi) -> *const c_char;
    pub fn sapca_multi_n_devices(mh: sapca_multi) -> u32;
    pub fn sapca_multi_member(mh: sapca_multi, i: u32) -> sapca_handle;
    pub fn sapca_multi_uses_rccl(mh: sapca_multi) -> c_int;
    pub fn sapca_multi_set_mask(mh: sapca_multi, mask: *const u8, n: usize) -> c_int;
    pub fn sapca_multi_fit_csr_f32(mh: sapca_multi, m: u64, n: u64, nnz: u64, row_offsets: *const u64, col_indices: *const u64,
                                   values: *const f32) -> c_int;
    pub fn sapca_multi_fit_csr_f64(mh: sapca_multi, m: u64, n: u64, nnz: u64, row_offsets: *const u64, col_indices: *const u64,
                                   values: *const f64) -> c_int;
    pub fn sapca_multi_transform_csr_f32(mh: sapca_multi, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                         col_indices: *const u64, values: *const f32, out: *mut f32) -> c_int;
    pub fn sapca_multi_transform_csr_f64(mh: sapca_multi, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                         col_indices: *const u64, values: *const f64, out: *mut f64) -> c_int;
    pub fn sapca_multi_fit_transform_csr_f32(mh: sapca_multi, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                             col_indices: *const u64, values: *const f32, out: *mut f32) -> c_int;
    pub fn sapca_multi_fit_transform_csr_f64(mh: sapca_multi, m: u64, n: u64, nnz: u64, row_offsets: *const u64,
                                             col_indices: *const u64, values: *const f64, out: *mut f64) -> c_int;
    pub fn sapca_get_components_f32(h: sapca_handle, out: *mut f32, cap: usize) -> c_int;
    pub fn sapca_get_components_f64(h: sapca_handle, out: *mut f64, cap: usize) -> c_int;
    pub fn sapca_get_explained_variance_f32(h: sapca_handle, out: *mut f32, cap: usize) -> c_int;
    pub fn sapca_get_explained_variance_f64(h: sapca_handle, out: *mut f64, cap: usize) -> c_int;
    pub fn sapca_get_mean_f32(h: sapca_handle, out: *mut f32, cap: usize) -> c_int;
    pub fn sapca_get_mean_f64(h: sapca_handle, out: *mut f64, cap: usize) -> c_int;
    pub fn sapca_get_explained_variance_ratio_f32(h: sapca_handle, out: *mut f32, cap: usize) -> c_int;
    pub fn sapca_get_explained_variance_ratio_f64(h: sapca_handle, out: *mut f64, cap: usize) -> c_int;
    pub fn sapca_get_cumulative_explained_variance_ratio_f32(h: sapca_handle, out: *mut f32, cap: usize) -> c_int;
    pub fn sapca_get_cumulative_explained_variance_ratio_f64(h: sapca_handle, out: *mut f64, cap: usize) -> c_int;
    pub fn sapca_get_feature_importances_f32(h: sapca_handle, out: *mut f32, cap: usize) -> c_int;
    pub fn sapca_get_feature_importances_f64(h: sapca_handle, out: *mut f64, cap: usize) -> c_int;

    // device-resident workflow: upload once, then Normalize / Log1P / MatrixSum / MatrixNonZero / MatrixMinMax and
    // the fits on the resident copy (src/lib.rs:28-33 of single-algebra: normalize -> log1p -> PCA)
    pub fn sapca_upload_csr_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64, col_indices: *const u64,
                                values: *const f32, d_row_offsets: *mut *const i64, d_col_indices: *mut *const i32,
                                d_values: *mut *mut f32) -> c_int;
    pub fn sapca_upload_csr_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const u64, col_indices: *const u64,
                                values: *const f64, d_row_offsets: *mut *const i64, d_col_indices: *mut *const i32,
                                d_values: *mut *mut f64) -> c_int;
    pub fn sapca_normalize_csr_device_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const i64,
                                          col_indices: *const i32, values: *mut f32, sums: *const f64, sums_len: u64,
                                          target: f64, direction: i32) -> c_int;
    pub fn sapca_normalize_csr_device_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const i64,
                                          col_indices: *const i32, values: *mut f64, sums: *const f64, sums_len: u64,
                                          target: f64, direction: i32) -> c_int;
    pub fn sapca_log1p_csr_device_f32(h: sapca_handle, nnz: u64, values: *mut f32) -> c_int;
    pub fn sapca_log1p_csr_device_f64(h: sapca_handle, nnz: u64, values: *mut f64) -> c_int;
    pub fn sapca_upload_values_changed(h: sapca_handle) -> c_int;
    pub fn sapca_stats_csr_device_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const i64,
                                      col_indices: *const i32, values: *const f32, direction: i32, sum: *mut f64,
                                      sum_squared: *mut f64, nonzero: *mut u64, min_out: *mut f32, max_out: *mut f32) -> c_int;
    pub fn sapca_stats_csr_device_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const i64,
                                      col_indices: *const i32, values: *const f64, direction: i32, sum: *mut f64,
                                      sum_squared: *mut f64, nonzero: *mut u64, min_out: *mut f64, max_out: *mut f64) -> c_int;
    pub fn sapca_fit_transform_csr_device_f32(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const i64,
                                              col_indices: *const i32, values: *const f32, out: *mut f32) -> c_int;
    pub fn sapca_fit_transform_csr_device_f64(h: sapca_handle, m: u64, n: u64, nnz: u64, row_offsets: *const i64,
                                              col_indices: *const i32, values: *const f64, out: *mut f64) -> c_int;
}
