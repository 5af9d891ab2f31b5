//! Same surface as `single_algebra::dimred::pca` (reference v0.9.2):
//! `SVDMethod`, `PowerIterationNormalizer`, `SparsePCABuilder<T>`/`SparsePCA<T>`,
//! `MaskedSparsePCABuilder<T>`/`MaskedSparsePCA<T>`, `fit` / `transform` / `fit_transform` /
//! `feature_importances` / `explained_variance_ratio` / `cumulative_explained_variance_ratio`,
//! returning `anyhow::Result` with the reference's messages -- the work happens in libsapca.so.
//!
//! The reference types are generic over `T: SvdFloat + FloatOpsTS + RealField + ...`
//! (src/dimred/pca/sparse/mod.rs:33-36, sparse_masked/mod.rs:179-182); in practice T is f32 or f64.
//! Here T is bounded by the sealed trait `SapcaFloat`, implemented for exactly those two: it carries the
//! `_f32` / `_f64` entry points of include/sapca.h as associated functions.
//!
//! `build()` is infallible, as in the reference (sparse/mod.rs:470-483): a builder only records its parameters; the GPU
//! handle is created by the first `fit` / `fit_transform`, whose `Result` carries a creation failure.  The README's
//! examples (README.md:48-96) compile against this crate with the `use` paths changed from `single_algebra::` to
//! `sapca::` (the module tree `dimred::pca::{sparse, sparse_masked}` is mirrored below).
//!
//! UNTESTED SOURCE: written against include/sapca.h, never compiled (no rustc in the build image).
// `CsrMatrix::row_offsets()` / `col_indices()` are `&[usize]` and cross the FFI as `*const u64` (include/sapca.h takes the
// reference's index arrays zero-copy): only sound where usize IS 64 bits wide.
#[cfg(not(target_pointer_width = "64"))]
compile_error!("sapca passes nalgebra_sparse's usize index arrays to libsapca as u64: 64-bit targets only");

use anyhow::{anyhow, Result};
use nalgebra_sparse::CsrMatrix;
use ndarray::{Array1, Array2};
use sapca_sys as ffi;
use std::ffi::CStr;
use std::marker::PhantomData;

#[derive(Debug, Clone, Copy, PartialEq)]
pub enum PowerIterationNormalizer { QR, LU, None }

/// src/dimred/pca/mod.rs:49-68
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum SVDMethod {
    Lanczos,
    Random { n_oversamples: usize, n_power_iterations: usize, normalizer: PowerIterationNormalizer },
}
impl Default for SVDMethod { fn default() -> Self { Self::Lanczos } }

mod sealed { pub trait Sealed {} impl Sealed for f32 {} impl Sealed for f64 {} }

/// The value types libsapca is built for.  One associated function per typed entry point of the C ABI.
pub trait SapcaFloat: sealed::Sealed + Copy + Default + num_traits::Zero + 'static {
    fn to_f64(self) -> f64;
    unsafe fn fit(h: ffi::sapca_handle, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self) -> i32;
    unsafe fn transform(h: ffi::sapca_handle, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self,
                        out: *mut Self) -> i32;
    unsafe fn fit_transform(h: ffi::sapca_handle, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self,
                            out: *mut Self) -> i32;
    // the same three calls on a sapca_multi (one handle, several GPUs: the host matrix is sharded by rows inside the library)
    unsafe fn multi_fit(mh: ffi::sapca_multi, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self) -> i32;
    unsafe fn multi_transform(mh: ffi::sapca_multi, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self,
                              out: *mut Self) -> i32;
    unsafe fn multi_fit_transform(mh: ffi::sapca_multi, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self,
                                  out: *mut Self) -> i32;
    unsafe fn feature_importances(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32;
    unsafe fn explained_variance_ratio(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32;
    unsafe fn cumulative_explained_variance_ratio(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32;
    unsafe fn components(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32;
    unsafe fn explained_variance(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32;
    unsafe fn mean(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32;
}

macro_rules! impl_sapca_float {
    ($t:ty, $fit:ident, $transform:ident, $fit_transform:ident, $mfit:ident, $mtransform:ident, $mfit_transform:ident, $fi:ident, $evr:ident, $cevr:ident, $comp:ident, $ev:ident, $mean:ident) => {
        impl SapcaFloat for $t {
            fn to_f64(self) -> f64 { self as f64 }
            unsafe fn fit(h: ffi::sapca_handle, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self) -> i32 {
                ffi::$fit(h, m, n, nnz, ro, ci, v)
            }
            unsafe fn transform(h: ffi::sapca_handle, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self,
                                out: *mut Self) -> i32 {
                ffi::$transform(h, m, n, nnz, ro, ci, v, out)
            }
            unsafe fn fit_transform(h: ffi::sapca_handle, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64,
                                    v: *const Self, out: *mut Self) -> i32 {
                ffi::$fit_transform(h, m, n, nnz, ro, ci, v, out)
            }
            unsafe fn multi_fit(mh: ffi::sapca_multi, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64, v: *const Self) -> i32 {
                ffi::$mfit(mh, m, n, nnz, ro, ci, v)
            }
            unsafe fn multi_transform(mh: ffi::sapca_multi, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64,
                                      v: *const Self, out: *mut Self) -> i32 {
                ffi::$mtransform(mh, m, n, nnz, ro, ci, v, out)
            }
            unsafe fn multi_fit_transform(mh: ffi::sapca_multi, m: u64, n: u64, nnz: u64, ro: *const u64, ci: *const u64,
                                          v: *const Self, out: *mut Self) -> i32 {
                ffi::$mfit_transform(mh, m, n, nnz, ro, ci, v, out)
            }
            unsafe fn feature_importances(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32 { ffi::$fi(h, out, cap) }
            unsafe fn explained_variance_ratio(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32 { ffi::$evr(h, out, cap) }
            unsafe fn cumulative_explained_variance_ratio(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32 {
                ffi::$cevr(h, out, cap)
            }
            unsafe fn components(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32 { ffi::$comp(h, out, cap) }
            unsafe fn explained_variance(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32 { ffi::$ev(h, out, cap) }
            unsafe fn mean(h: ffi::sapca_handle, out: *mut Self, cap: usize) -> i32 { ffi::$mean(h, out, cap) }
        }
    };
}
impl_sapca_float!(f32, sapca_fit_csr_f32, sapca_transform_csr_f32, sapca_fit_transform_csr_f32, sapca_multi_fit_csr_f32,
                  sapca_multi_transform_csr_f32, sapca_multi_fit_transform_csr_f32, sapca_get_feature_importances_f32,
                  sapca_get_explained_variance_ratio_f32, sapca_get_cumulative_explained_variance_ratio_f32,
                  sapca_get_components_f32, sapca_get_explained_variance_f32, sapca_get_mean_f32);
impl_sapca_float!(f64, sapca_fit_csr_f64, sapca_transform_csr_f64, sapca_fit_transform_csr_f64, sapca_multi_fit_csr_f64,
                  sapca_multi_transform_csr_f64, sapca_multi_fit_transform_csr_f64, sapca_get_feature_importances_f64,
                  sapca_get_explained_variance_ratio_f64, sapca_get_cumulative_explained_variance_ratio_f64,
                  sapca_get_components_f64, sapca_get_explained_variance_f64, sapca_get_mean_f64);

/// `.0`: the handle the getters read (with several devices: member 0 -- the fitted state is replicated on all members);
/// `.1`: the sapca_multi that owns the members, null for a single device.
struct Handle(ffi::sapca_handle, ffi::sapca_multi);
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { if self.1.is_null() { ffi::sapca_destroy(self.0) } else { ffi::sapca_multi_destroy(self.1) } }
    }
}
unsafe impl Send for Handle {}

fn check_multi(mh: ffi::sapca_multi, status: i32) -> Result<()> {
    if status == ffi::SAPCA_OK { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(ffi::sapca_multi_last_error(mh)) }.to_string_lossy().into_owned();
    Err(anyhow!(msg))
}

fn check(h: ffi::sapca_handle, status: i32) -> Result<()> {
    if status == ffi::SAPCA_OK { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(ffi::sapca_last_error(h)) }.to_string_lossy().into_owned();
    Err(anyhow!(msg))   // carries the reference's own strings, e.g. "Must be fitted before transform!"
}

fn create(n_components: usize, alpha: f64, tolerance: f64, seed: u32, center: bool, verbose: bool,
          method: SVDMethod, mask: Option<&[bool]>, devices: &[i32]) -> Result<Handle> {
    let mut o: ffi::sapca_options = unsafe { std::mem::zeroed() };
    unsafe { ffi::sapca_options_default(&mut o) };
    o.n_components = n_components as u64;
    o.alpha = alpha; o.tolerance = tolerance; o.random_seed = seed;
    o.center = center as u8; o.verbose = verbose as u8;
    match method {
        SVDMethod::Lanczos => o.method = ffi::SAPCA_LANCZOS,
        SVDMethod::Random { n_oversamples, n_power_iterations, normalizer } => {
            o.method = ffi::SAPCA_RANDOM;
            o.n_oversamples = n_oversamples as u64;
            o.n_power_iterations = n_power_iterations as u64;
            o.normalizer = match normalizer {
                PowerIterationNormalizer::QR => ffi::SAPCA_NORM_QR,
                PowerIterationNormalizer::LU => ffi::SAPCA_NORM_LU,
                PowerIterationNormalizer::None => ffi::SAPCA_NORM_NONE,
            };
        }
    }
    if devices.len() > 1 {
        // one estimator, several GPUs: rows of the CsrMatrix are range-partitioned inside the library
        let mut mh: ffi::sapca_multi = std::ptr::null_mut();
        let st = unsafe { ffi::sapca_multi_create(&o, devices.as_ptr(), devices.len() as u32, &mut mh) };
        if st != ffi::SAPCA_OK {
            let msg = unsafe { CStr::from_ptr(ffi::sapca_multi_last_error(std::ptr::null_mut())) }.to_string_lossy().into_owned();
            return Err(anyhow!("sapca_multi_create failed with status {st}: {msg}"));
        }
        let h = Handle(unsafe { ffi::sapca_multi_member(mh, 0) }, mh);
        if let Some(m) = mask {
            let bytes: Vec<u8> = m.iter().map(|&b| b as u8).collect();
            check_multi(mh, unsafe { ffi::sapca_multi_set_mask(mh, bytes.as_ptr(), bytes.len()) })?;
        }
        return Ok(h);
    }
    if let Some(&d) = devices.first() { o.device_id = d; }
    let mut h: ffi::sapca_handle = std::ptr::null_mut();
    let st = unsafe { ffi::sapca_create(&o, &mut h) };
    if st != ffi::SAPCA_OK { return Err(anyhow!("sapca_create failed with status {st}")); }
    let h = Handle(h, std::ptr::null_mut());
    if let Some(m) = mask {
        let bytes: Vec<u8> = m.iter().map(|&b| b as u8).collect();
        check(h.0, unsafe { ffi::sapca_set_mask(h.0, bytes.as_ptr(), bytes.len()) })?;
    }
    Ok(h)
}

/// What a builder records (sparse/mod.rs:37-46: the estimator's own fields); the handle is made from it on first use.
#[derive(Clone)]
struct Config {
    n_components: usize, alpha: f64, tolerance: f64, random_seed: u32, center: bool, verbose: bool,
    svdmethod: SVDMethod, mask: Option<Vec<bool>>, devices: Vec<i32>,
}
impl Config {
    fn create(&self) -> Result<Handle> {
        create(self.n_components, self.alpha, self.tolerance, self.random_seed, self.center, self.verbose, self.svdmethod,
               self.mask.as_deref(), &self.devices)
    }
}

/// SparsePCA<T> (sparse/mod.rs:33-359)
pub struct SparsePCA<T: SapcaFloat> { cfg: Config, h: Option<Handle>, _t: PhantomData<T> }

impl<T: SapcaFloat> SparsePCA<T> {
    /// sparse/mod.rs:63-84 (same argument order, `tollerance` spelled as there)
    pub fn new(n_components: usize, alpha: T, tollerance: Option<T>, random_seed: Option<u32>, center: bool, verbose: bool,
               svdmethod: SVDMethod) -> Self {
        Self::from_config(Config { n_components, alpha: alpha.to_f64(), tolerance: tollerance.map(|t| t.to_f64()).unwrap_or(1e-6),
                                   random_seed: random_seed.unwrap_or(42), center, verbose, svdmethod, mask: None, devices: Vec::new() })
    }
    fn from_config(cfg: Config) -> Self { Self { cfg, h: None, _t: PhantomData } }
    /// the handle, created on the first fit (a creation failure -- no GPU, no memory -- surfaces in that call's Result)
    fn handle_mut(&mut self) -> Result<&Handle> {
        if self.h.is_none() { self.h = Some(self.cfg.create()?); }
        Ok(self.h.as_ref().unwrap())
    }
    /// for the `&self` methods: no handle yet means nothing was fitted -- the reference's own messages
    fn handle(&self, msg: &'static str) -> Result<&Handle> { self.h.as_ref().ok_or_else(|| anyhow!(msg)) }
    /// sparse/mod.rs:102-242
    pub fn fit(&mut self, x: &CsrMatrix<T>) -> Result<&mut Self> {
        let (ro, ci, v) = (x.row_offsets(), x.col_indices(), x.values());
        let h = self.handle_mut()?;
        if !h.1.is_null() {
            check_multi(h.1, unsafe {
                T::multi_fit(h.1, x.nrows() as u64, x.ncols() as u64, v.len() as u64, ro.as_ptr() as *const u64,
                             ci.as_ptr() as *const u64, v.as_ptr())
            })?;
            return Ok(self);
        }
        check(h.0, unsafe {
            T::fit(h.0, x.nrows() as u64, x.ncols() as u64, v.len() as u64, ro.as_ptr() as *const u64,
                   ci.as_ptr() as *const u64, v.as_ptr())
        })?;
        Ok(self)
    }
    /// sparse/mod.rs:255-285 (the count-weighted projection as written there is the default; see include/sapca.h)
    pub fn transform(&self, x: &CsrMatrix<T>) -> Result<Array2<T>> {
        let h = self.handle("Must be fitted before transform!")?;   // sparse/mod.rs:259,263
        let mut out = Array2::<T>::zeros((x.nrows(), self.cfg.n_components));
        let (ro, ci, v) = (x.row_offsets(), x.col_indices(), x.values());
        if !h.1.is_null() {
            check_multi(h.1, unsafe {
                T::multi_transform(h.1, x.nrows() as u64, x.ncols() as u64, v.len() as u64, ro.as_ptr() as *const u64,
                                   ci.as_ptr() as *const u64, v.as_ptr(), out.as_mut_ptr())
            })?;
            return Ok(out);
        }
        check(h.0, unsafe {
            T::transform(h.0, x.nrows() as u64, x.ncols() as u64, v.len() as u64, ro.as_ptr() as *const u64,
                         ci.as_ptr() as *const u64, v.as_ptr(), out.as_mut_ptr())
        })?;
        Ok(out)
    }
    /// sparse/mod.rs:355-358
    pub fn fit_transform(&mut self, x: &CsrMatrix<T>) -> Result<Array2<T>> {
        let mut out = Array2::<T>::zeros((x.nrows(), self.cfg.n_components));
        let (ro, ci, v) = (x.row_offsets(), x.col_indices(), x.values());
        let h = self.handle_mut()?;
        if !h.1.is_null() {
            check_multi(h.1, unsafe {
                T::multi_fit_transform(h.1, x.nrows() as u64, x.ncols() as u64, v.len() as u64, ro.as_ptr() as *const u64,
                                       ci.as_ptr() as *const u64, v.as_ptr(), out.as_mut_ptr())
            })?;
            return Ok(out);
        }
        check(h.0, unsafe {
            T::fit_transform(h.0, x.nrows() as u64, x.ncols() as u64, v.len() as u64, ro.as_ptr() as *const u64,
                             ci.as_ptr() as *const u64, v.as_ptr(), out.as_mut_ptr())
        })?;
        Ok(out)
    }
    fn dims(&self) -> Result<(ffi::sapca_handle, usize, usize, usize)> {
        let h = self.handle("Model must be fitted first!")?.0;   // sparse/mod.rs:299,316
        let (mut k, mut nu, mut nc) = (0u64, 0u64, 0u64);
        check(h, unsafe { ffi::sapca_get_dims(h, &mut k, &mut nu, &mut nc) })?;
        Ok((h, k as usize, nu as usize, nc as usize))
    }
    /// sparse/mod.rs:295-302
    pub fn feature_importances(&self) -> Result<Array2<T>> {
        let (h, k, nu, _) = self.dims()?;
        let mut out = Array2::<T>::zeros((k, nu));
        check(h, unsafe { T::feature_importances(h, out.as_mut_ptr(), k * nu) })?;
        Ok(out)
    }
    /// sparse/mod.rs:312-322
    pub fn explained_variance_ratio(&self) -> Result<Array1<T>> {
        let (h, k, _, _) = self.dims()?;
        let mut out = Array1::<T>::zeros(k);
        check(h, unsafe { T::explained_variance_ratio(h, out.as_mut_ptr(), k) })?;
        Ok(out)
    }
    /// sparse/mod.rs:333-343
    pub fn cumulative_explained_variance_ratio(&self) -> Result<Array1<T>> {
        let (h, k, _, _) = self.dims()?;
        let mut out = Array1::<T>::zeros(k);
        check(h, unsafe { T::cumulative_explained_variance_ratio(h, out.as_mut_ptr(), k) })?;
        Ok(out)
    }
    /// The fields the reference keeps private (sparse/mod.rs:41-43), for callers that need them.
    pub fn components(&self) -> Result<Array2<T>> {
        let (h, k, nu, _) = self.dims()?;
        let mut out = Array2::<T>::zeros((k, nu));
        check(h, unsafe { T::components(h, out.as_mut_ptr(), k * nu) })?;
        Ok(out)
    }
    pub fn explained_variance(&self) -> Result<Array1<T>> {
        let (h, k, _, _) = self.dims()?;
        let mut out = Array1::<T>::zeros(k);
        check(h, unsafe { T::explained_variance(h, out.as_mut_ptr(), k) })?;
        Ok(out)
    }
    pub fn mean(&self) -> Result<Array1<T>> {
        let (h, _, _, nc) = self.dims()?;
        let mut out = Array1::<T>::zeros(nc);
        check(h, unsafe { T::mean(h, out.as_mut_ptr(), nc) })?;
        Ok(out)
    }
}

/// SparsePCABuilder<T> (sparse/mod.rs:375-484; defaults :392-401)
pub struct SparsePCABuilder<T: SapcaFloat> {
    n_components: usize, alpha: f64, tolerance: f64, random_seed: Option<u32>, center: bool, verbose: bool,
    svdmethod: SVDMethod, devices: Vec<i32>, _t: PhantomData<T>,
}
impl<T: SapcaFloat> Default for SparsePCABuilder<T> {
    fn default() -> Self {
        Self { n_components: 50, alpha: 1.0, tolerance: 1e-6, random_seed: Some(42), center: true, verbose: false,
               svdmethod: SVDMethod::default(), devices: Vec::new(), _t: PhantomData }
    }
}
impl<T: SapcaFloat> SparsePCABuilder<T> {
    pub fn new() -> Self { Self::default() }
    pub fn n_components(mut self, n: usize) -> Self { self.n_components = n; self }
    pub fn alpha(mut self, a: T) -> Self { self.alpha = a.to_f64(); self }
    pub fn tolerance(mut self, t: T) -> Self { self.tolerance = t.to_f64(); self }
    pub fn random_seed(mut self, s: u32) -> Self { self.random_seed = Some(s); self }
    pub fn center(mut self, c: bool) -> Self { self.center = c; self }
    pub fn verbose(mut self, v: bool) -> Self { self.verbose = v; self }
    pub fn svd_method(mut self, m: SVDMethod) -> Self { self.svdmethod = m; self }
    /// Extension (not in the reference): the HIP devices to run on.  Empty = the current device; one = that device; several =
    /// the rows of every matrix passed to fit / transform / fit_transform are range-partitioned over them inside the
    /// library (include/sapca.h, sapca_multi_*), the call itself unchanged.
    pub fn devices(mut self, d: Vec<i32>) -> Self { self.devices = d; self }
    fn config(&self, mask: Option<Vec<bool>>) -> Config {
        Config { n_components: self.n_components, alpha: self.alpha, tolerance: self.tolerance,
                 random_seed: self.random_seed.unwrap_or(42), center: self.center, verbose: self.verbose,
                 svdmethod: self.svdmethod, mask, devices: self.devices.clone() }
    }
    /// sparse/mod.rs:470-483: infallible, records the parameters (the handle is created by the first fit)
    pub fn build(self) -> SparsePCA<T> { SparsePCA::from_config(self.config(None)) }
}

/// MaskedSparsePCA<T> (sparse_masked/mod.rs:179-620): same calls on a handle that carries the mask.
/// The reference rejects ANY mask/column-count mismatch, including an empty mask (:258-262); the C ABI
/// treats an empty mask as "no mask", so that check lives here.
pub struct MaskedSparsePCA<T: SapcaFloat> { inner: SparsePCA<T>, mask_len: usize }
impl<T: SapcaFloat> MaskedSparsePCA<T> {
    /// sparse_masked/mod.rs:214-237 (same argument order)
    pub fn new(n_components: usize, alpha: T, tollerance: Option<T>, random_seed: Option<u32>, mask: Vec<bool>, center: bool,
               verbose: bool, svd_method: SVDMethod) -> Self {
        let mask_len = mask.len();
        let cfg = Config { n_components, alpha: alpha.to_f64(), tolerance: tollerance.map(|t| t.to_f64()).unwrap_or(1e-6),
                           random_seed: random_seed.unwrap_or(42), center, verbose, svdmethod: svd_method, mask: Some(mask),
                           devices: Vec::new() };
        Self { inner: SparsePCA::from_config(cfg), mask_len }
    }
    fn check_mask(&self, x: &CsrMatrix<T>) -> Result<()> {
        if x.ncols() != self.mask_len {
            return Err(anyhow!("The mask vector length and the number of features (columns) have to be the same!"));
        }
        Ok(())
    }
    pub fn fit(&mut self, x: &CsrMatrix<T>) -> Result<&mut Self> { self.check_mask(x)?; self.inner.fit(x)?; Ok(self) }
    /// sparse_masked/mod.rs:438-546: the mask-length check comes first (:440-444), then "Must be fitted before transform!"
    pub fn transform(&self, x: &CsrMatrix<T>) -> Result<Array2<T>> { self.check_mask(x)?; self.inner.transform(x) }
    pub fn fit_transform(&mut self, x: &CsrMatrix<T>) -> Result<Array2<T>> { self.check_mask(x)?; self.inner.fit_transform(x) }
    pub fn feature_importances(&self) -> Result<Array2<T>> { self.inner.feature_importances() }
    pub fn explained_variance_ratio(&self) -> Result<Array1<T>> { self.inner.explained_variance_ratio() }
    pub fn cumulative_explained_variance_ratio(&self) -> Result<Array1<T>> { self.inner.cumulative_explained_variance_ratio() }
    /// `cols_to_use` and the original -> masked map of sparse_masked/mod.rs:264-271, :462-466 (exact integers)
    pub fn mask_index_maps(&self) -> Result<(Vec<usize>, Vec<i64>)> {
        let (h, _, nu, nc) = self.inner.dims()?;
        let mut cols = vec![0u64; nu.max(1)];
        let mut o2m = vec![0i64; nc.max(1)];
        check(h, unsafe { ffi::sapca_get_mask_index_maps(h, cols.as_mut_ptr(), cols.len(), o2m.as_mut_ptr(), o2m.len()) })?;
        cols.truncate(nu);
        o2m.truncate(nc);
        Ok((cols.into_iter().map(|c| c as usize).collect(), o2m))
    }
}

/// MaskedSparsePCABuilder<T> (sparse_masked/mod.rs:37-160)
pub struct MaskedSparsePCABuilder<T: SapcaFloat> { base: SparsePCABuilder<T>, mask: Vec<bool> }
impl<T: SapcaFloat> Default for MaskedSparsePCABuilder<T> {
    fn default() -> Self { Self { base: SparsePCABuilder::default(), mask: Vec::new() } }
}
impl<T: SapcaFloat> MaskedSparsePCABuilder<T> {
    pub fn new() -> Self { Self::default() }
    pub fn n_components(mut self, n: usize) -> Self { self.base = self.base.n_components(n); self }
    pub fn alpha(mut self, a: T) -> Self { self.base = self.base.alpha(a); self }
    pub fn tolerance(mut self, t: T) -> Self { self.base = self.base.tolerance(t); self }
    pub fn random_seed(mut self, s: u32) -> Self { self.base = self.base.random_seed(s); self }
    pub fn center(mut self, c: bool) -> Self { self.base = self.base.center(c); self }
    pub fn verbose(mut self, v: bool) -> Self { self.base = self.base.verbose(v); self }
    pub fn svd_method(mut self, m: SVDMethod) -> Self { self.base = self.base.svd_method(m); self }
    pub fn mask(mut self, mask: Vec<bool>) -> Self { self.mask = mask; self }
    pub fn devices(mut self, d: Vec<i32>) -> Self { self.base = self.base.devices(d); self }
    /// sparse_masked/mod.rs:146-160: infallible
    pub fn build(self) -> MaskedSparsePCA<T> {
        let mask_len = self.mask.len();
        let cfg = self.base.config(Some(self.mask));
        MaskedSparsePCA { inner: SparsePCA::from_config(cfg), mask_len }
    }
}

/// The reference's module tree (src/dimred/pca/mod.rs:36-42; README.md:51-53 imports
/// `dimred::pca::{SparsePCABuilder, SVDMethod}` and `dimred::pca::sparse::PowerIterationNormalizer`).
pub mod dimred {
    pub mod pca {
        pub use crate::{MaskedSparsePCA, MaskedSparsePCABuilder, PowerIterationNormalizer, SVDMethod, SparsePCA, SparsePCABuilder};
        pub mod sparse { pub use crate::{PowerIterationNormalizer, SparsePCA, SparsePCABuilder}; }
        pub mod sparse_masked { pub use crate::{MaskedSparsePCA, MaskedSparsePCABuilder}; }
    }
}

/// Type aliases for callers that used the two instantiations by name.
pub type SparsePCAf32 = SparsePCA<f32>;
pub type SparsePCAf64 = SparsePCA<f64>;
