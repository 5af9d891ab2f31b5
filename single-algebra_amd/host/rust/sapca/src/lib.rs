//! Same surface as `single_algebra::dimred::pca` (reference v0.9.2):
//! `SVDMethod`, `PowerIterationNormalizer`, `SparsePCABuilder`/`SparsePCA`,
//! `MaskedSparsePCABuilder`/`MaskedSparsePCA`, `fit` / `transform` / `fit_transform` /
//! `feature_importances` / `explained_variance_ratio` / `cumulative_explained_variance_ratio`,
//! returning `anyhow::Result` with the reference's messages -- the work happens in libsapca.so.
//!
//! UNTESTED SOURCE: written against include/sapca.h, never compiled (no rustc in the build image).
//! Only the f32 side is spelled out; f64 is the same with the `_f64` symbols.
use anyhow::{anyhow, Result};
use nalgebra_sparse::CsrMatrix;
use ndarray::{Array1, Array2};
use sapca_sys as ffi;
use std::ffi::CStr;

#[derive(Debug, Clone, Copy, PartialEq)]
pub enum PowerIterationNormalizer { QR, LU, None }

/// src/dimred/pca/mod.rs:49-68
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum SVDMethod {
    Lanczos,
    Random { n_oversamples: usize, n_power_iterations: usize, normalizer: PowerIterationNormalizer },
}
impl Default for SVDMethod { fn default() -> Self { Self::Lanczos } }

struct Handle(ffi::sapca_handle);
impl Drop for Handle { fn drop(&mut self) { unsafe { ffi::sapca_destroy(self.0) } } }
unsafe impl Send for Handle {}

fn check(h: ffi::sapca_handle, status: i32) -> Result<()> {
    if status == ffi::SAPCA_OK { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(ffi::sapca_last_error(h)) }.to_string_lossy().into_owned();
    Err(anyhow!(msg))   // carries the reference's own strings, e.g. "Must be fitted before transform!"
}

fn create(n_components: usize, alpha: f64, tolerance: f64, seed: u32, center: bool, verbose: bool,
          method: SVDMethod, mask: Option<&[bool]>) -> Result<Handle> {
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
    let mut h: ffi::sapca_handle = std::ptr::null_mut();
    let st = unsafe { ffi::sapca_create(&o, &mut h) };
    if st != ffi::SAPCA_OK { return Err(anyhow!("sapca_create failed with status {st}")); }
    let h = Handle(h);
    if let Some(m) = mask {
        let bytes: Vec<u8> = m.iter().map(|&b| b as u8).collect();
        check(h.0, unsafe { ffi::sapca_set_mask(h.0, bytes.as_ptr(), bytes.len()) })?;
    }
    Ok(h)
}

/// SparsePCA<f32> (sparse/mod.rs:33-359)
pub struct SparsePCA { h: Handle, n_components: usize }

impl SparsePCA {
    pub fn fit(&mut self, x: &CsrMatrix<f32>) -> Result<&mut Self> {
        let (ro, ci, v) = (x.row_offsets(), x.col_indices(), x.values());
        check(self.h.0, unsafe {
            ffi::sapca_fit_csr_f32(self.h.0, x.nrows() as u64, x.ncols() as u64, v.len() as u64,
                                   ro.as_ptr() as *const u64, ci.as_ptr() as *const u64, v.as_ptr())
        })?;
        Ok(self)
    }
    pub fn transform(&self, x: &CsrMatrix<f32>) -> Result<Array2<f32>> {
        let mut out = Array2::<f32>::zeros((x.nrows(), self.n_components));
        let (ro, ci, v) = (x.row_offsets(), x.col_indices(), x.values());
        check(self.h.0, unsafe {
            ffi::sapca_transform_csr_f32(self.h.0, x.nrows() as u64, x.ncols() as u64, v.len() as u64,
                                         ro.as_ptr() as *const u64, ci.as_ptr() as *const u64, v.as_ptr(),
                                         out.as_mut_ptr())
        })?;
        Ok(out)
    }
    pub fn fit_transform(&mut self, x: &CsrMatrix<f32>) -> Result<Array2<f32>> {
        let mut out = Array2::<f32>::zeros((x.nrows(), self.n_components));
        let (ro, ci, v) = (x.row_offsets(), x.col_indices(), x.values());
        check(self.h.0, unsafe {
            ffi::sapca_fit_transform_csr_f32(self.h.0, x.nrows() as u64, x.ncols() as u64, v.len() as u64,
                                             ro.as_ptr() as *const u64, ci.as_ptr() as *const u64, v.as_ptr(),
                                             out.as_mut_ptr())
        })?;
        Ok(out)
    }
    fn dims(&self) -> Result<(usize, usize)> {
        let (mut k, mut nu, mut nc) = (0u64, 0u64, 0u64);
        check(self.h.0, unsafe { ffi::sapca_get_dims(self.h.0, &mut k, &mut nu, &mut nc) })?;
        Ok((k as usize, nu as usize))
    }
    pub fn feature_importances(&self) -> Result<Array2<f32>> {
        let (k, nu) = self.dims()?;
        let mut out = Array2::<f32>::zeros((k, nu));
        check(self.h.0, unsafe { ffi::sapca_get_feature_importances_f32(self.h.0, out.as_mut_ptr(), k * nu) })?;
        Ok(out)
    }
    pub fn explained_variance_ratio(&self) -> Result<Array1<f32>> {
        let (k, _) = self.dims()?;
        let mut out = Array1::<f32>::zeros(k);
        check(self.h.0, unsafe { ffi::sapca_get_explained_variance_ratio_f32(self.h.0, out.as_mut_ptr(), k) })?;
        Ok(out)
    }
    pub fn cumulative_explained_variance_ratio(&self) -> Result<Array1<f32>> {
        let (k, _) = self.dims()?;
        let mut out = Array1::<f32>::zeros(k);
        check(self.h.0, unsafe { ffi::sapca_get_cumulative_explained_variance_ratio_f32(self.h.0, out.as_mut_ptr(), k) })?;
        Ok(out)
    }
}

/// SparsePCABuilder<f32> (sparse/mod.rs:375-484; defaults :392-401)
pub struct SparsePCABuilder {
    n_components: usize, alpha: f32, tolerance: f32, random_seed: Option<u32>, center: bool, verbose: bool,
    svdmethod: SVDMethod,
}
impl Default for SparsePCABuilder {
    fn default() -> Self {
        Self { n_components: 50, alpha: 1.0, tolerance: 1e-6, random_seed: Some(42), center: true, verbose: false,
               svdmethod: SVDMethod::default() }
    }
}
impl SparsePCABuilder {
    pub fn new() -> Self { Self::default() }
    pub fn n_components(mut self, n: usize) -> Self { self.n_components = n; self }
    pub fn alpha(mut self, a: f32) -> Self { self.alpha = a; self }
    pub fn tolerance(mut self, t: f32) -> Self { self.tolerance = t; self }
    pub fn random_seed(mut self, s: u32) -> Self { self.random_seed = Some(s); self }
    pub fn center(mut self, c: bool) -> Self { self.center = c; self }
    pub fn verbose(mut self, v: bool) -> Self { self.verbose = v; self }
    pub fn svd_method(mut self, m: SVDMethod) -> Self { self.svdmethod = m; self }
    /// The reference's `build()` is infallible; creating the GPU handle is not, hence the Result.
    pub fn build(self) -> Result<SparsePCA> {
        let h = create(self.n_components, self.alpha as f64, self.tolerance as f64, self.random_seed.unwrap_or(42),
                       self.center, self.verbose, self.svdmethod, None)?;
        Ok(SparsePCA { h, n_components: self.n_components })
    }
}

/// MaskedSparsePCA<f32> (sparse_masked/mod.rs:179-620): same calls on a handle that carries the mask.
/// The reference rejects ANY mask/column-count mismatch, including an empty mask (:258-262); the C ABI
/// treats an empty mask as "no mask", so that check lives here.
pub struct MaskedSparsePCA { inner: SparsePCA, mask_len: usize }
impl MaskedSparsePCA {
    fn check_mask(&self, x: &CsrMatrix<f32>) -> Result<()> {
        if x.ncols() != self.mask_len {
            return Err(anyhow!("The mask vector length and the number of features (columns) have to be the same!"));
        }
        Ok(())
    }
    pub fn fit(&mut self, x: &CsrMatrix<f32>) -> Result<&mut Self> { self.check_mask(x)?; self.inner.fit(x)?; Ok(self) }
    pub fn transform(&self, x: &CsrMatrix<f32>) -> Result<Array2<f32>> { self.check_mask(x)?; self.inner.transform(x) }
    pub fn fit_transform(&mut self, x: &CsrMatrix<f32>) -> Result<Array2<f32>> { self.check_mask(x)?; self.inner.fit_transform(x) }
    pub fn feature_importances(&self) -> Result<Array2<f32>> { self.inner.feature_importances() }
    pub fn explained_variance_ratio(&self) -> Result<Array1<f32>> { self.inner.explained_variance_ratio() }
    pub fn cumulative_explained_variance_ratio(&self) -> Result<Array1<f32>> { self.inner.cumulative_explained_variance_ratio() }
}

/// MaskedSparsePCABuilder<f32> (sparse_masked/mod.rs:37-160)
pub struct MaskedSparsePCABuilder { base: SparsePCABuilder, mask: Vec<bool> }
impl MaskedSparsePCABuilder {
    pub fn new() -> Self { Self { base: SparsePCABuilder::default(), mask: Vec::new() } }
    pub fn n_components(mut self, n: usize) -> Self { self.base = self.base.n_components(n); self }
    pub fn alpha(mut self, a: f32) -> Self { self.base = self.base.alpha(a); self }
    pub fn tolerance(mut self, t: f32) -> Self { self.base = self.base.tolerance(t); self }
    pub fn random_seed(mut self, s: u32) -> Self { self.base = self.base.random_seed(s); self }
    pub fn center(mut self, c: bool) -> Self { self.base = self.base.center(c); self }
    pub fn verbose(mut self, v: bool) -> Self { self.base = self.base.verbose(v); self }
    pub fn svd_method(mut self, m: SVDMethod) -> Self { self.base = self.base.svd_method(m); self }
    pub fn mask(mut self, mask: Vec<bool>) -> Self { self.mask = mask; self }
    pub fn build(self) -> Result<MaskedSparsePCA> {
        let b = self.base;
        let h = create(b.n_components, b.alpha as f64, b.tolerance as f64, b.random_seed.unwrap_or(42), b.center,
                       b.verbose, b.svdmethod, Some(&self.mask))?;
        Ok(MaskedSparsePCA { inner: SparsePCA { h, n_components: b.n_components }, mask_len: self.mask.len() })
    }
}
