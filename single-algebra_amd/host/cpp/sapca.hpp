// sapca.hpp -- C++ mirror of single-algebra's sparse-PCA API over the C ABI (include/sapca.h).
// Same names and argument meaning as the reference (src/dimred/pca: SVDMethod, SparsePCABuilder,
// MaskedSparsePCABuilder, fit / transform / fit_transform / feature_importances /
// explained_variance_ratio / cumulative_explained_variance_ratio); errors surface as sapca::Error
// carrying the reference's messages.  Header-only; link with -lsapca.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/sapca.h"

namespace sapca {

struct Error : std::runtime_error {
  sapca_status status;
  Error(sapca_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};

enum class PowerIterationNormalizer { QR = SAPCA_NORM_QR, LU = SAPCA_NORM_LU, None = SAPCA_NORM_NONE };

// enum SVDMethod { Lanczos, Random { n_oversamples, n_power_iterations, normalizer } }   (pca/mod.rs:49-68)
struct SVDMethod {
  bool random = false;
  size_t n_oversamples = 10, n_power_iterations = 4;
  PowerIterationNormalizer normalizer = PowerIterationNormalizer::QR;
  static SVDMethod Lanczos() { return {}; }
  static SVDMethod Random(size_t p, size_t q, PowerIterationNormalizer n = PowerIterationNormalizer::QR) { return {true, p, q, n}; }
};

// nalgebra_sparse::CsrMatrix<T> as three borrowed arrays (usize == uint64_t indices)
template <typename T>
struct CsrRef {
  uint64_t nrows, ncols, nnz;
  const uint64_t* row_offsets;
  const uint64_t* col_indices;
  const T* values;
};

template <typename T> struct Abi;
#define SAPCA_ABI(SUF, T)                                                                                        \
  template <> struct Abi<T> {                                                                                    \
    static sapca_status fit(sapca_handle h, const CsrRef<T>& x) { return sapca_fit_csr_##SUF(h, x.nrows, x.ncols, x.nnz, x.row_offsets, x.col_indices, x.values); } \
    static sapca_status transform(sapca_handle h, const CsrRef<T>& x, T* o) { return sapca_transform_csr_##SUF(h, x.nrows, x.ncols, x.nnz, x.row_offsets, x.col_indices, x.values, o); } \
    static sapca_status fit_transform(sapca_handle h, const CsrRef<T>& x, T* o) { return sapca_fit_transform_csr_##SUF(h, x.nrows, x.ncols, x.nnz, x.row_offsets, x.col_indices, x.values, o); } \
    static sapca_status importances(sapca_handle h, T* o, size_t c) { return sapca_get_feature_importances_##SUF(h, o, c); } \
    static sapca_status ratio(sapca_handle h, T* o, size_t c) { return sapca_get_explained_variance_ratio_##SUF(h, o, c); } \
    static sapca_status cumulative(sapca_handle h, T* o, size_t c) { return sapca_get_cumulative_explained_variance_ratio_##SUF(h, o, c); } \
  };
SAPCA_ABI(f32, float)
SAPCA_ABI(f64, double)
#undef SAPCA_ABI

// Direction of the reference's Normalize / statistics traits (single-algebra src/utils.rs)
enum class Direction : int32_t { ROW = 0, COLUMN = 1 };

// A CsrMatrix uploaded once into buffers owned by a handle: Normalize / Log1P / MatrixSum / MatrixNonZero /
// MatrixMinMax (src/sparse/csr.rs:23-134, 259-392, 558-630, 917-1078) run on the resident copy and the device
// entry points of the estimators take the same arrays (src/lib.rs:28-33: normalize -> log1p -> PCA).
template <typename T> struct ResidentAbi;
#define SAPCA_RES(SUF, T)                                                                                          \
  template <> struct ResidentAbi<T> {                                                                              \
    static sapca_status upload(sapca_handle h, const CsrRef<T>& x, const int64_t** p, const int32_t** i, T** v) { return sapca_upload_csr_##SUF(h, x.nrows, x.ncols, x.nnz, x.row_offsets, x.col_indices, x.values, p, i, v); } \
    static sapca_status normalize(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p, const int32_t* i, T* v, const double* s, uint64_t sl, double t, int32_t d) { return sapca_normalize_csr_device_##SUF(h, m, n, nnz, p, i, v, s, sl, t, d); } \
    static sapca_status log1p(sapca_handle h, uint64_t nnz, T* v) { return sapca_log1p_csr_device_##SUF(h, nnz, v); } \
    static sapca_status stats(sapca_handle h, uint64_t m, uint64_t n, uint64_t nnz, const int64_t* p, const int32_t* i, const T* v, int32_t d, double* s, double* q, uint64_t* c, T* lo, T* hi) { return sapca_stats_csr_device_##SUF(h, m, n, nnz, p, i, v, d, s, q, c, lo, hi); } \
  };
SAPCA_RES(f32, float)
SAPCA_RES(f64, double)
#undef SAPCA_RES

template <typename T>
class ResidentCsr {
 public:
  ResidentCsr(sapca_handle h, const CsrRef<T>& x) : h_(h), m_(x.nrows), n_(x.ncols), nnz_(x.nnz) {
    check(ResidentAbi<T>::upload(h_, x, &ptr_, &idx_, &val_));
  }
  void normalize(const std::vector<double>& sums, double target, Direction d) {
    check(ResidentAbi<T>::normalize(h_, m_, n_, nnz_, ptr_, idx_, val_, sums.data(), sums.size(), target, (int32_t)d));
  }
  void log1p_normalize() { check(ResidentAbi<T>::log1p(h_, nnz_, val_)); }
  // values() is writable: after editing them with a kernel of your own, say so before the next fit
  void values_changed() { check(sapca_upload_values_changed(h_)); }
  std::vector<double> sum(Direction d) const {
    std::vector<double> s(d == Direction::COLUMN ? n_ : m_);
    check(ResidentAbi<T>::stats(h_, m_, n_, nnz_, ptr_, idx_, val_, (int32_t)d, s.data(), nullptr, nullptr, nullptr, nullptr));
    return s;
  }
  std::vector<uint64_t> nonzero(Direction d) const {
    std::vector<uint64_t> c(d == Direction::COLUMN ? n_ : m_);
    check(ResidentAbi<T>::stats(h_, m_, n_, nnz_, ptr_, idx_, val_, (int32_t)d, nullptr, nullptr, c.data(), nullptr, nullptr));
    return c;
  }
  const int64_t* row_offsets() const { return ptr_; }
  const int32_t* col_indices() const { return idx_; }
  T* values() const { return val_; }

 private:
  void check(sapca_status st) const {
    if (st != SAPCA_OK) throw Error(st, sapca_last_error(h_));
  }
  sapca_handle h_;
  uint64_t m_, n_, nnz_;
  const int64_t* ptr_ = nullptr;
  const int32_t* idx_ = nullptr;
  T* val_ = nullptr;
};

template <typename T>
class SparsePCA {
 public:
  SparsePCA(size_t n_components, T alpha, T tolerance, uint32_t seed, bool center, bool verbose, SVDMethod m,
            const std::vector<bool>* mask = nullptr)
      : k_(n_components), mask_len_(mask ? mask->size() : 0), masked_(mask != nullptr) {
    sapca_options o;
    sapca_options_default(&o);
    o.n_components = n_components; o.alpha = alpha; o.tolerance = tolerance; o.random_seed = seed;
    o.center = center; o.verbose = verbose;
    o.method = m.random ? SAPCA_RANDOM : SAPCA_LANCZOS;
    o.n_oversamples = m.n_oversamples; o.n_power_iterations = m.n_power_iterations; o.normalizer = (int32_t)m.normalizer;
    sapca_status st = sapca_create(&o, &h_);
    if (st != SAPCA_OK) throw Error(st, sapca_last_error(nullptr));
    if (mask && !mask->empty()) {
      std::vector<uint8_t> b(mask->begin(), mask->end());
      check(sapca_set_mask(h_, b.data(), b.size()));
    }
  }
  ~SparsePCA() { sapca_destroy(h_); }
  SparsePCA(const SparsePCA&) = delete;
  SparsePCA& operator=(const SparsePCA&) = delete;

  SparsePCA& fit(const CsrRef<T>& x) { mask_check(x); check(Abi<T>::fit(h_, x)); return *this; }
  std::vector<T> transform(const CsrRef<T>& x) const {                     // m x k row-major
    mask_check(x);
    std::vector<T> out(x.nrows * k_);
    check(Abi<T>::transform(h_, x, out.data()));
    return out;
  }
  std::vector<T> fit_transform(const CsrRef<T>& x) {
    mask_check(x);
    std::vector<T> out(x.nrows * k_);
    check(Abi<T>::fit_transform(h_, x, out.data()));
    return out;
  }
  std::vector<T> feature_importances() const {                             // k x n_used row-major
    uint64_t k, nu, nc;
    check(sapca_get_dims(h_, &k, &nu, &nc));
    std::vector<T> out(k * nu);
    check(Abi<T>::importances(h_, out.data(), out.size()));
    return out;
  }
  std::vector<T> explained_variance_ratio() const { return vec(&Abi<T>::ratio); }
  std::vector<T> cumulative_explained_variance_ratio() const { return vec(&Abi<T>::cumulative); }
  sapca_handle handle() const { return h_; }

 private:
  void check(sapca_status st) const { if (st != SAPCA_OK) throw Error(st, sapca_last_error(h_)); }
  void mask_check(const CsrRef<T>& x) const {   // the reference rejects any mismatch, an empty mask included (masked :258-262)
    if (masked_ && x.ncols != mask_len_)
      throw Error(SAPCA_ERR_MASK_LEN, "The mask vector length and the number of features (columns) have to be the same!");
  }
  std::vector<T> vec(sapca_status (*get)(sapca_handle, T*, size_t)) const {
    uint64_t k, nu, nc;
    check(sapca_get_dims(h_, &k, &nu, &nc));
    std::vector<T> out(k);
    check(get(h_, out.data(), out.size()));
    return out;
  }
  sapca_handle h_ = nullptr;
  size_t k_, mask_len_;
  bool masked_;
};

// SparsePCABuilder<T> (sparse/mod.rs:375-484) and MaskedSparsePCABuilder<T> (sparse_masked/mod.rs:37-160)
template <typename T, bool Masked>
class BuilderT {
 public:
  BuilderT& n_components(size_t n) { k_ = n; return *this; }
  BuilderT& alpha(T a) { alpha_ = a; return *this; }
  BuilderT& tolerance(T t) { tol_ = t; return *this; }
  BuilderT& random_seed(uint32_t s) { seed_ = s; return *this; }
  BuilderT& center(bool c) { center_ = c; return *this; }
  BuilderT& verbose(bool v) { verbose_ = v; return *this; }
  BuilderT& svd_method(SVDMethod m) { method_ = m; return *this; }
  BuilderT& mask(std::vector<bool> m) { static_assert(Masked, "mask() belongs to MaskedSparsePCABuilder"); mask_ = std::move(m); return *this; }
  SparsePCA<T>* build() const { return new SparsePCA<T>(k_, alpha_, tol_, seed_, center_, verbose_, method_, Masked ? &mask_ : nullptr); }

 private:
  size_t k_ = 50; T alpha_ = 1; T tol_ = (T)1e-6; uint32_t seed_ = 42; bool center_ = true, verbose_ = false;   // :392-401
  SVDMethod method_{};
  std::vector<bool> mask_;
};
template <typename T> using SparsePCABuilder = BuilderT<T, false>;
template <typename T> using MaskedSparsePCABuilder = BuilderT<T, true>;

}  // namespace sapca
