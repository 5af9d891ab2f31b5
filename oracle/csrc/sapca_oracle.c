/*
 * C / OpenMP restatement of the reference's sparse-PCA CPU path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  PARITY UNPINNED: see the
 * header of oracle/sapca_oracle.py -- the reference's arithmetic for this path
 * lives in single-svdlib 1.0.9, which is not available; this file restates the
 * in-tree semantics line by line and the published algorithm for the rest.
 * It is validated against oracle/sapca_oracle.py (numpy/LAPACK) in tests/.
 *
 * Paths below are relative to /root/reference.
 *
 *   orc_sum_col / orc_sum_col_squared    src/sparse/csr.rs:259-312, 558-608
 *   orc_randomized_fit                   src/dimred/pca/sparse/mod.rs:102-242
 *                                        (Random branch :161-198) with the
 *                                        single-svdlib call restated after
 *                                        scikit-learn extmath.py:287-353,374-590
 *   orc_transform_masked                 src/dimred/pca/sparse_masked/mod.rs:438-546
 *   orc_transform_sparse                 src/dimred/pca/sparse/mod.rs:255-285 (closed form)
 *
 * Build: oracle/Makefile  ->  oracle/build/liborc.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PARALLEL_THRESHOLD 200000 /* csr.rs:19 */
#define RAYON_CHUNK 8192          /* csr.rs:289, 585 */

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#define DEFINE_FOR(T, SUF)                                                                        \
  /* R1/R2: column sums.  sq=0: sum, sq=1: sum of squares.  Below the threshold: one serial */   \
  /* scatter-add (csr.rs:273-284).  Above: chunks of 8192 entries, each with a freshly zeroed */ \
  /* n-vector that is then added into the running result (csr.rs:286-308) -- the O(chunks*n) */  \
  /* cost is part of the reference's path, so it is kept.                                      */ \
  void orc_sum_col_##SUF(uint64_t nnz, const uint64_t* col, const T* val, uint64_t n, int sq,     \
                         T* out) {                                                                \
    memset(out, 0, n * sizeof(T));                                                                \
    if (nnz == 0 || n == 0) return;                                                               \
    if (nnz < PARALLEL_THRESHOLD) {                                                               \
      for (uint64_t e = 0; e < nnz; ++e) out[col[e]] += sq ? val[e] * val[e] : val[e];            \
      return;                                                                                     \
    }                                                                                             \
    uint64_t nchunks = (nnz + RAYON_CHUNK - 1) / RAYON_CHUNK;                                     \
    _Pragma("omp parallel")                                                                       \
    {                                                                                             \
      T* acc = (T*)calloc(n, sizeof(T));                                                          \
      T* loc = (T*)malloc(n * sizeof(T));                                                         \
      _Pragma("omp for schedule(static)")                                                         \
      for (uint64_t c = 0; c < nchunks; ++c) {                                                    \
        memset(loc, 0, n * sizeof(T));                                                            \
        uint64_t e0 = c * RAYON_CHUNK, e1 = e0 + RAYON_CHUNK < nnz ? e0 + RAYON_CHUNK : nnz;      \
        for (uint64_t e = e0; e < e1; ++e) loc[col[e]] += sq ? val[e] * val[e] : val[e];          \
        for (uint64_t j = 0; j < n; ++j) acc[j] += loc[j];                                        \
      }                                                                                           \
      _Pragma("omp critical")                                                                     \
      for (uint64_t j = 0; j < n; ++j) out[j] += acc[j];                                          \
      free(acc);                                                                                  \
      free(loc);                                                                                  \
    }                                                                                             \
  }                                                                                               \
                                                                                                  \
  /* R8: Y = A X - 1 c^T, c = X^T mu (c == NULL: uncentred).  X is n x l, Y is m x l. */          \
  static void spmm_##SUF(uint64_t m, const uint64_t* ptr, const uint64_t* col, const T* val,      \
                         const T* X, uint64_t l, const T* c, T* Y) {                              \
    _Pragma("omp parallel for schedule(dynamic, 256)")                                            \
    for (uint64_t i = 0; i < m; ++i) {                                                            \
      T* y = Y + i * l;                                                                           \
      for (uint64_t j = 0; j < l; ++j) y[j] = c ? -c[j] : (T)0;                                   \
      for (uint64_t e = ptr[i]; e < ptr[i + 1]; ++e) {                                            \
        const T a = val[e];                                                                       \
        const T* x = X + col[e] * l;                                                              \
        for (uint64_t j = 0; j < l; ++j) y[j] += a * x[j];                                        \
      }                                                                                           \
    }                                                                                             \
  }                                                                                               \
                                                                                                  \
  /* R9: Z = A^T Y - mu s^T, s = 1^T Y.  Thread-private n x l accumulators, then reduce. */       \
  static void spmmt_##SUF(uint64_t m, uint64_t n, const uint64_t* ptr, const uint64_t* col,       \
                          const T* val, const T* Y, uint64_t l, const T* mu, T* Z) {              \
    memset(Z, 0, n * l * sizeof(T));                                                              \
    _Pragma("omp parallel")                                                                       \
    {                                                                                             \
      T* zl = (T*)calloc(n * l, sizeof(T));                                                       \
      _Pragma("omp for schedule(dynamic, 256)")                                                   \
      for (uint64_t i = 0; i < m; ++i) {                                                          \
        const T* y = Y + i * l;                                                                   \
        for (uint64_t e = ptr[i]; e < ptr[i + 1]; ++e) {                                          \
          const T a = val[e];                                                                     \
          T* z = zl + col[e] * l;                                                                 \
          for (uint64_t j = 0; j < l; ++j) z[j] += a * y[j];                                      \
        }                                                                                         \
      }                                                                                           \
      _Pragma("omp critical")                                                                     \
      for (uint64_t t = 0; t < n * l; ++t) Z[t] += zl[t];                                         \
      free(zl);                                                                                   \
    }                                                                                             \
    if (mu) {                                                                                     \
      T* s = (T*)calloc(l, sizeof(T));                                                            \
      for (uint64_t i = 0; i < m; ++i)                                                            \
        for (uint64_t j = 0; j < l; ++j) s[j] += Y[i * l + j];                                    \
      _Pragma("omp parallel for")                                                                 \
      for (uint64_t r = 0; r < n; ++r)                                                            \
        for (uint64_t j = 0; j < l; ++j) Z[r * l + j] -= mu[r] * s[j];                            \
      free(s);                                                                                    \
    }                                                                                             \
  }                                                                                               \
                                                                                                  \
  /* R10 (QR): thin Householder QR of the r x l row-major panel P, Q returned in place.  */       \
  /* nalgebra's QR (what the reference's dependency calls) is unblocked Householder too;  */      \
  /* the reflector applications are OpenMP-parallel over columns here.                    */      \
  static void householder_q_##SUF(T* P, uint64_t r, uint64_t l) {                                 \
    T* V = (T*)malloc(r * l * sizeof(T)); /* reflectors, column k in V[:,k] (rows >= k) */        \
    T* beta = (T*)calloc(l, sizeof(T));                                                           \
    for (uint64_t k = 0; k < l && k < r; ++k) {                                                   \
      double nrm = 0;                                                                             \
      for (uint64_t i = k; i < r; ++i) nrm += (double)P[i * l + k] * P[i * l + k];                \
      nrm = sqrt(nrm);                                                                            \
      T x0 = P[k * l + k];                                                                        \
      T alpha = (T)(x0 >= 0 ? -nrm : nrm);                                                        \
      double vn = 0;                                                                              \
      for (uint64_t i = k; i < r; ++i) {                                                          \
        T v = P[i * l + k] - (i == k ? alpha : (T)0);                                             \
        V[i * l + k] = v;                                                                         \
        vn += (double)v * v;                                                                      \
      }                                                                                           \
      beta[k] = vn > 0 ? (T)(2.0 / vn) : (T)0;                                                    \
      _Pragma("omp parallel for schedule(static)")                                                \
      for (uint64_t j = k; j < l; ++j) {                                                          \
        double d = 0;                                                                             \
        for (uint64_t i = k; i < r; ++i) d += (double)V[i * l + k] * P[i * l + j];                \
        T f = (T)(d * beta[k]);                                                                   \
        for (uint64_t i = k; i < r; ++i) P[i * l + j] -= f * V[i * l + k];                        \
      }                                                                                           \
    }                                                                                             \
    /* accumulate Q = H_0 ... H_{l-1} [I; 0] */                                                   \
    memset(P, 0, r * l * sizeof(T));                                                              \
    for (uint64_t j = 0; j < l && j < r; ++j) P[j * l + j] = 1;                                   \
    for (uint64_t kk = (l < r ? l : r); kk-- > 0;) {                                              \
      _Pragma("omp parallel for schedule(static)")                                                \
      for (uint64_t j = kk; j < l; ++j) {                                                         \
        double d = 0;                                                                             \
        for (uint64_t i = kk; i < r; ++i) d += (double)V[i * l + kk] * P[i * l + j];              \
        T f = (T)(d * beta[kk]);                                                                  \
        for (uint64_t i = kk; i < r; ++i) P[i * l + j] -= f * V[i * l + kk];                      \
      }                                                                                           \
    }                                                                                             \
    free(V);                                                                                      \
    free(beta);                                                                                   \
  }                                                                                               \
                                                                                                  \
  /* R10 (LU): the row-permuted unit-lower factor of P (scipy lu(permute_l=True)). */             \
  static void lu_pl_##SUF(T* P, uint64_t r, uint64_t l) {                                         \
    uint64_t* perm = (uint64_t*)malloc(r * sizeof(uint64_t));                                     \
    T* W = (T*)malloc(r * l * sizeof(T));                                                         \
    memcpy(W, P, r * l * sizeof(T));                                                              \
    for (uint64_t i = 0; i < r; ++i) perm[i] = i;                                                 \
    uint64_t steps = l < r ? l : r;                                                               \
    for (uint64_t k = 0; k < steps; ++k) {                                                        \
      uint64_t piv = k;                                                                           \
      T best = (T)fabs((double)W[k * l + k]);                                                     \
      for (uint64_t i = k + 1; i < r; ++i) {                                                      \
        T a = (T)fabs((double)W[i * l + k]);                                                      \
        if (a > best) { best = a; piv = i; }                                                      \
      }                                                                                           \
      if (piv != k) {                                                                             \
        for (uint64_t j = 0; j < l; ++j) { T t = W[k * l + j]; W[k * l + j] = W[piv * l + j]; W[piv * l + j] = t; } \
        uint64_t t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;                                 \
      }                                                                                           \
      T d = W[k * l + k];                                                                         \
      if (d == 0) continue;                                                                       \
      _Pragma("omp parallel for schedule(static)")                                                \
      for (uint64_t i = k + 1; i < r; ++i) {                                                      \
        T f = W[i * l + k] / d;                                                                   \
        W[i * l + k] = f;                                                                         \
        for (uint64_t j = k + 1; j < l; ++j) W[i * l + j] -= f * W[k * l + j];                    \
      }                                                                                           \
    }                                                                                             \
    for (uint64_t i = 0; i < r; ++i)                                                              \
      for (uint64_t j = 0; j < l; ++j)                                                            \
        P[perm[i] * l + j] = j < i ? W[i * l + j] : (j == i ? (T)1 : (T)0);                       \
    free(W);                                                                                      \
    free(perm);                                                                                   \
  }                                                                                               \
                                                                                                  \
  static void normalize_##SUF(T* P, uint64_t r, uint64_t l, int normalizer) {                     \
    if (normalizer == 0) householder_q_##SUF(P, r, l);                                            \
    else if (normalizer == 1) lu_pl_##SUF(P, r, l);                                               \
  }                                                                                               \
                                                                                                  \
  /* R11: SVD of B^T (n x l, row-major in Z) by one-sided Jacobi (Hestenes): on exit the */       \
  /* columns of Z are sigma_j * (right-vectors-of-B)_j; returns sigma sorted descending   */      \
  /* with order[] giving the column permutation.                                          */      \
  static void jacobi_cols_##SUF(T* Zrm, uint64_t n, uint64_t l, double* sig, uint64_t* order) {   \
    /* the rotations work on a column-major copy: a column is one contiguous run of n values   */   \
    /* (row-major, every pass over a column pair touched n cache lines per column: minutes at  */   \
    /* l = 110).  Same rotations, same summation order along a column.                          */  \
    T* Z = (T*)malloc(n * l * sizeof(T));                                                         \
    _Pragma("omp parallel for schedule(static)")                                                  \
    for (uint64_t j = 0; j < l; ++j)                                                              \
      for (uint64_t i = 0; i < n; ++i) Z[j * n + i] = Zrm[i * l + j];                             \
    for (int sweep = 0; sweep < 60; ++sweep) {                                                    \
      double off = 0;                                                                             \
      for (uint64_t p = 0; p + 1 < l; ++p)                                                        \
        for (uint64_t q = p + 1; q < l; ++q) {                                                    \
          T* zp_ = Z + p * n;                                                                     \
          T* zq_ = Z + q * n;                                                                     \
          double a = 0, b = 0, g = 0;                                                             \
          _Pragma("omp parallel for reduction(+ : a, b, g) schedule(static)")                     \
          for (uint64_t i = 0; i < n; ++i) {                                                      \
            double zp = zp_[i], zq = zq_[i];                                                      \
            a += zp * zp; b += zq * zq; g += zp * zq;                                             \
          }                                                                                       \
          if (a == 0 || b == 0) continue;                                                         \
          double r = fabs(g) / sqrt(a * b);                                                       \
          if (r > off) off = r;                                                                   \
          if (r < 1e-15) continue;                                                                \
          double zeta = (b - a) / (2.0 * g);                                                      \
          double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));           \
          double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;                                       \
          _Pragma("omp parallel for schedule(static)")                                            \
          for (uint64_t i = 0; i < n; ++i) {                                                      \
            double zp = zp_[i], zq = zq_[i];                                                      \
            zp_[i] = (T)(cs * zp - sn * zq);                                                      \
            zq_[i] = (T)(sn * zp + cs * zq);                                                      \
          }                                                                                       \
        }                                                                                         \
      if (off < (sizeof(T) == 4 ? 5e-7 : 1e-14)) break;                                           \
    }                                                                                             \
    for (uint64_t j = 0; j < l; ++j) {                                                            \
      double a = 0;                                                                               \
      for (uint64_t i = 0; i < n; ++i) a += (double)Z[j * n + i] * Z[j * n + i];                  \
      sig[j] = sqrt(a);                                                                           \
      order[j] = j;                                                                               \
    }                                                                                             \
    _Pragma("omp parallel for schedule(static)")                                                  \
    for (uint64_t i = 0; i < n; ++i)                                                              \
      for (uint64_t j = 0; j < l; ++j) Zrm[i * l + j] = Z[j * n + i];                             \
    free(Z);                                                                                      \
    for (uint64_t i = 0; i < l; ++i)                                                              \
      for (uint64_t j = i + 1; j < l; ++j)                                                        \
        if (sig[order[j]] > sig[order[i]]) { uint64_t t = order[i]; order[i] = order[j]; order[j] = t; } \
  }                                                                                               \
                                                                                                  \
  /* R4 (Random branch) + R3 + R13 + R14.  omega: n x l row-major (required: the caller    */     \
  /* injects it; the reference's rand-0.9 stream is not reproducible).  Outputs:            */    \
  /* components k x n, sing k, expl_var k, mean n.  Returns 0, or 1 when rank < k.          */    \
  int orc_randomized_fit_##SUF(uint64_t m, uint64_t n, const uint64_t* ptr, const uint64_t* col,  \
                               const T* val, uint64_t k, uint64_t p, uint64_t q, int normalizer,  \
                               int center, const T* omega, T* components, T* sing, T* expl_var,   \
                               T* mean, T* total_var_out) {                                       \
    uint64_t nnz = ptr[m], l = k + p;                                                             \
    T* s1 = (T*)malloc(n * sizeof(T));                                                            \
    T* s2 = (T*)malloc(n * sizeof(T));                                                            \
    T total_var = 0;                                                                              \
    if (center) {                                                                                 \
      orc_sum_col_##SUF(nnz, col, val, n, 0, s1);                 /* sparse/mod.rs:107 */         \
      for (uint64_t j = 0; j < n; ++j) mean[j] = s1[j] / (T)m;    /* :108-114 */                  \
      orc_sum_col_##SUF(nnz, col, val, n, 0, s1);                 /* :121 (second pass) */        \
      orc_sum_col_##SUF(nnz, col, val, n, 1, s2);                 /* :122 */                      \
      for (uint64_t j = 0; j < n; ++j) {                          /* :126-130 */                  \
        T mj = s1[j] / (T)m;                                                                      \
        total_var += (s2[j] - mj * s1[j]) / ((T)m - (T)1);                                        \
      }                                                                                           \
    } else {                                                                                      \
      memset(mean, 0, n * sizeof(T));                                                             \
    }                                                                                             \
    free(s1);                                                                                     \
    free(s2);                                                                                     \
    const T* mu = center ? mean : NULL;                                                           \
    T* Q = (T*)malloc(n * l * sizeof(T));                                                         \
    T* Y = (T*)malloc(m * l * sizeof(T));                                                         \
    T* c = (T*)malloc(l * sizeof(T));                                                             \
    memcpy(Q, omega, n * l * sizeof(T));                                                          \
    for (uint64_t it = 0; it <= q; ++it) {                                                        \
      if (mu) {                                                                                   \
        for (uint64_t j = 0; j < l; ++j) c[j] = 0;                                                \
        for (uint64_t r = 0; r < n; ++r)                                                          \
          for (uint64_t j = 0; j < l; ++j) c[j] += mu[r] * Q[r * l + j];                          \
      }                                                                                           \
      spmm_##SUF(m, ptr, col, val, Q, l, mu ? c : NULL, Y);                                       \
      if (it == q) break;                                                                         \
      normalize_##SUF(Y, m, l, normalizer);                                                       \
      spmmt_##SUF(m, n, ptr, col, val, Y, l, mu, Q);                                              \
      normalize_##SUF(Q, n, l, normalizer);                                                       \
    }                                                                                             \
    householder_q_##SUF(Y, m, l);                       /* final range basis: always QR */        \
    spmmt_##SUF(m, n, ptr, col, val, Y, l, mu, Q);      /* Q now holds B^T = Ac^T Q, n x l */     \
    double* sg = (double*)malloc(l * sizeof(double));                                             \
    uint64_t* ord = (uint64_t*)malloc(l * sizeof(uint64_t));                                      \
    jacobi_cols_##SUF(Q, n, l, sg, ord);                                                          \
    int rc = 0;                                                                                   \
    for (uint64_t r = 0; r < k; ++r) {                                                            \
      uint64_t j = ord[r];                                                                        \
      double sv = sg[j];                                                                          \
      sing[r] = (T)sv;                                                                            \
      if (!(sv > 0)) rc = 1;                                                                      \
      /* svd_flip, v-based (sparse/mod.rs:201-206): largest |.| entry of the row made positive */ \
      uint64_t arg = 0;                                                                           \
      double best = -1;                                                                           \
      for (uint64_t i = 0; i < n; ++i) {                                                          \
        double a = fabs((double)Q[i * l + j]);                                                    \
        if (a > best) { best = a; arg = i; }                                                      \
      }                                                                                           \
      double sgn = Q[arg * l + j] < 0 ? -1.0 : 1.0;                                               \
      for (uint64_t i = 0; i < n; ++i) components[r * n + i] = (T)(sgn * Q[i * l + j] / (sv > 0 ? sv : 1.0)); \
      expl_var[r] = (T)(sv * sv) / (T)(m - 1);                    /* :210-216 */                  \
    }                                                                                             \
    if (!center) {                                                /* :218-223 (over the k kept) */\
      total_var = 0;                                                                              \
      for (uint64_t r = 0; r < k; ++r) total_var += expl_var[r];                                  \
    }                                                                                             \
    if (total_var_out) *total_var_out = total_var;                                                \
    free(sg); free(ord); free(Q); free(Y); free(c);                                               \
    return rc;                                                                                    \
  }                                                                                               \
                                                                                                  \
  /* One centred sparse x dense sweep on its own (for timing the CPU SpMM in GB/s). */            \
  void orc_spmm_##SUF(uint64_t m, const uint64_t* ptr, const uint64_t* col, const T* val,         \
                      const T* X, uint64_t l, const T* c, T* Y) {                                 \
    spmm_##SUF(m, ptr, col, val, X, l, c, Y);                                                     \
  }                                                                                               \
  void orc_spmmt_##SUF(uint64_t m, uint64_t n, const uint64_t* ptr, const uint64_t* col,          \
                       const T* val, const T* Y, uint64_t l, const T* mu, T* Z) {                 \
    spmmt_##SUF(m, n, ptr, col, val, Y, l, mu, Z);                                                \
  }                                                                                               \
  void orc_householder_q_##SUF(T* P, uint64_t r, uint64_t l) { householder_q_##SUF(P, r, l); }    \
  void orc_lu_pl_##SUF(T* P, uint64_t r, uint64_t l) { lu_pl_##SUF(P, r, l); }                    \
                                                                                                  \
  /* R16: t_ik = sum over stored (j, a) of row i with o2m[j] >= 0 of                        */    \
  /*      (a - [center] mu_j) * V[k, o2m[j]]   (sparse_masked/mod.rs:488-529).              */    \
  /* o2m == NULL: identity map (unmasked).  comps is k x n_used row-major, out m x k.       */    \
  void orc_transform_masked_##SUF(uint64_t m, const uint64_t* ptr, const uint64_t* col,           \
                                  const T* val, const int64_t* o2m, uint64_t n_used, uint64_t k,  \
                                  const T* comps, const T* mean, int center, T* out) {            \
    _Pragma("omp parallel for schedule(dynamic, 64)")                                             \
    for (uint64_t i = 0; i < m; ++i) {                                                            \
      T* t = out + i * k;                                                                         \
      for (uint64_t kk = 0; kk < k; ++kk) t[kk] = 0;                                              \
      for (uint64_t e = ptr[i]; e < ptr[i + 1]; ++e) {                                            \
        int64_t mi = o2m ? o2m[col[e]] : (int64_t)col[e];                                         \
        if (mi < 0) continue;                                                                     \
        T eff = center ? val[e] - mean[col[e]] : val[e];                                          \
        for (uint64_t kk = 0; kk < k; ++kk) t[kk] += eff * comps[kk * n_used + (uint64_t)mi];     \
      }                                                                                           \
    }                                                                                             \
  }                                                                                               \
                                                                                                  \
  /* R15 closed form: t_ik = sum_j cnt_j (x_ij - [center] mu_j) V_kj  (sparse/mod.rs:268-282). */ \
  void orc_transform_sparse_##SUF(uint64_t m, uint64_t n, const uint64_t* ptr,                    \
                                  const uint64_t* col, const T* val, uint64_t k, const T* comps,  \
                                  const T* mean, int center, T* out) {                            \
    uint64_t nnz = ptr[m];                                                                        \
    T* cnt = (T*)calloc(n, sizeof(T));                                                            \
    for (uint64_t e = 0; e < nnz; ++e) cnt[col[e]] += 1;                                          \
    T* W = (T*)malloc(n * k * sizeof(T));                                                         \
    T* c = (T*)calloc(k, sizeof(T));                                                              \
    for (uint64_t j = 0; j < n; ++j)                                                              \
      for (uint64_t kk = 0; kk < k; ++kk) {                                                       \
        W[j * k + kk] = cnt[j] * comps[kk * n + j];                                               \
        if (center) c[kk] += mean[j] * W[j * k + kk];                                             \
      }                                                                                           \
    spmm_##SUF(m, ptr, col, val, W, k, center ? c : NULL, out);                                   \
    free(cnt); free(W); free(c);                                                                  \
  }

DEFINE_FOR(float, f32)
DEFINE_FOR(double, f64)
